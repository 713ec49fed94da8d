"""The 4x4x1 stream kernels (csrc/stream_r8.h, spline_r8.h, flow_r8.h) keep their weight ring in AGPRs that inline-asm loads
fill behind hipcc's back, waited for with hand-counted s_waitcnt.  That is only sound while the compiler never reads, moves or
re-allocates such a register between its load and the wait that covers it - a property of the GENERATED code, so it is
checked on the ISA of the in-tree build (tools/check_r8_isa.py) for every instantiation of the kernels; no GPU needed."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_r8_isa", os.path.join(ROOT, "tools", "check_r8_isa.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)


def test_checker_flags_a_copy_of_an_in_flight_ring_register_and_a_scalar_base_hazard():
    ok = ["global_load_dwordx4 a[0:3], v1, s[2:3]", "global_load_dwordx4 a[4:7], v1, s[2:3] offset:1024",
          "s_waitcnt vmcnt(1)", "v_mfma_f32_4x4x1_16b_f32 a[8:11], v2, a0, a[8:11]", "s_waitcnt vmcnt(0)",
          "v_mfma_f32_4x4x1_16b_f32 a[8:11], v2, a4, a[8:11]"]
    assert chk.check_kernel("ok", ok) == []
    copied = ok[:2] + ["v_accvgpr_mov_b32 a20, a5"] + ok[2:]                     # a5's load is still in flight
    assert [b[1] for b in chk.check_kernel("copied", copied)] == ["touches an AGPR with a load in flight"]
    early = ok[:2] + ["v_mfma_f32_4x4x1_16b_f32 a[8:11], v2, a4, a[8:11]"] + ok[2:]   # used before its wait
    assert len(chk.check_kernel("early", early)) == 1
    hazard = ["v_readlane_b32 s2, v9, 0", "s_nop 1", "global_load_dwordx4 a[0:3], v1, s[2:3]", "s_waitcnt vmcnt(0)"]
    assert [b[1] for b in chk.check_kernel("hazard", hazard)] == ["scalar-base hazard"]
    padded = ["v_readlane_b32 s2, v9, 0", "s_nop 4", "global_load_dwordx4 a[0:3], v1, s[2:3]", "s_waitcnt vmcnt(0)"]
    assert chk.check_kernel("padded", padded) == []


@pytest.mark.parametrize("which", sorted(chk.SOURCES))
def test_no_instruction_touches_a_ring_register_whose_load_is_in_flight(which):
    src, patterns = chk.SOURCES[which]
    obj = os.path.join(ROOT, "fab_torch_amd", "build", os.path.splitext(src)[0] + ".o")
    if not os.path.exists(obj) and not shutil.which("hipcc"):
        pytest.skip("needs the in-tree build or hipcc")
    if not os.path.exists(os.path.join(chk.LLVM, "llvm-objdump")):
        pytest.skip("needs the ROCm LLVM tools")
    text = chk.disassemble(src)
    kernels, cur = {}, None
    for raw in text.splitlines():
        m = chk.re.match(r"^[0-9a-f]+ <(\S+)>:", raw)
        if m:
            cur = kernels.setdefault(m.group(1), []) if any(p in m.group(1) for p in patterns) else None
        elif cur is not None:
            ins = raw.split("//")[0].strip()
            if ins:
                cur.append(ins)
    assert kernels, f"{src}: no kernel matching {patterns}"
    for name, lines in kernels.items():
        assert sum(1 for l in lines if l.startswith("global_load_dwordx4 a[")) >= 100, f"{name}: the ring loads are gone"
        assert chk.check_kernel(name, lines) == [], name
