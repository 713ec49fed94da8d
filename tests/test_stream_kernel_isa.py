"""The flow / spline kernels keep weight rings in registers that inline-asm loads fill behind hipcc's back (csrc/flow_device.h,
flow_r4.h, stream_r8.h, spline_r8.h, flow_r8.h), waited for with hand-counted s_waitcnt.  That is only sound while the compiler
never reads, moves or re-allocates such a register between its load and the wait that covers it - a property of the GENERATED
code, so it is checked on the ISA of the in-tree build (fab_torch_amd/_isa_check.py; the build itself runs the same check and
refuses to produce a library that fails it) for EVERY kernel of the library, along every control-flow path; no GPU needed."""
import os

import pytest

from fab_torch_amd import _isa_check as chk

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "fab_torch_amd", "build")


def test_checker_flags_a_copy_of_an_in_flight_ring_register_and_a_scalar_base_hazard():
    ok = ["global_load_dwordx4 a[0:3], v1, s[2:3]", "global_load_dwordx4 a[4:7], v1, s[2:3] offset:1024",
          "s_waitcnt vmcnt(1)", "v_mfma_f32_4x4x1_16b_f32 a[8:11], v2, a0, a[8:11]", "s_waitcnt vmcnt(0)",
          "v_mfma_f32_4x4x1_16b_f32 a[8:11], v2, a4, a[8:11]"]
    assert chk.check_kernel("ok", ok) == []
    copied = ok[:2] + ["v_accvgpr_mov_b32 a20, a5"] + ok[2:]                     # a5's load is still in flight
    assert [b[1] for b in chk.check_kernel("copied", copied)] == ["touches a register with a load in flight"]
    early = ok[:2] + ["v_mfma_f32_4x4x1_16b_f32 a[8:11], v2, a4, a[8:11]"] + ok[2:]   # used before its wait
    assert len(chk.check_kernel("early", early)) == 1
    vgpr = ["global_load_dwordx4 v[10:13], v1, s[2:3]", "v_add_f32_e32 v4, v11, v5", "s_waitcnt vmcnt(0)"]   # VGPR rings too (flow_r4.h)
    assert len(chk.check_kernel("vgpr", vgpr)) == 1
    store_counts = ["global_load_dwordx4 v[10:13], v1, s[2:3]", "global_store_dword v[2:3], v4, off", "s_waitcnt vmcnt(1)",
                    "v_add_f32_e32 v4, v11, v5"]                                # the store is the newest entry: the load has landed
    assert chk.check_kernel("store", store_counts) == []
    hazard = ["v_readlane_b32 s2, v9, 0", "s_nop 1", "global_load_dwordx4 a[0:3], v1, s[2:3]", "s_waitcnt vmcnt(0)"]
    assert [b[1] for b in chk.check_kernel("hazard", hazard)] == ["scalar-base hazard"]
    padded = ["v_readlane_b32 s2, v9, 0", "s_nop 4", "global_load_dwordx4 a[0:3], v1, s[2:3]", "s_waitcnt vmcnt(0)"]
    assert chk.check_kernel("padded", padded) == []


def test_checker_flags_a_bf16_mfma_that_writes_over_its_own_operand():
    """gfx950 returns a wrong first row per 4-row group when a bf16 MFMA's destination overlaps its A / B operand (round 6: found
    with the accumulators in architectural registers, -amdgpu-mfma-vgpr-form); the build refuses such code."""
    mk = lambda lines: chk.from_lines(lines)                                     # noqa: E731
    ok = mk(["v_mfma_f32_16x16x32_bf16 v[36:39], v[52:55], v[40:43], v[36:39]",
             "v_mfma_f32_4x4x4_16b_bf16 v[8:11], v[16:17], a[68:69], v[8:11]",
             "v_mfma_f32_16x16x4_f32 v[92:95], v95, v75, v[96:99]"])            # (the fp32 forms are unaffected)
    assert chk.check_bf16_mfma_overlap("ok", ok) == []
    bad = mk(["v_mfma_f32_16x16x32_bf16 v[36:39], v[52:55], v[36:39], v[48:51]",
              "v_mfma_f32_4x4x4_16b_bf16 v[106:109], v[106:107], a[26:27], 0"])
    assert [b[1] for b in chk.check_bf16_mfma_overlap("bad", bad)] == ["bf16 MFMA destination overlaps its B operand",
                                                                       "bf16 MFMA destination overlaps its A operand"]


def test_checker_follows_loop_back_edges():
    """A ring slot requested at the bottom of a loop and read at its top, before the wait: only visible along the back edge."""
    loop = [(0, "s_mov_b32 s0, 0", None), (4, "v_add_f32_e32 v4, v10, v5", None), (8, "s_waitcnt vmcnt(0)", None),
            (12, "global_load_dwordx4 v[10:13], v1, s[2:3]", None), (16, "s_cbranch_scc1 65533", 4), (20, "s_waitcnt vmcnt(0)", None),
            (24, "s_endpgm", None)]
    assert [b[0] for b in chk.check_kernel("loop", loop)] == [1]
    fixed = list(loop)
    fixed[1], fixed[2] = (4, "s_waitcnt vmcnt(0)", None), (8, "v_add_f32_e32 v4, v10, v5", None)
    assert chk.check_kernel("fixed", fixed) == []


def test_checker_pins_the_lowering_of_the_in_kernel_step_size_rule():
    """hmc_adapt_last (csrc/ais_kernels.hip) relies on how gfx950 lowers device-scope relaxed atomics: write-through (sc1) stores,
    a ticket drawn behind `s_waitcnt vmcnt(0)`, sc1 loads in the last wave (outside the HIP memory model).  The
    build's ISA check fails when a compiler lowers them differently."""
    name = "_ZN3fab13k_hmc_step_r4ILi5ELb0ELi2EEE"
    good = (["global_store_dword v[6:7], v1, off sc1", "global_store_dword v[8:9], v1, off sc1", "s_waitcnt vmcnt(0)",
             "v_mov_b32_e32 v5, 0", "s_waitcnt vmcnt(2)", "global_atomic_add v5, v5, v6, s[40:41] sc0"] +
            [f"global_load_dword v{10 + k}, v[8:9], off offset:{4 * k} sc1" for k in range(32)] + ["s_waitcnt vmcnt(0)"])
    assert chk.check_adapt_fold(name, chk.from_lines(good)) == []
    assert chk.check_adapt_fold("some_other_kernel", chk.from_lines(good[2:])) == []          # only the kernels that hold the rule
    plain_stores = [g.replace(" sc1", "") if g.startswith("global_store") else g for g in good]
    assert any("write-through" in b[1] for b in chk.check_adapt_fold(name, chk.from_lines(plain_stores)))
    plain_loads = [g.replace(" sc1", "") if g.startswith("global_load") else g for g in good]
    assert any("sc1 statistic loads" in b[1] for b in chk.check_adapt_fold(name, chk.from_lines(plain_loads)))
    no_wait = [g for g in good if g != "s_waitcnt vmcnt(0)" or good.index(g) > 5]
    assert any("ticket" in b[1] for b in chk.check_adapt_fold(name, chk.from_lines(no_wait)))
    late_store = good[:3] + ["global_store_dword v[8:9], v1, off sc1"] + good[3:]              # a store between the wait and the ticket
    assert any("ticket" in b[1] for b in chk.check_adapt_fold(name, chk.from_lines(late_store)))


@pytest.mark.parametrize("stem", ["flow_kernels", "ais_kernels", "spline_kernels", "train_kernels"])
def test_no_instruction_touches_a_register_whose_load_is_in_flight(stem):
    obj = os.path.join(BUILD, stem + ".o")
    if not os.path.exists(obj):
        pytest.skip("needs the in-tree build (fab_torch_amd/build/*.o)")
    if not chk.tools_available():
        pytest.skip("needs the ROCm LLVM tools")
    res = chk.check_object(obj)
    assert res, f"{stem}: no kernels found"
    names = " ".join(res)
    if stem == "ais_kernels":
        for k in ("k_hmc_step_r4", "k_ais_init_r4", "k_hmc_step_r8", "k_ais_init_r8", "k_hmc_stepILi5E"):
            assert k in names, f"{k} is not in the library any more"
    if stem == "spline_kernels":
        assert "k_spline_logprob_r8" in names
    bad = {k: v for k, v in res.items() if v}
    assert not bad, bad
