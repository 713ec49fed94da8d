"""Static check of the 8-chain spline kernel's ISA (fab_torch_amd/csrc/spline_r8.h): the weight ring lives in AGPRs that
inline-asm loads fill behind hipcc's back, so the compiler must never read or move such a register between the load and
the hand-counted `s_waitcnt vmcnt(N)` that covers it.  Walks every k_spline_logprob_r8 kernel linearly (the GEMM stages are
straight-line), tracks the loads in flight and flags (1) any non-load instruction that reads or writes an AGPR whose load may
still be in flight, (2) a VMEM load whose scalar base was written by v_readlane / v_readfirstlane fewer than 5 wait states
earlier.  Usage: python tools/check_r8_isa.py [spline] [flow]  (compiles the .hip for gfx950 into /tmp; no GPU needed)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


SOURCES = {"spline": ("spline_kernels.hip", ("k_spline_logprob_r8",)),
           "flow": ("ais_kernels.hip", ("k_hmc_step_r8", "k_ais_init_r8"))}


def disassemble(src):
    """ISA text of the gfx950 code object of `src`: taken from the in-tree build (fab_torch_amd/build/<stem>.o, when it is
    newer than every source under csrc/) or compiled into a temporary directory."""
    tmp = tempfile.mkdtemp(prefix="r8isa")
    elf = os.path.join(tmp, "dev.elf")
    csrc = os.path.join(ROOT, "fab_torch_amd", "csrc")
    obj = os.path.join(ROOT, "fab_torch_amd", "build", os.path.splitext(src)[0] + ".o")
    newest = max(os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc))
    if os.path.exists(obj) and os.path.getmtime(obj) >= newest and not os.environ.get("R8ISA_RECOMPILE"):
        fat = os.path.join(tmp, "fat.bin")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj])
        bundle = fat
    else:
        bundle = os.path.join(tmp, "sp.co")
        subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-c",
                               os.path.join(csrc, src), "-I", os.path.join(ROOT, "include"), "-w", "-o", bundle])
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={bundle}",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={elf}"])
    return subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", elf], text=True)


def agprs(text):
    out = set()
    for m in re.finditer(r"\ba(\d+)\b|\ba\[(\d+):(\d+)\]", text):
        if m.group(1):
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check_kernel(name, lines):
    inflight = []          # list of register sets, oldest first (asm loads only; other VMEM ops only delay completion)
    bad = []
    for i, l in enumerate(lines):
        m = re.match(r"global_load_dwordx4 a\[(\d+):(\d+)\], v\d+, s\[(\d+):(\d+)\]", l)
        if m:
            inflight.append(set(range(int(m.group(1)), int(m.group(2)) + 1)))
            base = {"s" + m.group(3), "s" + m.group(4)}
            ws = 0
            for j in range(i - 1, max(i - 8, -1), -1):
                p = lines[j]
                if p.startswith("s_nop"):
                    ws += int(p.split()[1]) + 1
                    continue
                mm = re.match(r"v_read(?:first)?lane_b32 (s\d+)", p)
                if mm and mm.group(1) in base:
                    if ws < 5:
                        bad.append((i, "scalar-base hazard", l))
                    break
                ws += 1
                if ws >= 5:
                    break
            continue
        m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", l)
        if m:
            n = int(m.group(1))
            if len(inflight) > n:
                inflight = inflight[len(inflight) - n:] if n else []
            continue
        if l.startswith("s_barrier") or l.startswith("s_cbranch") or l.startswith("s_branch"):
            continue
        used = agprs(l)
        if used and inflight:
            fl = set().union(*inflight)
            if used & fl:
                bad.append((i, "touches an AGPR with a load in flight", l))
    return bad


def main():
    rc = 0
    for which in (sys.argv[1:] or sorted(SOURCES)):
        rc |= check_source(*SOURCES[which])
    return rc


def check_source(src, patterns):
    text = disassemble(src)
    kernels, cur, name = {}, None, None
    for raw in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", raw)
        if m:
            name = m.group(1)
            cur = kernels.setdefault(name, []) if any(p in name for p in patterns) else None
            continue
        if cur is not None:
            ins = raw.split("//")[0].strip()
            if ins:
                cur.append(ins)
    if not kernels:
        print(f"{src}: no kernel matching {patterns} found")
        return 1
    rc = 0
    for name, lines in sorted(kernels.items()):
        bad = check_kernel(name, lines)
        n_ld = sum(1 for l in lines if l.startswith("global_load_dwordx4 a["))
        print(f"{name}: {len(lines)} instructions, {n_ld} ring loads, {len(bad)} findings")
        for b in bad[:12]:
            print("   ", b)
        rc |= bool(bad)
    return rc


if __name__ == "__main__":
    sys.exit(main())
