"""Build, in-tree with hipcc: (1) libfabhip.so — the HIP kernels (gfx950) behind the plain C ABI of include/fabhip.h,
no torch headers involved; (2) _fabhip_torch.so — the TORCH_LIBRARY(fabhip) custom-op layer over that ABI
(csrc/torch_ops.cpp, host C++ compiled against the running torch)."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libfabhip.so")
SOURCES = ["flow_kernels.hip", "ais_kernels.hip", "reduce_resample.hip", "train_kernels.hip", "topk.hip",
           "generic_kernels.hip", "spline_kernels.hip", "train_step.hip"]
ARCH = "gfx950"
# -amdgpu-mfma-vgpr-form (round 6): MFMA accumulators in ARCHITECTURAL registers.  hipcc's default puts them in the accumulation
# file, whose only tenant here should be the weight rings (inline-asm loads): every epilogue then starts with one v_accvgpr_read
# per accumulator register (32 per wide stage of the 8-chain kernels, in front of VALU work nothing overlaps) and every zeroing is
# a v_accvgpr_write.  With the flag: k_hmc_step_r8<5> 0.673 -> 0.638 ms at 2048 chains, k_hmc_step_r4 0.525 -> 0.511 at 512; the
# register budgets still hold (the build refuses spills) and the ring's ISA check runs on the result as before.
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-result", "-Wno-pass-failed",
         "-Rpass-analysis=kernel-resource-usage", "-mllvm", "-amdgpu-mfma-vgpr-form"] + \
    os.environ.get("FABHIP_EXTRA_FLAGS", "").split()


def spilling_kernels(remarks: str):
    """(kernel, spilled VGPRs) of every kernel hipcc's resource-usage remarks report with a register spill: each
    instantiation in the library can be selected by some shape, and a spilling one turns its hot loop into scratch
    traffic - the build fails instead (shipped k_hmc_step_r4<8, *> spilled 163 - 772 VGPRs)."""
    import re
    out, name = [], None
    for line in remarks.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"remark:\s+VGPRs Spill: (\d+)", line)
        if m and name and int(m.group(1)) > 0:
            out.append((name, int(m.group(1))))
    return out


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build libfabhip.so)")


STAMP = LIB + ".srchash"
TORCH_LIB = os.path.join(HERE, "_fabhip_torch.so")      # TORCH_LIBRARY(fabhip) op layer over the C ABI
TORCH_SRC = "torch_ops.cpp"


def _src_hash():
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC))
    try:                                     # the op layer is compiled against torch's headers
        import torch
        h.update(torch.__version__.encode())
    except ImportError:
        pass
    files.append(os.path.join(os.path.dirname(HERE), "include", "fabhip.h"))
    for f in files:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


_HIPCC_VERSION = None


def _version_digest(v):
    import hashlib
    return hashlib.sha256(v.encode()).hexdigest()[:16]


def _hipcc_version():
    global _HIPCC_VERSION
    if _HIPCC_VERSION is None:
        try:
            _HIPCC_VERSION = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True).stdout.strip()
        except Exception:                    # no compiler here (a box that only loads the prebuilt library): the stamp decides
            _HIPCC_VERSION = ""
    return _HIPCC_VERSION


def is_stale():
    """True when libfabhip.so is missing or was built from different sources (content hash, not mtimes:
    the snapshot that travels to the GPU box does not preserve them)."""
    if not (os.path.exists(LIB) and os.path.exists(TORCH_LIB) and os.path.exists(STAMP)):
        return True
    with open(STAMP) as fh:
        lines = fh.read().splitlines()
    if not lines or lines[0].strip() != _src_hash():
        return True
    # another compiler = other machine code (whose ISA has not been checked): rebuild.  A box without hipcc can only load.
    here = _hipcc_version()
    return bool(here) and len(lines) > 1 and lines[1].strip() != _version_digest(here)


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every .hip source for gfx950 and link libfabhip.so next to this file."""
    if not force and not is_stale():
        return LIB
    hipcc = _hipcc()
    os.makedirs(BUILD, exist_ok=True)
    # several ranks of one node may get here at once (torchrun on a box without a prebuilt library): one builds,
    # the others wait on the lock and find the result
    import fcntl
    lock = open(os.path.join(BUILD, ".lock"), "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if not force and not is_stale():
            return LIB
        return _build_locked(hipcc, verbose)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _torch_flags():
    import torch
    from torch.utils import cpp_extension as ce
    inc = []
    for d in ce.include_paths():
        inc += ["-I", d]
    return ["-O2", "-std=c++17", "-fPIC", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
            f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-Wno-unused-result"] + inc + \
           ["-I", "/opt/rocm/include"]


def _check_isa(obj):
    """The hand-counted waits of the ring / stream kernels are only sound for machine code in which no instruction touches a
    register whose load is still in flight (_isa_check.py): verified on every object right after it is compiled - a library
    whose code fails the check is never produced.  Own process: the objects are checked in parallel."""
    from . import _isa_check
    if os.environ.get("FABHIP_SKIP_ISA_CHECK") == "1":
        return
    if not _isa_check.tools_available():
        print(f"[fab_torch_amd] WARNING: ROCm LLVM tools not found under {_isa_check.LLVM}: the ISA of {os.path.basename(obj)} "
              "was NOT checked for ring registers used before their load has landed", file=sys.stderr)
        return
    r = subprocess.run([sys.executable, "-m", "fab_torch_amd._isa_check", obj], capture_output=True, text=True,
                       cwd=os.path.dirname(HERE))
    if r.returncode != 0:
        raise RuntimeError(f"{os.path.basename(obj)}: generated code touches a register whose load may still be in flight "
                           f"(hand-counted s_waitcnt no longer matches what hipcc emitted):\n{r.stdout[-4000:]}{r.stderr[-2000:]}")


def _build_locked(hipcc, verbose):

    def compile_one(src):
        if src == TORCH_SRC:                 # host C++ against torch's headers (no device code)
            obj = os.path.join(BUILD, "torch_ops.o")
            cmd = [hipcc] + _torch_flags() + ["-c", os.path.join(CSRC, src), "-o", obj]
        else:
            obj = os.path.join(BUILD, src.replace(".hip", ".o"))
            cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-8000:]}")
        spills = spilling_kernels(r.stderr) if src != TORCH_SRC else []
        if spills and os.environ.get("FABHIP_ALLOW_SPILLS") != "1":
            raise RuntimeError(f"{src}: kernels with register spills (remove the instantiation or cut its registers): "
                               + ", ".join(f"{k} ({n} VGPRs)" for k, n in spills))
        if src != TORCH_SRC:
            _check_isa(obj)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES) + 1) as ex:
        objs = list(ex.map(compile_one, SOURCES + [TORCH_SRC]))
    torch_obj = objs.pop()
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr[-8000:]}")
    _link_torch_ops(hipcc, torch_obj)
    with open(STAMP, "w") as fh:
        fh.write(_src_hash() + "\n" + _version_digest(_hipcc_version()) + "\n")
    if verbose:
        print(f"[fab_torch_amd] built {LIB} and {TORCH_LIB}", file=sys.stderr)
    return LIB


def _link_torch_ops(hipcc, obj):
    """fab_torch_amd/_fabhip_torch.so: linked to libfabhip.so next to it (rpath $ORIGIN) and to the torch libraries
    of the running interpreter."""
    import torch
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = [hipcc, "-shared", "-fPIC", obj, "-o", TORCH_LIB, "-L", HERE, "-lfabhip", "-L", tlib, "-lc10", "-ltorch_cpu",
           "-ltorch", "-lc10_hip", "-ltorch_hip", "-Wl,-rpath,$ORIGIN"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"linking the torch op layer failed:\n{r.stderr[-8000:]}")


if __name__ == "__main__":
    build(force="--force" in sys.argv)
