"""Build libfabhip.so (HIP, gfx950) in-tree with hipcc.  No torch headers are involved: the library is
a plain C-ABI shared object (include/fabhip.h) loaded through ctypes."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libfabhip.so")
SOURCES = ["flow_kernels.hip", "ais_kernels.hip", "reduce_resample.hip", "train_kernels.hip", "topk.hip"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-result", "-Wno-pass-failed"] + \
    os.environ.get("FABHIP_EXTRA_FLAGS", "").split()


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build libfabhip.so)")


STAMP = LIB + ".srchash"


def _src_hash():
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC))
    files.append(os.path.join(os.path.dirname(HERE), "include", "fabhip.h"))
    for f in files:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def is_stale():
    """True when libfabhip.so is missing or was built from different sources (content hash, not mtimes:
    the snapshot that travels to the GPU box does not preserve them)."""
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return True
    with open(STAMP) as fh:
        return fh.read().strip() != _src_hash()


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


def _build_locked(hipcc, verbose):

    def compile_one(src):
        obj = os.path.join(BUILD, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-8000:]}")
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr[-8000:]}")
    with open(STAMP, "w") as fh:
        fh.write(_src_hash())
    if verbose:
        print(f"[fab_torch_amd] built {LIB}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
