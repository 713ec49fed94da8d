"""Recompile csrc/torch_ops.cpp and relink _fabhip_torch.so only (development shortcut: `python -m fab_torch_amd._build` is the
real build and rewrites the stamp; this leaves the stamp alone, so the library reads as stale until then)."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fab_torch_amd import _build as b                                                               # noqa: E402

hipcc = b._hipcc()
obj = os.path.join(b.BUILD, "torch_ops.o")
r = subprocess.run([hipcc] + b._torch_flags() + ["-c", os.path.join(b.CSRC, b.TORCH_SRC), "-o", obj], capture_output=True, text=True)
if r.returncode != 0:
    sys.exit(r.stderr[-6000:])
b._link_torch_ops(hipcc, obj)
print("relinked", b.TORCH_LIB)
