"""Dev tool: stage timeline (s_memtime) of one forward layer and one backward layer of the flow kernel."""
import ctypes as C, os, sys
os.environ["FABHIP_TIMELINE"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fab_torch_amd as fa
from fab_torch_amd import _lib
dev = torch.device("cuda", 0)
nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 10
torch.manual_seed(0)
flow = fa.RealNVP(32, 10, nodes).to(dev).requires_grad_(False)
x = torch.randn(1024, 32, device=dev)
if os.environ.get("FAST") == "1":
    fa.fast_mode(True)
for _ in range(5):
    flow.native_log_prob(x, with_grad=True)
torch.cuda.synchronize()
buf = (C.c_int64 * 64)()
_lib.check(_lib.load().fabhip_debug_timeline(buf, 64), "timeline")
ts = list(buf)
names = {0: "fwd: layer start", 1: "affine GEMM done", 2: "barrier", 3: "GEMM1+epilogue done", 4: "barrier",
         5: "GEMM2+epilogue done", 6: "barrier", 7: "GEMM3 k-split done", 8: "barrier", 9: "(mlp return)",
         10: "coupling elementwise done", 11: "barrier (layer end)",
         16: "bwd: layer start", 17: "d-params elementwise done", 18: "barrier", 19: "dH2 GEMM (K=32)+mask done",
         20: "barrier", 21: "dH1 GEMM (K=W)+mask done", 22: "barrier", 23: "dz1 k-split done", 24: "barrier",
         25: "partial sums done", 26: "barrier", 27: "affine^T GEMM done", 28: "barrier (layer end)"}
prev = None
for i in sorted(names):
    if ts[i] == 0:
        continue
    d = "" if prev is None or i in (0, 16) else f"+{(ts[i] - prev):7d} ticks"
    print(f"{i:2d} {names[i]:32s} {d}")
    prev = ts[i]
print("fwd layer total ticks:", ts[11] - ts[0], " bwd layer total ticks:", ts[28] - ts[16], "(s_memtime ticks = shader cycles on gfx950, ~2.4 GHz: MI355X_MICROARCH.md)")
if ts[40] and ts[41] and ts[42]:
    print(f"inside the forward W x W GEMM: ring prologue issued at +{ts[40] - ts[4]} after the stage start, main loop "
          f"{ts[41] - ts[40]} ticks (MFMA floor 12800 at W=320), post-loop requests {ts[42] - ts[41]}, "
          f"epilogue {ts[5] - ts[42]}")
