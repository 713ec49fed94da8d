"""Dev tool: stage timeline (s_memtime) of one forward and one reverse layer of k_spline_logprob (cfg-3 shape)."""
import ctypes as C, os, sys
os.environ["FABHIP_TIMELINE"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fab_torch_amd as fa
from fab_torch_amd import _lib
D = int(os.environ.get("D", 32))
flow = fa.make_wrapped_normflow_spline(D, 12, 256, (), 5.0).to("cuda").requires_grad_(False)
x = torch.randn(2048, D, device="cuda")
if os.environ.get("FAST") == "1":
    fa.fast_mode(True)
for _ in range(5):
    flow.log_prob_and_grad(x)
torch.cuda.synchronize()
buf = (C.c_int64 * 32)()
_lib.check(_lib.load().fabhip_debug_spline_timeline(buf, 32), "timeline")
ts = list(buf)
names = {0: "fwd: layer start", 1: "state save + identity features", 2: "hidden (W0, Wa, Wb GEMMs)", 3: "final GEMM chunks -> P (LDS)",
         4: "spline forward of the tile", 8: "bwd: layer start", 9: "tile loads (state, P, ReLU signs)", 10: "spline reverse -> dP",
         11: "WfT GEMM (K = NFP)", 12: "WbT, WaT GEMMs + masks", 13: "W0T k-split", 14: "partial sums + periodic features"}
prev = None
for i in sorted(names):
    if not ts[i]:
        continue
    print(f"{i:2d} {names[i]:40s}" + ("" if prev is None or i in (0, 8) else f" +{ts[i] - prev:7d} ticks"))
    prev = ts[i]
print("fwd layer", ts[4] - ts[0], "ticks; bwd layer", ts[14] - ts[8], "ticks")
