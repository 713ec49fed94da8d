"""Dev tool: stage timeline (s_memtime) of one forward and one reverse layer of the 4-chain-tile HMC kernel."""
import ctypes as C, os, sys
os.environ["FABHIP_TIMELINE"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fab_torch_amd as fa
from fab_torch_amd import _lib
from fab_torch_amd.transition_operators import create_point
dev = torch.device("cuda", 0)
torch.manual_seed(0)
D, B = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
flow = fa.RealNVP(D, 10, 10).to(dev).requires_grad_(False)
target = fa.ManyWellEnergy(D)
hmc = fa.HamiltonianMonteCarlo(8, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.05, L=5).to(dev)
hmc.set_eval_mode(True)
x0, _ = flow.native_sample(torch.randn(B, D, device=dev))
pt = create_point(x0, flow, target, with_grad=True)
for _ in range(5):
    hmc.transition(pt, 4, 0.5)
torch.cuda.synchronize()
buf = (C.c_int64 * 64)()
_lib.check(_lib.load().fabhip_debug_timeline(buf, 64), "timeline")
ts = list(buf)
from fab_torch_amd import _ops
fused = int(_ops.load().get_option(_ops.OPT_R4_STREAM)) >= 2
names = {0: "fwd layer start", 1: "affine done", 2: "W1 (d -> W) done", 3: "W2 (W x W) done", 4: "W3 (W -> 64) done",
         5: "coupling done", 16: "rev layer start", 17: "d-params done", 18: "W3T (64 -> W) done", 19: "W2T (W x W) done",
         20: "W1T (W -> d) done", 21: "add + barrier", 22: "affine^T done"}
if fused:                            # flow_r4f.h: six stages per layer pair
    names = {0: "fwd layer start", 1: "S1 y -> h1, z (K = 32) done", 3: "S2 W2 (W x W) done", 5: "S3 W3 + coupling done",
             16: "rev layer start", 18: "S4 W3T (32 -> W) done", 19: "S5 W2T (W x W) done", 22: "S6 [W1'T ; AT] (W + D -> D) done"}
prev = None
for i in sorted(names, key=lambda i: ts[i]):
    if not ts[i]:
        continue
    print(f"{i:2d} {names[i]:36s}", "" if prev is None or i in (0, 16) else f"+{ts[i] - prev:6d} ticks")
    prev = ts[i]
print("fwd layer:", ts[5] - ts[0], "rev layer:", ts[22] - ts[16], "ticks")
if any(ts[32:64]):                   # -DFAB_R4F_WAVETL (dev build): per-wave stamps inside S1, relative to wave 0's stage start
    lab = ["start", "item 0 landed", "quad 0 issued", "refill 0 issued", "item 1 + quad 1", "refill 1 issued",
           "A tile + refill 2", "barrier passed"]
    t0 = min(ts[32 + 8 * w] for w in range(4))
    print("S1 per wave (ticks since the first wave's start): " + " | ".join(lab))
    for w in range(4):
        print(f"  wave {w}: " + " ".join(f"{ts[32 + 8 * w + k] - t0:6d}" for k in range(8)))
