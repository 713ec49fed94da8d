"""Dev tool: stage timeline (s_memtime) of one forward and one reverse layer of the 8-chain-tile HMC kernel (flow_r8.h)."""
import ctypes as C, os, sys
os.environ["FABHIP_TIMELINE"] = "1"
os.environ["FABHIP_TILE"] = "8"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fab_torch_amd as fa
from fab_torch_amd import _lib
from fab_torch_amd.transition_operators import create_point
dev = torch.device("cuda", 0)
torch.manual_seed(0)
D, B = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 2048
nodes = int(sys.argv[2]) if len(sys.argv) > 2 else 10
flow = fa.RealNVP(D, 10, nodes).to(dev).requires_grad_(False)
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
names = {0: "fwd layer start", 1: "affine (every wave, 4 dense tiles)", 2: "W1 (d -> W, 4 tiles)", 3: "W2 (W x W, 16 G tiles)",
         4: "W3 (K split, 2 G dense tiles) + next ring", 5: "coupling", 16: "rev layer start", 17: "W3T (8 tiles)",
         18: "W2T (16 G tiles)", 19: "W1T (K split, G dense tiles)", 20: "add", 21: "affine^T (4 dense tiles) + next ring"}
if fused:       # flow_r8.h FUSED: y -> z and y -> h1 in one stage; dh1 W1'^T + g_z W'^T in one K-split stage + its consumer
    names.update({2: "S1: z = y W' + ac | h1 = relu(y W1' + b1')", 19: "S6: dh1 W1'^T + g_z W'^T (K split)",
                  21: "S6 consumer: 8 partials + next cotangents"})
prev = None
for i in sorted(names):
    if not ts[i]:
        continue
    print(f"{i:2d} {names[i]:36s}", "" if prev is None or i in (0, 16) else f"+{ts[i] - prev:6d} ticks")
    prev = ts[i]
print("fwd layer:", ts[5] - ts[0], "rev layer:", ts[21] - ts[16], "ticks")
