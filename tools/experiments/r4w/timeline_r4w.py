"""Dev tool: stage timeline (s_memtime) of one forward and one reverse layer of the second-generation 4 / 8-chain tiles
(flow_r4w.h) through fabhip_flow_log_prob.  Usage: python tools/timeline_r4w.py [chains] [4|8]"""
import ctypes as C, os, sys
os.environ["FABHIP_TIMELINE"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import fab_torch_amd as fa
from fab_torch_amd import _lib, _ops
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
shape = int(sys.argv[2]) if len(sys.argv) > 2 else 4
flow = bench.build_flow_state(0).to(dev).requires_grad_(False)
x = torch.randn(B, 32, device=dev)
with _ops.option(_ops.OPT_TILE_SHAPE, shape):
    for _ in range(5):
        flow.log_prob_and_grad(x)
torch.cuda.synchronize()
buf = (C.c_int64 * 64)()
_lib.check(_lib.load().fabhip_debug_timeline(buf, 64), "timeline")
ts = list(buf)
names = {0: "fwd layer start", 6: "(body top)", 7: "deferred requests issued", 8: "affine: MFMAs of k-quads s = 0 issued",
         9: "affine: MFMAs s = 1 issued", 10: "affine: epilogue stored", 1: "affine done", 2: "W1 slice done", 3: "W2 (W x W) + slice reduce done", 4: "W3 partials done",
         5: "coupling done", 16: "rev layer start", 17: "d-params done", 18: "W3T slice done", 19: "W2T (W x W) + reduce done",
         20: "W1T + add done", 22: "affine^T done"}
prev = None
for i in (0, 6, 7, 8, 9, 10, 1, 2, 3, 4, 5, 16, 17, 18, 19, 20, 22):
    if not ts[i]:
        continue
    print(f"{i:2d} {names[i]:34s}", "" if prev is None or i in (0, 16) else f"+{ts[i] - prev:6d} ticks")
    prev = ts[i]
print(f"B={B} tile shape {shape}: fwd layer:", ts[5] - ts[0], "rev layer:", ts[22] - ts[16], "ticks")
