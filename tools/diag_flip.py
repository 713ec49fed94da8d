"""dev diagnostic: which chains of a workload slice differ from the oracle in a teacher-forced transition, and why."""
import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fab_torch_amd as fa
import test_gpu_workloads as tw

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3_manywell32_k12_m12_2048"
kw = dict(tw.WORKLOADS.get(name, dict(D=32, K=12, nodes=10, M=12, B=16384, eps=0.12)))
seed = len(name) if name in tw.WORKLOADS else 4
if len(sys.argv) > 2:
    kw["eps"] = float(sys.argv[2])
w = tw.Workload(name, seed=seed, **kw)
b = tw.SLICE
DEV = "cuda"
e0, na, nb = w.eps0[:b], w.noise_a[:, :, :b].contiguous(), w.noise_b[:, :, :b].contiguous()
opt, olw, oinfo = w.oa.sample_and_log_weights(e0, na, nb, keep_snapshots=True)
snaps, margins = w.oa.snapshots, w.oa.margins
for j in range(1, w.M + 1):
    p_in, lw_in = snaps[j - 1]
    p_ref, lw_ref = snaps[j]
    pt = fa.Point(p_in.x.clone().to(DEV), p_in.log_q.clone().to(DEV), p_in.log_p.clone().to(DEV),
                  p_in.grad_log_q.clone().to(DEV), p_in.grad_log_p.clone().to(DEV))
    lw = lw_in.clone().to(DEV)
    w.hop.transition(pt, j, float(w.ais.B_space[j]), log_w=lw, beta_next=float(w.ais.B_space[j + 1]),
                     noise_p=na[j - 1].to(DEV), noise_e=nb[j - 1].to(DEV))
    err = (pt.x.cpu() - p_ref.x).abs().max(1).values / max(1.0, float(p_ref.x.abs().max()))
    lwerr = (lw.cpu() - lw_ref).abs() / lw_ref.abs().clamp(min=1)
    bad = (err > 1e-4) | (lwerr > 1e-4)
    for r in bad.nonzero().flatten().tolist():
        print(f"tr {j} row {r}: x err {float(err[r]):.2e} lw err {float(lwerr[r]):.2e} margin {float(margins[j][r]):.4g} "
              f"|x_in|max {float(p_in.x[r].abs().max()):.3g} |x_ref|max {float(p_ref.x[r].abs().max()):.3g} "
              f"lq_in {float(p_in.log_q[r]):.5g} lp_in {float(p_in.log_p[r]):.5g} lw_ref {float(lw_ref[r]):.6g} lw_hip {float(lw[r]):.6g} "
              f"|gq_in|max {float(p_in.grad_log_q[r].abs().max()):.3g} |gp_in|max {float(p_in.grad_log_p[r].abs().max()):.3g}")
    print(f"tr {j}: max x err {float(err.max()):.2e}  max lw err {float(lwerr.max()):.2e}  |x|max {float(p_ref.x.abs().max()):.3g}")
