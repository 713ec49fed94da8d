"""Can one fused AIS call (noise generation + fabhip::ais_run) be captured in a HIP graph through torch.cuda.graph?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fab_torch_amd as fa

DEV = "cuda"
for name, D, K, nodes, M, B, op in (("gmm-cfg1", 2, 4, 40, 4, 512, "met"), ("mw32-headline", 32, 10, 10, 8, 1024, "hmc")):
    torch.manual_seed(0)
    flow = fa.RealNVP(D, K, nodes).to(DEV).requires_grad_(False)
    target = fa.GMM(D, 40, 40.0, 1.0).to(DEV) if op == "met" else fa.ManyWellEnergy(D)
    if op == "met":
        tr = fa.Metropolis(M, D, flow.log_prob, target.log_prob, 1, alpha=2.0, p_target=False, max_step_size=5.0,
                           min_step_size=5.0, adjust_step_size=False).to(DEV)
    else:
        tr = fa.HamiltonianMonteCarlo(M, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.1, L=5).to(DEV)
    ais = fa.AnnealedImportanceSampler(flow, target.log_prob, tr, False, 2.0, M)

    def eager():
        return ais.run(B)

    for _ in range(5):
        eager()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        eager()
    torch.cuda.synchronize()
    t_eager = (time.perf_counter() - t0) / 50
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            eager()
    torch.cuda.current_stream().wait_stream(s)
    try:
        with torch.cuda.graph(g):
            out = eager()
        g.replay(); torch.cuda.synchronize()
        lw1 = out[1].clone()
        g.replay(); torch.cuda.synchronize()
        lw2 = out[1].clone()
        t0 = time.perf_counter()
        for _ in range(50):
            g.replay()
        torch.cuda.synchronize()
        t_graph = (time.perf_counter() - t0) / 50
        print(name, "eager %.3f ms, graph replay %.3f ms, fresh noise per replay: %s, finite: %s" %
              (1e3 * t_eager, 1e3 * t_graph, not torch.equal(lw1, lw2), bool(torch.isfinite(lw2).all())))
    except Exception as e:  # noqa: BLE001
        print(name, "capture failed:", type(e).__name__, str(e)[:300])
