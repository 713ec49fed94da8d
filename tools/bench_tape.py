"""Cost of the training tape: flow.log_prob_and_grad (no tape) vs flow.log_prob_with_tape (+ the parameter-gradient
call) on the headline architecture, per minibatch size."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fab_torch_amd as fa
from bench import build_flow_state

DEV = "cuda"
flow = build_flow_state(0).to(DEV).requires_grad_(False)


def timeit(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


for B in (1024, 2048, 4096):
    x, _ = flow.sample_and_log_prob((B,))
    coef = torch.randn(B, device=DEV) / B
    row = {"B": B, "log_prob_and_grad_ms": timeit(lambda: flow.log_prob_and_grad(x)),
           "log_prob_with_tape_ms": timeit(lambda: flow.log_prob_with_tape(x))}
    lq, tape = flow.log_prob_with_tape(x)
    row["param_grad_flat_ms"] = timeit(lambda: flow.param_grad_flat(tape, coef))
    print(json.dumps(row))
