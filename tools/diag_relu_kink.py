"""Diagnostic: when a parameter gradient of the HIP path differs from the oracle by more than rounding, is it a
ReLU kink (a hidden pre-activation within rounding distance of 0 whose sign differs between two fp32 summation
orders)?  Finds the samples whose fp64 oracle pre-activations come within `thr` of zero, zeroes their coefficient
and repeats the comparison.  Usage (GPU box): python tools/diag_relu_kink.py D K nodes B"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T  # noqa: E402


def main():
    D, K, nodes, B = (int(a) for a in sys.argv[1:5])
    thr = 2e-5
    nf = T.seeded_flow(D, K, nodes, 300 + D + K)
    with torch.no_grad():
        g = torch.Generator().manual_seed(9)
        nf.q0.loc.add_(0.3 * torch.randn(nf.q0.loc.shape, generator=g))
        nf.q0.log_scale.add_(0.2 * torch.randn(nf.q0.log_scale.shape, generator=g))
    hf = T.hip_flow_from_oracle(nf).requires_grad_(True)
    torch.manual_seed(11)
    with torch.no_grad():
        x = nf.sample_eps(torch.randn(B, D))[0] + 0.1 * torch.randn(B, D)
    coef = torch.randn(B) / B
    near = T.relu_kink_samples(nf, x, nodes * D, thr)
    print(f"samples with a hidden pre-activation within {thr:g} of zero: {int(near.sum())} of {B}")
    names = [n for n, _ in nf.named_parameters()]
    hp = dict(hf._nf_model.named_parameters())
    for label, c in (("all samples", coef), ("kink samples removed", torch.where(near, torch.zeros_like(coef), coef))):
        _, g_o, _ = T._param_grads(nf, [p for _, p in nf.named_parameters()], x, c)
        _, g_h, _ = T._param_grads(hf, [hp[n] for n in names], x.to("cuda"), c.to("cuda"))
        worst = max((float((a.cpu() - b).abs().max()) / max(float(b.abs().max()), 1e-6), n)
                    for n, a, b in zip(names, g_h, g_o))
        print(f"{label}: worst relative error {worst[0]:.2e} ({worst[1]})")


if __name__ == "__main__":
    main()
