"""Shared helpers for the test-suite (fixtures loading, tolerances)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# BASELINE.json north_star: "log-weights and flow log-probs within 1e-4 relative fp32"
RTOL = 1e-4


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def close(a, b, rtol=RTOL, atol=None, atol_scale=1.0):
    """Element-wise |a-b| <= atol + rtol*|b| (numpy.isclose form) with rtol = 1e-4 (north_star) and a SMALL absolute
    floor for entries near zero: atol = 2e-6 * max(1, max|b|) by default, i.e. ~16 fp32 ulps of the largest entry
    (entries of one tensor are sums of terms of that magnitude, so a cancelling entry cannot be more accurate than
    that in fp32 whatever the summation order) — 50x tighter than the former "1e-4 of the largest entry".
    Non-finite patterns must agree exactly."""
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    if a.shape != b.shape:
        return False
    if not np.array_equal(np.isfinite(a), np.isfinite(b)):
        return False
    fin = np.isfinite(b)
    if not fin.any():
        return True
    a64, b64 = a[fin].astype(np.float64), b[fin].astype(np.float64)
    if atol is None:
        atol = 2e-6 * max(1.0, float(np.abs(b64).max())) * atol_scale
    return bool(np.all(np.abs(a64 - b64) <= atol + rtol * np.abs(b64)))


def worst(a, b, rtol=RTOL):
    """(max |a-b| / (atol + rtol |b|)) diagnostic for assertion messages: <= 1 passes `close`."""
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    fin = np.isfinite(b)
    if not fin.any():
        return 0.0
    a64, b64 = a[fin].astype(np.float64), b[fin].astype(np.float64)
    atol = 2e-6 * max(1.0, float(np.abs(b64).max()))
    return float(np.max(np.abs(a64 - b64) / (atol + rtol * np.abs(b64))))


def max_rel_err(a, b):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    fin = np.isfinite(b)
    scale = max(1.0, float(np.abs(b[fin]).max())) if fin.any() else 1.0
    return float(np.abs(a[fin].astype(np.float64) - b[fin].astype(np.float64)).max()) / scale if fin.any() else 0.0


def oracle_flow_from_golden(g):
    """Rebuild the oracle RealNVP whose state_dict is stored in a fixture under 'flow.*'."""
    from oracle import flow as oflow
    sd = {k[len("flow."):]: torch.tensor(v) for k, v in g.items() if k.startswith("flow.")}
    D = sd["q0.loc"].shape[1]
    K = len([k for k in sd if k.endswith(".log_S")])
    W = sd["flows.0.flows.1.param_map.net.0.weight"].shape[0]
    assert W % D == 0
    nf = oflow.make_realnvp(D, K, W // D)
    nf.load_state_dict(sd)
    return nf


def seeded_oracle_flow(dim, n_layers, nodes, seed, std=0.05):
    """The oracle RealNVP with seeded parameters: torch.manual_seed(seed) -> nn.Linear default initialisation and the
    QR-based InvertibleAffine of oracle.flow.make_realnvp, last coupling Linear re-drawn N(0, std^2) with seed + 1 (the
    reference's `init_zeros` makes an untrained flow the identity).  tests/golden/make_golden.py builds the flow of the
    g14 fixtures with this very function, so the fixtures store the seed instead of 4.8 MB of weights."""
    from oracle import flow as oflow
    torch.manual_seed(seed)
    nf = oflow.make_realnvp(dim, n_layers, nodes)
    oflow.randomize_last_layers(nf, std=std, seed=seed + 1)
    return nf


def flow_from_g14(g):
    """The oracle flow of a g14 fixture, rebuilt from its seed and checked against the stored weight probe."""
    nf = seeded_oracle_flow(int(g["D"]), int(g["K"]), int(g["nodes"]), int(g["flow_seed"]), float(g["flow_std"]))
    probe = torch.stack([nf.flows[0].flows[1].param_map.net[2].weight[0, :8].detach(),
                         nf.flows[-2].flows[1].param_map.net[4].weight[1, :8].detach()])
    assert np.array_equal(probe.numpy(), g["flow_probe"]), "seeded flow differs from the one the fixture was made with"
    return nf
