"""Self-consistency of the oracle RQ-spline coupling flow (oracle/spline.py).  normflows is absent from /root/reference
and the reference holds no vector at this boundary, so these properties are what pins the restatement (parity with the
reference: UNPINNED, said so in the oracle header and in DESIGN.md)."""
import math

import pytest
import torch

from oracle import spline as osp


def _flow(D=8, layers=4, hidden=16, circ=(1, 4, 6), seed=0, dtype=torch.float64):
    torch.set_default_dtype(dtype)
    try:
        tb = torch.full((D,), 5.0)
        tb[list(circ)] = torch.tensor([math.pi / 0.7, math.pi / 1.3, math.pi])[:len(circ)]
        torch.manual_seed(seed)
        f = osp.make_circular_coupled_flow(D, layers, hidden, circ, tb, num_bins=8, seed=seed)
        osp.randomize(f, 0.4, seed + 1)
    finally:
        torch.set_default_dtype(torch.float32)
    return f


def _noise(B, D, seed, dtype=torch.float64):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(B, D, generator=g, dtype=dtype), torch.randn(B, D, generator=g, dtype=dtype)


def test_identity_initialised_flow_is_the_identity_map():
    torch.set_default_dtype(torch.float64)
    try:
        f = osp.make_circular_coupled_flow(6, 4, 8, (2,), torch.tensor([5., 5., 3.0, 5., 5., 5.]), circ_shift=None)
        u, e = _noise(32, 6, 0)
        z, lq0 = f.q0.forward_eps(u, e)
        x, lq = f.sample_eps(u, e)
        assert torch.allclose(x, z, atol=1e-12) and torch.allclose(lq, lq0, atol=1e-10)
    finally:
        torch.set_default_dtype(torch.float32)


def test_log_prob_of_samples_equals_returned_log_q_and_round_trip():
    f = _flow()
    u, e = _noise(200, 8, 1)
    x, lq = f.sample_eps(u, e)
    assert torch.allclose(f.log_prob(x), lq, atol=1e-9)
    # layer-wise inverse o forward = identity (circular coordinates modulo the period)
    z, _ = f.q0.forward_eps(u, e)
    for layer in f.flows:
        y, ld_f = layer(z)
        zb, ld_i = layer.inverse(y)
        if isinstance(layer, osp.PeriodicWrap):
            continue
        assert torch.allclose(zb, z, atol=1e-9), type(layer).__name__
        assert torch.allclose(ld_f + ld_i, torch.zeros_like(ld_f), atol=1e-9)
        z = y


def test_log_det_matches_the_autograd_jacobian():
    f = _flow(D=6, layers=2, hidden=8, circ=(0, 3))
    u, e = _noise(5, 6, 2)
    x, _ = f.sample_eps(u, e)
    for layer in f.flows:
        if not isinstance(layer, osp.CircularCoupledRationalQuadraticSpline):
            continue
        for b in range(x.shape[0]):
            xb = x[b:b + 1].clone()
            J = torch.autograd.functional.jacobian(lambda t: layer.inverse(t)[0], xb)[0, :, 0, :]
            _, ld = layer.inverse(xb)
            assert abs(float(torch.linalg.slogdet(J)[1]) - float(ld)) < 1e-8


def test_density_is_periodic_in_the_circular_coordinates_and_gradient_matches_finite_differences():
    f = _flow()
    u, e = _noise(16, 8, 3)
    x, _ = f.sample_eps(u, e)
    wrap = f.flows[-1]
    assert isinstance(wrap, osp.PeriodicWrap)
    shift = torch.zeros_like(x)
    shift[:, wrap.ind] = 2 * wrap.bound
    assert torch.allclose(f.log_prob(x + shift), f.log_prob(x), atol=1e-8)
    xg = x.clone().requires_grad_(True)
    (g,) = torch.autograd.grad(f.log_prob(xg).sum(), xg)
    h = 1e-6
    for j in (0, 1, 5):
        d = torch.zeros_like(x); d[:, j] = h
        fd = (f.log_prob(x + d) - f.log_prob(x - d)) / (2 * h)
        assert torch.allclose(fd, g[:, j], atol=1e-4, rtol=1e-4)


def test_state_dict_keys_follow_normflows_naming():
    f = _flow(dtype=torch.float32)
    keys = set(f.state_dict())
    for k in ("flows.0.prqct.transform_net.initial_layer.weight", "flows.0.prqct.transform_net.blocks.0.linear_layers.1.bias",
              "flows.0.prqct.transform_net.final_layer.weight", "flows.0.prqct.unconditional_transform.unnormalized_widths",
              "flows.0.prqct.unconditional_transform.unnormalized_derivatives"):
        assert k in keys, k
    assert any(k.endswith("transform_net.preprocessing.weights") for k in keys)
