"""Pin the CPU oracle against fixtures produced by the IMPORTED reference
(tests/golden/make_golden.py).  CPU only; no HIP involved."""
import os

import numpy as np
import pytest
import torch

from oracle import ais as oais
from oracle import flow as oflow
from oracle import numerical as onum
from oracle import targets as otgt

from helpers import load_golden, oracle_flow_from_golden, RTOL, close


def test_g1_beta_schedules():
    g = load_golden("g1_beta.npz")
    for key, ref in g.items():
        spacing, M = key.rsplit("_", 1)
        got = oais.beta_schedule(int(M), spacing)
        assert got.dtype == torch.float64
        np.testing.assert_array_equal(got.numpy(), ref)
    # SURVEY §8(a)-R2 known answers
    np.testing.assert_allclose(oais.beta_schedule(4).numpy(), [0, .2, .4, .6, .8, 1], atol=1e-15)


def test_g2_intermediate_log_prob_and_grad():
    g = load_golden("g2_intermediate.npz")
    pt = oais.Point(*(torch.tensor(g[k]) for k in ("x", "log_q", "log_p", "grad_log_q", "grad_log_p")))
    betas = torch.tensor(g["betas"])
    for ai, alpha in enumerate((2.0, 0.5)):
        for p_target in (False, True):
            for bi, beta in enumerate(betas):
                lp = oais.intermediate_log_prob(pt, beta, alpha, p_target)
                gr = oais.grad_intermediate_log_prob(pt, beta, alpha, p_target)
                assert lp.dtype == torch.float32
                np.testing.assert_array_equal(lp.numpy(), g[f"lp_a{ai}_p{int(p_target)}_b{bi}"])
                np.testing.assert_array_equal(gr.numpy(), g[f"gr_a{ai}_p{int(p_target)}_b{bi}"])


def test_g3_targets():
    g = load_golden("g3_targets.npz")
    for D in (6, 32):
        t = otgt.ManyWell(D)
        x = torch.tensor(g[f"mw{D}_x"])
        np.testing.assert_array_equal(t.log_prob(x).numpy(), g[f"mw{D}_lp"])
        gr = t.grad_log_prob(x).numpy()
        ref = g[f"mw{D}_g"]
        fin = np.isfinite(ref)
        np.testing.assert_allclose(gr[fin], ref[fin], rtol=2e-6, atol=1e-6)
        assert abs(t.log_Z - float(g[f"mw{D}_logZ"])) < 1e-9
    # KATs from SURVEY §8(a)-R10
    assert abs(otgt.ManyWell(32).log_prob(torch.full((1, 32), 1.7)).item() - 134.28639) < 1e-3
    gm = otgt.GMM(2, 40, 40.0, 1.0, seed=0)
    np.testing.assert_array_equal(gm.locs.numpy(), g["gmm_locs"])
    np.testing.assert_allclose(gm.scales.numpy(), g["gmm_scales"], rtol=1e-7)
    lp = gm.log_prob(torch.tensor(g["gmm_x"])).numpy()
    ref = g["gmm_lp"]
    assert np.array_equal(np.isnan(lp), np.isnan(ref))
    assert np.array_equal(np.isneginf(lp), np.isneginf(ref)) and np.isneginf(ref).any()
    fin = np.isfinite(ref)
    np.testing.assert_allclose(lp[fin], ref[fin], rtol=2e-6, atol=1e-5)
    assert abs(lp[0] - (-23.3163)) < 1e-3            # SURVEY §8(a)-R11 KAT


def test_g4_ess_logz():
    g = load_golden("g4_ess.npz")
    for i in range(5):
        lw = torch.tensor(g[f"lw{i}"])
        np.testing.assert_allclose(onum.effective_sample_size(lw).item(), g[f"ess{i}"], rtol=1e-6)
        np.testing.assert_allclose(onum.log_Z(lw, lw.shape[0]).item(), g[f"logZ{i}"], rtol=1e-6)
    assert abs(onum.effective_sample_size(torch.tensor([0., 1., 2., 3.])).item() - 0.5215276) < 1e-6


def test_g5_multinomial_bit_exact_vs_reference_resample():
    g = load_golden("g5_multinomial.npz")
    for N in (64, 1024, 4096, 16384):
        probs = onum.categorical_probs(torch.tensor(g[f"lw_{N}"])).numpy()
        np.testing.assert_array_equal(probs, g[f"probs_{N}"])
        idx = onum.multinomial_torch_compat(g[f"probs_{N}"], g[f"u_{N}"])
        np.testing.assert_array_equal(idx, g[f"idx_{N}"])


def test_fixed_point_resamplers_properties():
    rng = np.random.default_rng(0)
    lw = (rng.standard_normal(5000) * 3).astype(np.float32)
    lw[7] = -np.inf
    lw[9] = np.nan
    u = rng.random(5000)
    idx = onum.multinomial_fixed(lw, u)
    assert idx.min() >= 0 and idx.max() < 5000 and 7 not in idx and 9 not in idx
    # agrees with the torch-compatible path except (rarely) within rounding of a bucket edge
    p = onum.categorical_probs(torch.tensor(np.where(np.isnan(lw), -np.inf, lw))).numpy()
    idx_t = onum.multinomial_torch_compat(p, u)
    assert (idx != idx_t).mean() < 2e-3
    s = onum.systematic_fixed(lw, 0.37)
    assert np.all(np.diff(s) >= 0) and 7 not in s
    counts = np.bincount(s, minlength=5000)
    expect = p.astype(np.float64) * 5000
    assert np.all(np.abs(counts - expect) <= 1.0 + 1e-3 * expect)      # systematic: |n_i - N p_i| < 1


def test_oracle_flow_self_consistency():
    torch.manual_seed(3)
    nf = oflow.make_realnvp(6, 3, 5)
    oflow.randomize_last_layers(nf, 0.2, seed=4)
    eps = torch.randn(32, 6)
    with torch.no_grad():
        x, lq = nf.sample_eps(eps)
        lq2 = nf.log_prob(x)
    assert close(lq2, lq, 1e-5)
    # log-det against the dense Jacobian of the inverse map on one point
    def inv(xx):
        z = xx[None]
        for i in range(len(nf.flows) - 1, -1, -1):
            z, _ = nf.flows[i].inverse(z)
        return z[0]
    J = torch.autograd.functional.jacobian(inv, x[0])
    z0 = inv(x[0])[None]
    want = torch.linalg.slogdet(J)[1] + nf.q0.log_prob(z0)[0]
    assert abs(want.item() - lq2[0].item()) < 1e-4


@pytest.mark.parametrize("tag", ["d6", "d32", "d6_outer2"])
def test_g6_hmc_transition_matches_reference(tag):
    g = load_golden(f"g6_hmc_{tag}.npz")
    nf = oracle_flow_from_golden(g)
    D = g["in_x"].shape[1]
    target = otgt.ManyWell(D)
    hmc = oais.HMC(int(g["M"]), D, nf.log_prob, target.log_prob, alpha=float(g["alpha"]),
                   p_target=bool(g["p_target"]), n_outer=int(g["n_outer"]), L=int(g["L"]))
    hmc.epsilons = torch.tensor(g["in_epsilons"])
    hmc.common_epsilon = torch.tensor(g["in_common_epsilon"])
    pt = oais.Point(*(torch.tensor(g[k]) for k in ("in_x", "in_log_q", "in_log_p", "in_gq", "in_gp")))
    res = hmc.transition(pt, int(g["i"]), torch.tensor(g["beta"]), torch.tensor(g["noise_p"]),
                         torch.tensor(g["noise_e"]))
    assert close(res.x, g["out_x"], 1e-5) and close(res.log_q, g["out_log_q"], 1e-5)
    assert close(res.log_p, g["out_log_p"], 1e-5) and close(res.grad_log_q, g["out_gq"], 1e-4)
    assert close(res.grad_log_p, g["out_gp"], 1e-5)
    np.testing.assert_array_equal(hmc.epsilons.numpy(), g["out_epsilons"])
    np.testing.assert_array_equal(hmc.common_epsilon.numpy(), g["out_common_epsilon"])
    changed = (g["out_x"] != g["in_x"]).any(1)
    assert 0 < changed.sum() < len(changed)          # the fixture exercises accept AND reject


def test_g7_metropolis_transition_matches_reference():
    g = load_golden("g7_metropolis.npz")
    nf = oracle_flow_from_golden(g)
    target = otgt.GMM(2, 40, 40.0, 1.0, seed=0)
    np.testing.assert_array_equal(target.locs.numpy(), g["gmm_locs"])
    met = oais.Metropolis(int(g["M"]), 2, nf.log_prob, target.log_prob, int(g["n_updates"]),
                          alpha=float(g["alpha"]), p_target=False, max_step_size=5.0, min_step_size=1.0)
    np.testing.assert_array_equal(met.noise_scalings.numpy(), g["in_noise_scalings"])
    pt = oais.Point(*(torch.tensor(g[k]) for k in ("in_x", "in_log_q", "in_log_p")))
    res = met.transition(pt, int(g["i"]), torch.tensor(g["beta"]), torch.tensor(g["noise_x"]),
                         torch.tensor(g["noise_u"]))
    assert close(res.x, g["out_x"], 1e-6) and close(res.log_q, g["out_log_q"], 1e-5)
    assert close(res.log_p, g["out_log_p"], 1e-5)
    np.testing.assert_array_equal(met.noise_scalings.numpy(), g["out_noise_scalings"])
    changed = (g["out_x"] != g["in_x"]).any(1)
    assert 0 < changed.sum() < len(changed)


@pytest.mark.parametrize("tag", ["mw6_hmc_m4", "mw6_hmc_m8geo_ptarget", "mw32_hmc_m8"])
def test_g8_full_ais_hmc_matches_reference(tag):
    g = load_golden(f"g8_ais_{tag}.npz")
    nf = oracle_flow_from_golden(g)
    D = g["eps0"].shape[1]
    target = otgt.ManyWell(D)
    M = int(g["M"])
    hmc = oais.HMC(M, D, nf.log_prob, target.log_prob, alpha=float(g["alpha"]),
                   p_target=bool(g["p_target"]), L=int(g["L"]))
    hmc.epsilons = torch.tensor(g["in_epsilons"])
    hmc.common_epsilon = torch.tensor(g["in_common_epsilon"])
    ais = oais.AIS(lambda e: _noq(nf, e), nf.log_prob, target.log_prob, hmc, bool(g["p_target"]),
                   float(g["alpha"]), M, str(g["spacing"]))
    np.testing.assert_array_equal(ais.B_space.numpy(), g["B_space"])
    pt, log_w, info = ais.sample_and_log_weights(torch.tensor(g["eps0"]), torch.tensor(g["noise_p"]),
                                                 torch.tensor(g["noise_e"]))
    assert close(pt.x, g["out_x"], 1e-4) and close(log_w, g["log_w"], 1e-4)
    assert close(pt.log_q, g["out_log_q"], 1e-4) and close(pt.log_p, g["out_log_p"], 1e-4)
    np.testing.assert_array_equal(hmc.epsilons.numpy(), g["out_epsilons"])
    np.testing.assert_array_equal(hmc.common_epsilon.numpy(), g["out_common_epsilon"])
    assert abs(info.ess_ais - float(g["ess_ais"])) <= 0.01 * float(g["ess_ais"])
    assert abs(info.ess_base - float(g["ess_base"])) <= 0.01 * float(g["ess_base"])
    assert abs(info.log_Z - float(g["log_Z"])) <= RTOL * abs(float(g["log_Z"])) + 1e-4


@pytest.mark.parametrize("fixture", ["g14_ais_headline.npz", "g15_ais_headline_mild.npz", "g16_ais_headline_rejecting.npz"])
def test_g14_headline_architecture_ais_matches_reference(fixture):
    """(g15: the same call in the mild regime the zero-waiver GPU test uses; g16: step size 0.26 with the tuning frozen, 64 chains
    selected from a pool of 512 as well-conditioned in every transition - 59 % of their proposals rejected.)  The reference's AIS call at the HEADLINE flow architecture (10 x (16-320-320-32) + InvertibleAffine, D = 32, M = 8,
    L = 5; fab/experiments/config/many_well.yaml:7-10,25-29) with the weights rebuilt from the fixture's seed: the oracle
    replays it - every transition's snapshot, the adapted step sizes bit for bit."""
    from helpers import flow_from_g14
    g = load_golden(fixture)
    nf = flow_from_g14(g)
    D, M = int(g["D"]), int(g["M"])
    target = otgt.ManyWell(D)
    hmc = oais.HMC(M, D, nf.log_prob, target.log_prob, alpha=float(g["alpha"]), p_target=False, L=int(g["L"]),
                   eval_mode="g16" in fixture)
    hmc.epsilons = torch.tensor(g["in_epsilons"])
    hmc.common_epsilon = torch.tensor(g["in_common_epsilon"])
    ais = oais.AIS(lambda e: _noq(nf, e), nf.log_prob, target.log_prob, hmc, False, float(g["alpha"]), M)
    pt, log_w, info = ais.sample_and_log_weights(torch.tensor(g["eps0"]), torch.tensor(g["noise_p"]),
                                                 torch.tensor(g["noise_e"]), keep_snapshots=True)
    assert len(ais.snapshots) == M + 1 == g["snap_x"].shape[0]
    for j, (sp, _) in enumerate(ais.snapshots):
        assert close(sp.x, g["snap_x"][j], 1e-4), f"snapshot {j}"
        assert close(sp.log_q, g["snap_log_q"][j], 1e-4) and close(sp.log_p, g["snap_log_p"][j], 1e-4)
    assert close(pt.x, g["out_x"], 1e-4) and close(log_w, g["log_w"], 1e-4)
    np.testing.assert_array_equal(hmc.epsilons.numpy(), g["out_epsilons"])
    np.testing.assert_array_equal(hmc.common_epsilon.numpy(), g["out_common_epsilon"])
    assert abs(info.ess_ais - float(g["ess_ais"])) <= 0.01 * float(g["ess_ais"])
    assert abs(info.log_Z - float(g["log_Z"])) <= RTOL * abs(float(g["log_Z"])) + 1e-4


def test_g17_multi_call_step_size_trajectory_matches_reference():
    """g17: six reference calls in a row from `init_step_size: 1.0`, tuning on, at the headline architecture -
    every proposal rejected, every transition divides its step sizes, each call starts from the previous call's state.  The oracle
    (ONE sampler, its own state carried from call to call) reproduces every call's outputs and the step sizes bit for bit."""
    from helpers import flow_from_g14
    g = load_golden("g17_step_size_trajectory.npz")
    nf = flow_from_g14(g)
    D, M = int(g["D"]), int(g["M"])
    target = otgt.ManyWell(D)
    hmc = oais.HMC(M, D, nf.log_prob, target.log_prob, alpha=float(g["alpha"]), p_target=False, epsilon=1.0, L=int(g["L"]))
    ais = oais.AIS(lambda e: _noq(nf, e), nf.log_prob, target.log_prob, hmc, False, float(g["alpha"]), M)
    for c in range(g["eps0"].shape[0]):
        np.testing.assert_array_equal(hmc.epsilons.numpy(), g["in_epsilons"][c])
        np.testing.assert_array_equal(hmc.common_epsilon.numpy(), g["in_common_epsilon"][c])
        pt, log_w, info = ais.sample_and_log_weights(torch.tensor(g["eps0"][c]), torch.tensor(g["noise_p"][c]),
                                                     torch.tensor(g["noise_e"][c]))
        assert close(pt.x, g["out_x"][c], 1e-5) and close(log_w, g["log_w"][c], 1e-4)
        assert close(pt.log_q, g["out_log_q"][c], 1e-4) and close(pt.log_p, g["out_log_p"][c], 1e-4)
        np.testing.assert_array_equal(hmc.epsilons.numpy(), g["out_epsilons"][c])
        np.testing.assert_array_equal(hmc.common_epsilon.numpy(), g["out_common_epsilon"][c])
    assert not g["moved"].any()


def _noq(nf, e):
    with torch.no_grad():
        return nf.sample_eps(e)


def test_g8_full_ais_metropolis_matches_reference():
    g = load_golden("g8_ais_gmm_metropolis.npz")
    nf = oracle_flow_from_golden(g)
    target = otgt.GMM(2, 40, 40.0, 1.0, seed=0)
    M = int(g["M"])
    met = oais.Metropolis(M, 2, nf.log_prob, target.log_prob, int(g["n_updates"]), alpha=float(g["alpha"]),
                          p_target=False, max_step_size=5.0, min_step_size=2.0)
    ais = oais.AIS(lambda e: _noq(nf, e), nf.log_prob, target.log_prob, met, False, float(g["alpha"]), M)
    pt, log_w, info = ais.sample_and_log_weights(torch.tensor(g["eps0"]), torch.tensor(g["noise_x"]),
                                                 torch.tensor(g["noise_u"]))
    assert close(pt.x, g["out_x"], 1e-5) and close(log_w, g["log_w"], 1e-4)
    np.testing.assert_array_equal(met.noise_scalings.numpy(), g["out_noise_scalings"])
    assert abs(info.ess_ais - float(g["ess_ais"])) <= 0.01 * float(g["ess_ais"])
    assert abs(info.log_Z - float(g["log_Z"])) <= 1e-4 * abs(float(g["log_Z"])) + 1e-4


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_oracle_trainer_iteration_vs_reference_traces(seed):
    """R14: oracle/train.py replays the reference PrioritisedBufferTrainer traces (g12: 5 iterations, B = 64, D = 6,
    captured noise): sampled indices bit-exact, loss / grad_norm / buffer log_w / log_q_old after the adjust and the
    final parameters to 1e-5 (same eager CPU ops)."""
    from oracle import train as otrain
    g = load_golden(f"g12_trainer_seed{seed}.npz")
    D, M, L, B = int(g["D"]), int(g["M"]), int(g["L"]), int(g["B"])
    alpha, n_iter, n_batches = float(g["alpha"]), int(g["n_iter"]), int(g["n_batches"])
    nf = oracle_flow_from_golden(g)
    target = otgt.ManyWell(D)
    hmc = oais.HMC(M, D, nf.log_prob, target.log_prob, alpha=alpha, p_target=False, epsilon=0.2, L=L)
    assert np.array_equal(hmc.epsilons.numpy(), g["in_epsilons"])
    ais = oais.AIS(lambda e: tuple(t.detach() for t in nf.sample_eps(e)), nf.log_prob, target.log_prob, hmc,
                   False, alpha, M)
    buf = otrain.Buffer(D, int(g["buf_len"]), int(g["buf_min"]))
    n_init = int(g["n_init_calls"])
    T = torch.tensor
    for c in range(n_init):                                   # initial_sampler fills the buffer (setup_run.py:119-122)
        pt, lw, _ = ais.sample_and_log_weights(T(g[f"call{c}_eps0"]), T(g[f"call{c}_noise_p"]), T(g[f"call{c}_noise_e"]))
        assert close(lw, g[f"call{c}_log_w"], 1e-5)
        buf.add(pt.x.detach(), lw.detach(), pt.log_q.detach())
    assert buf.can_sample
    params = list(nf.parameters())
    opt = torch.optim.Adam(params, lr=float(g["lr"]))
    for it in range(n_iter):
        c = n_init + it
        noise = dict(eps0=T(g[f"call{c}_eps0"]), noise_p=T(g[f"call{c}_noise_p"]), noise_e=T(g[f"call{c}_noise_e"]),
                     gumbel=T(g[f"it{it}_gumbel"]), perm=T(g[f"it{it}_perm"]))
        out = otrain.train_iteration(ais, nf.log_prob, params, opt, buf, alpha, B, n_batches, noise,
                                     float(g["max_gradient_norm"]), float(g["w_adjust_max_clip"]))
        assert np.array_equal(out["indices"].numpy(), g[f"it{it}_indices"]), f"iteration {it}: sampled indices"
        for key in ("loss", "grad_norm", "ess_ais", "log_Z", "w_adjust_mean", "log_q_x_mean"):
            ref = float(g[f"it{it}_{key}"])
            assert abs(out[key] - ref) <= 1e-5 * max(1.0, abs(ref)), (it, key, out[key], ref)
        assert close(buf.log_w, g[f"it{it}_buf_log_w"], 1e-5) and close(buf.log_q_old, g[f"it{it}_buf_log_q_old"], 1e-5)
    assert np.array_equal(hmc.epsilons.numpy(), g["out_epsilons"])
    for k, v in nf.state_dict().items():
        assert close(v, g["final." + k], 1e-5), k
