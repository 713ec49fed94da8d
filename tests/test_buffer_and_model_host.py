"""Host-side "next" rows: prioritised replay buffer against the reference's own buffer
(golden fixture g9) + sampling properties; FABModel glue contracts.  CPU only (the buffer is device-agnostic)."""
import numpy as np
import pytest
import torch

from helpers import load_golden

import fab_torch_amd as fa


def test_buffer_add_wraparound_and_adjust_match_reference():
    g = load_golden("g9_buffer.npz")
    batches = [(torch.tensor(g[f"b{k}_x"]), torch.tensor(g[f"b{k}_lw"]), torch.tensor(g[f"b{k}_lq"])) for k in range(4)]
    it = iter(batches)
    buf = fa.PrioritisedReplayBuffer(int(g["dim"]), int(g["max_length"]), int(g["min_sample_length"]), lambda: next(it))
    buf.add(*batches[2]); buf.add(*batches[3])
    buf.adjust(torch.tensor(g["adj"]), torch.tensor(g["lq"]), torch.tensor(g["idx"]))
    np.testing.assert_array_equal(buf.buffer.x.numpy(), g["x"])
    np.testing.assert_array_equal(buf.buffer.log_w.numpy(), g["log_w"])
    np.testing.assert_array_equal(buf.buffer.log_q_old.numpy(), g["log_q_old"])
    assert buf.current_index == int(g["current_index"]) and buf.is_full == bool(g["is_full"])
    assert buf.can_sample == bool(g["can_sample"])


def test_buffer_sampling_needs_the_gpu():
    """Gumbel-top-k sampling is fabhip_topk: a host-resident buffer cannot be sampled (no torch.topk fallback).
    The sampling properties themselves are checked on the GPU (tests/test_gpu_workloads.py)."""
    from fab_torch_amd._lib import FabhipError
    data = (torch.randn(100, 2), torch.randn(100), torch.randn(100))
    buf = fa.PrioritisedReplayBuffer(2, 101, 99, lambda: data)
    with pytest.raises(FabhipError, match="no CPU path"):
        buf.sample(10)
    with pytest.raises(Exception):
        fa.PrioritisedReplayBuffer(2, 10, 5, lambda: data, fill_buffer_during_init=False).sample(2)


def test_fabmodel_contracts_without_gpu():
    flow = fa.RealNVP(6, 2, 5)
    target = fa.ManyWellEnergy(6, use_gpu=False)
    with pytest.raises(Exception, match="transition operator must be provided"):
        fa.FABModel(flow, target, 4)
    hmc = fa.HamiltonianMonteCarlo(4, 6, flow.log_prob, target.log_prob, alpha=2.0, p_target=True)
    model = fa.FABModel(flow, target, 4, alpha=2.0, transition_operator=hmc, loss_type="fab_alpha_div")
    ais = model.annealed_importance_sampler
    assert ais.p_target is False and hmc.p_target is False and ais.n_intermediate_distributions == 4
    model.set_ais_target(min_is_target=False)
    assert ais.p_target is True and hmc.p_target is True
    model.set_ais_target(min_is_target=True)
    assert ais.p_target is False and hmc.p_target is False
    assert list(model.parameters())[0] is list(flow.parameters())[0]
    assert model.get_iter_info() == {}
    x = torch.randn(8, 6)
    from fab_torch_amd._lib import FabhipError
    with pytest.raises(FabhipError, match="no CPU path"):                     # the product has no CPU / ATen density
        model.forward_kl(x)
    with pytest.raises(FabhipError, match="no CPU path"):
        with torch.no_grad():
            flow.log_prob(x)
    with pytest.raises(FabhipError, match="no CPU path"):
        target.log_prob(x)


def test_fabmodel_save_load_roundtrip(tmp_path):
    flow = fa.RealNVP(6, 2, 5)
    target = fa.ManyWellEnergy(6, use_gpu=False)
    hmc = fa.HamiltonianMonteCarlo(4, 6, flow.log_prob, target.log_prob, alpha=2.0)
    model = fa.FABModel(flow, target, 4, transition_operator=hmc)
    with torch.no_grad():
        hmc.epsilons.mul_(0.3); flow._nf_model.q0.loc.add_(1.5)
    path = str(tmp_path / "model.pt")
    model.save(path)
    ckpt = torch.load(path)
    assert set(ckpt) == {"flow", "trans_op"} and set(ckpt["trans_op"]) == {"common_epsilon", "epsilons", "mass_vector"}
    flow2 = fa.RealNVP(6, 2, 5)
    hmc2 = fa.HamiltonianMonteCarlo(4, 6, flow2.log_prob, target.log_prob, alpha=2.0)
    model2 = fa.FABModel(flow2, target, 4, transition_operator=hmc2)
    model2.load(path)
    assert torch.equal(hmc2.epsilons, hmc.epsilons) and torch.equal(flow2._nf_model.q0.loc, flow._nf_model.q0.loc)
    assert model2.annealed_importance_sampler.transition_operator is hmc2


def test_manywell_eval_helpers_match_reference_golden():
    """ManyWellEnergy.get_modes_test_set_iterator / performance_metrics (log-Z part) / sample against fixtures
    produced by the imported reference (tests/golden/make_golden.py:g10_manywell_eval); many_well.py:24-36,96-115."""
    g = load_golden("g10_manywell_eval.npz")
    target = fa.ManyWellEnergy(dim=6, use_gpu=False)
    modes = torch.cat(target.get_modes_test_set_iterator(batch_size=3))
    assert np.array_equal(modes.numpy(), g["modes"])
    assert abs(float(target.log_Z) - float(g["log_Z"])) < 1e-5
    info = target.performance_metrics(None, torch.tensor(g["log_w"]))
    assert abs(info["relative_MSE_Z_estimate"] - float(g["relative_MSE_Z_estimate"])) < 1e-6
    assert abs(info["abs_MSE_log_Z_estimate"] - float(g["abs_MSE_log_Z_estimate"])) < 1e-6
    # the exact sampler draws from the same distribution (different RNG stream): moments within 5 standard errors
    torch.manual_seed(0)
    n = 200_000
    xs = target.sample((n,))
    assert xs.shape == (n, 6)
    se = g["sample_std"] / np.sqrt(n) * np.sqrt(2)
    assert np.all(np.abs(xs.mean(0).numpy() - g["sample_mean"]) < 5 * se)
    assert np.all(np.abs(xs.std(0).numpy() - g["sample_std"]) < 0.01)
    assert abs(float((xs[:, 0::2] > 0).float().mean()) - float(g["sample_frac_deep_well"])) < 0.005


def test_gmm_eval_helpers_match_reference_golden():
    """quadratic_function / importance_weighted_expectation / GMM.performance_metrics / effective_sample_size_over_p
    against the imported reference (g11; gmm.py:68-100, utils/numerical.py:25-64)."""
    from fab_torch_amd.numerical import quadratic_function, effective_sample_size_over_p
    g = load_golden("g11_gmm_eval.npz")
    x, log_w = torch.tensor(g["x"]), torch.tensor(g["log_w"])
    state = torch.get_rng_state()
    fx = quadratic_function(x)
    assert torch.equal(torch.get_rng_state(), state)           # unlike the reference, the global RNG is untouched
    assert np.allclose(fx.numpy(), g["fx"], rtol=1e-6, atol=1e-3)
    torch.manual_seed(0)
    target = fa.GMM(dim=2, n_mixes=40, loc_scaling=40.0, log_var_scaling=1.0, use_gpu=False,
                    true_expectation_estimation_n_samples=int(2e5))
    assert np.array_equal(target.locs.numpy(), g["locs"])       # same seeded means as gmm.py:22
    est = float(target.true_expectation)                        # own Monte-Carlo estimate: statistically equal
    assert abs(est - float(g["true_expectation"])) < 0.03 * abs(float(g["true_expectation"]))
    target._true_expectation = torch.tensor(float(g["true_expectation"]))
    info = target.performance_metrics(x, log_w)
    assert abs(info["bias_normed"] - float(g["bias_normed"])) < 1e-5
    assert abs(info["bias_no_correction"] - float(g["bias_no_correction"])) < 1e-5
    assert abs(float(effective_sample_size_over_p(0.5 * log_w)) - float(g["ess_over_p"])) < 1e-6


def test_trainers_accept_the_reference_logger_objects_and_plain_callables():
    """fab/utils/logging.py:12-30: the reference's trainers call `logger.write(info)` / `logger.close()`; a plain callable
    (this repo's own convention) keeps working."""
    from fab_torch_amd.train import _log, _close

    class ListLogger:                                      # the reference's default logger, reduced to its interface
        def __init__(self):
            self.history, self.closed = [], False

        def write(self, data):
            self.history.append(data)

        def close(self):
            self.closed = True
    lg = ListLogger()
    _log(lg, {"loss": 1.0}); _close(lg)
    assert lg.history == [{"loss": 1.0}] and lg.closed
    seen = []
    _log(seen.append, {"loss": 2.0}); _close(seen.append); _log(None, {}); _close(None)
    assert seen == [{"loss": 2.0}]
