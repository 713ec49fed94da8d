"""Checkpoint / resume of a training run on the GPU (fab/core.py:222-260 `FABModel.save/load`, the optimiser state
and the replay buffer, fab/utils/prioritised_replay_buffer.py:133-153): a run interrupted after n iterations, saved,
re-created from scratch and resumed must end with exactly the parameters of the uninterrupted run (every kernel is
deterministic and the noise comes from torch's seeded device generator)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

fa = pytest.importorskip("fab_torch_amd")

DEV = "cuda"
D, B = 6, 128


def make(seed, optimiser):
    torch.manual_seed(seed)
    flow = fa.RealNVP(D, 3, 6).to(DEV)
    target = fa.ManyWellEnergy(D)
    hmc = fa.HamiltonianMonteCarlo(2, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.3,
                                   L=3).to(DEV)
    model = fa.FABModel(flow, target, 2, alpha=2.0, transition_operator=hmc, loss_type="fab_alpha_div")
    opt = fa.FlatAdam(flow, lr=1e-3) if optimiser == "flat_adam" else torch.optim.Adam(flow.parameters(), lr=1e-3)
    return flow, hmc, model, opt


def make_buffer(model):
    def init_sampler():
        pt, lw = model.annealed_importance_sampler.sample_and_log_weights(B, logging=False)
        return pt.x, lw, pt.log_q
    return fa.PrioritisedReplayBuffer(D, 12 * B, 2 * B, init_sampler, device=DEV)


@pytest.mark.parametrize("optimiser", ["flat_adam", "torch_adam"])
def test_interrupted_run_resumes_bit_identically(optimiser, tmp_path):
    # uninterrupted: 6 iterations
    flow, hmc, model, opt = make(0, optimiser)
    buf = make_buffer(model)
    trainer = fa.PrioritisedBufferTrainer(model, opt, buf, alpha=2.0, n_batches_buffer_sampling=2,
                                          max_gradient_norm=100.0)
    trainer.run(3, B)
    rng_mid = torch.cuda.get_rng_state()
    model.save(str(tmp_path / "model.pt"))
    torch.save(opt.state_dict(), str(tmp_path / "opt.pt"))
    buf.save(str(tmp_path / "buffer.pt"))
    trainer.run(6, B, start_iter=3)
    ref = {k: v.detach().clone() for k, v in flow.state_dict().items()}
    ref_eps = hmc.epsilons.clone()

    # resumed: fresh objects with different initial values, everything restored from the files
    flow2, hmc2, model2, opt2 = make(123, optimiser)
    assert any(not torch.equal(v, flow2.state_dict()[k]) for k, v in ref.items())
    buf2 = make_buffer(model2)
    model2.load(str(tmp_path / "model.pt"))
    opt2.load_state_dict(torch.load(str(tmp_path / "opt.pt")))
    buf2.load(str(tmp_path / "buffer.pt"))
    trainer2 = fa.PrioritisedBufferTrainer(model2, opt2, buf2, alpha=2.0, n_batches_buffer_sampling=2,
                                           max_gradient_norm=100.0)
    torch.cuda.set_rng_state(rng_mid)
    trainer2.run(6, B, start_iter=3)
    for k, v in ref.items():
        assert torch.equal(v, flow2.state_dict()[k]), k
    assert torch.equal(ref_eps, hmc2.epsilons)
    # the kernel image follows the restored parameters
    x = torch.randn(32, D, device=DEV)
    with torch.no_grad():
        assert torch.equal(flow.native_log_prob(x)[0], flow2.native_log_prob(x)[0])


def test_autograd_free_training_entry_points_equal_the_autograd_function():
    """RealNVP.log_prob_with_tape / param_grad_flat (what the fused trainer calls) against the torch.autograd.Function
    wrapping the same kernels: identical log q and, for the same coefficients, a bitwise identical gradient image."""
    flow, _, _, opt = make(3, "flat_adam")
    x = torch.randn(200, D, device=DEV)
    coef = torch.randn(200, device=DEV) / 200
    opt.zero_grad()
    lq_a = flow.log_prob(x)
    (lq_a * coef).sum().backward()
    g_a = opt.theta.grad.clone()
    with torch.no_grad():
        lq_b, tape = flow.log_prob_with_tape(x)
        g_b = flow.param_grad_flat(tape, coef)
    assert torch.equal(lq_a.detach(), lq_b) and torch.equal(g_a, g_b)
    assert float(g_b.abs().max()) > 0


def test_fused_fab_alpha_div_step_equals_the_autograd_step():
    """Plain Trainer with FlatAdam: the autograd-free fab_alpha_div step against model.loss(B).backward() + the same
    optimiser step on an identical copy fed the same device noise (fab/core.py:112-128, fab/train.py:100-111)."""
    flow_a, _, model_a, opt_a = make(4, "flat_adam")
    flow_b, _, model_b, opt_b = make(4, "flat_adam")
    torch.manual_seed(77)
    opt_a.zero_grad()
    loss_a = model_a.loss(B)
    loss_a.backward()
    opt_a.step(max_grad_norm=50.0)
    torch.manual_seed(77)
    info = fa.Trainer(model_b, opt_b, max_gradient_norm=50.0).step(1, B)
    assert abs(info["loss"] - float(loss_a.detach())) <= 1e-6 * max(1.0, abs(float(loss_a.detach())))
    for (k, va), (_, vb) in zip(flow_a.state_dict().items(), flow_b.state_dict().items()):
        assert torch.equal(va, vb), k
    # like core.py:123-128, the sampler is left on the p target (evaluation) after the loss
    assert model_b.annealed_importance_sampler.p_target is model_a.annealed_importance_sampler.p_target is True


@pytest.mark.parametrize("kind", ["buffer", "plain"])
def test_run_takes_the_reference_arguments_time_limit_plots_logger(kind, tmp_path):
    """`run(n_iterations, batch_size, eval_batch_size, n_eval, n_plot, n_checkpoints, save, tlimit, start_time,
    start_iter)` as in fab/train.py:63-136 / train_with_prioritised_buffer.py:106-255: the caller's `plot(model)` is
    driven at the plot iterations, a `Logger` object gets write() / close(), and a time limit stops the run with a
    checkpoint of the iteration it stopped at."""
    import os

    class Fig:
        saved = []

        def savefig(self, path):
            Fig.saved.append(os.path.basename(path))

    class ListLogger:
        def __init__(self):
            self.rows, self.closed = [], False

        def write(self, d):
            self.rows.append(d)

        def close(self):
            self.closed = True
    flow, hmc, model, opt = make(0, "torch_adam")
    lg, calls = ListLogger(), []

    def plot(m):
        calls.append(m)
        return [Fig(), Fig()]
    if kind == "buffer":
        trainer = fa.PrioritisedBufferTrainer(model, opt, make_buffer(model), 2.0, 2, None, lg, plot, 100.0, 10.0, False,
                                              str(tmp_path))          # positional, in the reference's order
    else:
        trainer = fa.Trainer(model, opt, None, lg, plot, 100.0, str(tmp_path))
    trainer.run(n_iterations=4, batch_size=B, eval_batch_size=2 * B, n_eval=2, n_plot=2, n_checkpoints=2, save=True,
                tlimit=None, start_time=None, start_iter=0)
    assert len(calls) == 2 and calls[0] is model and sorted(set(Fig.saved)) == ["0_iter_1.png", "0_iter_4.png", "1_iter_1.png",
                                                                               "1_iter_4.png"]
    assert lg.closed and len(lg.rows) >= 4 and any("eval_ess_ais" in r or "eval_ess_ais_p_target" in r for r in lg.rows)
    assert sorted(os.listdir(tmp_path / "model_checkpoints")) == ["iter_1", "iter_4"]
    # a time limit of ~0 hours: one iteration, then a checkpoint of that iteration and a closed logger
    lg2 = ListLogger()
    trainer.logger = lg2
    hist_before = len(trainer.history)
    trainer.run(n_iterations=50, batch_size=B, n_checkpoints=2, tlimit=1e-9, start_iter=10)
    assert lg2.closed and len(trainer.history) == hist_before + 1
    assert "iter_11" in os.listdir(tmp_path / "model_checkpoints")
