"""FAB training loop with the prioritised buffer — the iteration of
fab/train_with_prioritised_buffer.py:138-216 (PrioritisedBufferTrainer.run) without the plotting / wandb /
directory plumbing: AIS (fused HIP call) -> buffer.add -> Gumbel-top-k minibatches -> for each minibatch
loss = -mean(clip(exp((1-alpha)(log q(x) - log_q_old))) * log q(x)), clipped-gradient optimiser step, buffer
weight adjustment.  NaN loss / non-finite gradient norm skip the step like the reference (:172-181)."""
from typing import Callable, Dict, List, Optional

import torch

from .buffer import PrioritisedReplayBuffer
from .core import FABModel
from .optim import FlatAdam


def _log(logger, info: Dict):
    """`logger` is the reference's `Logger` (fab/utils/logging.py:12-30: `.write(dict)`, `.close()`) or a plain callable."""
    if logger is None:
        return
    write = getattr(logger, "write", None)
    (write if callable(write) else logger)(info)


def _make_and_save_plots(trainer, i: int, save: bool):
    """train.py:47-54 / train_with_prioritised_buffer.py:70-77: `plot(model)` returns figures; saved under plots/ or
    shown.  (Plotting itself is the caller's code - this repo ships no plotter.)"""
    import os
    figures = trainer.plot(trainer.model)
    for j, figure in enumerate(figures):
        if save:
            os.makedirs(os.path.join(trainer.save_dir, "plots"), exist_ok=True)
            figure.savefig(os.path.join(trainer.save_dir, "plots", f"{j}_iter_{i}.png"))
        else:
            _pyplot_call("show")                           # train.py:52-53: plt.show() when not saving
        if not _is_matplotlib_figure(figure) or not _pyplot_call("close", figure):     # train.py:54: plt.close(figure)
            close = getattr(figure, "close", None)         # (a caller's non-matplotlib figure object)
            if callable(close):
                close()


def _is_matplotlib_figure(figure) -> bool:
    try:
        from matplotlib.figure import Figure
    except ImportError:
        return False
    return isinstance(figure, Figure)


def _pyplot_call(name: str, *args) -> bool:
    """Call matplotlib.pyplot.<name>(*args) if matplotlib is importable (the caller's `plot` made the figures with
    it); False otherwise."""
    try:
        import matplotlib.pyplot as plt
    except ImportError:
        return False
    getattr(plt, name)(*args)
    return True


def _time_is_up(tlimit, start_time, max_it_time) -> Optional[float]:
    """train.py:121-131: hours past if the next iteration would cross the limit, else None."""
    if tlimit is None:
        return None
    from time import time
    past = (time() - start_time) / 3600
    return past if past + max_it_time / 3600 > tlimit else None


def _close(logger):
    close = getattr(logger, "close", None)
    if callable(close):
        close()                                            # train_with_prioritised_buffer.py:255, train.py:135


class PrioritisedBufferTrainer:
    def __init__(self, model: FABModel, optimizer: torch.optim.Optimizer, buffer: PrioritisedReplayBuffer,
                 alpha: float, n_batches_buffer_sampling: int = 2, optim_schedular=None, logger=None, plot=None,
                 max_gradient_norm: Optional[float] = 5.0, w_adjust_max_clip: Optional[float] = 10.0,
                 w_adjust_in_buffer_after_update: bool = False, save_path: str = ""):
        # (argument order of train_with_prioritised_buffer.py:23-36)
        self.model, self.optimizer, self.buffer, self.alpha = model, optimizer, buffer, alpha
        self.plot = plot
        self.save_dir = save_path
        self.model.annealed_importance_sampler.p_target = False          # AIS targets p^alpha q^(1-alpha)
        self.model.annealed_importance_sampler.transition_operator.p_target = False
        self.optim_schedular = optim_schedular
        self.max_gradient_norm = max_gradient_norm if max_gradient_norm else float("inf")
        self.n_batches_buffer_sampling = n_batches_buffer_sampling
        self.max_adjust_w_clip = w_adjust_max_clip
        self.w_adjust_in_buffer_after_update = w_adjust_in_buffer_after_update
        self.logger = logger
        self.history: List[Dict] = []
        self._fused = isinstance(optimizer, FlatAdam) and optimizer.native      # tape + flat-image kernels: RealNVP
        self._flat = isinstance(optimizer, FlatAdam)                            # fused clip + Adam for any flow

    def step(self, i: int, batch_size: int, noise: Optional[Dict] = None) -> Dict:
        """One iteration of train_with_prioritised_buffer.py:138-198.  `noise` (optional, parity replays): the random
        draws of the iteration as explicit inputs — eps0 / noise_a / noise_b for the AIS call, gumbel / perm for the
        buffer's sampling without replacement."""
        model, buf = self.model, self.buffer
        noise = noise or {}
        self.optimizer.zero_grad()
        point_ais, log_w_ais = model.annealed_importance_sampler.sample_and_log_weights(
            batch_size, eps0=noise.get("eps0"), noise_a=noise.get("noise_a"), noise_b=noise.get("noise_b"))
        buf.add(point_ais.x.detach(), log_w_ais.detach(), point_ais.log_q.detach())
        if self._fused and self._one_op_minibatch():
            return self._step_fused(i, batch_size, noise)
        info = model.get_iter_info()
        mini_dataset = buf.sample_n_batches(batch_size=batch_size, n_batches=self.n_batches_buffer_sampling,
                                            gumbel=noise.get("gumbel"), perm=noise.get("perm"))
        self.last_indices = torch.cat([m[3] for m in mini_dataset])
        loss = grad_norm = None
        for (x, log_w, log_q_old, indices) in mini_dataset:
            self.optimizer.zero_grad()
            if self._fused:
                # FlatAdam path, no autograd graph at all: w_adjust is detached in the reference's loss (:164-170), so
                # d loss / d log_q_b = -w_adjust_b / B exactly, which is fed straight to the parameter-gradient
                # kernels.  Clipping, the finite-norm check and Adam run on the device; a non-finite loss gives a
                # non-finite gradient norm, which skips the update there (same outcome as the two host checks of the
                # reference, :172-181, without synchronising every minibatch).
                with torch.no_grad():
                    log_q_x, tape = model.flow.log_prob_with_tape(x)
                    log_w_adjust = (1 - self.alpha) * (log_q_x - log_q_old)
                    w_adjust_pre_clip = torch.exp(log_w_adjust)
                    w_adjust = (torch.clip(w_adjust_pre_clip, max=self.max_adjust_w_clip)
                                if self.max_adjust_w_clip is not None else w_adjust_pre_clip)
                    loss = - torch.mean(w_adjust * log_q_x)
                    # a non-finite loss must skip the update even when the gradient image happens to be finite
                    # (e.g. a row with log_q = -inf whose coefficient is 0 or clipped): poison the coefficients, the
                    # on-device finite-norm check of fabhip_adam_clip_step then skips (reference :172-181)
                    poison = torch.where(torch.isfinite(loss), 1.0, float("nan"))
                    flat = model.flow.param_grad_flat(tape, w_adjust * (-1.0 / x.shape[0]) * poison)
                grad_norm = self.optimizer.step(max_grad_norm=self.max_gradient_norm, flat_grad=flat)
                if not self.w_adjust_in_buffer_after_update:
                    buf.adjust(log_w_adjust, log_q_x, indices)
                continue
            log_q_x = model.flow.log_prob(x)
            log_w_adjust = (1 - self.alpha) * (log_q_x.detach() - log_q_old)
            w_adjust_pre_clip = torch.exp(log_w_adjust)
            w_adjust = (torch.clip(w_adjust_pre_clip, max=self.max_adjust_w_clip)
                        if self.max_adjust_w_clip is not None else w_adjust_pre_clip)
            loss = - torch.mean(w_adjust * log_q_x)
            if self._flat:
                # FlatAdam on a non-RealNVP flow: autograd for the gradients, then ONE fused clip + Adam launch whose
                # on-device finite-norm check skips the update (a non-finite loss poisons the gradient: reference :172-181)
                (loss * torch.where(torch.isfinite(loss.detach()), 1.0, float("nan"))).backward()
                grad_norm = self.optimizer.step(max_grad_norm=self.max_gradient_norm)
            elif torch.isfinite(loss):
                loss.backward()
                grad_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), self.max_gradient_norm)
                if torch.isfinite(grad_norm):
                    self.optimizer.step()
                else:
                    print("nan grad norm in replay step")
            else:
                print("nan loss in replay step")
            if not self.w_adjust_in_buffer_after_update:
                buf.adjust(log_w_adjust, log_q_x.detach(), indices)
        info.update(loss=loss.item(), step=i, grad_norm=float(grad_norm) if grad_norm is not None else float("nan"),
                    sampled_log_w_std=torch.std(log_w).item(), sampled_log_w_mean=torch.mean(log_w).item(),
                    w_adjust_mean=torch.mean(w_adjust_pre_clip).item(), w_adjust_min=torch.min(w_adjust_pre_clip).item(),
                    w_adjust_max=torch.max(w_adjust_pre_clip).item(), log_q_x_mean=torch.mean(log_q_x).item())
        if self.w_adjust_in_buffer_after_update:
            with torch.no_grad():
                for (x, log_w, log_q_old, indices) in mini_dataset:
                    log_q_new = model.flow.log_prob(x)
                    buf.adjust((1 - self.alpha) * (log_q_new - log_q_old), log_q_new, indices)
        # NB: like the reference, this trainer never steps `optim_schedular` (it is only checkpointed, :59-68);
        # fab/train.py:110-111 is the loop that steps it.
        return info

    def _one_op_minibatch(self) -> bool:
        """The whole minibatch body as ONE op (fabhip::buffer_train_step): RealNVP + FlatAdam, the buffer on the flow's device and
        sampled WITHOUT replacement (the op reads x / log_q_old in place and adjusts the buffer before the next minibatch reads it,
        which equals the reference's gather-everything-first only when no row is drawn twice), weights adjusted on the fly."""
        buf = self.buffer
        return (not buf.sample_with_replacement and not self.w_adjust_in_buffer_after_update and buf.buffer.x.is_cuda
                and buf.buffer.x.device == self.optimizer.theta.device and getattr(self, "one_op_minibatch", True))

    def _step_fused(self, i: int, batch_size: int, noise: Dict) -> Dict:
        """train_with_prioritised_buffer.py:153-198 with every minibatch as one `fabhip::buffer_train_step` call: no gathered copies
        of the minibatches, no autograd graph, no host synchronisation until the iteration's logging values are read (once)."""
        from . import _ops
        model, buf, opt = self.model, self.buffer, self.optimizer
        ops = _ops.load()
        flow = model.flow
        nb = self.n_batches_buffer_sampling
        indices = buf.sample_indices(batch_size * nb, gumbel=noise.get("gumbel"), perm=noise.get("perm"))
        self.last_indices = indices
        chunks = torch.chunk(indices, nb)
        log_w_last = buf.buffer.log_w[chunks[-1]]               # (logging: the last minibatch's weights as sampled, :190-191)
        packed, D, K, W = flow.native(need_inverse=False)       # registers the parameter set / makes the image current
        opt._check_alias()
        handle = flow._own_handle()
        grp = opt.param_groups[0]
        clip = float(self.max_adjust_w_clip) if self.max_adjust_w_clip is not None else 0.0
        mx = 0.0 if self.max_gradient_norm == float("inf") else float(self.max_gradient_norm)
        stats = None
        with torch.no_grad():
            theta = opt.theta.detach()
            for j, rows in enumerate(chunks):
                _, _, stats = ops.buffer_train_step(
                    handle, packed, D, K, W, j > 0, buf.buffer.x, rows.contiguous(), buf.buffer.log_q_old, True, float(self.alpha),
                    clip, buf.buffer.log_w, buf.buffer.log_q_old, theta, opt.m, opt.v, float(grp["lr"]),
                    float(grp["betas"][0]), float(grp["betas"][1]), float(grp["eps"]), opt.steps, mx)
        flow._packed_key = None                                 # the parameters moved behind autograd's version counters (and the
        flow._packed_has_inverse = False                        # image holds the training tiles of the last-but-one parameters)
        opt.grad_norm.copy_(stats[5:6])
        # the AIS call's logging values (:150; they do not change after the call): read here, behind the enqueued minibatches, so
        # that the device-to-host read of the transition operator's statistics does not stall the queue in front of them
        info = model.get_iter_info()
        lw = torch.stack([torch.std(log_w_last), torch.mean(log_w_last)])
        host = torch.cat([stats[:6], lw]).tolist()              # ONE device-to-host read for the iteration's logging values
        info.update(loss=host[0], step=i, grad_norm=host[5], sampled_log_w_std=host[6], sampled_log_w_mean=host[7],
                    w_adjust_mean=host[1], w_adjust_min=host[2], w_adjust_max=host[3], log_q_x_mean=host[4])
        return info

    def save_checkpoint(self, i: int):
        """model.pt / optimizer.pt / buffer.pt under model_checkpoints/iter_{i}/, scheduler.pt next to the
        iteration directories (train_with_prioritised_buffer.py:59-68)."""
        import os
        ckpt_dir = os.path.join(self.save_dir, "model_checkpoints")
        path = os.path.join(ckpt_dir, f"iter_{i}")
        os.makedirs(path, exist_ok=False)
        self.model.save(os.path.join(path, "model.pt"))
        torch.save(self.optimizer.state_dict(), os.path.join(path, "optimizer.pt"))
        self.buffer.save(os.path.join(path, "buffer.pt"))
        if self.optim_schedular:
            torch.save(self.optim_schedular.state_dict(), os.path.join(ckpt_dir, "scheduler.pt"))

    def make_and_save_plots(self, i: int, save: bool):
        _make_and_save_plots(self, i, save)

    def perform_eval(self, i: int, eval_batch_size: int, batch_size: int) -> Dict:
        """train_with_prioritised_buffer.py:79-101: frozen step sizes; p as the AIS target, then the practical target."""
        ais = self.model.annealed_importance_sampler
        ais.transition_operator.set_eval_mode(True)
        info_p = self.model.get_eval_info(outer_batch_size=eval_batch_size, inner_batch_size=batch_size,
                                          set_p_target=True)
        assert ais.p_target is False and ais.transition_operator.p_target is False
        info_g = self.model.get_eval_info(outer_batch_size=eval_batch_size, inner_batch_size=batch_size,
                                          set_p_target=False, ais_only=True)
        ais.transition_operator.set_eval_mode(False)
        out = {k + "_p_target": v for k, v in info_p.items()}
        out.update({k + "_min_var_target": v for k, v in info_g.items()})
        out.update(step=i)
        return out

    def run(self, n_iterations: int, batch_size: int, eval_batch_size: Optional[int] = None,
            n_eval: Optional[int] = None, n_plot: Optional[int] = None, n_checkpoints: Optional[int] = None,
            save: bool = True, tlimit: Optional[float] = None, start_time: Optional[float] = None,
            start_iter: int = 0) -> List[Dict]:
        """train_with_prioritised_buffer.py:106-255, same arguments (`tlimit` in hours: stop + checkpoint before the next
        iteration would cross it; `n_plot` / `save` drive the caller's `plot(model)`)."""
        import numpy as np
        from time import time
        if start_iter >= n_iterations:
            raise Exception("Not running training as start_iter >= total training iterations")
        eval_iter = list(np.linspace(1, n_iterations, n_eval, dtype="int")) if n_eval is not None else []
        plot_iter = list(np.linspace(1, n_iterations, n_plot, dtype="int")) if n_plot is not None and self.plot else []
        ckpt_iter = list(np.linspace(1, n_iterations, n_checkpoints, dtype="int")) if n_checkpoints else []
        if n_eval is not None:
            assert eval_batch_size is not None
        if tlimit is not None:
            assert n_checkpoints is not None, "Time limited specified but not checkpoints are being saved."
        start_time = time()                                 # (the reference overwrites a supplied start_time too, :129-130)
        max_it_time = 0.0
        for i in range(start_iter + 1, n_iterations + 1):
            it_start = time()
            info = self.step(i, batch_size)
            self.history.append(info)
            _log(self.logger, info)
            if i in eval_iter:
                ev = self.perform_eval(i, eval_batch_size, batch_size)
                self.history.append(ev)
                _log(self.logger, ev)
            if i in plot_iter:
                _make_and_save_plots(self, i, save)
            if i in ckpt_iter:
                self.save_checkpoint(i)
            max_it_time = max(max_it_time, time() - it_start)
            past = _time_is_up(tlimit, start_time, max_it_time)
            if past is not None:
                if i not in ckpt_iter:
                    self.save_checkpoint(i)
                _close(self.logger)
                print(f"\nEnding training at iteration {i}, after training for {past:.2f} hours as timelimit "
                      f"{tlimit:.2f} hours has been reached.\n")
                return self.history
        _close(self.logger)
        return self.history


class Trainer:
    """The plain FAB loop of fab/train.py:16-136 (what the shipped ManyWell config runs: `fab_alpha_div` on fresh
    AIS samples, no buffer): zero_grad -> model.loss(batch_size) -> backward -> clip -> step, with the NaN-loss and
    non-finite-gradient skips, evaluation (`model.get_eval_info`) and checkpoints at linearly spaced iterations.
    Plotting / tqdm / time limits are left to the caller."""

    def __init__(self, model: FABModel, optimizer: torch.optim.Optimizer, optim_schedular=None,
                 logger=None, plot=None, max_gradient_norm: Optional[float] = 5.0,
                 save_path: str = ""):
        self.model, self.optimizer, self.optim_schedular, self.logger = model, optimizer, optim_schedular, logger
        self.plot = plot
        self.max_gradient_norm = max_gradient_norm if max_gradient_norm else float("inf")
        self.save_dir = save_path
        self.history: List[Dict] = []
        self._fused = isinstance(optimizer, FlatAdam) and optimizer.native      # tape + flat-image kernels: RealNVP
        self._flat = isinstance(optimizer, FlatAdam)                            # fused clip + Adam for any flow

    def save_checkpoint(self, i: int):
        import os
        path = os.path.join(self.save_dir, "model_checkpoints", f"iter_{i}")
        os.makedirs(path, exist_ok=False)
        self.model.save(os.path.join(path, "model.pt"))
        torch.save(self.optimizer.state_dict(), os.path.join(path, "optimizer.pt"))
        if self.optim_schedular:
            torch.save(self.optim_schedular.state_dict(),
                       os.path.join(self.save_dir, "model_checkpoints", "scheduler.pt"))

    def make_and_save_plots(self, i: int, save: bool):
        _make_and_save_plots(self, i, save)

    def perform_eval(self, i: int, eval_batch_size: int, batch_size: int) -> Dict:
        """fab/train.py:56-60."""
        ev = self.model.get_eval_info(outer_batch_size=eval_batch_size, inner_batch_size=batch_size)
        ev.update(step=i)
        _log(self.logger, ev)
        return ev

    def step(self, i: int, batch_size: int) -> Dict:
        self.optimizer.zero_grad()
        grad_norm = torch.tensor(float("nan"))
        if self._fused and self.model.loss_type == "fab_alpha_div":
            # fab_alpha_div (core.py:112-128) without an autograd graph: the AIS weights are detached, so
            # d loss / d log_q_b = -sign(alpha) softmax(log_w)_b / B goes straight to the parameter-gradient kernels
            model = self.model
            with torch.no_grad():
                model.set_ais_target(min_is_target=True)
                point, log_w = model.annealed_importance_sampler.sample_and_log_weights(batch_size)
                log_q_x, tape = model.flow.log_prob_with_tape(point.x)
                w = torch.softmax(log_w, dim=-1)
                sign = float(1.0 if model.alpha > 0 else (-1.0 if model.alpha < 0 else 0.0))
                loss = -sign * torch.mean(w * log_q_x)
                poison = torch.where(torch.isfinite(loss), 1.0, float("nan"))     # non-finite loss => skipped update
                flat = model.flow.param_grad_flat(tape, w * (-sign / log_q_x.shape[0]) * poison)
                model.set_ais_target(min_is_target=False)
            grad_norm = self.optimizer.step(max_grad_norm=self.max_gradient_norm, flat_grad=flat)
            if self.optim_schedular:          # the host reads the loss for logging below anyway
                if bool(torch.isfinite(loss)):
                    self.optim_schedular.step()
            self.optimizer.zero_grad()
            info = self.model.get_iter_info()
            info.update(loss=loss.item(), step=i, grad_norm=float(grad_norm))
            return info
        loss = self.model.loss(batch_size)
        if self._flat and bool(torch.isfinite(loss)):
            loss.backward()
            grad_norm = self.optimizer.step(max_grad_norm=self.max_gradient_norm)
            if self.optim_schedular:
                self.optim_schedular.step()
        elif self._flat:
            print("nan loss encountered")
        elif not torch.isnan(loss) and not torch.isinf(loss):
            loss.backward()
            grad_norm = torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.max_gradient_norm)
            if torch.isfinite(grad_norm):
                self.optimizer.step()
            else:
                print("encountered inf grad norm")
            if self.optim_schedular:
                self.optim_schedular.step()
        else:
            print("nan loss encountered")
        self.optimizer.zero_grad()
        info = self.model.get_iter_info()
        info.update(loss=loss.item(), step=i, grad_norm=float(grad_norm))
        return info

    def run(self, n_iterations: int, batch_size: int, eval_batch_size: Optional[int] = None,
            n_eval: Optional[int] = None, n_plot: Optional[int] = None, n_checkpoints: Optional[int] = None,
            save: bool = True, tlimit: Optional[float] = None, start_time: Optional[float] = None,
            start_iter: int = 0) -> List[Dict]:
        """fab/train.py:63-136, same arguments."""
        import numpy as np
        from time import time
        if start_iter >= n_iterations:
            raise Exception("Not running training as start_iter >= total training iterations")
        eval_iter = list(np.linspace(1, n_iterations, n_eval, dtype="int")) if n_eval is not None else []
        plot_iter = list(np.linspace(1, n_iterations, n_plot, dtype="int")) if n_plot is not None and self.plot else []
        ckpt_iter = list(np.linspace(1, n_iterations, n_checkpoints, dtype="int")) if n_checkpoints else []
        if n_eval is not None:
            assert eval_batch_size is not None
        if tlimit is not None:
            assert n_checkpoints is not None, "Time limited specified but not checkpoints are being saved."
        start_time = time()
        max_it_time = 0.0
        for i in range(start_iter + 1, n_iterations + 1):
            it_start = time()
            info = self.step(i, batch_size)
            if i in eval_iter:
                info.update(self.model.get_eval_info(outer_batch_size=eval_batch_size, inner_batch_size=batch_size))
            self.history.append(info)
            _log(self.logger, info)
            if i in plot_iter:
                _make_and_save_plots(self, i, save)
            if i in ckpt_iter:
                self.save_checkpoint(i)
            max_it_time = max(max_it_time, time() - it_start)
            past = _time_is_up(tlimit, start_time, max_it_time)
            if past is not None:
                if i not in ckpt_iter:
                    self.save_checkpoint(i)
                _close(self.logger)
                print(f"\nEnding training at iteration {i}, after training for {past:.2f} hours as timelimit "
                      f"{tlimit:.2f} hours has been reached.\n")
                return self.history
        _close(self.logger)
        return self.history
