"""Oracle prioritised-buffer training iteration (PyTorch CPU, explicit noise) — TEST INFRASTRUCTURE.

Restates, in meaning,
* PrioritisedReplayBuffer.add / sample / adjust ... fab/utils/prioritised_replay_buffer.py:71-131
  (`sample_without_replacement` :10-17: top-n of log_w + Gumbel noise, then a random permutation)
* one iteration of PrioritisedBufferTrainer.run ... fab/train_with_prioritised_buffer.py:138-198
  (AIS -> buffer.add -> sample n_batches minibatches -> per minibatch: loss = -mean(clip(exp((1-alpha)
  (log q(x) - log_q_old))) * log q(x)), NaN/inf-loss and non-finite-grad-norm skips, clip_grad_norm_, optimiser
  step, on-the-fly buffer adjust).
Random draws are arguments: the AIS noise (oracle/ais.py), the buffer's Gumbel noise `gumbel [len]` and the
permutation `perm [n]`.  Pinned against the imported reference by tests/golden/g12_trainer_seed*.npz.
"""
import torch


class Buffer:
    def __init__(self, dim, max_length, min_sample_length):
        self.x = torch.zeros(max_length, dim)
        self.log_w = torch.zeros(max_length)
        self.log_q_old = torch.zeros(max_length)
        self.max_length, self.min_sample_length = max_length, min_sample_length
        self.current_index, self.is_full, self.can_sample = 0, False, False

    def add(self, x, log_w, log_q_old):                                   # :71-85
        n = x.shape[0]
        idx = (torch.arange(n) + self.current_index) % self.max_length
        self.x[idx], self.log_w[idx], self.log_q_old[idx] = x, log_w, log_q_old
        new_index = self.current_index + n
        if not self.is_full:
            self.is_full = new_index >= self.max_length
            self.can_sample = new_index >= self.min_sample_length
        self.current_index = new_index % self.max_length

    def sample(self, n, gumbel, perm):                                    # :88-103, :10-17
        max_index = self.max_length if self.is_full else self.current_index
        keys = gumbel + self.log_w[:max_index]
        idx = torch.topk(keys, n, sorted=False).indices
        idx = idx[perm]
        return self.x[idx], self.log_w[idx], self.log_q_old[idx], idx

    def adjust(self, log_w_adjustment, log_q, indices):                   # :117-131
        valid = torch.isfinite(log_w_adjustment) & torch.isfinite(log_q)
        vi = indices[valid]
        self.log_w[vi] += log_w_adjustment[valid]
        self.log_q_old[vi] = log_q[valid]
        self.log_w[indices[~valid]] = -float("inf")


def train_iteration(ais, flow_log_prob, params, optimizer, buffer: Buffer, alpha, batch_size, n_batches, noise,
                    max_gradient_norm=5.0, w_adjust_max_clip=10.0):
    """One iteration of train_with_prioritised_buffer.py:138-198.  `ais`: oracle AIS; `noise` = dict(eps0, noise_p,
    noise_e, gumbel, perm).  Returns the logged scalars + the sampled indices."""
    optimizer.zero_grad()
    pt, log_w_ais, info = ais.sample_and_log_weights(noise["eps0"], noise["noise_p"], noise["noise_e"])
    buffer.add(pt.x.detach(), log_w_ais.detach(), pt.log_q.detach())
    x, log_w, log_q_old, indices = buffer.sample(batch_size * n_batches, noise["gumbel"], noise["perm"])
    loss = grad_norm = None
    for xb, lwb, lqb, ib in zip(torch.chunk(x, n_batches), torch.chunk(log_w, n_batches),
                                torch.chunk(log_q_old, n_batches), torch.chunk(indices, n_batches)):
        optimizer.zero_grad()
        log_q_x = flow_log_prob(xb)
        log_w_adjust = (1 - alpha) * (log_q_x.detach() - lqb)
        w_pre = torch.exp(log_w_adjust)
        w_adjust = torch.clip(w_pre, max=w_adjust_max_clip) if w_adjust_max_clip is not None else w_pre
        loss = -torch.mean(w_adjust * log_q_x)
        if not torch.isnan(loss) and not torch.isinf(loss):
            loss.backward()
            grad_norm = torch.nn.utils.clip_grad_norm_(params, max_gradient_norm)
            if torch.isfinite(grad_norm):
                optimizer.step()
        with torch.no_grad():
            buffer.adjust(log_w_adjust, log_q_x.detach(), ib)
    return dict(loss=float(loss.detach()), grad_norm=float(grad_norm), indices=indices, ess_ais=info.ess_ais,
                ess_base=info.ess_base, log_Z=info.log_Z, w_adjust_mean=float(w_pre.mean()),
                log_q_x_mean=float(log_q_x.mean()), sampled_log_w_mean=float(lwb.mean()))
