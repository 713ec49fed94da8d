"""FABModel — host-side glue with the reference's interface (fab/core.py:18-260): wires flow + target + AIS,
toggles the AIS target (`set_ais_target`, core.py:102-110), the FAB alpha-divergence loss
(`fab_alpha_div_inner`, core.py:112-118: -sign(alpha) * mean(softmax(log_w) * log q(x))), iteration / evaluation
info and checkpoints `{'flow': state_dict, 'trans_op': state_dict}` (core.py:222-260).

The AIS call is the fused HIP path; `flow.log_prob(x)` inside the losses is the differentiable custom op
`fabhip::realnvp_logprob_tape` (HIP forward with a tape, HIP parameter-gradient kernels as its registered backward), the
reparameterised baseline losses go through `fabhip::realnvp_sample_tape`.  The reference's experimental losses
(core.py:134-170: `flow_alpha_2_div`, `flow_alpha_2_div_unbiased`, `fab_ub_alpha_2_div`), which its constructor refuses, are not
carried."""
import warnings
from typing import Any, Dict, Optional

import numpy as np
import torch

from .ais import AnnealedImportanceSampler
from .numerical import effective_sample_size
from .point import Point
from .transition_operators import TransitionOperator

ALPHA_DIV_TARGET_LOSSES = ["fab_alpha_div"]
LOSSES_USING_AIS = ["fab_alpha_div", "fab_ub_alpha_2_div", None]
EXPERIMENTAL_LOSSES = ["flow_alpha_2_div_unbiased", "flow_alpha_2_div", "fab_ub_alpha_2_div"]     # core.py:15
SUPPORTED_LOSSES = [None, "fab_alpha_div", "forward_kl", "flow_reverse_kl", "flow_alpha_2_div_nis",
                    "target_forward_kl"] + EXPERIMENTAL_LOSSES


class FABModel:
    def __init__(self, flow, target_distribution, n_intermediate_distributions: int, alpha: float = 2.,
                 transition_operator: Optional[TransitionOperator] = None, ais_distribution_spacing: str = "linear",
                 loss_type: Optional[str] = None, use_ais: bool = True):
        assert loss_type in SUPPORTED_LOSSES
        if loss_type in EXPERIMENTAL_LOSSES:               # same refusal as the reference (core.py:50-51)
            raise Exception("Running using experiment loss not used within the main FAB paper.")
        if loss_type in ALPHA_DIV_TARGET_LOSSES:
            assert alpha is not None, "Alpha must be specified if using the alpha div loss."
        self.alpha, self.loss_type = alpha, loss_type
        self.flow, self.target_distribution = flow, target_distribution
        self.n_intermediate_distributions = n_intermediate_distributions
        self.ais_distribution_spacing = ais_distribution_spacing
        assert len(flow.event_shape) == 1, "Currently only 1D distributions are supported"
        if use_ais or loss_type in LOSSES_USING_AIS:
            if transition_operator is None:
                raise Exception("If using AIS, transition operator must be provided.")
            self.transition_operator = transition_operator
            self.annealed_importance_sampler = self._make_ais()

    def _make_ais(self):
        self.transition_operator.p_target = False
        self.transition_operator.alpha = self.alpha
        return AnnealedImportanceSampler(
            base_distribution=self.flow, target_log_prob=self.target_distribution.log_prob,
            transition_operator=self.transition_operator, p_target=False, alpha=self.alpha,
            n_intermediate_distributions=self.n_intermediate_distributions,
            distribution_spacing_type=self.ais_distribution_spacing)

    def parameters(self):
        return self.flow.parameters()

    # ---- losses ---------------------------------------------------------------------------------------
    def loss(self, args) -> torch.Tensor:
        if self.loss_type is None:
            raise NotImplementedError("If loss_type is None, then the loss must be manually calculated.")
        return {"fab_alpha_div": self.fab_alpha_div, "forward_kl": self.forward_kl,
                "flow_reverse_kl": self.flow_reverse_kl, "flow_alpha_2_div_nis": self.flow_alpha_2_div_nis,
                "target_forward_kl": self.target_forward_kl}[self.loss_type](args)

    def set_ais_target(self, min_is_target: bool = True):
        """min_is_target: AIS targets p^alpha q^(1-alpha); otherwise p."""
        p_target = not min_is_target
        self.annealed_importance_sampler.p_target = p_target
        self.annealed_importance_sampler.transition_operator.p_target = p_target

    def fab_alpha_div_inner(self, point: Point, log_w_ais: torch.Tensor) -> torch.Tensor:
        log_q_x = self.flow.log_prob(point.x)
        return - np.sign(self.alpha) * torch.mean(torch.softmax(log_w_ais, dim=-1) * log_q_x)

    def fab_alpha_div(self, batch_size: int) -> torch.Tensor:
        self.set_ais_target(min_is_target=True)
        point_ais, log_w_ais = self.annealed_importance_sampler.sample_and_log_weights(batch_size)
        loss = self.fab_alpha_div_inner(point_ais, log_w_ais)
        self.set_ais_target(min_is_target=False)          # evaluation runs with the target p
        return loss

    def inner_loss(self, point: Point, log_w_ais) -> torch.Tensor:
        if self.loss_type == "fab_alpha_div":
            return self.fab_alpha_div_inner(point, log_w_ais)
        raise NotImplementedError                          # ("fab_ub_alpha_2_div": experimental in the reference, refused at construction)

    def flow_reverse_kl(self, batch_size: int) -> torch.Tensor:
        x, log_q = self.flow.sample_and_log_prob((batch_size,))
        return torch.mean(log_q) - torch.mean(self.target_distribution.log_prob(x))

    def flow_alpha_2_div_nis(self, batch_size: int) -> torch.Tensor:
        x, log_q_x = self.flow.sample_and_log_prob((batch_size,))
        log_p_x = self.target_distribution.log_prob(x)
        return - torch.mean(torch.exp(2 * (log_p_x - log_q_x)).detach() * log_q_x)

    def target_forward_kl(self, batch_size: int) -> torch.Tensor:
        return self.forward_kl(self.target_distribution.sample((batch_size,)))

    def forward_kl(self, x_p: torch.Tensor) -> torch.Tensor:
        return -torch.mean(self.flow.log_prob(x_p))

    # ---- info -----------------------------------------------------------------------------------------
    def get_iter_info(self) -> Dict[str, Any]:
        ais = getattr(self, "annealed_importance_sampler", None)
        if ais is not None and hasattr(ais, "_logging_info"):
            return ais.get_logging_info()
        return {}

    def get_eval_info(self, outer_batch_size: int, inner_batch_size: int, set_p_target: bool = True,
                      ais_only: bool = False) -> Dict[str, Any]:
        if not hasattr(self, "annealed_importance_sampler"):
            raise NotImplementedError
        if set_p_target:
            self.set_ais_target(min_is_target=False)
        base_samples, base_log_w, ais_samples, ais_log_w = \
            self.annealed_importance_sampler.generate_eval_data(outer_batch_size, inner_batch_size)
        info = dict(zip(("eval_ess_flow", "eval_ess_ais"),
                        torch.stack([effective_sample_size(base_log_w), effective_sample_size(ais_log_w)]).tolist()))
        metrics = getattr(self.target_distribution, "performance_metrics", None)
        if metrics is not None:
            if not ais_only:
                flow_info = metrics(base_samples, base_log_w, self.flow.log_prob, batch_size=inner_batch_size)
                info.update({"flow_" + k: v for k, v in flow_info.items()})
            info.update({"ais_" + k: v for k, v in metrics(ais_samples, ais_log_w).items()})
        self.set_ais_target(min_is_target=True)
        return info

    # ---- checkpoints ----------------------------------------------------------------------------------
    def save(self, path: str):
        torch.save({'flow': self.flow.state_dict(), 'trans_op': self.transition_operator.state_dict()}, path)

    def load(self, path: str, map_location: Optional[str] = None):
        checkpoint = torch.load(path, map_location=map_location)
        try:
            self.flow.load_state_dict(checkpoint['flow'])
        except RuntimeError:
            try:
                self.flow._nf_model.load_state_dict(checkpoint['flow'])      # raw normflows checkpoint
            except RuntimeError:
                raise RuntimeError('Flow could not be loaded. Perhaps there is a mismatch in the architectures.')
        try:
            self.transition_operator.load_state_dict(checkpoint['trans_op'])
        except RuntimeError:
            warnings.warn('Transition operator could not be loaded. Perhaps there is a mismatch in the architectures.')
        if getattr(self, "annealed_importance_sampler", None) is not None:
            self.annealed_importance_sampler = self._make_ais()
