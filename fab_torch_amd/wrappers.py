"""`WrappedTorchDist` - a `torch.distributions` object behind the `Distribution` plug-in interface, as in
fab/wrappers/torch.py:7-24 (the base distribution of the reference's own AIS tests, ais_test.py:98-99).  It is a
GENERIC plug-in: the sampler evaluates it with its own torch code and runs the transitions through the generic HIP
path (transition_operators.py)."""
from typing import Tuple

import torch


class WrappedTorchDist:
    def __init__(self, torch_dist: torch.distributions.Distribution):
        self._torch_dist = torch_dist

    def sample_and_log_prob(self, shape: Tuple[int, ...]) -> Tuple[torch.Tensor, torch.Tensor]:
        samples = self._torch_dist.sample(shape)
        return samples, self._torch_dist.log_prob(samples)

    def sample(self, shape: Tuple) -> torch.Tensor:
        return self._torch_dist.sample(shape)

    def log_prob(self, x: torch.Tensor) -> torch.Tensor:
        return self._torch_dist.log_prob(x)

    @property
    def event_shape(self) -> Tuple[int, ...]:
        return self._torch_dist.event_shape
