"""Test infrastructure (not part of the package: the reference's alternative flow adapters are out of scope, SURVEY.md section 2
row 4): a `torch.distributions` object behind the `Distribution` plug-in interface (fab/types_.py:8-27), the kind of base
distribution the reference's own AIS tests use (ais_test.py:98-99) - a GENERIC plug-in for tests/test_gpu_generic_path.py: the
sampler evaluates it with its own torch code and runs the transitions through the generic HIP path."""
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
