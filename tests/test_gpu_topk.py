"""fabhip_topk (radix select + index-ordered compaction + LDS bitonic sort) against the exact specification:
the k largest keys in descending order, ties by ascending index — numpy lexsort on the CPU; and against
torch.topk's key multiset.  Bit-exact indices (integer work)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

fa = pytest.importorskip("fab_torch_amd")
from fab_torch_amd.buffer import topk_indices, sample_without_replacement  # noqa: E402

DEV = "cuda"


def spec_topk(keys: np.ndarray, k: int) -> np.ndarray:
    order = np.lexsort((np.arange(len(keys)), -keys.astype(np.float64)))     # by key descending, then index ascending
    return order[:k]


@pytest.mark.parametrize("n,k", [(1, 1), (5, 3), (4096, 4096), (4097, 17), (100_000, 1000), (512_000, 16384),
                                 (1 << 20, 16384), (70_000, 16000), (300_000, 40_000)])
def test_topk_matches_the_specification(n, k):
    g = torch.Generator().manual_seed(n + k)
    keys = torch.randn(n, generator=g) * 5
    spec = spec_topk(keys.numpy(), k)
    if k <= 16384:
        assert np.array_equal(topk_indices(keys.to(DEV), k, sorted=True).cpu().numpy(), spec)
    # unsorted mode: the same set, in ascending index order
    assert np.array_equal(topk_indices(keys.to(DEV), k).cpu().numpy(), np.sort(spec))


def test_topk_ties_infinities_and_duplicates():
    g = torch.Generator().manual_seed(3)
    n, k = 50_000, 9000
    keys = torch.randint(-20, 20, (n,), generator=g).float()                 # ~1250 copies of every value
    keys[::7] = -float("inf")                                                # dead buffer entries
    keys[5::1001] = float("inf")
    spec = spec_topk(keys.numpy(), k)
    assert np.array_equal(topk_indices(keys.to(DEV), k, sorted=True).cpu().numpy(), spec)
    assert np.array_equal(topk_indices(keys.to(DEV), k).cpu().numpy(), np.sort(spec))
    # all keys equal: the first k indices
    same = torch.full((20_000,), 1.5)
    assert np.array_equal(topk_indices(same.to(DEV), 700).cpu().numpy(), np.arange(700))
    # same key multiset as torch.topk
    keys2 = torch.randn(300_000, generator=g)
    mine = keys2[topk_indices(keys2.to(DEV), 12345, sorted=True).cpu()]
    ref = torch.topk(keys2, 12345, sorted=True).values
    assert torch.equal(mine, ref)


def test_buffer_sampling_without_replacement_uses_the_kernel():
    torch.manual_seed(0)
    logits = torch.randn(200_000, device=DEV)
    logits[:1000] = -float("inf")
    idx = sample_without_replacement(logits, 16384)
    assert idx.shape == (16384,) and idx.dtype == torch.int64
    assert len(torch.unique(idx)) == 16384                                   # without replacement
    assert int((idx < 1000).sum()) == 0                                      # dead entries are never drawn
    # heavier logits are drawn more often: mean logit of the sample exceeds the population mean
    assert float(logits[idx].mean()) > float(logits[1000:].mean()) + 0.3
    big = sample_without_replacement(logits, 50000)                          # unsorted mode has no k limit
    assert len(torch.unique(big)) == 50000 and int((big < 1000).sum()) == 0
