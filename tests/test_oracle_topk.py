"""The C restatement of torch.topk's CPU tie order (oracle/topk_ref.c) against torch.topk itself,
which is what the reference calls (SFTS.py:155, Frequency.py:58).  Order AND set (fixture F7 of
SURVEY.md 8(c), generated live because torch CPU is available everywhere)."""
import pytest
import torch


@pytest.mark.parametrize("n", [16, 128, 129, 192, 512])
@pytest.mark.parametrize("k", [1, 2, 10, 17])
def test_topk_matches_torch(oracle, n, k):
    if k > n:
        pytest.skip("k>n")
    g = torch.Generator().manual_seed(n * 131 + k)
    cases = [torch.randint(90, 110, (256, n), generator=g, dtype=torch.int32),     # tie-heavy counts
             torch.rand(256, n, generator=g),
             torch.randint(0, 5, (256, n), generator=g).float(),                  # tie-heavy floats
             torch.zeros(4, n)]                                                   # all equal
    for x in cases:
        ref = torch.topk(x, k, dim=1).indices
        got = oracle.topk_indices(x, k)
        assert torch.equal(ref, got)


def test_topk_nan_first(oracle):
    x = torch.rand(8, 128)
    x[:, 5] = float("nan")
    assert torch.equal(torch.topk(x, 2, dim=1).indices, oracle.topk_indices(x, 2))
