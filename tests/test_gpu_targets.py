"""GPU: class centers + farthest-point proxies (dali_class_targets) against the reference's golden picks and the oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import trainstep as TS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from daliid_amd import train_encodersKIT
    return train_encodersKIT


class _PinFirst:
    """Pins np.random.choice to a scripted sequence (the reference draws the first proxy from numpy's global stream)."""

    def __init__(self, picks):
        self.picks, self.i, self.orig = list(picks), 0, np.random.choice

    def __enter__(self):
        def choice(n, *a, **k):
            v = self.picks[self.i]; self.i += 1
            assert 0 <= v < n
            return v
        np.random.choice = choice
        return self

    def __exit__(self, *exc):
        np.random.choice = self.orig


def test_picks_match_reference_golden(T):
    z = load_golden("proxies.npz")                                   # made by the reference's selectProxiesByTriagulation
    for name, tol in (("X40", 1e-4), ("X3", 1e-5), ("X1", 0.0)):
        X = torch.from_numpy(z[name]).cuda()
        first = int(z[name + "_first"]) if name + "_first" in z else 0
        with _PinFirst([first]):
            idx, md = T.selectProxiesByTriagulation(X, num_proxies=5)
        assert idx.dtype == torch.long and idx.tolist() == z[name + "_idx"].tolist()
        assert abs(md - float(z[name + "_maxdist"])) <= tol


@pytest.mark.parametrize("d,sizes", [(2048, [17, 1, 3, 100, 5, 4, 6, 33]), (64, [2, 9, 300, 5]), (768, [12] * 40)])
def test_build_centers_and_proxies_matches_oracle(T, d, sizes):
    rng = np.random.default_rng(3)
    labels = np.repeat(np.arange(len(sizes)) * 7 + 3, sizes)
    perm = rng.permutation(labels.shape[0])
    labels = labels[perm]                                            # identities interleaved, as in a real train list
    g = torch.Generator().manual_seed(4)
    fvs = torch.randn(labels.shape[0], d, generator=g) * 2.0 + torch.randn(1, d, generator=g)
    picks = [int(rng.integers(0, n)) for n in sizes]
    c_ref, cl_ref, p_ref, pl_ref = TS.build_centers_and_proxies(fvs, labels, 5, picks)
    with _PinFirst(picks):
        c, cl, p, pl, mean_max = T.build_centers_and_proxies(fvs.cuda(), labels, 5)
    assert cl.tolist() == cl_ref.tolist() and pl.tolist() == pl_ref.tolist()
    assert p.shape == p_ref.shape
    torch.testing.assert_close(c.cpu(), c_ref, rtol=0, atol=2e-6)
    torch.testing.assert_close(p.cpu(), p_ref, rtol=0, atol=2e-6)    # same rows picked, same normalisation
    md = []
    for ci, lab in enumerate(cl_ref):
        rows = fvs[torch.from_numpy(labels == lab)]
        md.append(TS.select_proxies_farthest_point(rows, 5, picks[ci])[1])
    assert abs(mean_max - float(np.mean(md))) < 1e-3


def test_large_identity_and_unit_norms(T):
    g = torch.Generator().manual_seed(9)
    fvs = torch.randn(3000, 256, generator=g)
    labels = np.array([0] * 2900 + [1] * 100)
    with _PinFirst([2899, 0]):
        c, cl, p, pl, _ = T.build_centers_and_proxies(fvs.cuda(), labels, 5)
    c_ref, _, p_ref, _ = TS.build_centers_and_proxies(fvs, labels, 5, [2899, 0])
    torch.testing.assert_close(p.cpu(), p_ref, rtol=0, atol=2e-6)
    torch.testing.assert_close(c.cpu(), c_ref, rtol=0, atol=2e-6)
    assert torch.allclose(p.norm(dim=1), torch.ones(10, device="cuda"), atol=1e-6)


def test_rejects_bad_arguments(T):
    from daliid_amd import ops_eval, _lib
    fvs = torch.randn(8, 6, device="cuda")                           # d not a multiple of 4
    order = torch.arange(8, dtype=torch.int32, device="cuda")
    bounds = torch.tensor([0, 8], dtype=torch.int32, device="cuda")
    first = torch.zeros(1, dtype=torch.int32, device="cuda")
    with pytest.raises(_lib.DaliError):
        ops_eval.class_targets(fvs, order, bounds, first, 5)
    with pytest.raises(_lib.DaliError):
        ops_eval.class_targets(torch.randn(8, 8, device="cuda"), order, bounds, first, 17)
