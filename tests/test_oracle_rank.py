"""market1501 CMC/mAP oracle: hand-computed known answers + brute-force twin + identities.

torchreid.metrics.evaluate_rank is third-party and absent here, and the reference holds no fixture
for it: these known-answer cases are the pin."""
import numpy as np
import torch

from oracle import evalrank as E


def test_known_answer_single_query():
    # gallery sorted by distance: idx 2 (d=.1, pid 7 cam 1 -> same pid+cam as query: junk),
    # idx 0 (.2, pid 3), idx 3 (.3, pid 7 cam 2 -> match), idx 1 (.4, pid 7 cam 0 -> match), idx 4 (.5, pid 9)
    dist = np.array([[0.2, 0.4, 0.1, 0.3, 0.5]], dtype=np.float32)
    cmc, mAP = E.eval_market1501(dist, np.array([7]), np.array([3, 7, 7, 7, 9]), np.array([1]),
                                 np.array([0, 0, 1, 2, 0]), max_rank=4)
    # kept order: [pid3, match, match, pid9] -> cmc = [0,1,1,1]; AP = (1/2 + 2/3)/2
    np.testing.assert_allclose(cmc, [0, 1, 1, 1])
    assert abs(mAP - (0.5 + 2.0 / 3.0) / 2.0) < 1e-12


def test_known_answer_two_queries_and_invalid_query():
    g_pids = np.array(["a", "b", "a", "c"])
    g_cams = np.array(["0", "0", "1", "0"])
    dist = np.array([[0.9, 0.1, 0.5, 0.3],     # q0 pid a cam 0: gallery 0 is junk; order b, c, a(idx2) -> rank 3
                     [0.2, 0.8, 0.1, 0.4],     # q1 pid b cam 1: order a(2), a(0), c, b -> rank 4
                     [0.1, 0.2, 0.3, 0.4]])    # q2 pid z: no match -> skipped
    cmc, mAP = E.eval_market1501(dist, np.array(["a", "b", "z"]), g_pids, np.array(["0", "1", "5"]), g_cams, max_rank=4)
    np.testing.assert_allclose(cmc, [0.0, 0.0, 0.5, 1.0])
    assert abs(mAP - (1.0 / 3.0 + 1.0 / 4.0) / 2.0) < 1e-12


def test_bruteforce_twin_random():
    rng = np.random.default_rng(0)
    for trial in range(5):
        nq, ng = 17, 61
        dist = rng.random((nq, ng)).astype(np.float32)
        qp, gp = rng.integers(0, 6, nq), rng.integers(0, 6, ng)
        qc, gc = rng.integers(0, 3, nq), rng.integers(0, 3, ng)
        a = E.eval_market1501(dist, qp, gp, qc, gc, max_rank=50)
        b = E.eval_market1501_bruteforce(dist, qp, gp, qc, gc, max_rank=50)
        np.testing.assert_allclose(a[0], b[0], atol=1e-6)
        assert abs(a[1] - b[1]) < 1e-12


def test_perfectly_separated_ids_give_map_one():
    q, g, qp, gp, qc, gc = E.synthetic_reid_set(20, 8, 2, 64, noise=0.01, seed=3)
    d = E.validate_features(q, g)
    cmc, mAP = E.eval_market1501(d.numpy(), qp, gp, qc, gc)
    assert abs(mAP - 1.0) < 1e-12 and cmc[0] == 1.0


def test_l2sq_equals_2x_cosine_for_unit_rows():
    g = torch.Generator().manual_seed(0)
    q = E.l2_normalize_rows(torch.randn(9, 33, generator=g))
    r = E.l2_normalize_rows(torch.randn(14, 33, generator=g))
    np.testing.assert_allclose(E.l2sq_distmat(q, r).numpy(), 2 * E.cosine_distmat(q, r).numpy(), atol=2e-6)


def test_fused_distmat_known_answer():
    """evaluateCleanATModels.py:154-157 by hand: one query, two gallery rows, two models."""
    import torch
    q1, g1 = torch.tensor([[2.0, 0.0]]), torch.tensor([[3.0, 0.0], [0.0, 5.0]])          # d1 = [0, 1]
    q2, g2 = torch.tensor([[0.0, 1.0]]), torch.tensor([[1.0, 0.0], [0.0, 4.0]])          # d2 = [1, 0]
    simple = E.fused_distmat(q1, g1, q2, g2)
    assert torch.allclose(simple, torch.tensor([[0.5, 0.5]]))
    m1 = (torch.tensor([[2.0]]), torch.tensor([[3.0], [5.0]]))                             # w1 = max(2,[3,5]) = [3,5]
    m2 = (torch.tensor([[1.0]]), torch.tensor([[1.0], [4.0]]))                             # w2 = max(1,[1,4]) = [1,4]
    fused = E.fused_distmat(q1, g1, q2, g2, m1, m2)
    assert torch.allclose(fused, torch.tensor([[(3 * 0 + 1 * 1) / 4.0, (5 * 1 + 4 * 0) / 9.0]]))
