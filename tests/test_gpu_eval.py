"""GPU parity: evaluation path (l2norm, pair distance on MFMA, market1501 ranking) through the C ABI
vs the CPU oracle.  Tolerances are stated per precision mode."""
import numpy as np
import pytest
import torch

from oracle import evalrank as E

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from daliid_amd import ops_eval
    return ops_eval


@pytest.mark.parametrize("n,d,eps", [(1, 7, 0.0), (5, 33, 1e-9), (300, 2048, 0.0), (257, 768, 1e-9)])
def test_l2norm_rows_fwd_bwd(ops, n, d, eps):
    g = torch.Generator().manual_seed(n * 131 + d)
    x = torch.randn(n, d, generator=g) * 3.0
    dy = torch.randn(n, d, generator=g)
    xr = x.clone().requires_grad_(True)
    ref = xr / (torch.norm(xr, dim=1, keepdim=True) + eps)       # train_encodersKIT.py:198 / validateModels.py:41
    (ref * dy).sum().backward()
    y, nrm = ops.l2norm_rows(x.cuda(), eps, return_norms=True)
    np.testing.assert_allclose(y.cpu().numpy(), ref.detach().numpy(), rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(nrm.cpu().numpy(), x.norm(dim=1).numpy(), rtol=2e-6)
    dx = ops.l2norm_rows_bwd(x.cuda(), dy.cuda(), eps)
    np.testing.assert_allclose(dx.cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
@pytest.mark.parametrize("nq,ng,d", [(1, 5, 32), (130, 259, 96), (3, 1031, 64), (129, 128, 160), (520, 416, 96), (768, 1280, 64)])
def test_pairdist_exact_on_small_integers(ops, prec, nq, ng, d):
    """Small integers are exact in bf16 and their dot products exact in fp32: the MFMA fragment layout,
    the LDS swizzle, the XCD tile map and every edge tile must reproduce the oracle BIT-EXACTLY.
    (520, 416) and (768, 1280) have INTERIOR 128 x 256 tiles (and ng % 32 == 0): those leave through the persistent kernel's
    line-staged store (pairdist_epilogue_lines), the others through the per-lane epilogue of the edge tiles."""
    g = torch.Generator().manual_seed(nq + 7 * ng + d)
    q = torch.randint(-3, 4, (nq, d), generator=g).float()
    gal = torch.randint(-3, 4, (ng, d), generator=g).float()
    gal[:, 0] += torch.arange(ng).float() % 5            # asymmetric, row-dependent
    q[:, 1] -= torch.arange(nq).float() % 3
    for metric, ref in (("cosine", E.cosine_distmat(q, gal)), ("l2sq", E.l2sq_distmat(q, gal))):
        out = ops.pairdist(q.cuda(), gal.cuda(), metric=metric, precision=prec)
        assert torch.equal(out.cpu(), ref), (metric, (out.cpu() - ref).abs().max())


# Tolerances on unit rows (|q.g| <= 1).  bf16: each operand carries <= 2^-9 relative rounding, so
# |err| <= ~2^-8 * sum|q_i||g_i| <= 3.9e-3 (Cauchy-Schwarz); observed max 1.8e-3.  bf16x3 keeps hi+lo (16 mantissa
# bits) and drops only lo*lo: worst case ~2e-5, observed max 3.5e-6 at d=33 (few, large components) and < 1e-6 at
# d >= 768; the fp32 oracle itself carries ~1e-7 * sqrt(d) of its own.
@pytest.mark.parametrize("prec,atol", [("bf16x3", None), ("bf16", 4e-3)])
@pytest.mark.parametrize("nq,ng,d", [(64, 200, 33), (257, 1031, 2048), (500, 700, 768)])
def test_pairdist_matches_oracle(ops, prec, atol, nq, ng, d):
    if atol is None:
        atol = 8e-6 if d < 128 else 2e-6
    g = torch.Generator().manual_seed(d)
    q = torch.randn(nq, d, generator=g)
    gal = torch.randn(ng, d, generator=g)
    # fused normalise + 1 - q.g == validateModels.py:41-47
    ref = E.validate_features(q, gal)
    out = ops.pairdist(q.cuda(), gal.cuda(), precision=prec, normalize=True).cpu()
    assert (out - ref).abs().max().item() <= atol
    # unnormalised L2^2 (values O(d)): relative tolerance
    ref2 = E.l2sq_distmat(q, gal)
    out2 = ops.pairdist(q.cuda(), gal.cuda(), metric="l2sq", precision=prec).cpu()
    rel = ((out2 - ref2).abs() / ref2.abs().clamp(min=1.0)).max().item()
    assert rel <= (8e-6 if prec == "bf16x3" else 8e-3)


@pytest.mark.parametrize("nq,ng,d", [(256, 751, 2048), (256, 2253, 2048), (256, 2048, 751), (1, 5, 16), (33, 65, 40), (70, 31, 23), (384, 751, 768), (5, 3, 7)])
def test_dot_similarities_of_the_loss_heads_exact_and_on_either_kernel(ops, nq, ng, d):
    """metric="dot" (the heads' fn @ centers^T, fn @ proxies^T, dS @ centers: losses.py:62, :277).  Problems whose 128 x 256 tiles would
    occupy at most a quarter of the CUs leave through the small fp32-MFMA kernel (dot_small_kernel: exact fp32 products, no operand
    pre-pass; d < 16 stays on the distance kernel), DALI_PAIRDIST_SMALL=0 keeps them on the persistent distance kernel: both are bit-exact
    on small integers -- ragged row counts (751 = 23 x 32 + 15), a K that is no multiple of 16 (751: 46 whole steps + a masked one, rows
    that are not 16-byte aligned), fewer k-steps than waves -- and agree with fp64 to fp32 grade on random rows."""
    import os
    from daliid_amd import _lib
    g = torch.Generator().manual_seed(nq + 7 * ng + d)
    q = torch.randint(-3, 4, (nq, d), generator=g).float()
    gal = torch.randint(-3, 4, (ng, d), generator=g).float()
    gal[:, 0] += torch.arange(ng).float() % 5
    q[:, d - 1] -= torch.arange(nq).float() % 3
    ref = q @ gal.t()
    qr, gr = torch.randn(nq, d, generator=g), torch.randn(ng, d, generator=g)
    ref_r = (qr.double() @ gr.double().t())
    try:
        for flag in ("1", "0"):
            os.environ["DALI_PAIRDIST_SMALL"] = flag
            _lib.lib().dali_debug_reload_env()
            out = ops.pairdist(q.cuda(), gal.cuda(), metric="dot")
            assert torch.equal(out.cpu(), ref), (flag, (out.cpu() - ref).abs().max())
            out_r = ops.pairdist(qr.cuda(), gr.cuda(), metric="dot").cpu().double()
            # exact fp32 products summed in fp32 on the small kernel; hi/lo split operands (the lo x lo product dropped, 2^-16 relative) otherwise
            small = flag == "1" and d >= 16
            rel = (out_r - ref_r).abs().max().item() / ref_r.abs().max().item()
            assert rel <= (2e-6 if small else 5e-5), (flag, rel)
    finally:
        os.environ.pop("DALI_PAIRDIST_SMALL", None)
        _lib.lib().dali_debug_reload_env()


def test_pairdist_prepared_equals_one_shot(ops):
    g = torch.Generator().manual_seed(5)
    q = torch.randn(100, 256, generator=g).cuda()
    gal = torch.randn(333, 256, generator=g).cuda()
    a = ops.pairdist(q, gal, normalize=True)
    b = ops.pairdist_prepared(ops.PreparedRows(q, True), ops.PreparedRows(gal, True))
    assert torch.equal(a, b)


def test_pairdist_empty(ops):
    q = torch.zeros(0, 64).cuda()
    gal = torch.randn(10, 64).cuda()
    assert ops.pairdist(q, gal).shape == (0, 10)


def _rank_case(seed, nq, ng, n_ids, n_cams, ties=False):
    rng = np.random.default_rng(seed)
    dist = rng.random((nq, ng)).astype(np.float32)
    if ties:
        dist = np.round(dist * 8) / 8          # many exact ties -> exercises the index tie-break
    return dist, rng.integers(0, n_ids, nq), rng.integers(0, n_ids, ng), rng.integers(0, n_cams, nq), rng.integers(0, n_cams, ng)


@pytest.mark.parametrize("seed,nq,ng,n_ids,n_cams,ties", [(0, 17, 61, 6, 3, False), (1, 40, 1000, 12, 2, True),
                                                          (2, 9, 5000, 3, 4, False), (3, 5, 300, 400, 2, False)])
def test_rank_eval_matches_oracle_on_same_distmat(ops, seed, nq, ng, n_ids, n_cams, ties):
    dist, qp, gp, qc, gc = _rank_case(seed, nq, ng, n_ids, n_cams, ties)
    if seed == 3:
        qp[0] = gp[0]; qc[0] = gc[0] + 1       # guarantee at least one valid query among many invalid ones
    ref_cmc, ref_map = E.eval_market1501(dist, qp, gp, qc, gc, max_rank=50)
    cmc, mAP = ops.rank_eval(torch.from_numpy(dist).cuda(), qp, gp, qc, gc, max_rank=50)
    np.testing.assert_allclose(cmc, ref_cmc, atol=1e-6)
    assert abs(mAP - ref_map) < 1e-6


def test_rank_eval_string_ids_and_known_answer(ops):
    g_pids = np.array(["a", "b", "a", "c"]); g_cams = np.array(["0", "0", "1", "0"])
    dist = torch.tensor([[0.9, 0.1, 0.5, 0.3], [0.2, 0.8, 0.1, 0.4], [0.1, 0.2, 0.3, 0.4]])
    cmc, mAP = ops.rank_eval(dist.cuda(), np.array(["a", "b", "z"]), g_pids, np.array(["0", "1", "5"]), g_cams, max_rank=4)
    np.testing.assert_allclose(cmc, [0.0, 0.0, 0.5, 1.0])
    assert abs(mAP - (1 / 3 + 1 / 4) / 2) < 1e-7


def test_validate_pipeline_map_within_1e3(ops):
    """normalise -> distmat -> CMC/mAP end to end on synthetic ids (north star: mAP within 1e-3)."""
    q, g, qp, gp, qc, gc = E.synthetic_reid_set(60, 20, 3, 64, noise=2.0, seed=12)
    ref_d = E.validate_features(q, g)
    ref_cmc, ref_map = E.eval_market1501(ref_d.numpy(), qp, gp, qc, gc)
    assert 0.08 < ref_map < 0.999         # a non-trivial ranking problem
    for prec, tol in (("bf16x3", 2e-5), ("bf16", 1e-3)):
        d = ops.pairdist(q.cuda(), g.cuda(), precision=prec, normalize=True)
        cmc, mAP = ops.rank_eval(d, qp, gp, qc, gc)
        assert abs(mAP - ref_map) < tol, (prec, mAP, ref_map)
        np.testing.assert_allclose(cmc, ref_cmc, atol=(1.0 / len(qp) + 1e-6) if prec == "bf16" else 2e-5)


def test_config5_full_size_properties(ops):
    """BASELINE config 5 (10k x 100k x 2048): size-independent checks.
    (a) sampled entries vs fp64 dot products; (b) linearity: sum_g out[q,g] = ng - q . sum_g g;
    (c) role symmetry: D(Q,G)[i,j] == D(G,Q)[j,i] on a sampled block."""
    nq, ng, d = 10000, 100000, 2048
    gen = torch.Generator(device="cuda").manual_seed(12)
    q = torch.randn(nq, d, device="cuda", generator=gen)
    g = torch.randn(ng, d, device="cuda", generator=gen)
    out = ops.pairdist(q, g, precision="bf16x3", normalize=True)
    qn = (q / q.norm(dim=1, keepdim=True)).double()
    gn = (g / g.norm(dim=1, keepdim=True)).double()
    idx_q = torch.randint(0, nq, (4096,), device="cuda", generator=gen)
    idx_g = torch.randint(0, ng, (4096,), device="cuda", generator=gen)
    ref = 1.0 - (qn[idx_q] * gn[idx_g]).sum(1)
    assert (out[idx_q, idx_g].double() - ref).abs().max().item() < 2e-6
    rows = idx_q[:64]
    lin = ng - qn[rows] @ gn.sum(0)
    assert (out[rows].double().sum(1) - lin).abs().max().item() < 1e-2   # 1e5 terms of ~1.0 +- 1e-7
    sub_q, sub_g = q[:300], g[5000:5600]
    a = ops.pairdist(sub_q, sub_g, normalize=True)
    b = ops.pairdist(sub_g, sub_q, normalize=True)
    assert (a - b.T).abs().max().item() < 1e-6
    assert (out[:300, 5000:5600] - a).abs().max().item() == 0.0           # tile position does not change values
    del out
    torch.cuda.empty_cache()


def test_rank_eval_large_identities_second_tier_and_limit(ops):
    """The ranking kernel keeps a query's same-identity gallery entries in LDS: up to 512 in the first launch, up to 4096 in the
    second; beyond that the documented limit is an error, not a wrong number."""
    from daliid_amd._lib import DaliError
    rng = np.random.default_rng(3)
    ng, nq = 6000, 12
    gp = np.concatenate((np.zeros(700, np.int64), np.ones(3000, np.int64), rng.integers(2, 40, ng - 3700)))   # identity 0: 700 entries, identity 1: 3000
    rng.shuffle(gp)
    gc = rng.integers(0, 3, ng)
    qp = np.array([0, 1, 0, 1, 5, 7, 9, 1, 0, 38, 39, 1000])                                # the last query's identity is not in the gallery
    qc = rng.integers(0, 3, nq)
    dist = rng.random((nq, ng)).astype(np.float32)
    ref_cmc, ref_map = E.eval_market1501(dist, qp, gp, qc, gc, max_rank=50)
    cmc, mAP = ops.rank_eval(torch.from_numpy(dist).cuda(), qp, gp, qc, gc, max_rank=50)
    np.testing.assert_allclose(cmc, ref_cmc, atol=1e-6)
    assert abs(mAP - ref_map) < 1e-6
    gp2 = np.zeros(5000, np.int64)
    with pytest.raises(DaliError, match="4096"):
        ops.rank_eval(torch.from_numpy(dist[:, :5000].copy()).cuda(), qp, gp2, qc, gc[:5000])
    # sparse identity codes straight through the code-level entry: refused with status 2, never mis-indexed
    t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.int32)).cuda()
    wide = gp.copy(); wide[0] = 1 << 29
    o = ops.rank_eval_codes(torch.from_numpy(dist).cuda(), t(qp), t(wide), t(qc), t(gc))
    assert int(o["status"].item()) == 2


def test_config5_rank_eval_full_size_properties(ops):
    """configs[4], CMC/mAP leg at 10k x 100k (1000 identities x 100 gallery entries, 10 queries per identity): per-query AP and
    first-hit rank of 64 sampled queries against a brute-force argsort of their rows; the aggregate against the per-query values;
    perfectly separated identities -> mAP = 1, rank-1 = 1."""
    nq, ng, n_ids = 10000, 100000, 1000
    rng = np.random.default_rng(12)
    g_pids = np.repeat(np.arange(n_ids), 100); q_pids = np.repeat(np.arange(n_ids), 10)
    g_cams = rng.integers(0, 6, ng); q_cams = rng.integers(0, 6, nq)
    gen = torch.Generator(device="cuda").manual_seed(12)
    dist = torch.rand(nq, ng, device="cuda", generator=gen)
    cmc, mAP, ap, first = ops.rank_eval(dist, q_pids, g_pids, q_cams, g_cams, max_rank=50, return_per_query=True)
    valid = first >= 0
    assert valid.sum() == nq                                        # 100 entries over 6 cameras: every query keeps a match
    assert abs(mAP - float(ap[valid].astype(np.float64).mean())) < 1e-6
    hist = np.bincount(first[valid], minlength=ng)[:50].cumsum() / valid.sum()
    np.testing.assert_allclose(cmc, hist, atol=1e-6)
    sample = rng.choice(nq, 64, replace=False)
    rows = dist[torch.from_numpy(sample).cuda()].cpu().numpy()
    for r, q in zip(rows, sample):
        order = np.lexsort((np.arange(ng), r))                       # (distance, index): the kernel's tie order
        keep = ~((g_pids[order] == q_pids[q]) & (g_cams[order] == q_cams[q]))
        m = (g_pids[order] == q_pids[q])[keep]
        pos = np.flatnonzero(m)
        ref_ap = float(np.mean((np.arange(len(pos)) + 1) / (pos + 1)))
        assert first[q] == pos[0] and abs(ap[q] - ref_ap) < 1e-6, (q, first[q], pos[0], ap[q], ref_ap)
    # separated identities: same-identity pairs closest
    same = torch.from_numpy(q_pids).cuda()[:, None] == torch.from_numpy(g_pids).cuda()[None, :]
    dist.mul_(0.5).add_(0.5)
    dist[same] -= 0.5
    del same
    cmc2, mAP2 = ops.rank_eval(dist, q_pids, g_pids, q_cams, g_cams)
    assert abs(mAP2 - 1.0) < 1e-6 and abs(cmc2[0] - 1.0) < 1e-6
    del dist
    torch.cuda.empty_cache()


# ---- gallery-sharded ranking: bins summed over shards == the single-GPU ranking, bit for bit --------------------------
def _shard_case(seed, nq, ng, n_ids, n_cams, ties):
    rng = np.random.default_rng(seed)
    q_pids = rng.integers(0, n_ids, nq); g_pids = rng.integers(0, n_ids, ng)
    q_pids[:3] = n_ids + 5                                   # queries whose identity is not in the gallery at all: invalid
    q_cams = rng.integers(0, n_cams, nq); g_cams = rng.integers(0, n_cams, ng)
    d = rng.random((nq, ng), dtype=np.float32)
    if ties:
        d = np.round(d * 20) / 20                            # many exact ties: ordered by GLOBAL gallery index
    return torch.from_numpy(d), q_pids.astype(str), g_pids.astype(str), q_cams.astype(str), g_cams.astype(str)


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("nq,ng,n_ids,ties", [(37, 1000, 11, False), (64, 5000, 40, True), (5, 300, 2, True)])
def test_sharded_ranking_equals_whole_gallery_ranking(ops, world, nq, ng, n_ids, ties):
    """The three device steps of rank_eval_sharded run for every shard in ONE process (the two collectives replaced by a stack and a sum):
    per-query AP, first-hit rank, CMC and mAP must equal dali_rank_eval's on the whole matrix bit for bit -- string ids, junk removal,
    invalid queries, exact ties, an empty last shard (world 8 on 300 entries = 128-row slices) included."""
    dist, qp_s, gp_s, qc_s, gc_s = _shard_case(nq + ng, nq, ng, n_ids, 3, ties)
    D = dist.cuda()
    cmc_ref, map_ref, ap_ref, fr_ref = ops.rank_eval(D, qp_s, gp_s, qc_s, gc_s, max_rank=50, return_per_query=True)
    bounds = ops.shard_bounds(ng, world)
    t = lambda a: torch.from_numpy(a).cuda()
    shards = []
    for r in range(world):
        lo, hi = bounds[r], bounds[r + 1]
        qp, gp = ops.factorize_ids(qp_s, gp_s[lo:hi]); qc, gc = ops.factorize_ids(qc_s, gc_s[lo:hi])
        shards.append((D[:, lo:hi].contiguous(), t(qp), t(gp), t(qc), t(gc), lo))
    cap = max(int(np.unique(gp_s[bounds[r]:bounds[r + 1]], return_counts=True)[1].max()) if bounds[r + 1] > bounds[r] else 1 for r in range(world))
    ks, cs = zip(*[ops.rank_shard_matches(*sh, cap)[:2] for sh in shards])
    keys_all, counts_all = torch.stack(ks), torch.stack(cs)
    bins_cap = min(world * cap, ops.RANK_PMAX)
    bins = sum(ops.rank_shard_bins(*sh, keys_all, counts_all, bins_cap)[0] for sh in shards)
    o = ops.rank_shard_finish(bins, counts_all, min(50, ng))
    assert np.array_equal(o["ap"].cpu().numpy(), ap_ref) and np.array_equal(o["first_rank"].cpu().numpy(), fr_ref)
    assert np.array_equal(o["cmc"].cpu().numpy(), cmc_ref) and float(o["map64"].item()) == map_ref
    assert (fr_ref[:3] == -1).all() and (fr_ref[3:] >= 0).any()
