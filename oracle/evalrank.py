"""Oracle: gallery distance + market1501 CMC/mAP on CPU (fp32 torch / numpy).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Distance follows validateModels.py:41-47 (``q/|q|``, ``g/|g|``, ``1 - q @ g.T``; no epsilon).
Ranking restates the third-party ``torchreid.metrics.evaluate_rank(...,
use_metric_cuhk03=False)`` (KaiyangZhou/deep-person-reid; unpinned, absent from the image;
reference call sites validateModels.py:68, evaluate.py:312): the public market1501 protocol.
The reference holds no fixture for it -> "parity unpinned" at that boundary; pinned by the
hand-computed cases in tests/test_oracle_rank.py and by ``eval_market1501_bruteforce``.

Tie rule: the third party uses ``np.argsort`` (introsort, order of exact ties unspecified).
This oracle and the HIP path both break exact distance ties by ascending gallery index
(``kind='stable'``); goldens avoid exact ties between a match and a non-match.
"""
import numpy as np
import torch


def l2_normalize_rows(x, eps=0.0):
    """validateModels.py:41-42 (eps=0) / train_encodersKIT.py:198 (eps=1e-9, added to the norm)."""
    return x / (torch.norm(x, dim=1, keepdim=True) + eps)


def cosine_distmat(q, g):
    """validateModels.py:47 on already-normalised rows."""
    return 1.0 - torch.mm(q, g.T)


def l2sq_distmat(q, g):
    """Squared euclidean distance; for unit rows equals 2*(1-q.g) (the commented cdist variant,
    validateModels.py:45, is its square root and ranks identically)."""
    return (q * q).sum(1, keepdim=True) + (g * g).sum(1).reshape(1, -1) - 2.0 * torch.mm(q, g.T)


def validate_features(q_fvs, g_fvs):
    """validateModels.py:41-47 from raw features to distmat."""
    return cosine_distmat(l2_normalize_rows(q_fvs), l2_normalize_rows(g_fvs))


def fused_distmat(q1, g1, q2, g2, mags1=None, mags2=None):
    """evaluateCleanATModels.py:114-126 and :154-157: d_m = 1 - unit(q_m) @ unit(g_m).T for the two models; with
    magnitudes (norms of the embeddings under a pooling mode, :249-256) w_m[i,j] = max(qmag_m[i], gmag_m[j]) and
    distmat = (w1*d1 + w2*d2)/(w1 + w2); without, the simple ensemble (d1 + d2)/2."""
    d1, d2 = validate_features(q1, g1), validate_features(q2, g2)
    if mags1 is None:
        return (d1 + d2) / 2
    w1 = torch.maximum(mags1[0].reshape(-1, 1).repeat(1, g1.shape[0]), mags1[1].reshape(1, -1).repeat(q1.shape[0], 1))
    w2 = torch.maximum(mags2[0].reshape(-1, 1).repeat(1, g2.shape[0]), mags2[1].reshape(1, -1).repeat(q2.shape[0], 1))
    return (w1 * d1 + w2 * d2) / (w1 + w2)


def _codes(*arrays):
    """Map arbitrary (string) id arrays to shared int64 codes; equality is all that matters."""
    allv = np.concatenate([np.asarray(a).ravel() for a in arrays])
    _, inv = np.unique(allv, return_inverse=True)
    out, o = [], 0
    for a in arrays:
        n = np.asarray(a).size
        out.append(inv[o:o + n].astype(np.int64))
        o += n
    return out


def eval_market1501(distmat, q_pids, g_pids, q_camids, g_camids, max_rank=50):
    """market1501 protocol -> (cmc[max_rank] float32, mAP float).

    For every query: order the gallery by ascending distance; drop gallery entries with the
    same pid AND the same camid as the query; skip the query if no match remains; CMC row =
    cumulative matches clipped to 1, cut to ``max_rank``; AP = mean over matches of
    (matches so far / position).  CMC = mean over valid queries, mAP = mean AP.
    """
    distmat = np.asarray(distmat)
    nq, ng = distmat.shape
    q_pids, g_pids = _codes(q_pids, g_pids)
    q_camids, g_camids = _codes(q_camids, g_camids)
    max_rank = min(max_rank, ng)
    cmc_sum = np.zeros(max_rank, dtype=np.float64)
    aps = []
    for qi in range(nq):
        order = np.argsort(distmat[qi], kind="stable")
        same_pid = g_pids[order] == q_pids[qi]
        junk = same_pid & (g_camids[order] == q_camids[qi])
        hits = same_pid[~junk].astype(np.float64)
        if not hits.any():
            continue
        c = np.minimum(hits.cumsum(), 1.0)
        row = np.ones(max_rank)
        row[:min(max_rank, c.size)] = c[:max_rank]
        if c.size < max_rank:          # fewer kept entries than max_rank: extend the last value
            row[c.size:] = c[-1]
        cmc_sum += row
        prec = hits.cumsum() / np.arange(1, hits.size + 1)
        aps.append(float((prec * hits).sum() / hits.sum()))
    if not aps:
        raise AssertionError("Error: all query identities do not appear in gallery")
    return (cmc_sum / len(aps)).astype(np.float32), float(np.mean(aps))


def eval_market1501_bruteforce(distmat, q_pids, g_pids, q_camids, g_camids, max_rank=50):
    """O(Nq*Ng^2) twin written from the definition (no argsort): rank of a kept gallery entry =
    number of kept entries strictly closer, plus equal-distance kept entries with a smaller index."""
    distmat = np.asarray(distmat, dtype=np.float64)
    nq, ng = distmat.shape
    q_pids, g_pids = _codes(q_pids, g_pids)
    q_camids, g_camids = _codes(q_camids, g_camids)
    max_rank = min(max_rank, ng)
    cmc_sum = np.zeros(max_rank)
    aps = []
    for qi in range(nq):
        keep = ~((g_pids == q_pids[qi]) & (g_camids == q_camids[qi]))
        pos = np.where(keep & (g_pids == q_pids[qi]))[0]
        if pos.size == 0:
            continue
        ranks = []
        for p in pos:
            d = distmat[qi, p]
            before = keep & ((distmat[qi] < d) | ((distmat[qi] == d) & (np.arange(ng) < p)))
            ranks.append(int(before.sum()))
        ranks = np.sort(np.asarray(ranks))
        row = (np.arange(max_rank) >= ranks[0]).astype(np.float64)
        cmc_sum += row
        aps.append(float(np.mean((np.arange(ranks.size) + 1.0) / (ranks + 1.0))))
    return (cmc_sum / len(aps)).astype(np.float32), float(np.mean(aps))


def synthetic_reid_set(n_ids, per_id_gallery, per_id_query, dim, noise=0.5, n_cams=6, seed=12,
                       dtype=torch.float32):
    """SURVEY 8(d) config-5 generator: rows = normalize(id_centroid + noise*randn)."""
    g = torch.Generator().manual_seed(seed)
    cent = torch.randn(n_ids, dim, generator=g, dtype=dtype)
    g_pids = torch.arange(n_ids).repeat_interleave(per_id_gallery)
    q_pids = torch.arange(n_ids).repeat_interleave(per_id_query)
    gal = cent[g_pids] + noise * torch.randn(g_pids.numel(), dim, generator=g, dtype=dtype)
    qry = cent[q_pids] + noise * torch.randn(q_pids.numel(), dim, generator=g, dtype=dtype)
    g_cams = torch.randint(0, n_cams, (g_pids.numel(),), generator=g)
    q_cams = torch.randint(0, n_cams, (q_pids.numel(),), generator=g)
    return qry, gal, q_pids.numpy(), g_pids.numpy(), q_cams.numpy(), g_cams.numpy()
