"""Python entry points of the evaluation kernels (thin wrappers over the C ABI)."""
import numpy as np
import torch

from . import _lib

METRIC_COSINE, METRIC_L2SQ, METRIC_DOT = 0, 1, 2
PREC_BF16X3, PREC_BF16 = 0, 1
_PREC = {"bf16x3": PREC_BF16X3, "bf16": PREC_BF16}
_METRIC = {"cosine": METRIC_COSINE, "l2sq": METRIC_L2SQ, "dot": METRIC_DOT}


def l2norm_rows(x, eps=0.0, return_norms=False):
    """y = x / (|x| + eps) per row (validateModels.py:41-42 eps=0; train_encodersKIT.py:198 eps=1e-9)."""
    assert x.dim() == 2
    y = torch.empty_like(x)
    norms = torch.empty(x.shape[0], device=x.device, dtype=torch.float32) if return_norms else None
    _lib.check(_lib.lib().dali_l2norm_rows(_lib.ctx(x.device), _lib.stream_ptr(), _lib.ptr(x, torch.float32, "x"),
                                            x.shape[0], x.shape[1], float(eps), _lib.ptr(y), _lib.ptr(norms)),
               "dali_l2norm_rows")
    return (y, norms) if return_norms else y


def l2norm_rows_bwd(x, dy, eps=0.0):
    dx = torch.empty_like(x)
    _lib.check(_lib.lib().dali_l2norm_rows_bwd(_lib.ctx(x.device), _lib.stream_ptr(), _lib.ptr(x, torch.float32, "x"),
                                                _lib.ptr(dy, torch.float32, "dy"), x.shape[0], x.shape[1], float(eps),
                                                _lib.ptr(dx)), "dali_l2norm_rows_bwd")
    return dx


def pairdist(q, g, metric="cosine", precision="bf16x3", normalize=False, out=None):
    """distmat[nq,ng] fp32 on the GPU: 1 - q@g.T (validateModels.py:47) or squared L2."""
    assert q.dim() == 2 and g.dim() == 2 and q.shape[1] == g.shape[1]
    nq, ng, d = q.shape[0], g.shape[0], q.shape[1]
    if out is None:
        out = torch.empty(nq, ng, device=q.device, dtype=torch.float32)
    if nq == 0 or ng == 0:
        return out
    _lib.check(_lib.lib().dali_pairdist(_lib.ctx(q.device), _lib.stream_ptr(), _lib.ptr(q, torch.float32, "q"),
                                         _lib.ptr(g, torch.float32, "g"), nq, ng, d, _METRIC[metric], _PREC[precision],
                                         int(bool(normalize)), _lib.ptr(out, torch.float32, "out")), "dali_pairdist")
    return out


class PreparedRows:
    """bf16 operand image of a feature matrix (dali_pairdist_prepare): opaque bytes + the squared row norms."""

    def __init__(self, x, normalize=False, precision="bf16x3"):
        assert x.dim() == 2
        self.n, self.d = x.shape
        self.precision = precision
        L = _lib.lib()
        nbytes = int(L.dali_pairdist_operand_bytes(self.n, self.d, _PREC[precision]))
        self.image = torch.empty(max(nbytes, 16), device=x.device, dtype=torch.uint8)
        self.sq = torch.empty(max(self.n, 1), device=x.device, dtype=torch.float32)
        _lib.check(L.dali_pairdist_prepare(_lib.ctx(x.device), _lib.stream_ptr(), _lib.ptr(x, torch.float32, "x"), self.n, self.d,
                                           int(bool(normalize)), _PREC[precision], _lib.ptr(self.image), _lib.ptr(self.sq)),
                   "dali_pairdist_prepare")


def pairdist_prepared(qp, gp, metric="cosine", out=None):
    assert qp.d == gp.d and qp.precision == gp.precision
    if out is None:
        out = torch.empty(qp.n, gp.n, device=qp.image.device, dtype=torch.float32)
    _lib.check(_lib.lib().dali_pairdist_prepared(_lib.ctx(out.device), _lib.stream_ptr(), _lib.ptr(qp.image), _lib.ptr(qp.sq),
                                                  _lib.ptr(gp.image), _lib.ptr(gp.sq), qp.n, gp.n, qp.d, _METRIC[metric],
                                                  _PREC[qp.precision], _lib.ptr(out, torch.float32, "out")),
               "dali_pairdist_prepared")
    return out


def factorize_ids(*arrays):
    """Shared int32 codes for id columns (the reference carries pids / camids as numpy strings)."""
    flat = np.concatenate([np.asarray(a).ravel() for a in arrays])
    _, inv = np.unique(flat, return_inverse=True)
    out, o = [], 0
    for a in arrays:
        n = np.asarray(a).size
        out.append(inv[o:o + n].astype(np.int32))
        o += n
    return out


def rank_eval_codes(distmat, qp, gp, qc, gc, max_rank=50):
    """Device-side part of rank_eval: ids already int32 codes on the GPU.  -> dict of device tensors (no host sync)."""
    nq, ng = distmat.shape
    dev = distmat.device
    max_rank = min(max_rank, ng)
    out = dict(cmc=torch.empty(max_rank, device=dev, dtype=torch.float32), mAP=torch.empty(1, device=dev, dtype=torch.float32),
               map64=torch.empty(1, device=dev, dtype=torch.float64), nvalid=torch.empty(1, device=dev, dtype=torch.int32),
               status=torch.empty(1, device=dev, dtype=torch.int32), ap=torch.empty(nq, device=dev, dtype=torch.float32),
               first_rank=torch.empty(nq, device=dev, dtype=torch.int32))
    _lib.check(_lib.lib().dali_rank_eval(_lib.ctx(dev), _lib.stream_ptr(), _lib.ptr(distmat, torch.float32, "distmat"),
                                          _lib.ptr(qp, torch.int32), _lib.ptr(gp, torch.int32), _lib.ptr(qc, torch.int32), _lib.ptr(gc, torch.int32),
                                          nq, ng, max_rank, _lib.ptr(out["cmc"]), _lib.ptr(out["mAP"]), _lib.ptr(out["map64"]),
                                          _lib.ptr(out["nvalid"]), _lib.ptr(out["ap"]), _lib.ptr(out["first_rank"]), _lib.ptr(out["status"])),
               "dali_rank_eval")
    return out


def rank_eval(distmat, q_pids, g_pids, q_camids, g_camids, max_rank=50, return_per_query=False):
    """market1501 CMC/mAP of torchreid.metrics.evaluate_rank (validateModels.py:68) on the GPU.
    distmat: CUDA fp32 [nq,ng].  Returns (cmc numpy float32 [max_rank], mAP float)."""
    dev = distmat.device
    qp, gp = factorize_ids(q_pids, g_pids)
    qc, gc = factorize_ids(q_camids, g_camids)
    t = lambda a: torch.from_numpy(a).to(dev)
    o = rank_eval_codes(distmat, t(qp), t(gp), t(qc), t(gc), max_rank)
    st = int(o["status"].item())
    if st != 0:
        raise _lib.DaliError("dali_rank_eval: " + ("a query's identity has more than 4096 gallery entries (documented limit)" if st == 1 else
                                                   "identity codes span more than 2^20 values (documented limit)"))
    if int(o["nvalid"].item()) == 0:
        raise AssertionError("Error: all query identities do not appear in gallery")
    res = (o["cmc"].cpu().numpy(), float(o["map64"].item()))
    if return_per_query:
        return res + (o["ap"].cpu().numpy(), o["first_rank"].cpu().numpy())
    return res


RANK_PMAX = 4096          # matches + junk of one query the ranking kernels hold in LDS (csrc/eval.hip)


def shard_bounds(n, world):
    """Contiguous gallery slices of (almost) equal size, multiples of 128 rows (the distance kernel's gallery tile) where possible:
    -> list of world + 1 offsets."""
    per = -(-n // world)
    per = -(-per // 128) * 128
    return [min(r * per, n) for r in range(world + 1)]


def rank_shard_matches(dist_shard, qp, gp, qc, gc, g_offset, cap):
    """Step 1 (device, int32 code tensors): -> (keys int64 [nq, cap], counts int32 [nq], status int32 [1])."""
    nq, ng = dist_shard.shape
    dev = dist_shard.device
    keys = torch.full((nq, cap), -1, device=dev, dtype=torch.int64)
    counts = torch.zeros(nq, device=dev, dtype=torch.int32)
    status = torch.zeros(1, device=dev, dtype=torch.int32)
    if ng > 0:
        _lib.check(_lib.lib().dali_rank_shard_matches(_lib.ctx(dev), _lib.stream_ptr(), _lib.ptr(dist_shard, torch.float32, "dist_shard"),
                                                       _lib.ptr(qp, torch.int32), _lib.ptr(gp, torch.int32), _lib.ptr(qc, torch.int32),
                                                       _lib.ptr(gc, torch.int32), nq, ng, int(g_offset), cap, _lib.ptr(keys), _lib.ptr(counts),
                                                       _lib.ptr(status)), "dali_rank_shard_matches")
    return keys, counts, status


def rank_shard_bins(dist_shard, qp, gp, qc, gc, g_offset, keys_all, counts_all, bins_cap):
    """Step 3: keys_all int64 [world, nq, cap], counts_all int32 [world, nq] (all shards, rank-major) -> (bins int32 [nq, bins_cap + 1], status)."""
    nq, ng = dist_shard.shape
    dev = dist_shard.device
    world, _, cap = keys_all.shape
    bins = torch.zeros(nq, bins_cap + 1, device=dev, dtype=torch.int32)
    status = torch.zeros(1, device=dev, dtype=torch.int32)
    if ng > 0:
        _lib.check(_lib.lib().dali_rank_shard_bins(_lib.ctx(dev), _lib.stream_ptr(), _lib.ptr(dist_shard, torch.float32, "dist_shard"),
                                                    _lib.ptr(qp, torch.int32), _lib.ptr(gp, torch.int32), _lib.ptr(qc, torch.int32),
                                                    _lib.ptr(gc, torch.int32), nq, ng, int(g_offset), _lib.ptr(keys_all.contiguous(), torch.int64),
                                                    _lib.ptr(counts_all.contiguous(), torch.int32), world, cap, _lib.ptr(bins), bins_cap,
                                                    _lib.ptr(status)), "dali_rank_shard_bins")
    return bins, status


def rank_shard_finish(bins, counts_all, max_rank):
    """Step 5: the SUMMED bins -> dict of device tensors (cmc, mAP, map64, nvalid, ap, first_rank)."""
    dev = bins.device
    nq, bins_cap = bins.shape[0], bins.shape[1] - 1
    o = dict(cmc=torch.empty(max_rank, device=dev, dtype=torch.float32), mAP=torch.empty(1, device=dev, dtype=torch.float32),
             map64=torch.empty(1, device=dev, dtype=torch.float64), nvalid=torch.empty(1, device=dev, dtype=torch.int32),
             ap=torch.empty(nq, device=dev, dtype=torch.float32), first_rank=torch.empty(nq, device=dev, dtype=torch.int32))
    _lib.check(_lib.lib().dali_rank_shard_finish(_lib.ctx(dev), _lib.stream_ptr(), _lib.ptr(bins.contiguous(), torch.int32),
                                                  _lib.ptr(counts_all.contiguous(), torch.int32), counts_all.shape[0], nq, bins_cap, max_rank,
                                                  _lib.ptr(o["cmc"]), _lib.ptr(o["mAP"]), _lib.ptr(o["map64"]), _lib.ptr(o["nvalid"]),
                                                  _lib.ptr(o["ap"]), _lib.ptr(o["first_rank"])), "dali_rank_shard_finish")
    return o


def rank_eval_sharded(dist_shard, q_pids, g_pids_shard, q_camids, g_camids_shard, g_offset, group=None, max_rank=50, ng_total=None,
                      return_per_query=False):
    """market1501 CMC / mAP with the GALLERY sharded over the ranks of ``group`` (validateModels.py:41-47,61-69): this rank
    holds ``dist_shard`` [nq, ng_local] = all queries against its gallery slice, whose first entry has global index ``g_offset``.  Every
    rank passes all query ids and ITS slice of the gallery ids.  Two collectives: an all-gather of the per-query match keys (8 bytes per
    match) and an all-reduce (SUM) of integer bins; the result is bit-identical to ``rank_eval`` on the whole matrix and the same on
    every rank.  -> (cmc numpy float32 [max_rank], mAP float)"""
    import torch.distributed as dist
    dev = dist_shard.device
    nq, ng = dist_shard.shape
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    qp, gp = factorize_ids(q_pids, g_pids_shard)                 # codes only need to agree between q and g on THIS rank
    qc, gc = factorize_ids(q_camids, g_camids_shard)
    t = lambda a: torch.from_numpy(a).to(dev)
    codes = (t(qp), t(gp), t(qc), t(gc))
    # cap: the largest identity of any shard (an upper bound of a query's matches inside one shard), agreed over the ranks
    caps = torch.tensor([int(np.bincount(gp).max()) if ng > 0 else 1, ng], dtype=torch.int64, device=dev)
    if world > 1:
        gathered = [torch.empty_like(caps) for _ in range(world)]
        dist.all_gather(gathered, caps, group=group)
        cap, ng_all = int(max(int(c[0]) for c in gathered)), int(sum(int(c[1]) for c in gathered))
    else:
        cap, ng_all = int(caps[0]), ng
    if ng_total is not None and ng_all != ng_total:
        raise _lib.DaliError("rank_eval_sharded: the gallery shards hold %d entries, expected %d" % (ng_all, ng_total))
    cap = max(1, min(cap, RANK_PMAX))
    bins_cap = min(world * cap, RANK_PMAX)
    keys, counts, st1 = rank_shard_matches(dist_shard, *codes, g_offset, cap)
    if world > 1:
        keys_l = [torch.empty_like(keys) for _ in range(world)]
        counts_l = [torch.empty_like(counts) for _ in range(world)]
        dist.all_gather(keys_l, keys, group=group)
        dist.all_gather(counts_l, counts, group=group)
        keys_all, counts_all = torch.stack(keys_l), torch.stack(counts_l)
    else:
        keys_all, counts_all = keys.unsqueeze(0), counts.unsqueeze(0)
    bins, st2 = rank_shard_bins(dist_shard, *codes, g_offset, keys_all, counts_all, bins_cap)
    status = torch.maximum(st1, st2)
    if world > 1:
        dist.all_reduce(bins, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(status, op=dist.ReduceOp.MAX, group=group)
    o = rank_shard_finish(bins, counts_all, min(max_rank, ng_all))
    stv = int(status.item())
    if stv != 0:
        raise _lib.DaliError("rank_eval_sharded: " + ("a query has more than %d matches (documented limit)" % RANK_PMAX if stv == 1 else
                                                      "identity codes span more than 2^20 values (documented limit)"))
    if int(o["nvalid"].item()) == 0:
        raise AssertionError("Error: all query identities do not appear in gallery")
    res = (o["cmc"].cpu().numpy(), float(o["map64"].item()))
    if return_per_query:
        return res + (o["ap"].cpu().numpy(), o["first_rank"].cpu().numpy())
    return res


def class_targets(fvs, order, bounds, first_pick, num_proxies=5):
    """Class centers + farthest-point proxies in one launch (train_encodersKIT.py:113-156, :252-284).
    fvs [N,D] fp32 CUDA; order [N] int32 (rows sorted by identity), bounds [NC+1] int32, first_pick [NC] int32 (position
    of the first proxy inside each identity's slice), all CUDA.
    -> centers [NC,D], proxies [NC*num_proxies,D] (zero rows where an identity has fewer images), proxy_rows
    [NC*num_proxies] int32 (row of fvs or -1), max_dist [NC]."""
    assert fvs.dim() == 2 and fvs.is_contiguous()
    n, d = fvs.shape
    nc = bounds.numel() - 1
    assert order.numel() == n and first_pick.numel() == nc
    dev = fvs.device
    centers = torch.empty(nc, d, device=dev, dtype=torch.float32)
    proxies = torch.empty(nc * num_proxies, d, device=dev, dtype=torch.float32)
    proxy_rows = torch.empty(nc * num_proxies, device=dev, dtype=torch.int32)
    max_dist = torch.empty(nc, device=dev, dtype=torch.float32)
    _lib.check(_lib.lib().dali_class_targets(_lib.ctx(dev), _lib.stream_ptr(), _lib.ptr(fvs, torch.float32, "fvs"), n, d,
                                              _lib.ptr(order, torch.int32, "order"), _lib.ptr(bounds, torch.int32, "bounds"), nc,
                                              _lib.ptr(first_pick, torch.int32, "first_pick"), int(num_proxies), _lib.ptr(centers),
                                              _lib.ptr(proxies), _lib.ptr(proxy_rows), _lib.ptr(max_dist)), "dali_class_targets")
    return centers, proxies, proxy_rows, max_dist


def pairdist_blend(distmat, q, g, mags_prev=None, mags=None, precision="bf16x3", normalize=True):
    """In place: distmat <- (w1*distmat + w2*(1 - q@g.T)) / (w1 + w2), w_m = max(qmag_m[:,None], gmag_m[None,:])
    (evaluateCleanATModels.py:154-157), computed in the distance kernel's epilogue.  mags_prev = (q_mag, g_mag) of the
    model that produced ``distmat``, mags = those of this model; both None -> plain average (:126)."""
    assert q.dim() == 2 and g.dim() == 2 and q.shape[1] == g.shape[1] and tuple(distmat.shape) == (q.shape[0], g.shape[0])
    assert (mags_prev is None) == (mags is None)
    nq, ng, d = q.shape[0], g.shape[0], q.shape[1]
    if nq == 0 or ng == 0:
        return distmat
    f32 = torch.float32
    m = [None] * 4 if mags is None else [t.reshape(-1).contiguous().float() for t in (*mags_prev, *mags)]
    if mags is not None:
        assert m[0].numel() == nq and m[1].numel() == ng and m[2].numel() == nq and m[3].numel() == ng
    _lib.check(_lib.lib().dali_pairdist_blend(_lib.ctx(q.device), _lib.stream_ptr(), _lib.ptr(q, f32, "q"), _lib.ptr(g, f32, "g"), nq, ng, d,
                                               _PREC[precision], int(bool(normalize)), _lib.ptr(m[0]), _lib.ptr(m[1]), _lib.ptr(m[2]),
                                               _lib.ptr(m[3]), _lib.ptr(distmat, f32, "distmat")), "dali_pairdist_blend")
    return distmat
