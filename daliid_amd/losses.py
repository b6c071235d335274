"""Mirror of the reference's ``losses.py`` for the two losses the trainer calls (train_encodersKIT.py:200-208) plus
the cosine schedule and balanced accuracy, on HIP kernels (dali_center_loss_* / dali_proxy_loss_* / dali_pairdist).

Same names, argument order and return values as the reference:
  getValueFromCosineSchedule(t_cur, t_max, n_min=0.0, n_max=1.0)                         losses.py:5-7
  BatchWeightedCenterLoss(batch_fvs, batch_labels, samples_distortion, centers, centers_labels, current_epoch,
                          number_of_epoches, is_clean_training, tau=0.1, gpu_index=0)
      -> (loss, acc_bal, avg_max_prob)                                                     losses.py:39-88
  BatchWeightedProxyLoss(batch_fvs, batch_labels, samples_distortion, all_proxies, proxies_labels, current_epoch,
                         number_of_epoches, top_negs=50, tau=0.1, gpu_index=0) -> loss    losses.py:273-341
  BatchWeightedSoftmaxTripletLoss(batch_fvs, batch_labels, samples_distortion, current_epoch, number_of_epoches,
                                  tau=0.1, gpu_index=0) -> loss   (optional head)          losses.py:607-654
``loss`` is a differentiable 0-dim CUDA tensor (gradient wrt ``batch_fvs``).  ``LossHeads`` is the fused form the
trainer mirror uses: both heads, one pass, no host synchronisation, data-parallel normalisers.
"""
import numpy as np
import torch

from . import _lib
from .ops_eval import pairdist

_KMAX = None


def _kmax():
    global _KMAX
    if _KMAX is None:
        _KMAX = int(_lib.lib().dali_proxy_kmax())
    return _KMAX


def getValueFromCosineSchedule(t_cur, t_max, n_min=0.0, n_max=1.0):
    """losses.py:5-7."""
    return n_min + 0.5 * (n_max - n_min) * (1 + np.cos(((t_max - t_cur) / t_max) * np.pi))


def distortion_weights(current_epoch, number_of_epoches):
    """losses.py:42-49 / :279-286: fp32 table [w0..w5] indexed by the distortion level."""
    mins = (0.8, 0.6, 0.4, 0.2, 0.1)
    return torch.tensor([1.0] + [getValueFromCosineSchedule(current_epoch, number_of_epoches, n_min=m, n_max=1.0) for m in mins],
                        dtype=torch.float32)


def getACCBal(predicted_labels, gt_labels):
    """losses.py:190-203 (host side, logging only)."""
    predicted_labels, gt_labels = np.asarray(predicted_labels), np.asarray(gt_labels)
    all_labels = np.union1d(np.unique(predicted_labels), np.unique(gt_labels))
    n = len(all_labels)
    cm = np.zeros((n, n))
    np.add.at(cm, (np.searchsorted(all_labels, gt_labels), np.searchsorted(all_labels, predicted_labels)), 1.0)
    return np.trace(cm / (np.sum(cm, axis=1) + 1e-7)) / n


def _codes(t, device):
    """integer id codes on the device (the reference carries ids as float tensors / numpy int arrays)."""
    if isinstance(t, torch.Tensor):
        return t.to(device=device).round().to(torch.int32).contiguous()
    return torch.as_tensor(np.asarray(t).astype(np.int64), device=device).to(torch.int32).contiguous()


def _sample_weights(samples_distortion, current_epoch, number_of_epoches, device):
    table = distortion_weights(current_epoch, number_of_epoches).to(device)
    idx = samples_distortion if isinstance(samples_distortion, torch.Tensor) else torch.as_tensor(np.asarray(samples_distortion))
    return table[idx.to(device).long()].contiguous()


def center_fwd(S, labels, center_labels, w, tau):
    nb, NC = S.shape
    rowstat = torch.empty(nb, 4, device=S.device, dtype=torch.float32)
    sums = torch.empty(2, device=S.device, dtype=torch.float32)
    _lib.check(_lib.lib().dali_center_loss_fwd(_lib.ctx(S.device), _lib.stream_ptr(), _lib.ptr(S, torch.float32, "S"), _lib.ptr(labels, torch.int32),
                                                _lib.ptr(center_labels, torch.int32), _lib.ptr(w, torch.float32), float(tau), nb, NC,
                                                _lib.ptr(rowstat), _lib.ptr(sums)), "dali_center_loss_fwd")
    return rowstat, sums


def center_bwd(S, labels, center_labels, w, tau, denom, gscale=1.0):
    nb, NC = S.shape
    dS = torch.empty_like(S)
    _lib.check(_lib.lib().dali_center_loss_bwd(_lib.ctx(S.device), _lib.stream_ptr(), _lib.ptr(S, torch.float32), _lib.ptr(labels, torch.int32),
                                                _lib.ptr(center_labels, torch.int32), _lib.ptr(w, torch.float32), float(tau), nb, NC,
                                                _lib.ptr(denom, torch.float32), float(gscale), _lib.ptr(dS)), "dali_center_loss_bwd")
    return dS


def proxy_fwd(S, labels, proxy_labels, w, tau):
    nb, NP = S.shape
    dev = S.device
    k = _kmax()
    rowstat = torch.empty(nb, 2, device=dev, dtype=torch.float32)
    sums = torch.empty(2, device=dev, dtype=torch.float32)
    sel_idx = torch.empty(nb, 2 * k, device=dev, dtype=torch.int32)
    sel_coef = torch.empty(nb, 2 * k, device=dev, dtype=torch.float32)
    status = torch.empty(1, device=dev, dtype=torch.int32)
    _lib.check(_lib.lib().dali_proxy_loss_fwd(_lib.ctx(dev), _lib.stream_ptr(), _lib.ptr(S, torch.float32, "S"), _lib.ptr(labels, torch.int32),
                                               _lib.ptr(proxy_labels, torch.int32), _lib.ptr(w, torch.float32), float(tau), nb, NP,
                                               _lib.ptr(rowstat), _lib.ptr(sums), _lib.ptr(sel_idx), _lib.ptr(sel_coef), _lib.ptr(status)),
               "dali_proxy_loss_fwd")
    return rowstat, sums, sel_idx, sel_coef, status


def proxy_bwd(sel_idx, sel_coef, proxies, denom, gscale=1.0, out=None, accumulate=False):
    nb = sel_idx.shape[0]
    D = proxies.shape[1]
    if out is None:
        out = torch.empty(nb, D, device=proxies.device, dtype=torch.float32)
        accumulate = False
    _lib.check(_lib.lib().dali_proxy_loss_bwd(_lib.ctx(proxies.device), _lib.stream_ptr(), _lib.ptr(sel_idx, torch.int32), _lib.ptr(sel_coef, torch.float32),
                                               _lib.ptr(proxies, torch.float32, "proxies"), nb, D, _lib.ptr(denom, torch.float32), float(gscale),
                                               int(accumulate), _lib.ptr(out)), "dali_proxy_loss_bwd")
    return out


class _CenterLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fvs, centers, labels, clabels, w, tau):
        S = pairdist(fvs.contiguous(), centers, metric="dot")
        rowstat, sums = center_fwd(S, labels, clabels, w, tau)
        ctx.save_for_backward(S, centers, labels, clabels, w, sums)
        ctx.tau = tau
        ctx.rowstat = rowstat
        return sums[0] / sums[1], rowstat

    @staticmethod
    def backward(ctx, g, _g_rowstat):
        S, centers, labels, clabels, w, sums = ctx.saved_tensors
        dS = center_bwd(S, labels, clabels, w, ctx.tau, sums[1:2])
        dfn = pairdist(dS, centers.t().contiguous(), metric="dot")          # dS @ centers
        return dfn * g, None, None, None, None, None


class _ProxyLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fvs, proxies, labels, plabels, w, tau):
        S = pairdist(fvs.contiguous(), proxies, metric="dot")
        rowstat, sums, sel_idx, sel_coef, status = proxy_fwd(S, labels, plabels, w, tau)
        if int(status.item()) != 0:
            raise _lib.DaliError("BatchWeightedProxyLoss: an identity has more than %d proxies (documented limit)" % _kmax())
        ctx.save_for_backward(proxies, sel_idx, sel_coef, sums)
        return sums[0] / sums[1]

    @staticmethod
    def backward(ctx, g):
        proxies, sel_idx, sel_coef, sums = ctx.saved_tensors
        return proxy_bwd(sel_idx, sel_coef, proxies, sums[1:2]) * g, None, None, None, None, None


def BatchWeightedCenterLoss(batch_fvs, batch_labels, samples_distortion, centers, centers_labels, current_epoch,
                            number_of_epoches, is_clean_training, tau=0.1, gpu_index=0):
    dev = batch_fvs.device
    w = _sample_weights(samples_distortion, current_epoch, number_of_epoches, dev)
    labels, clabels = _codes(batch_labels, dev), _codes(centers_labels, dev)
    loss, rowstat = _CenterLossFn.apply(batch_fvs, centers.contiguous(), labels, clabels, w, float(tau))
    rs = rowstat.detach().cpu().numpy()                      # the reference syncs here too (losses.py:65,84-86)
    pred = np.asarray(centers_labels)[rs[:, 2].astype(np.int64)]
    n_match = np.rint(rs[:, 1] / np.maximum(w.cpu().numpy(), 1e-30))
    has_one = n_match == 1
    gt = batch_labels.detach().cpu().numpy() if isinstance(batch_labels, torch.Tensor) else np.asarray(batch_labels)
    acc = getACCBal(pred[has_one], gt[has_one])
    return loss, acc, float(rs[:, 3].mean())


def BatchWeightedProxyLoss(batch_fvs, batch_labels, samples_distortion, all_proxies, proxies_labels, current_epoch,
                           number_of_epoches, top_negs=50, tau=0.1, gpu_index=0):
    dev = batch_fvs.device
    w = _sample_weights(samples_distortion, current_epoch, number_of_epoches, dev)
    return _ProxyLossFn.apply(batch_fvs, all_proxies.contiguous(), _codes(batch_labels, dev), _codes(proxies_labels, dev), w, float(tau))


def triplet_distortion_weights(current_epoch, number_of_epoches):
    """losses.py:613-627: the 13-entry table of BatchWeightedSoftmaxTripletLoss."""
    mins = (0.90, 0.85, 0.80, 0.75, 0.70, 0.6, 0.5, 0.4, 0.3, 0.2, 0.1, 0.1)
    return torch.tensor([1.0] + [getValueFromCosineSchedule(current_epoch, number_of_epoches, n_min=m, n_max=1.0) for m in mins],
                        dtype=torch.float32)


def triplet_fwd(S, labels, w, tau):
    nb, dev = S.shape[0], S.device
    rowstat = torch.empty(nb, 2, device=dev, dtype=torch.float32)
    sums = torch.empty(2, device=dev, dtype=torch.float32)
    sel_idx = torch.empty(nb, 2, device=dev, dtype=torch.int32)
    sel_coef = torch.empty(nb, device=dev, dtype=torch.float32)
    status = torch.empty(1, device=dev, dtype=torch.int32)
    _lib.check(_lib.lib().dali_triplet_loss_fwd(_lib.ctx(dev), _lib.stream_ptr(), _lib.ptr(S, torch.float32, "S"), _lib.ptr(labels, torch.int32),
                                                 _lib.ptr(w, torch.float32), float(tau), nb, _lib.ptr(rowstat), _lib.ptr(sums), _lib.ptr(sel_idx),
                                                 _lib.ptr(sel_coef), _lib.ptr(status)), "dali_triplet_loss_fwd")
    return rowstat, sums, sel_idx, sel_coef, status


def triplet_bwd(sel_idx, sel_coef, denom, gscale=1.0):
    nb = sel_idx.shape[0]
    dS = torch.empty(nb, nb, device=sel_idx.device, dtype=torch.float32)
    _lib.check(_lib.lib().dali_triplet_loss_bwd(_lib.ctx(dS.device), _lib.stream_ptr(), _lib.ptr(sel_idx, torch.int32), _lib.ptr(sel_coef, torch.float32),
                                                 nb, _lib.ptr(denom, torch.float32), float(gscale), _lib.ptr(dS)), "dali_triplet_loss_bwd")
    return dS


class _TripletLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fvs, labels, w, tau):
        fvs = fvs.contiguous()
        S = pairdist(fvs, fvs, metric="dot")
        _, sums, sel_idx, sel_coef, status = triplet_fwd(S, labels, w, tau)
        if int(status.item()) != 0:                # the reference's topk(k=1) on an empty negative set raises (losses.py:641)
            raise _lib.DaliError("BatchWeightedSoftmaxTripletLoss: a sample has no negative in the batch (single identity)")
        ctx.save_for_backward(fvs, sel_idx, sel_coef, sums)
        return sums[0] / sums[1]

    @staticmethod
    def backward(ctx, g):
        fvs, sel_idx, sel_coef, sums = ctx.saved_tensors
        dS = triplet_bwd(sel_idx, sel_coef, sums[1:2])
        return pairdist(dS, fvs.t().contiguous(), metric="dot") * g, None, None, None        # (dS + dS^T) @ fvs


def BatchWeightedSoftmaxTripletLoss(batch_fvs, batch_labels, samples_distortion, current_epoch, number_of_epoches, tau=0.1, gpu_index=0):
    """losses.py:607-654 (optional head, not called by the trainer): in-batch hardest positive / hardest negative."""
    dev = batch_fvs.device
    table = triplet_distortion_weights(current_epoch, number_of_epoches).to(dev)
    idx = samples_distortion if isinstance(samples_distortion, torch.Tensor) else torch.as_tensor(np.asarray(samples_distortion))
    w = table[idx.to(dev).long()].contiguous()
    return _TripletLossFn.apply(batch_fvs, _codes(batch_labels, dev), w, float(tau))


class LossHeads:
    """Both heads fused for the trainer hot loop: loss = center + lambda_proxy * proxy (train_encodersKIT.py:200-208),
    forward + gradient wrt the normalised embeddings in one pass, no host sync.  With ``process_group`` set the two
    normalisers (and the numerators, for logging) are summed across data-parallel ranks before the backward."""

    def __init__(self, centers, centers_labels, proxies, proxies_labels, tau, lambda_proxy, process_group=None):
        dev = centers.device
        self.centers = centers.contiguous()
        self.centers_t = centers.t().contiguous()
        self.proxies = proxies.contiguous()
        self.clabels, self.plabels = _codes(centers_labels, dev), _codes(proxies_labels, dev)
        self.tau, self.lam, self.pg = float(tau), float(lambda_proxy), process_group
        # the kernel keeps one identity's positives in registers (dali_proxy_kmax()); its device-side status word is not read in the hot
        # loop (no host sync there), so the limit is enforced here, once per epoch, on the host copy of the labels
        plab = np.asarray(proxies_labels)
        if plab.size and int(np.unique(plab, return_counts=True)[1].max()) > _kmax():
            raise _lib.DaliError("LossHeads: an identity has more than %d proxies (documented limit of dali_proxy_loss_fwd)" % _kmax())

    def __call__(self, fn, labels, w):
        """fn [nb,D] normalised embeddings, labels int32 codes, w [nb] -> (stats[4] = c_num, c_den, p_num, p_den (global),
        dfn [nb,D])."""
        Sc = pairdist(fn, self.centers, metric="dot")
        Sp = pairdist(fn, self.proxies, metric="dot")
        self.rowstat_c, sums_c = center_fwd(Sc, labels, self.clabels, w, self.tau)
        _, sums_p, sel_idx, sel_coef, self.status = proxy_fwd(Sp, labels, self.plabels, w, self.tau)
        stats = torch.cat((sums_c, sums_p))
        if self.pg is not None:
            from .parallel import allreduce_loss_stats
            allreduce_loss_stats(stats, self.pg)
        dS = center_bwd(Sc, labels, self.clabels, w, self.tau, stats[1:2])
        dfn = pairdist(dS, self.centers_t, metric="dot")
        proxy_bwd(sel_idx, sel_coef, self.proxies, stats[3:4], gscale=self.lam, out=dfn, accumulate=True)
        return stats, dfn

    @staticmethod
    def losses_from_stats(stats, lam):
        c, p = stats[0] / stats[1], stats[2] / stats[3]
        return c + lam * p, c, p
