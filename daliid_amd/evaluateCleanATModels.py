"""Mirror of the two-model ("clean" + "distortion") fusion of the reference's ``evaluateCleanATModels.py``
(:78-245 ``validate``, :249-256 ``getWeightsByMagnitude``, :259-276 ``calculateMetrics``): the paper's cross-domain
fusion result.  Distances come from the MFMA distance kernel; the magnitude-weighted blend of the two matrices runs in
the second kernel's epilogue (``dali_pairdist_blend``), so the second matrix is never materialised."""
import numpy as np
import torch

from . import ops_eval
from .getFeatures import extractFeatures

precision = "bf16x3"


def getWeightsByMagnitude(subset, pooling, img_height, img_width, model, gpu_indexes):
    """:249-256: features under ``model.module.feature = pooling`` -> (norms [N,1], unit rows); resets the pooling
    to "both".  Stays on the GPU."""
    net = getattr(model, "module", model)
    net.feature = pooling
    try:
        fvs = extractFeatures(subset, img_height, img_width, model, 500, gpu_index=gpu_indexes[0], keep_on_device=True, verbose=False)
    finally:
        net.feature = "both"
    unit, norms = ops_eval.l2norm_rows(fvs.contiguous(), 0.0, return_norms=True)
    return norms[:, None], unit


def calculateMetrics(queries_images, gallery_images, distmat, pooling=None, version=None, verbose=True):
    """:259-276: market1501 CMC / mAP of a distance matrix, ranks 1/5/10/20 printed."""
    cmc, mAP = ops_eval.rank_eval(distmat, queries_images[:, 1], gallery_images[:, 1], queries_images[:, 2], gallery_images[:, 2])
    if verbose:
        print("** Results **")
        print("mAP: {:.2%}".format(mAP))
        print("CMC curve")
        for r in (1, 5, 10, 20):
            print("Rank-{:<3}: {:.2%}".format(r, cmc[r - 1]))
    return cmc, mAP


def fuse_distmats(q_clean, g_clean, q_dist, g_dist, mags_clean=None, mags_dist=None):
    """(w_c*d_c + w_d*d_d)/(w_c + w_d) with w = max(query magnitude, gallery magnitude) (:154-157); without magnitudes
    the simple ensemble (d_c + d_d)/2 (:126).  q_*/g_* are the un-normalised "both" embeddings (:96-100, :114-124)."""
    distmat = ops_eval.pairdist(q_clean.contiguous(), g_clean.contiguous(), metric="cosine", precision=precision, normalize=True)
    return ops_eval.pairdist_blend(distmat, q_dist.contiguous(), g_dist.contiguous(), mags_clean, mags_dist, precision=precision, normalize=True)


def validate(queries_images, gallery_images, model_clean, model_distortion, img_height, img_width, gpu_indexes, verbose=True):
    """:78-245 in the reference's order: concatenated features, each model alone, simple ensemble, then the
    magnitude-weighted ensembles for GAP (the paper's number), GMP and GAP+GMP.  -> dict name -> (cmc, mAP)."""
    model_clean.eval(); model_distortion.eval()
    ex = lambda subset, model: extractFeatures(subset, img_height, img_width, model, 500, gpu_index=gpu_indexes[0], keep_on_device=True,
                                               verbose=False)
    q_c, q_d = ex(queries_images, model_clean), ex(queries_images, model_distortion)
    g_c, g_d = ex(gallery_images, model_clean), ex(gallery_images, model_distortion)
    out = {}
    metrics = lambda dm: calculateMetrics(queries_images, gallery_images, dm, verbose=verbose)
    out["concatenation"] = metrics(ops_eval.pairdist(torch.cat((q_c, q_d), 1), torch.cat((g_c, g_d), 1), precision=precision, normalize=True))
    out["clean"] = metrics(ops_eval.pairdist(q_c, g_c, precision=precision, normalize=True))
    out["distortion"] = metrics(ops_eval.pairdist(q_d, g_d, precision=precision, normalize=True))
    out["simple_ensemble"] = metrics(fuse_distmats(q_c, g_c, q_d, g_d))
    for pooling in ("gap", "gmp", "both"):
        qm_c, _ = getWeightsByMagnitude(queries_images, pooling, img_height, img_width, model_clean, gpu_indexes)
        qm_d, _ = getWeightsByMagnitude(queries_images, pooling, img_height, img_width, model_distortion, gpu_indexes)
        gm_c, _ = getWeightsByMagnitude(gallery_images, pooling, img_height, img_width, model_clean, gpu_indexes)
        gm_d, _ = getWeightsByMagnitude(gallery_images, pooling, img_height, img_width, model_distortion, gpu_indexes)
        out["ensemble_" + pooling] = metrics(fuse_distmats(q_c, g_c, q_d, g_d, (qm_c, gm_c), (qm_d, gm_d)))
    return out
