"""Mirror of the reference's ``validateModels.py`` for the path in scope (validateModels.py:26-76, 108-118):
``validationManager.getValidator(name)`` -> object with ``setParameters`` / ``validate(queries, gallery, model)`` /
``calculateMetrics``.  Feature normalisation + ``1 - q @ g.T`` run as ONE fused call on the MFMA distance kernel and
CMC/mAP on the GPU ranking kernel; the features never leave the device."""
import numpy as np
import torch

from .getFeatures import extractFeatures
from . import ops_eval


class validateModels:

    precision = "bf16x3"          # "bf16" trades ~1e-4 absolute distance error for ~2x distance throughput
    distmat_on_cpu = False        # the reference returns a CPU tensor (validateModels.py:58); 4 GB at config-5 size

    def setParameters(self, img_height, img_width, rerank, gpu_index):
        self.img_height = img_height
        self.img_width = img_width
        self.rerank = rerank
        self.gpu_index = gpu_index

    def validate(self, queries, gallery, model):
        model.eval()
        queries_fvs = extractFeatures(queries, self.img_height, self.img_width, model, 500, self.gpu_index, keep_on_device=True)
        gallery_fvs = extractFeatures(gallery, self.img_height, self.img_width, model, 500, self.gpu_index, keep_on_device=True)
        distmat = self.distance(queries_fvs, gallery_fvs)
        del queries_fvs, gallery_fvs
        cmc, mAP = self.calculateMetrics(distmat, queries, gallery)
        return cmc, mAP, (distmat.cpu() if self.distmat_on_cpu else distmat)

    def validate_sharded(self, queries, gallery, model, process_group=None):
        """``validate`` with the gallery split over the ranks of ``process_group`` (one process per GPU): every rank extracts
        the query features and the features of ITS contiguous gallery slice, computes its [Nq, Ng / N] block of ``1 - q @ g.T`` and the
        ranks merge per-query hit counts (ops_eval.rank_eval_sharded: one all-gather of match keys, one all-reduce of integer bins).
        -> (cmc, mAP, this rank's distance block); cmc / mAP are identical on every rank and, GIVEN the same model state on every rank,
        equal to the single-GPU result bit for bit.  The weights are identical by construction (parallel.py); the BatchNorm running
        statistics are rank-local during training and are made rank 0's at the end of every ``trainer.train`` epoch
        (parallel.sync_buffers_from_rank0, the nn.DataParallel semantics of Encoders.py:39-40).  A model whose statistics still differ
        between ranks is refused here, collectively, before any feature is extracted."""
        import torch.distributed as dist
        from . import _lib, parallel
        world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        model.eval()
        if not parallel.buffers_in_sync((model,), process_group):
            raise _lib.DaliError("validate_sharded: BatchNorm running statistics differ between ranks "
                                 "(call parallel.sync_buffers_from_rank0 or load the same checkpoint on every rank)")
        lo, hi = ops_eval.shard_bounds(len(gallery), world)[rank:rank + 2]
        queries_fvs = extractFeatures(queries, self.img_height, self.img_width, model, 500, self.gpu_index, keep_on_device=True)
        if hi > lo:
            gallery_fvs = extractFeatures(gallery[lo:hi], self.img_height, self.img_width, model, 500, self.gpu_index, keep_on_device=True)
        else:                  # trailing ranks of a small gallery hold no rows (128-row aligned slices): an empty block, same collectives
            gallery_fvs = queries_fvs.new_zeros(0, queries_fvs.shape[1])
        block = self.distance(queries_fvs, gallery_fvs)
        del queries_fvs, gallery_fvs
        cmc, mAP = ops_eval.rank_eval_sharded(block, queries[:, 1], gallery[lo:hi, 1], queries[:, 2], gallery[lo:hi, 2], lo, process_group,
                                              max_rank=50, ng_total=len(gallery))
        return cmc, mAP, block

    def distance(self, queries_fvs, gallery_fvs):
        """validateModels.py:41-47: q/|q|, g/|g|, 1 - q @ g.T (fused)."""
        return ops_eval.pairdist(queries_fvs.contiguous(), gallery_fvs.contiguous(), metric="cosine", precision=self.precision, normalize=True)

    def calculateMetrics(self, distmat, queries, gallery):
        """validateModels.py:61-76 -> torchreid.metrics.evaluate_rank(distmat, q_pids, g_pids, q_camids, g_camids,
        use_metric_cuhk03=False)."""
        if not isinstance(distmat, torch.Tensor):
            distmat = torch.as_tensor(np.asarray(distmat))
        distmat = distmat.to(torch.device("cuda", getattr(self, "gpu_index", 0)), dtype=torch.float32).contiguous()
        ranks = [1, 5, 10]
        print('Computing CMC and mAP ...')
        cmc, mAP = ops_eval.rank_eval(distmat, queries[:, 1], gallery[:, 1], queries[:, 2], gallery[:, 2], max_rank=50)
        print('** Results **')
        print('mAP: {:.2%}'.format(mAP))
        print('Ranks:')
        for r in ranks:
            if r <= len(cmc):
                print('Rank-{:<3}: {:.2%}'.format(r, cmc[r - 1]))
        return cmc, mAP


class MSMT17_validator:
    """validateModels.MSMT17_validator (validateModels.py:120-190): MSMT17's train/val balanced-accuracy validation.  mainKIT.main only
    builds it when ``dataset == 'MSMT17'`` (mainKIT.py:122-124); that dataset branch is outside the scope table.  The
    name exists so that ``from validateModels import validationManager, MSMT17_validator`` (mainKIT.py:28) keeps importing."""

    def __init__(self, train_images, val_images, trainer, dir_to_save):
        raise NotImplementedError("MSMT17_validator is out of scope of this build (SURVEY.md 2.1)")


class validationManager:

    @staticmethod
    def getValidator(name):
        """validateModels.py:108-118: the BRIAR / MSMT17 validators are outside the scope table."""
        if name in ("BRIAR", "MSMT17"):
            raise NotImplementedError("validator for %s is out of scope of this build (SURVEY.md 2.1)" % name)
        return validateModels()
