"""Synthetic ReID data in the reference's record format (datasetUtils.py:15: numpy string rows
``[img_path, pid, camid, kind]``) with an in-memory image loader -- there are no datasets in the build / bench
environment (datasetUtils.load_dataset's hard-coded /scratch paths, datasetUtils.py:110-112, are out of scope)."""
import numpy as np
import torch

from . import getFeatures


class SyntheticImages:
    """Deterministic per-identity images: a low-resolution identity pattern + per-image noise, generated on demand
    from the record key ``syn://<pid>/<idx>`` and kept on the GPU."""

    def __init__(self, n_ids, per_id, n_cams=6, seed=12, device="cuda", noise=0.5):
        self.n_ids, self.per_id, self.noise, self.device, self.seed = n_ids, per_id, noise, torch.device(device), seed
        rng = np.random.default_rng(seed)
        pids = np.repeat(np.arange(n_ids), per_id)
        idx = np.tile(np.arange(per_id), n_ids)
        cams = rng.integers(0, n_cams, pids.size)
        self.records = np.array([["syn://%d/%d" % (p, i), str(p), str(c), "person"] for p, i, c in zip(pids, idx, cams)])
        g = torch.Generator(device="cpu").manual_seed(seed)
        self.patterns = torch.randn(n_ids, 3, 8, 4, generator=g).to(self.device)

    def loader(self, paths, img_height, img_width, turb=None):
        pid = torch.tensor([int(p.split("/")[2]) for p in paths], device=self.device)
        idx = [int(p.split("/")[3]) for p in paths]
        base = torch.nn.functional.interpolate(self.patterns[pid], size=(img_height, img_width), mode="bilinear", align_corners=False)
        g = torch.Generator(device=self.device)
        out = torch.empty(len(paths), 3, img_height, img_width, device=self.device)
        strength = 0.0 if turb is None else 0.3 * float(turb[1])
        for i, (p, k) in enumerate(zip(pid.tolist(), idx)):
            g.manual_seed(self.seed * 1000003 + p * 1009 + k)
            out[i] = base[i] + (self.noise + strength) * torch.randn(3, img_height, img_width, device=self.device, generator=g)
        return out

    def install(self):
        """Serve both the evaluation loader (getFeatures.sample) and the training loader (samplePKBatches) from memory:
        synthetic tensors carry no augmentation, which has to be said explicitly (the default training loader augments)."""
        from . import train_encodersKIT
        getFeatures.set_image_loader(self.loader)
        train_encodersKIT.set_train_loader(self.loader)
        return self

    @staticmethod
    def uninstall():
        from . import train_encodersKIT
        getFeatures.set_image_loader(None)
        train_encodersKIT.set_train_loader(None)

    def split(self, n_query_per_id=1):
        """-> (train, gallery, query) record arrays: the last n_query_per_id images of every id are queries."""
        k = np.tile(np.arange(self.per_id), self.n_ids)
        q = k >= self.per_id - n_query_per_id
        return self.records, self.records[~q], self.records[q]
