"""Mirror of the reference's ``getFeatures.extractFeatures`` (getFeatures.py:47-71): eval-mode batched inference
over a dataset index, result concatenated in order.

Image decode / resize is outside the accelerated path: it goes through a pluggable loader.  The default
loader reproduces ``sample.transform_person`` (getFeatures.py:18-19: PIL bicubic resize, ToTensor, ImageNet
mean/std) with PIL + numpy; synthetic datasets register an in-memory loader with ``set_image_loader``.
"""
import time

import numpy as np
import torch

_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32).reshape(3, 1, 1)
_STD = np.array([0.229, 0.224, 0.225], dtype=np.float32).reshape(3, 1, 1)


def turb_path(path, turb):
    """The clean -> turbulence-simulated file pairing of getFeatures.py:24-33 / train_encodersKIT.py:367-377:
    ``<turb_dir>/<name>_turbstrength<k>.jpg`` (MSMT17: ``<pid>_<name>_turbstrength<k>.jpg``, pid = name up to the first
    underscore).  ``turb`` = (turbulance_dir_path, strength, dataset)."""
    import os
    turb_dir, strength, dataset = turb
    if strength is None:
        raise ValueError("turbulance_dir_path is set but turb_strength is None (getFeatures.py:30 formats it with %d)")
    name = path.split("/")[-1][:-4]
    if dataset == "MSMT17":
        name = name.split("_")[0] + "_" + name
    return os.path.join(turb_dir, name + "_turbstrength%d.jpg" % int(strength))


def pil_loader(paths, img_height, img_width, turb=None):
    """getFeatures.py:18-19 / :22-41: read_image (the turbulence-simulated file when ``turb`` is given) ->
    Resize((H,W), bicubic) -> ToTensor -> Normalize."""
    from PIL import Image
    if turb is not None:
        paths = [turb_path(p, turb) for p in paths]
    out = np.empty((len(paths), 3, img_height, img_width), dtype=np.float32)
    for i, p in enumerate(paths):
        img = Image.open(p).convert("RGB").resize((img_width, img_height), Image.BICUBIC)
        out[i] = (np.asarray(img, dtype=np.float32).transpose(2, 0, 1) / 255.0 - _MEAN) / _STD
    return torch.from_numpy(out)


_loader = pil_loader


def set_image_loader(fn):
    """fn(paths: sequence[str], img_height, img_width, turb=None) -> float tensor [n,3,H,W] (CPU or CUDA)."""
    global _loader
    _loader = fn if fn is not None else pil_loader


def get_image_loader():
    return _loader


def extractFeatures(subset, img_height, img_width, model, batch_size, gpu_index=0, dataset=None, turbulance_dir_path=None,
                    turb_strength=None, keep_on_device=False, verbose=True):
    """-> fp32 [N, D] features in dataset order; on the CPU like the reference (getFeatures.py:62) unless
    ``keep_on_device`` (the in-tree callers keep them on the GPU and skip the D2H/H2D round trip)."""
    model.eval()
    dev = torch.device("cuda", gpu_index)
    paths = [row[0] for row in subset]
    start = time.time()
    chunks = []
    turb = None if not turbulance_dir_path else (turbulance_dir_path, turb_strength, dataset)
    starts = list(range(0, len(paths), batch_size))
    batched = all(hasattr(_loader, a) for a in ("plan", "submit", "finish"))        # transforms.gpu_eval_loader: decode pool + one launch per batch
    with torch.no_grad():
        if batched:
            # batch i + 1 decodes on the pool while batch i's forward is enqueued and runs; its resize + normalise go to a side stream
            ahead = [_loader.submit(_loader.plan(paths[b:b + batch_size], img_height, img_width, turb)) for b in starts[:1]]
            for i in range(len(starts)):
                if i + 1 < len(starts):
                    b = starts[i + 1]
                    ahead.append(_loader.submit(_loader.plan(paths[b:b + batch_size], img_height, img_width, turb)))
                chunks.append(model(_loader.finish(ahead.pop(0), dev)))
        else:
            for b in starts:
                batch = _loader(paths[b:b + batch_size], img_height, img_width, turb)
                chunks.append(model(batch.to(dev, non_blocking=True)))
    fvs = torch.cat(chunks, 0) if chunks else torch.empty(0, 0, device=dev)
    if not keep_on_device:
        fvs = fvs.cpu()
    if verbose:
        print("Features extracted in %.2f seconds" % (time.time() - start))
    return fvs


def get_subset_one_encoder(selected_sample, train_set, topK, encoder, batch_size=500, gpu_index=0):
    """getFeatures.get_subset_one_encoder (getFeatures.py:306-356): top-K neighbours of one sample, a leftover of the clustering pipeline that
    mainKIT.py:39 imports and never calls.  Outside the scope table; the name exists so that the import keeps working."""
    raise NotImplementedError("get_subset_one_encoder is out of scope of this build (SURVEY.md 2.1)")
