"""The image side of the input pipeline on the GPU: what the reference does per image with
torchvision transforms on PIL images (``getFeatures.sample`` getFeatures.py:18-19; ``samplePKBatches.transform``
train_encodersKIT.py:313-320), batched on HIP kernels (``dali_resize_bicubic_u8``, ``dali_augment_batch``).

JPEG decode stays on the host.  Random parameters are drawn here, per image and in torchvision's call order, from
torch's global CPU generator (the reference's transforms draw from it too); the kernels are deterministic pixel
arithmetic that reproduces PIL's 8-bit results bit for bit.  torchvision is not installed in the build image and the
reference pins no version: the draw order follows torchvision 0.1x's ``RandomCrop.get_params``, ``RandomHorizontalFlip``,
``ColorJitter.get_params`` and ``RandomErasing.get_params``.
"""
import ctypes
import math
from functools import lru_cache

import numpy as np
import torch

from . import _lib

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
PRECISION_BITS = 32 - 8 - 2
AUG_WORDS = 16


# ---- Pillow's resampling coefficients (src/libImaging/Resample.c: bicubic_filter, precompute_coeffs, normalize_coeffs_8bpc)
def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


@lru_cache(maxsize=256)
def resize_coeffs(in_size, out_size):
    """-> (ksize, bounds int32 [out,2] = (first input index, taps), coefs int32 [out,ksize] in 22-bit fixed point)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coefs = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        if ww != 0.0:
            k = [w / ww for w in k]
        bounds[xx] = (xmin, xmax)
        for x, w in enumerate(k):
            coefs[xx, x] = int(-0.5 + w * (1 << PRECISION_BITS)) if w < 0 else int(0.5 + w * (1 << PRECISION_BITS))
    return ksize, bounds, coefs


def resize_u8_reference(img, out_h, out_w):
    """numpy restatement of the two 8-bit passes with the tables above (CPU check of the table builder against PIL)."""
    def one_pass(a, out_size):                      # resample axis 1 of [rows][n][3]
        ks, bounds, coefs = resize_coeffs(a.shape[1], out_size)
        out = np.empty((a.shape[0], out_size, a.shape[2]), dtype=np.uint8)
        ai = a.astype(np.int64)
        for xx in range(out_size):
            x0, n = bounds[xx]
            ssum = (1 << (PRECISION_BITS - 1)) + (ai[:, x0:x0 + n, :] * coefs[xx, :n].astype(np.int64)[None, :, None]).sum(1)
            out[:, xx, :] = np.clip(ssum >> PRECISION_BITS, 0, 255)
        return out
    tmp = one_pass(np.asarray(img), out_w)                                   # horizontal first
    return one_pass(tmp.transpose(1, 0, 2), out_h).transpose(1, 0, 2)       # then vertical


def resize_bicubic_u8(images, out_h, out_w, device=None, lane=0):
    """images: sequence of uint8 [h,w,3] arrays (numpy or CPU tensors) of any sizes -> uint8 CUDA tensor [N,out_h,out_w,3]
    = PIL ``img.resize((out_w, out_h), Image.BICUBIC)`` for each.  The horizontally resampled intermediate lives in the workspace of
    the context ``_lib.ctx(device, lane)``: a caller on a stream that runs beside the main one passes its own lane."""
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    arrs = [np.ascontiguousarray(np.asarray(im), dtype=np.uint8) for im in images]
    n = len(arrs)
    out = torch.empty(n, out_h, out_w, 3, device=dev, dtype=torch.uint8)
    if n == 0:
        return out
    sizes = sorted({a.shape[:2] for a in arrs})
    tab = {sz: i for i, sz in enumerate(sizes)}
    ks_h = max(resize_coeffs(w, out_w)[0] for _, w in sizes)
    ks_v = max(resize_coeffs(h, out_h)[0] for h, _ in sizes)
    bh = np.zeros((len(sizes), out_w, 2), np.int32); ch = np.zeros((len(sizes), out_w, ks_h), np.int32)
    bv = np.zeros((len(sizes), out_h, 2), np.int32); cv = np.zeros((len(sizes), out_h, ks_v), np.int32)
    for (h, w), i in tab.items():
        k, b, c = resize_coeffs(w, out_w); bh[i] = b; ch[i, :, :k] = c
        k, b, c = resize_coeffs(h, out_h); bv[i] = b; cv[i, :, :k] = c
    offs = np.zeros(n, np.int64)
    pos = 0
    for i, a in enumerate(arrs):
        assert a.ndim == 3 and a.shape[2] == 3, "RGB uint8 images expected"
        offs[i] = pos
        pos += a.size
    packed = torch.from_numpy(np.concatenate([a.reshape(-1) for a in arrs])).to(dev, non_blocking=True)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev, non_blocking=True)
    t_off, t_h, t_w = up(offs), up(np.array([a.shape[0] for a in arrs], np.int32)), up(np.array([a.shape[1] for a in arrs], np.int32))
    t_tab = up(np.array([tab[a.shape[:2]] for a in arrs], np.int32))
    t_bh, t_ch, t_bv, t_cv = up(bh), up(ch), up(bv), up(cv)
    max_in_h = max(a.shape[0] for a in arrs)
    _lib.check(_lib.lib().dali_resize_bicubic_u8(_lib.ctx(dev, lane), _lib.stream_ptr(), _lib.ptr(packed), _lib.ptr(t_off), _lib.ptr(t_h), _lib.ptr(t_w),
                                                  _lib.ptr(t_tab), n, max_in_h, _lib.ptr(t_bh), _lib.ptr(t_ch), ks_h, _lib.ptr(t_bv), _lib.ptr(t_cv),
                                                  ks_v, out_h, out_w, _lib.ptr(out)), "dali_resize_bicubic_u8")
    return out


# ---- random parameters in torchvision's order ------------------------------------------------------------------
def _f32_bits(x):
    return int(np.array([x], dtype=np.float32).view(np.int32)[0])


def sample_train_params(n, height, width, padding=10, brightness=0.4, contrast=0.3, saturation=0.4, erase_p=1.0,
                        erase_scale=(0.05, 0.30), erase_ratio=(0.3, 3.3)):
    """Per image: RandomCrop((H,W), padding) -> RandomHorizontalFlip(0.5) -> ColorJitter(b, c, s, hue=0) ->
    RandomErasing(p, scale, ratio) parameters (train_encodersKIT.py:313-320), int32 [n,16] (layout: dali_augment_batch)."""
    p = np.zeros((n, AUG_WORDS), dtype=np.int32)
    log_ratio = (math.log(erase_ratio[0]), math.log(erase_ratio[1]))
    area = height * width
    for i in range(n):
        # RandomCrop.get_params on the padded image
        top = int(torch.randint(0, 2 * padding + 1, size=(1,)).item())
        left = int(torch.randint(0, 2 * padding + 1, size=(1,)).item())
        flip = int(torch.rand(1).item() < 0.5)
        # ColorJitter.get_params: permutation first, then the factors; hue = 0 -> None (no draw)
        order = torch.randperm(4).tolist()
        b = float(torch.empty(1).uniform_(max(0.0, 1 - brightness), 1 + brightness))
        c = float(torch.empty(1).uniform_(max(0.0, 1 - contrast), 1 + contrast))
        s = float(torch.empty(1).uniform_(max(0.0, 1 - saturation), 1 + saturation))
        # RandomErasing.forward: the p draw, then get_params (up to 10 attempts)
        ei = ej = eh = ew = 0
        if torch.rand(1).item() < erase_p:
            for _ in range(10):
                erase_area = area * torch.empty(1).uniform_(erase_scale[0], erase_scale[1]).item()
                aspect = torch.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1])).item()
                h = int(round(math.sqrt(erase_area * aspect)))
                w = int(round(math.sqrt(erase_area / aspect)))
                if not (h < height and w < width):
                    continue
                ei = int(torch.randint(0, height - h + 1, size=(1,)).item())
                ej = int(torch.randint(0, width - w + 1, size=(1,)).item())
                eh, ew = h, w
                break
        p[i] = [top, left, flip, *order, ei, ej, eh, ew, _f32_bits(b), _f32_bits(c), _f32_bits(s), padding, 1]
    return p


def eval_params(n):
    p = np.zeros((n, AUG_WORDS), dtype=np.int32)
    p[:, 3:7] = -1
    return p


def augment(images_u8, params, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """images_u8: uint8 CUDA [N,H,W,3]; params int32 [N,16] (numpy or tensor) -> fp32 CUDA [N,3,H,W]."""
    n, h, w, _ = images_u8.shape
    dev = images_u8.device
    prm = torch.as_tensor(np.ascontiguousarray(params), dtype=torch.int32).to(dev) if not isinstance(params, torch.Tensor) else params.to(dev, torch.int32).contiguous()
    assert tuple(prm.shape) == (n, AUG_WORDS)
    out = torch.empty(n, 3, h, w, device=dev, dtype=torch.float32)
    m3, s3 = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    _lib.check(_lib.lib().dali_augment_batch(_lib.ctx(dev), _lib.stream_ptr(), _lib.ptr(images_u8.contiguous()), _lib.ptr(prm), n, h, w, m3, s3,
                                              _lib.ptr(out)), "dali_augment_batch")
    return out


# ---- loaders that plug into getFeatures.set_image_loader / train_encodersKIT.set_train_loader -------------------------
# Host side of the pipeline (the reference: torch DataLoader with 8 worker processes, train_encodersKIT.py:77-83, getFeatures.py:52).
# JPEG decode is the host's share: PIL releases the GIL while it decodes, so a thread pool scales with the cores, and a batch is
# SUBMITTED (its files start decoding) before the previous batch's GPU work is enqueued and FINISHED (one packed upload, ONE
# dali_resize_bicubic_u8 + ONE dali_augment_batch launch for the whole batch, on a side stream) when it is needed:
#     ticket = loader.submit(plan)   ...   images = loader.finish(ticket)
# Random augmentation parameters are drawn at PLAN time, on the calling thread, in the order the sequential per-call path draws
# them, so the batched path reproduces it bit for bit.
_pool = None
_side = {}


def decode_pool():
    """The process-wide decode pool: DALIID_DECODE_THREADS threads (default: the CPUs this process may run on, at most 16)."""
    global _pool
    if _pool is None:
        import os
        from concurrent.futures import ThreadPoolExecutor
        try:
            ncpu = len(os.sched_getaffinity(0))
        except AttributeError:
            ncpu = os.cpu_count() or 1
        _pool = ThreadPoolExecutor(max_workers=max(1, int(os.environ.get("DALIID_DECODE_THREADS", min(16, ncpu)))), thread_name_prefix="dali-decode")
    return _pool


def _decode_one(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"))                        # torchreid.utils.tools.read_image


def _decode(paths):
    if len(paths) <= 2:
        return [_decode_one(p) for p in paths]
    return list(decode_pool().map(_decode_one, paths))


def _turb_path(path, turb):
    from .getFeatures import turb_path
    return turb_path(path, turb)


def _side_stream(dev):
    key = (dev.type, dev.index)
    if key not in _side:
        _side[key] = torch.cuda.Stream(device=dev)
    return _side[key]


class ImagePlan:
    """What one loader call would do, with every random draw already made: files to decode + their [n,16] parameter rows."""
    __slots__ = ("files", "params", "height", "width")

    def __init__(self, files, params, height, width):
        self.files, self.params, self.height, self.width = files, params, height, width

    @staticmethod
    def concat(plans, order=None):
        """One plan for many (all of one output size); ``order``: permutation of the concatenated images (the final batch order)."""
        files = [f for p in plans for f in p.files]
        params = np.concatenate([p.params for p in plans], 0) if plans else np.zeros((0, AUG_WORDS), np.int32)
        if order is not None:
            files, params = [files[i] for i in order], params[np.asarray(order)]
        return ImagePlan(files, params, plans[0].height, plans[0].width)


class _Ticket:
    __slots__ = ("plan", "futures")

    def __init__(self, plan, futures):
        self.plan, self.futures = plan, futures


def submit(plan, decode=None):
    """Start decoding the plan's files on the pool (returns at once)."""
    fn = decode or _decode_one
    return _Ticket(plan, [decode_pool().submit(fn, f) for f in plan.files])


def finish(ticket, device=None, side_stream=True):
    """Wait for the decodes, then ONE resize launch + ONE augment launch for the whole plan.  With ``side_stream`` the upload and the two
    kernels run on the device's side stream (under whatever the current stream is computing) and the current stream waits for them."""
    arrs = [f.result() for f in ticket.futures]
    plan = ticket.plan
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    if not side_stream:
        return augment(resize_bicubic_u8(arrs, plan.height, plan.width, dev), plan.params)
    main = torch.cuda.current_stream(dev)
    side = _side_stream(dev)
    with torch.cuda.stream(side):
        # the side stream's own context: the resize keeps its intermediate in the context's workspace, and the main stream's kernels (Adam's
        # partial sums, the targets, the distance pre-pass) use THEIR context's workspace at the same time
        u8 = resize_bicubic_u8(arrs, plan.height, plan.width, dev, lane="side")
        out = augment(u8, plan.params)
    main.wait_stream(side)
    out.record_stream(main)
    return out


def plan_eval(paths, img_height, img_width, turb=None):
    files = [_turb_path(p, turb) for p in paths] if turb else list(paths)
    return ImagePlan(files, eval_params(len(files)), img_height, img_width)


def plan_train(paths, img_height, img_width, turb=None):
    files = [_turb_path(p, turb) for p in paths] if turb else list(paths)
    return ImagePlan(files, sample_train_params(len(files), img_height, img_width), img_height, img_width)


def gpu_eval_loader(paths, img_height, img_width, turb=None, decode=_decode):
    """getFeatures.sample.__getitem__ for a list of paths: decode (host) -> bicubic resize -> ToTensor -> Normalize (GPU)."""
    plan = plan_eval(paths, img_height, img_width, turb)
    return augment(resize_bicubic_u8(decode(plan.files), img_height, img_width), plan.params)


def gpu_train_loader(paths, img_height, img_width, turb=None, decode=_decode):
    """samplePKBatches.transform for a list of paths (train_encodersKIT.py:313-320)."""
    plan = plan_train(paths, img_height, img_width, turb)
    return augment(resize_bicubic_u8(decode(plan.files), img_height, img_width), plan.params)


# the batched protocol of the two loaders (train_encodersKIT.samplePKBatches.plan / getFeatures.extractFeatures use it when present)
gpu_eval_loader.plan, gpu_eval_loader.submit, gpu_eval_loader.finish = plan_eval, submit, finish
gpu_train_loader.plan, gpu_train_loader.submit, gpu_train_loader.finish = plan_train, submit, finish
