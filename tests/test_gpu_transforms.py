"""GPU: bicubic resize and the training transform against PIL (oracle/augment.py), bit for bit."""
import numpy as np
import pytest
import torch

from oracle import augment as OA

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from daliid_amd import transforms
    return transforms


def _images(rng, sizes):
    # smooth structure + noise so that enhancers / resampling see realistic neighbourhoods and saturated values
    out = []
    for h, w in sizes:
        yy, xx = np.mgrid[0:h, 0:w]
        base = 127 + 100 * np.sin(yy / 9.0)[..., None] * np.cos(xx / 7.0)[..., None] * np.array([1.0, 0.7, -0.8])
        out.append(np.clip(base + rng.normal(0, 40, size=(h, w, 3)), 0, 255).astype(np.uint8))
    return out


def test_resize_ragged_batch_equals_pil(T):
    rng = np.random.default_rng(1)
    sizes = [(128, 64), (128, 64), (300, 117), (64, 64), (700, 350), (256, 128), (17, 9), (128, 64)]
    imgs = _images(rng, sizes)
    out = T.resize_bicubic_u8(imgs, 256, 128).cpu().numpy()
    for i, im in enumerate(imgs):
        assert np.array_equal(out[i], OA.resize(im, 256, 128)), (i, sizes[i])
    out2 = T.resize_bicubic_u8(imgs[:3], 224, 224).cpu().numpy()
    for i in range(3):
        assert np.array_equal(out2[i], OA.resize(imgs[i], 224, 224))
    assert T.resize_bicubic_u8([], 256, 128).shape == (0, 256, 128, 3)


def test_eval_transform_equals_reference_loader(T):
    rng = np.random.default_rng(2)
    imgs = _images(rng, [(128, 64)] * 5 + [(200, 90)])
    u8 = T.resize_bicubic_u8(imgs, 256, 128)
    got = T.augment(u8, T.eval_params(len(imgs))).cpu().numpy()
    for i, im in enumerate(imgs):
        ref = OA.to_tensor_normalize(OA.resize(im, 256, 128))
        assert np.array_equal(got[i], ref)


@pytest.mark.parametrize("hw", [(256, 128), (224, 224), (64, 32)])
def test_train_transform_equals_pil_pipeline(T, hw):
    H, W = hw
    rng = np.random.default_rng(H + W)
    imgs = _images(rng, [(H, W)] * 24)
    torch.manual_seed(3)
    params = T.sample_train_params(len(imgs), H, W)
    # force the rare branches too: extrapolating factors, no erase, every op order start
    params[0, 11:14] = np.array([1.4, 1.3, 1.4], dtype=np.float32).view(np.int32)
    params[1, 11:14] = np.array([0.6, 0.7, 0.6], dtype=np.float32).view(np.int32)
    params[2, 9] = 0
    params[3, 3:7] = [3, 2, 1, 0]
    u8 = torch.from_numpy(np.stack(imgs)).cuda()
    got = T.augment(u8, params).cpu().numpy()
    for i, im in enumerate(imgs):
        ref = OA.train_transform(im, params[i])
        if not np.array_equal(got[i], ref):
            bad = np.argwhere(got[i] != ref)
            raise AssertionError("image %d params %s: %d differing values, first at %s: %r vs %r" %
                                 (i, params[i].tolist(), len(bad), bad[0], got[i][tuple(bad[0])], ref[tuple(bad[0])]))


def test_loaders_plug_into_the_mirrors(T, tmp_path):
    from PIL import Image
    from daliid_amd import getFeatures, train_encodersKIT
    rng = np.random.default_rng(9)
    paths = []
    for i, im in enumerate(_images(rng, [(128, 64)] * 6)):
        p = str(tmp_path / ("%04d_c1s1_%06d_00.png" % (i // 3, i)))          # lossless so that decode is exact
        Image.fromarray(im).save(p)
        paths.append(p)
    records = np.array([[p, str(i // 3), "0", "person"] for i, p in enumerate(paths)])
    getFeatures.set_image_loader(T.gpu_eval_loader)
    train_encodersKIT.set_train_loader(T.gpu_train_loader)
    try:
        x = getFeatures.get_image_loader()(paths, 256, 128, None)
        ref = np.stack([OA.to_tensor_normalize(OA.resize(np.asarray(Image.open(p).convert("RGB")), 256, 128)) for p in paths])
        assert x.is_cuda and np.array_equal(x.cpu().numpy(), ref)
        ds = train_encodersKIT.samplePKBatches("Synthetic", records, np.array([0, 0, 0, 1, 1, 1]), 256, 128, None, 0, K=3)
        imgs, labels, dist = ds[0]
        assert imgs.shape == (3, 3, 256, 128) and imgs.is_cuda and torch.isfinite(imgs).all() and (dist == 0).all()
        assert (imgs == ((0.0 - 0.485) / 0.229)).any()                         # an erased box (value 0 before Normalize)
    finally:
        getFeatures.set_image_loader(None)
        train_encodersKIT.set_train_loader(None)


def _write_dataset(tmp_path, n_ids, per_id, hw=(128, 64), turb=True, fmt="jpg"):
    from PIL import Image
    rng = np.random.default_rng(3)
    clean = tmp_path / "clean"; tdir = tmp_path / "turb"
    clean.mkdir(); tdir.mkdir()
    records = []
    for pid in range(n_ids):
        for k, im in enumerate(_images(rng, [hw] * per_id)):
            name = "%04d_c1s1_%06d_00" % (pid, k)
            Image.fromarray(im).save(str(clean / (name + "." + fmt)), quality=95)
            if turb:
                for s in range(1, 6):
                    Image.fromarray(np.roll(im, s, axis=1)).save(str(tdir / ("%s_turbstrength%d.jpg" % (name, s))), quality=95)
            records.append([str(clean / (name + "." + fmt)), str(pid), "0", "person"])
    return np.array(records), str(tdir)


def test_batched_pk_batch_equals_the_per_identity_calls(T, tmp_path):
    """One PK batch through plan_batch / finish_batch (decode pool, ONE resize + ONE augment launch for the whole batch, side stream)
    against the per-identity, per-distorted-image loader calls of samplePKBatches.__getitem__ (train_encodersKIT.py:365-400): same numpy /
    torch draws in the same order, so the tensors, labels and distortion levels are identical bit for bit."""
    from daliid_amd import train_encodersKIT as TK
    records, tdir = _write_dataset(tmp_path, 4, 5)
    labels = np.int32(records[:, 1])

    def run(batched):
        np.random.seed(21); torch.manual_seed(22)
        ds = TK.samplePKBatches("Market", records, labels, 64, 32, tdir, 1, K=3)
        ids = [2, 0, 3]
        if batched:
            imgs, lab, dist = ds.finish_batch(ds.plan_batch(ids, T.gpu_train_loader), T.gpu_train_loader, torch.device("cuda", 0))
        else:
            parts = [ds[i] for i in ids]
            imgs, lab, dist = torch.cat([p[0] for p in parts], 0), torch.cat([p[1] for p in parts], 0), np.concatenate([p[2] for p in parts])
        torch.cuda.synchronize()
        return imgs.cpu(), lab, dist, np.random.rand(), torch.rand(1).item()          # the generators end in the same state too

    TK.set_train_loader(None)
    a, b = run(False), run(True)
    assert a[0].shape == (18, 3, 64, 32) and torch.equal(a[0], b[0])
    assert torch.equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3] == b[3] and a[4] == b[4]
    assert (a[2].reshape(-1, 2)[:, 0] == 0).all() and (a[2].reshape(-1, 2)[:, 1] >= 1).all()


def test_prefetch_overlaps_decode_with_gpu_work(T, tmp_path, monkeypatch):
    """The PK loop of trainer.train with the batched loader: epoch wall time ~ max(host decode / threads, GPU), not their sum.  The decode is
    made artificially slow and GIL-free (sleep) so that the statement does not depend on the box: 6 batches x 24 files x 20 ms = 2.9 s of
    decode, 8 threads -> 0.36 s if parallel; the GPU step is stood in for by a 60 ms device spin per batch.  Sequential sum would be
    > 3.2 s; overlapped ~ max(0.36, 0.36) + fill."""
    import time
    from daliid_amd import train_encodersKIT as TK
    records, tdir = _write_dataset(tmp_path, 12, 4, hw=(64, 32))
    labels = np.int32(records[:, 1])
    monkeypatch.setenv("DALIID_DECODE_THREADS", "8")
    monkeypatch.setattr(T, "_pool", None)
    real = T._decode_one

    def slow_decode(path):
        time.sleep(0.02)
        return real(path)
    monkeypatch.setattr(T, "_decode_one", slow_decode)
    ds = TK.samplePKBatches("Market", records, labels, 64, 32, tdir, 1, K=4)
    loader = T.gpu_train_loader
    dev = torch.device("cuda", 0)
    spin = torch.zeros(1, device=dev)
    m = torch.randn(4096, 4096, device=dev)
    torch.mm(m, m); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        torch.mm(m, m)
    e1.record(); torch.cuda.synchronize()
    t_mm = e0.elapsed_time(e1) / 10 * 1e-3
    reps = max(1, int(round(0.06 / t_mm)))

    def gpu_step(imgs):                                  # ~60 ms of device time (calibrated above) that reads the batch
        spin.add_(imgs.sum())
        for _ in range(reps):
            torch.mm(m, m)

    def epoch(depth):
        np.random.seed(1); torch.manual_seed(1)
        batches = [[2 * b, 2 * b + 1] for b in range(6)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pending, nxt = [], 0
        for b in range(len(batches)):
            while nxt < len(batches) and len(pending) < 1 + depth:
                pending.append(ds.plan_batch(batches[nxt], loader)); nxt += 1
            imgs, _, _ = ds.finish_batch(pending.pop(0), loader, dev)
            gpu_step(imgs)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    epoch(2)                                             # warm-up (pool threads, allocator)
    t_seq_decode = 6 * 16 * 0.02                         # 16 files per batch (2 ids x 4 x (clean, distorted)), one after the other
    t_gpu = 6 * reps * t_mm
    t = epoch(2)
    print("prefetched epoch %.3f s (serial decode alone %.2f s, GPU alone %.2f s)" % (t, t_seq_decode, t_gpu))
    assert t < 0.5 * (t_seq_decode + t_gpu)              # nowhere near the sum
    assert t < 2.5 * max(t_seq_decode / 8, t_gpu) + 0.15


def test_extract_features_batched_path_equals_per_batch_loader_calls(T, tmp_path):
    """getFeatures.extractFeatures with the batched loader protocol (batch i + 1 decoding on the pool while batch i's forward runs, resize +
    normalise on a side stream) against the plain per-batch loader calls: identical features, in dataset order, ragged last batch included."""
    from daliid_amd import Encoders, getFeatures
    records, _ = _write_dataset(tmp_path, 5, 3, hw=(96, 40), turb=False, fmt="png")
    net = Encoders._DataParallelShim(Encoders.ResNet50ReID(layers=(1, 1, 1, 1), width=32, seed=3)).eval()
    try:
        getFeatures.set_image_loader(T.gpu_eval_loader)
        batched = getFeatures.extractFeatures(records, 64, 32, net, 4, gpu_index=0, verbose=False)          # 15 images: batches of 4, 4, 4, 3
        plain = lambda paths, h, w, turb=None: T.gpu_eval_loader(paths, h, w, turb)                       # same pixels, no plan / submit / finish attributes
        getFeatures.set_image_loader(plain)
        sequential = getFeatures.extractFeatures(records, 64, 32, net, 4, gpu_index=0, verbose=False)
    finally:
        getFeatures.set_image_loader(None)
    assert batched.shape == (15, 1024) and torch.equal(batched, sequential)


def test_side_stream_finish_beside_workspace_users_of_the_main_stream(T):
    """finish(side_stream=True) runs the resize on a side stream that does NOT wait for the main stream.  The resize keeps its horizontally
    resampled intermediate in a context workspace; dali_adam_step (its partial sums of sum theta^2, every step) and dali_class_targets use
    the workspace of the main stream's context from offset 0.  The side stream therefore has its own context (``_lib.ctx(dev, "side")``):
    with both running at the same time, the images equal the sequential path bit for bit and the Adam statistic equals its solo value."""
    from daliid_amd import _lib
    dev = torch.device("cuda", 0)
    assert _lib.ctx(dev, "side").value != _lib.ctx(dev).value
    rng = np.random.default_rng(7)
    imgs = _images(rng, [(160, 80)] * 48)
    plan = T.ImagePlan(["mem://%d" % i for i in range(len(imgs))], T.eval_params(len(imgs)), 256, 128)
    want = T.augment(T.resize_bicubic_u8(imgs, 256, 128, dev), plan.params)
    n = 24 * 1024 * 1024                                                     # a parameter buffer of ViT size: Adam's launch takes ~0.5 ms
    g = torch.Generator(device=dev).manual_seed(3)
    p0 = torch.randn(n, device=dev, generator=g) * 0.1
    grad = torch.randn(n, device=dev, generator=g) * 0.01
    L = _lib.lib()

    def adam(p, m, v, sq):
        _lib.check(L.dali_adam_step(_lib.ctx(dev), _lib.stream_ptr(), _lib.ptr(p), _lib.ptr(grad), _lib.ptr(m), _lib.ptr(v), n, 3.5e-4, 0.9, 0.999,
                                    1e-8, 5e-4, 1, 1.0, _lib.ptr(sq)), "dali_adam_step")
    p, m, v, sq_solo = p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev), torch.zeros(1, device=dev)
    adam(p, m, v, sq_solo)
    torch.cuda.synchronize()
    for rep in range(4):
        sqs = [torch.zeros(1, device=dev) for _ in range(6)]
        ticket = T.submit(plan, decode=lambda f: imgs[int(f.split("/")[-1])])
        torch.cuda.synchronize()
        for i in range(3):
            adam(p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev), sqs[i])
        got = T.finish(ticket, dev, side_stream=True)                       # enqueued while the Adam launches above are still running
        for i in range(3, 6):
            adam(p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev), sqs[i])
        torch.cuda.synchronize()
        assert torch.equal(got, want), rep
        for s in sqs:
            assert torch.equal(s, sq_solo), (rep, float(s), float(sq_solo))
