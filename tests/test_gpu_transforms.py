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
