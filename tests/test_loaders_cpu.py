"""CPU: the default image loaders read the files the reference reads.

getFeatures.sample.__getitem__ (getFeatures.py:22-38) and samplePKBatches.__getitem__ (train_encodersKIT.py:365-377) replace a
clean image by ``<turb_dir>/<name>_turbstrength<k>.jpg`` (MSMT17: ``<pid>_<name>_...``).  A loader that ignores the request
would feed clean pixels under a distortion label."""
import os

import numpy as np
import torch
import pytest
from PIL import Image

from daliid_amd import getFeatures, train_encodersKIT


def _write(path, value):
    Image.fromarray(np.full((12, 6, 3), value, dtype=np.uint8)).save(path, quality=100)


@pytest.fixture()
def files(tmp_path):
    clean = tmp_path / "clean"; turb = tmp_path / "turb"
    clean.mkdir(); turb.mkdir()
    _write(str(clean / "0002_c1s1_000451_03.jpg"), 20)
    for k in range(1, 6):
        _write(str(turb / ("0002_c1s1_000451_03_turbstrength%d.jpg" % k)), 40 * k)
        _write(str(turb / ("0002_0002_c1s1_000451_03_turbstrength%d.jpg" % k)), 40 * k + 10)      # the MSMT17 naming
    return str(clean / "0002_c1s1_000451_03.jpg"), str(turb)


def _level(t):
    """mean 0..255 pixel value back from the normalised tensor"""
    mean = np.array([0.485, 0.456, 0.406]).reshape(3, 1, 1); std = np.array([0.229, 0.224, 0.225]).reshape(3, 1, 1)
    return float(((t[0].numpy() * std + mean) * 255.0).mean())


def test_turb_path_naming(files):
    path, turb = files
    assert getFeatures.turb_path(path, (turb, 3, "Market")) == os.path.join(turb, "0002_c1s1_000451_03_turbstrength3.jpg")
    assert getFeatures.turb_path(path, (turb, 3, "MSMT17")) == os.path.join(turb, "0002_0002_c1s1_000451_03_turbstrength3.jpg")
    with pytest.raises(ValueError):
        getFeatures.turb_path(path, (turb, None, "Market"))


def test_default_eval_loader_reads_the_distorted_file(files):
    path, turb = files
    loader = getFeatures.get_image_loader()
    assert loader is getFeatures.pil_loader
    assert abs(_level(loader([path], 12, 6, None)) - 20) < 1.5
    for k in (1, 4):
        assert abs(_level(loader([path], 12, 6, (turb, k, "Market"))) - 40 * k) < 1.5
        assert abs(_level(loader([path], 12, 6, (turb, k, "MSMT17"))) - (40 * k + 10)) < 1.5
    with pytest.raises(FileNotFoundError):
        loader([path], 12, 6, (turb, 9, "Market"))


def test_default_train_loader_is_the_augmenting_one():
    """No silent fallback to the evaluation transform: the default training loader is the train_encodersKIT.py:313-320
    transform (GPU); anything else has to be installed explicitly."""
    from daliid_amd import transforms
    train_encodersKIT.set_train_loader(None)
    assert train_encodersKIT.get_train_loader() is transforms.gpu_train_loader
    marker = lambda *a, **k: None
    train_encodersKIT.set_train_loader(marker)
    try:
        assert train_encodersKIT.get_train_loader() is marker
    finally:
        train_encodersKIT.set_train_loader(None)


def test_pk_sampler_pairs_clean_with_distorted_file(files, monkeypatch):
    """samplePKBatches with kind_of_transform == 1 asks the loader for the turbulence file of a random strength 1..5 and
    labels the pair (0, strength)."""
    path, turb = files
    train_encodersKIT.set_train_loader(getFeatures.pil_loader)        # explicit: plain decode, no augmentation
    try:
        recs = np.array([[path, "2", "1", "person"]] * 3)
        ds = train_encodersKIT.samplePKBatches("Market", recs, np.array([2, 2, 2]), 12, 6, turb, 1, K=2)
        np.random.seed(0)
        imgs, labels, dist = ds[0]
        assert imgs.shape[0] == 4 and list(dist[0::2]) == [0, 0] and all(1 <= d <= 5 for d in dist[1::2])
        for j in range(2):
            assert abs(_level(imgs[2 * j:2 * j + 1]) - 20) < 1.5
            assert abs(_level(imgs[2 * j + 1:2 * j + 2]) - 40 * int(dist[2 * j + 1])) < 1.5
        assert float(labels[0]) == 2.0
    finally:
        train_encodersKIT.set_train_loader(None)


def test_pk_sampler_keeps_person_records_only():
    """train_encodersKIT.py:299,339,348: the K picks run over all rows of the identity, then rows whose kind (column 3) is not 'person' are
    dropped without a draw; the plan path (batched loader protocol) makes the same selection."""
    seen = []

    def loader(paths, h, w, turb=None):
        seen.append(list(paths))
        return torch.zeros(len(paths), 3, h, w)
    loader.plan = lambda paths, h, w, turb=None: type("P", (), {"files": list(paths), "concat": None})()
    train_encodersKIT.set_train_loader(loader)
    try:
        recs = np.array([["a%d" % i, "5", "0", "person" if i % 2 == 0 else "vehicle"] for i in range(6)])
        ds = train_encodersKIT.samplePKBatches("Market", recs, np.full(6, 5), 8, 4, None, 0, K=6)
        np.random.seed(1)
        imgs, labels, dist = ds[0]
        assert sorted(seen[-1]) == ["a0", "a2", "a4"] and imgs.shape[0] == 3 and labels.shape[0] == 3 and len(dist) == 3
        np.random.seed(1)
        plan, plabels, pdist = ds.plan(0, loader)
        assert plan.files == seen[-1] and plabels.shape[0] == 3                    # same picks, same order
        # an identity whose picks hold no person record is an error that says so (the reference dies on ``[].shape``, :397)
        recs2 = np.array([["b%d" % i, "7", "0", "vehicle"] for i in range(3)])
        ds2 = train_encodersKIT.samplePKBatches("Market", recs2, np.full(3, 7), 8, 4, None, 0, K=2)
        import pytest
        from daliid_amd._lib import DaliError
        with pytest.raises(DaliError):
            ds2[0]
        # record arrays without a kind column (in-memory synthetic sets) are all persons
        ds3 = train_encodersKIT.samplePKBatches("Synthetic", recs[:, :3], np.full(6, 5), 8, 4, None, 0, K=6)
        assert ds3[0][0].shape[0] == 6
    finally:
        train_encodersKIT.set_train_loader(None)


def test_loss_heads_refuse_more_proxies_per_identity_than_the_kernel_holds():
    """LossHeads never reads the kernel's status word in the hot loop (no host sync there): the documented limit dali_proxy_kmax() is
    checked on the host when the epoch's targets are built."""
    import pytest
    from daliid_amd import losses
    from daliid_amd._lib import DaliError
    k = losses._kmax()
    c = torch.nn.functional.normalize(torch.randn(2, 16))
    ok = torch.nn.functional.normalize(torch.randn(2 * k, 16))
    losses.LossHeads(c, np.arange(2), ok, np.repeat(np.arange(2), k), 0.05, 0.4)
    bad = torch.nn.functional.normalize(torch.randn(k + 2, 16))
    with pytest.raises(DaliError):
        losses.LossHeads(c, np.arange(2), bad, np.array([0] * (k + 1) + [1]), 0.05, 0.4)
