"""CPU: the resize coefficient tables (host logic of the GPU pipeline) reproduce PIL's bicubic resize bit for bit, and
the parameter sampler is well-formed and deterministic under torch's seed."""
import numpy as np
import pytest
import torch

from oracle import augment as OA


@pytest.fixture(scope="module")
def T():
    import importlib.util, os, sys, types
    # daliid_amd.transforms imports the ctypes table but not the GPU; load it without initialising anything
    from daliid_amd import transforms
    return transforms


@pytest.mark.parametrize("in_hw,out_hw", [((128, 64), (256, 128)), ((300, 117), (256, 128)), ((64, 64), (224, 224)), ((700, 350), (256, 128)),
                                          ((256, 128), (256, 128)), ((5, 3), (16, 8))])
def test_resize_tables_match_pil_bit_for_bit(T, in_hw, out_hw):
    rng = np.random.default_rng(sum(in_hw) + sum(out_hw))
    img = rng.integers(0, 256, size=(*in_hw, 3), dtype=np.uint8)
    mine = T.resize_u8_reference(img, *out_hw)
    assert np.array_equal(mine, OA.resize(img, *out_hw))


def test_train_params_layout_and_determinism(T):
    torch.manual_seed(5)
    a = T.sample_train_params(64, 256, 128)
    torch.manual_seed(5)
    b = T.sample_train_params(64, 256, 128)
    assert np.array_equal(a, b) and a.shape == (64, 16) and a.dtype == np.int32
    assert (a[:, 0] >= 0).all() and (a[:, 0] <= 20).all() and (a[:, 1] >= 0).all() and (a[:, 1] <= 20).all()
    assert set(np.unique(a[:, 2])) <= {0, 1} and 10 < a[:, 2].sum() < 54
    assert all(sorted(r) == [0, 1, 2, 3] for r in a[:, 3:7].tolist())
    f = a[:, 11:14].copy().view(np.float32)
    assert (f[:, 0] >= 0.6).all() and (f[:, 0] <= 1.4).all() and (f[:, 1] >= 0.7).all() and (f[:, 1] <= 1.3).all()
    eh, ew = a[:, 9], a[:, 10]
    frac = eh * ew / (256 * 128)
    assert (eh < 256).all() and (ew < 128).all() and (frac[eh > 0] > 0.03).all() and (frac < 0.34).all()
    assert (a[:, 7] + eh <= 256).all() and (a[:, 8] + ew <= 128).all()
    assert (a[:, 14] == 10).all() and (a[:, 15] == 1).all()
