"""GPU parity of the one-launch inference stem (daliid_amd/csrc/stem.hip, C ABI dali_stem_conv_bn_maxpool): conv1 7x7 / 2 -> bn1 by running
statistics (no ReLU: Encoders.py:321-322, :334) -> 3x3 / 2 max-pool, torchvision's conv1 / bn1 / maxpool under Encoders.py:33,36 as
getFeatures.py:56-67 forwards them.

Integer images / weights / shifts and power-of-two scales of both signs: bf16 holds the operands exactly, fp32 holds every sum exactly, so
the convolution's output is an exact fp32 number, its bf16 rounding (the rounding point of the three-launch form, which stores that tensor) is
the one torch's `.to(bfloat16)` makes, the affine of it is exact again, and the pooled tensor must equal torch's CPU fp32 convolution -> bf16 ->
affine -> max_pool2d -> bf16 BIT FOR BIT -- at image borders (windows clamped inside the kernel), for negative scales
(the kernel pools sign(scale) * raw and applies the affine to the selected element: the maximum of the affine values), with more tiles than resident workgroups (the persistent loop and its register-staged
prefetch) and at the benchmarked 256 x 128 size."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
bf16 = torch.bfloat16


@pytest.fixture(scope="module")
def nn():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from daliid_amd import ops_nn
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    return ops_nn


def _reference(img, w_ohwi, scale, shift):
    raw = F.conv2d(img, w_ohwi.permute(0, 3, 1, 2).contiguous(), stride=2, padding=3).to(bf16).float()
    z = raw * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    return F.max_pool2d(z, 3, 2, 1).to(bf16).float().permute(0, 2, 3, 1).contiguous()        # NHWC


@pytest.mark.parametrize("n,h,w", [(3, 64, 32), (5, 96, 64), (2, 256, 128), (40, 256, 128), (1, 32, 128), (530, 64, 32)])
def test_stem_conv_bn_maxpool_exact_integers(nn, n, h, w):
    from daliid_amd import _lib
    assert _lib.lib().dali_stem_fused_supported(n, h, w) == 1
    g = torch.Generator().manual_seed(1000 * n + h + w)
    img = torch.randint(-3, 4, (n, 3, h, w), generator=g).float()
    wt = torch.randint(-2, 3, (64, 7, 7, 3), generator=g).float()
    scale = torch.tensor([0.25, -0.5, 1.0, -0.125])[torch.randint(0, 4, (64,), generator=g)]
    shift = torch.randint(-40, 41, (64,), generator=g).float()
    want = _reference(img, wt, scale, shift)
    got = nn.stem_conv_bn_maxpool(img.cuda(), wt.cuda(), scale.cuda(), shift.cuda())
    torch.cuda.synchronize()
    assert got.shape == (n, h // 4, w // 4, 64) and got.dtype == bf16
    assert torch.equal(got.float().cpu(), want)
    # every output row / column / channel position differs from a constant (a stuck tile or channel block would pass a sum check, not this)
    assert want.std() > 1.0


def test_unsupported_shapes_are_refused_not_approximated(nn):
    from daliid_amd import _lib
    assert _lib.lib().dali_stem_fused_supported(4, 64, 48) == 0          # 24 convolution columns: not whole groups of 16
    assert _lib.lib().dali_stem_fused_supported(4, 64, 96) == 0          # 24 pooled columns: the kernel's pixel arithmetic is shifts
    assert _lib.lib().dali_stem_fused_supported(4, 64, 256) == 0         # wider than the register-staged patch covers
    img = torch.zeros(4, 3, 64, 48, device="cuda")
    with pytest.raises(_lib.DaliError):
        nn.stem_conv_bn_maxpool(img, torch.zeros(64, 7, 7, 3, device="cuda"), torch.ones(64, device="cuda"), torch.zeros(64, device="cuda"))


def test_net_plan_inference_stem_equals_the_three_launch_form():
    """dali_resnet_forward(training = 0) takes the one-launch stem; DALI_EVAL_STEM=0 keeps conv -> (stored bf16) -> bn + pool.  Same rounding points
    (bf16 of the convolution's output, fp32 affine, bf16 of the pooled value): the pooled tensors and the embeddings are identical."""
    from daliid_amd import Encoders, _lib
    net = Encoders.ResNet50ReID(seed=5).eval()
    with torch.no_grad():
        net.bn1.weight.mul_(torch.where(torch.arange(64, device="cuda") % 3 == 0, -1.0, 1.0))        # both signs of the scale
    x = torch.randn(6, 3, 64, 32, device="cuda")
    outs, embs = {}, {}
    for flag in ("1", "0"):
        os.environ["DALI_EVAL_STEM"] = flag
        _lib.lib().dali_debug_reload_env()
        with torch.no_grad():
            embs[flag] = net(x).clone()
        outs[flag] = net.debug_tensor("pool0", bf16, (6, 16, 8, 64)).float().clone()
    os.environ.pop("DALI_EVAL_STEM")
    _lib.lib().dali_debug_reload_env()
    assert outs["1"].abs().max() > 0 and outs["1"].std() > 0.01
    assert torch.equal(outs["1"], outs["0"])
    assert torch.equal(embs["1"], embs["0"])


@pytest.mark.parametrize("n,h", [(3, 64), (40, 256)])
def test_training_stem_convolution_and_statistics_exact_integers(n, h):
    """Training forward at width 64 and W = 128: conv1's output and the BatchNorm partial sums come from the patch kernel (stem.hip,
    stem_conv_stats_kernel; DALI_TRAIN_STEM=0 keeps the implicit-GEMM launch).  Integer images and weights: raw0 equals torch's CPU
    convolution bit for bit, the batch mean / variance (fp32 partial sums per wave and tile, fixed-order sums, fp64 finish) equal the exact
    ones to fp32 rounding; the two paths give identical raw0 and identical statistics' consumers (pool0) -- also with more tiles than
    resident workgroups (n = 40: 1280 tiles on 512 workgroups)."""
    from daliid_amd import Encoders, _lib
    g = torch.Generator().manual_seed(77 + n)
    net = Encoders.ResNet50ReID(layers=(1, 1, 1, 1), seed=3)
    sd = net.state_dict()
    sd["conv1.weight"] = torch.randint(-2, 3, tuple(sd["conv1.weight"].shape), generator=g).float().cuda()
    net.load_state_dict(sd)
    x = torch.randint(-3, 4, (n, 3, h, 128), generator=g).float()
    ref_raw0 = F.conv2d(x, sd["conv1.weight"].cpu(), stride=2, padding=3).permute(0, 2, 3, 1).contiguous()      # exact integers
    outs = {}
    for flag in ("1", "0"):
        os.environ["DALI_TRAIN_STEM"] = flag
        _lib.lib().dali_debug_reload_env()
        net.train()
        with torch.no_grad():
            net(x.cuda())
        raw0 = net.debug_tensor("raw0", bf16, (n, h // 2, 64, 64)).float().cpu()
        mean = net.debug_tensor("bn1.mean", torch.float32, (64,)).cpu()
        pool0 = net.debug_tensor("pool0", bf16, (n, h // 4, 32, 64)).float().cpu()
        outs[flag] = (raw0, mean, pool0)
    os.environ.pop("DALI_TRAIN_STEM")
    _lib.lib().dali_debug_reload_env()
    want = ref_raw0.to(bf16).float()
    for flag in ("1", "0"):
        assert torch.equal(outs[flag][0], want), flag
        assert torch.allclose(outs[flag][1].double(), ref_raw0.double().mean((0, 1, 2)), rtol=1e-6, atol=1e-6), flag
    assert torch.equal(outs["1"][2], outs["0"][2])
