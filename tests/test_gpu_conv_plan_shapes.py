"""GPU parity at the BENCHMARKED plan's own sizes (configs[1]: batch 256 of 256 x 128 images): every distinct convolution of
ResNet-50-ReID behind the stem, forward, data gradient (with a masked residual) and weight gradient on small integers, which bf16
holds exactly and whose sums fp32 holds exactly (the longest: 524288 pixels x |4| < 2^24), so every result must equal torch's CPU
fp32 convolution BIT FOR BIT.

Why beside tests/test_gpu_conv.py: tile shapes, XCD super-tile maps, grids, split-K counts and slab layouts are functions of the pixel
count, and at batch 256 they are not the ones the small cases take (4096-tile launches in layer1, 16-way splits of 524288 pixels, the
256 x 256 / 256 x 320 tiles).  A deterministic indexing error in one of them passes every property test of the full-size step (finite,
Adam-sized, bit-reproducible) and drowns in the train-mode tolerances of the end-to-end tests; here it is a failed torch.equal.
The C-ABI entry points take the same launchers and plans as the net (launch_igemm_conv / launch_igemm_wgrad + wgrad_plan)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
bf16 = torch.bfloat16
N = 256

# (h, w, cin, cout, r, stride, pad): ResNet-50 (3,4,6,3) with last_stride 1 on 256 x 128 inputs (Encoders.py:312-351): 64 x 32 after the stem + pool
PLAN = [
    (64, 32, 64, 64, 1, 1, 0), (64, 32, 64, 64, 3, 1, 1), (64, 32, 64, 256, 1, 1, 0), (64, 32, 256, 64, 1, 1, 0),            # layer1
    (64, 32, 256, 128, 1, 1, 0), (64, 32, 128, 128, 3, 2, 1), (32, 16, 128, 512, 1, 1, 0), (64, 32, 256, 512, 1, 2, 0),      # layer2, first block
    (32, 16, 512, 128, 1, 1, 0), (32, 16, 128, 128, 3, 1, 1),                                                               # layer2
    (32, 16, 512, 256, 1, 1, 0), (32, 16, 256, 256, 3, 2, 1), (16, 8, 256, 1024, 1, 1, 0), (32, 16, 512, 1024, 1, 2, 0),     # layer3, first block
    (16, 8, 1024, 256, 1, 1, 0), (16, 8, 256, 256, 3, 1, 1),                                                                # layer3
    (16, 8, 1024, 512, 1, 1, 0), (16, 8, 512, 512, 3, 1, 1), (16, 8, 512, 2048, 1, 1, 0), (16, 8, 1024, 2048, 1, 1, 0),      # layer4 (stride 1), first block
    (16, 8, 2048, 512, 1, 1, 0),                                                                                            # layer4
]


@pytest.fixture(scope="module")
def nn():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from daliid_amd import ops_nn
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    return ops_nn


def _ints(shape, gen, density=1.0):
    t = torch.randint(-2, 3, shape, generator=gen).float()
    if density < 1.0:
        t = t * (torch.rand(shape, generator=gen) < density).float()
    return t


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("case", PLAN, ids=lambda c: "%dx%d_%d-%d_k%d_s%d" % (c[0], c[1], c[2], c[3], c[4], c[5]))
def test_plan_conv_exact_at_batch_256(nn, case):
    h, w, cin, cout, r, stride, pad = case
    gen = torch.Generator().manual_seed(h * 7 + cin * 3 + cout + r * 11 + stride)
    x = _ints((N, cin, h, w), gen).requires_grad_(True)
    wt = _ints((cout, cin, r, r), gen, density=0.5).requires_grad_(True)
    y = F.conv2d(x, wt, stride=stride, padding=pad)
    dy = _ints(tuple(y.shape), gen, density=0.5)
    y.backward(dy)
    xg = nhwc(x.detach()).to(bf16).cuda()
    dyg = nhwc(dy).to(bf16).cuda()
    w_fwd = wt.detach().permute(0, 2, 3, 1).contiguous().to(bf16).cuda()           # [cout][r][s][cin]
    w_dg = wt.detach().permute(1, 2, 3, 0).contiguous().to(bf16).cuda()            # [cin][r][s][cout]
    # forward (the plan's launch carries the BatchNorm statistics epilogue)
    yk, stats = nn.conv2d_fwd(xg, w_fwd, stride, pad, want_stats=True)
    ref_y = nhwc(y.detach())
    assert torch.equal(yk.cpu(), ref_y.to(bf16)), (yk.cpu().float() - ref_y).abs().max()
    assert torch.equal(stats.double().sum(0)[:, 0].cpu(), ref_y.double().sum((0, 1, 2)))       # integer sums: exact in any order
    del yk, stats
    # the inference forward's launch of the same convolution: BatchNorm scale / shift + ReLU applied to the fp32 accumulators in the output stage
    # (dali_conv2d_bn_act; the 3x3 and the 64-channel kernels have their own fused instantiations).  Power-of-two scales and integer shifts keep
    # acc * scale + shift exact in fp32, so the result is the oracle's value rounded once to bf16, bit for bit.
    scale = torch.tensor([0.5, 1.0, 2.0, 0.25])[torch.randint(0, 4, (cout,), generator=gen)]
    shift = torch.randint(-3, 4, (cout,), generator=gen).float()
    ya = nn.conv2d_bn_act(xg, w_fwd, scale.cuda(), shift.cuda(), stride, pad, relu=True)
    ref_a = (ref_y * scale + shift).clamp_min(0)
    assert torch.equal(ya.cpu(), ref_a.to(bf16)), (ya.cpu().float() - ref_a).abs().max()
    del ya
    # data gradient with the identity path's masked residual (dy * (y > 0) formed in the epilogue)
    ref_dx = nhwc(x.grad)
    res = _ints((N, h, w, cin), gen)
    bits = torch.randint(0, 2, (N, h, w, cin), generator=gen)
    packed = (bits.reshape(-1, 8) << torch.arange(8)).sum(1).to(torch.uint8).cuda()
    dxm = nn.conv2d_dgrad(dyg, w_dg, (h, w), stride, pad, residual=res.to(bf16).cuda(), residual_mask=packed)
    assert torch.equal(dxm.cpu(), (ref_dx + res * bits).to(bf16)), (dxm.cpu().float() - (ref_dx + res * bits)).abs().max()
    del dxm
    # weight gradient (fp32, exact; split over the pixels, slabs reduced in a fixed order)
    dwk = nn.conv2d_wgrad(xg, dyg, (r, r), stride, pad)
    ref_dw = wt.grad.permute(0, 2, 3, 1).contiguous()
    assert torch.equal(dwk.cpu(), ref_dw), (dwk.cpu() - ref_dw).abs().max()
