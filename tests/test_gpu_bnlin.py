"""GPU parity of csrc/bnlin.hip and of the fused output stage of the 1x1 convolution kernels, through the C ABI, against plain
PyTorch fp32 / fp64 references of the same operations.

Reference arithmetic being accelerated: torchvision's Bottleneck tail as Encoders.ResNet50ReID runs it (Encoders.py:330-339):
out = bn3(conv3(a2)); out += identity; out = relu(out), training-mode BatchNorm, and its autograd backward."""
import numpy as np
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
    return ops_nn


def bits_of(y):
    """reference mask bytes: bit t of byte i = (y.flatten()[8 i + t] > 0)"""
    b = (y.flatten().float() > 0).to(torch.uint8).reshape(-1, 8)
    return (b * (2 ** torch.arange(8, dtype=torch.uint8, device=y.device))).sum(1).to(torch.uint8)


def ulp_close(got, ref_f32, ulps=1.0):
    """got bf16 vs an fp32 reference: within `ulps` bf16 ulps of the reference (+ tiny absolute slack around zero)"""
    err = (got.float() - ref_f32).abs()
    tol = ulps * 2.0 ** -8 * ref_f32.abs() + 1e-6 * float(ref_f32.abs().max())
    assert bool((err <= tol).all()), (float(err.max()), float((err / tol.clamp(min=1e-30)).max()))


# (pixels, cin, cout): lean staged path of the 128x128 kernel; edge tiles (general path); the 128x256 kernel; the 256x256 k-tile-64 kernel
@pytest.mark.parametrize("P,cin,cout", [(4096, 64, 256), (300, 32, 136), (32768, 512, 2048), (32768, 1024, 512)])
def test_conv1x1_fused_forward_stage(nn, P, cin, cout):
    g = torch.Generator().manual_seed(P + cin + cout)
    x = torch.randn(P, cin, generator=g).to(bf16).cuda()
    w = (torch.randn(cout, cin, generator=g) / cin ** 0.5).to(bf16).cuda()
    scale = (0.5 + torch.rand(cout, generator=g)).cuda()
    shift = (0.3 * torch.randn(cout, generator=g)).cuda()
    res = torch.randn(P, cout, generator=g).to(bf16).cuda()
    acc = x.float() @ w.float().T
    ref = torch.relu(acc * scale + shift + res.float())
    y, bits = nn.conv1x1_fused(x, w, out_scale=scale, out_shift=shift, residual=res, relu=True, want_bits=True)
    ulp_close(y, ref, 1.5)                       # fp32 summation order differs from torch's GEMM
    assert torch.equal(bits, bits_of(y))         # the mask describes the STORED tensor exactly
    # no residual, no relu: a plain affine of the accumulators
    y2 = nn.conv1x1_fused(x, w, out_scale=scale, out_shift=shift)
    ulp_close(y2, acc * scale + shift, 1.5)
    # the residual as another BatchNorm's raw input: res_scale * residual + bias (a bottleneck with a downsample branch)
    rs, rb = (0.5 + torch.rand(cout, generator=g)).cuda(), (0.3 * torch.randn(cout, generator=g)).cuda()
    y3, bits3 = nn.conv1x1_fused(x, w, out_scale=scale, out_shift=shift, bias=rb, residual=res, res_scale=rs, relu=True, want_bits=True)
    ulp_close(y3, torch.relu(acc * scale + shift + rb + rs * res.float()), 1.5)
    assert torch.equal(bits3, bits_of(y3))


@pytest.mark.parametrize("P,cin,cout", [(4096, 64, 256), (300, 32, 136), (32768, 512, 2048), (32768, 2048, 1024)])
def test_conv1x1_fused_masked_gradient_stage(nn, P, cin, cout):
    """the data-gradient use: out = (acc + residual) * mask, also accumulating in place (residual == out)"""
    g = torch.Generator().manual_seed(P + cin + cout + 1)
    x = torch.randn(P, cin, generator=g).to(bf16).cuda()
    w = (torch.randn(cout, cin, generator=g) / cin ** 0.5).to(bf16).cuda()
    res = torch.randn(P, cout, generator=g).to(bf16).cuda()
    keep = torch.rand(P, cout, generator=g).cuda() > 0.4
    mask = bits_of(keep.float())
    ref = (x.float() @ w.float().T + res.float()) * keep
    y = nn.conv1x1_fused(x, w, residual=res, out_mask=mask)
    ulp_close(y, ref, 1.5)
    assert bool((y[~keep] == 0).all())
    buf = res.clone()
    y2 = nn.conv1x1_fused(x, w, residual=buf, out_mask=mask, inplace=True)
    assert y2.data_ptr() == buf.data_ptr() and torch.equal(y2, y)
    # bias (the W^T Kc term of the bnlin data gradient) rides along
    bias = torch.randn(cout, generator=g).cuda()
    y3 = nn.conv1x1_fused(x, w, bias=bias, out_mask=mask)
    ulp_close(y3, (x.float() @ w.float().T + bias) * keep, 1.5)


# The persistent streaming kernel (csrc/fused1x1.h) takes these launches when they have >= 2 tiles of 128 x 128 per CU: the plan's own conv3 /
# masked conv1 gradient shapes at batch 256, a ragged pixel count (edge tile: predicated stores, zero-filled operand rows) and an inference
# batch (500 images: 62.5 pixel tiles per XCD).  Small integers: every product, sum, power-of-two scale and integer shift is exact in fp32,
# so the result is the reference rounded once to bf16 -- bit for bit, mask bits included.
@pytest.mark.parametrize("P,cin,cout", [(524288, 64, 256), (131072, 128, 512), (32768, 256, 1024), (32768, 512, 2048), (65000, 64, 256),
                                        (64000, 256, 1024), (40037, 128, 256), (33000, 512, 1024)])
def test_conv1x1_fused_streaming_kernel_exact_integers(nn, P, cin, cout):
    g = torch.Generator().manual_seed(P + cin + cout + 3)
    dev = "cuda"
    x = torch.randint(-2, 3, (P, cin), generator=g).to(dev)
    w = torch.randint(-1, 2, (cout, cin), generator=g).to(dev)
    res = torch.randint(-3, 4, (P, cout), generator=g, dtype=torch.int8).to(dev)
    scale = torch.tensor([0.5, 1.0, 2.0, 0.25])[torch.randint(0, 4, (cout,), generator=g)].to(dev)
    shift = torch.randint(-3, 4, (cout,), generator=g).float().to(dev)
    rs = torch.tensor([0.5, 1.0, 2.0])[torch.randint(0, 3, (cout,), generator=g)].to(dev)
    rb = torch.randint(-2, 3, (cout,), generator=g).float().to(dev)
    acc = x.float() @ w.float().T                                                   # exact: |acc| <= 2 * 512
    xb, wb, resb = x.to(bf16), w.to(bf16), res.to(bf16)
    # forward use: y = relu(acc * scale + shift + residual), mask bits of the stored tensor
    y, bits = nn.conv1x1_fused(xb, wb, out_scale=scale, out_shift=shift, residual=resb, relu=True, want_bits=True)
    ref = torch.relu(acc * scale + shift + res.float()).to(bf16)
    assert torch.equal(y, ref), (y.float() - ref.float()).abs().max()
    assert torch.equal(bits, bits_of(ref))
    del y, bits
    # a downsample block: the residual is another BatchNorm's raw input (res_scale * residual + bias)
    y3 = nn.conv1x1_fused(xb, wb, out_scale=scale, out_shift=shift, bias=rb, residual=resb, res_scale=rs, relu=True)
    assert torch.equal(y3, torch.relu(acc * scale + shift + rb + rs * res.float()).to(bf16))
    del y3
    # backward use: out = (acc + residual) gated by the mask bits, also in place over the residual
    keep = torch.rand(P, cout, generator=g).to(dev) > 0.4
    mask = bits_of(keep.float())
    ref_g = ((acc + res.float()) * keep).to(bf16)
    yg = nn.conv1x1_fused(xb, wb, residual=resb, out_mask=mask)
    assert torch.equal(yg, ref_g)
    buf = resb.clone()
    yi = nn.conv1x1_fused(xb, wb, residual=buf, out_mask=mask, inplace=True)
    assert yi.data_ptr() == buf.data_ptr() and torch.equal(yi, ref_g)
    del yg, yi, buf
    # mask without a residual (the merged data gradients of the Gram-scheme blocks hand on a masked result)
    ym = nn.conv1x1_fused(xb, wb, bias=rb, out_mask=mask)
    assert torch.equal(ym, ((acc + rb) * keep).to(bf16))


def hard_case(P, C, w, seed):
    """Inputs at the sizes the net plan USES the scheme at (layer1: P = 524288, w = 64; layer2: P = 131072, w = 128 at batch 256) and the
    w >= 512 shape whose column sums come from (split, n tile) rows, with heavy cancellation in var = E[raw^2] - mean^2: activations with
    mean 1 / variance 0.25 per feature, and the first 8 output channels with constant-sign weight rows (mean^2 / var = 4 w >= 256 there;
    a few of the random rows reach 50 as well)."""
    g = torch.Generator().manual_seed(seed)
    a = torch.relu(0.5 * torch.randn(P, w, generator=g) + 1.0).to(bf16).cuda()
    W = torch.randn(C, w, generator=g) / w ** 0.5
    W[:8] = (0.5 + torch.rand(8, 1, generator=g)) / w ** 0.5
    W = W.to(bf16).cuda()
    gamma, beta = (0.5 + torch.rand(C, generator=g)).cuda(), (0.2 * torch.randn(C, generator=g)).cuda()
    return g, a, W, gamma, beta


@pytest.mark.parametrize("P,C,w", [(524288, 256, 64), (131072, 512, 128), (32768, 2048, 512)])
def test_bnlin_forward_statistics_at_plan_sizes(nn, P, C, w):
    g, a, W, gamma, beta = hard_case(P, C, w, P + C + w)
    rm, rv = torch.zeros(C).cuda(), torch.ones(C).cuda()
    raw = a.double() @ W.double().T
    mean, var = raw.mean(0), raw.var(0, unbiased=False)
    ratio = (mean * mean / var)
    assert float(ratio[:8].min()) >= 50 and int((ratio >= 50).sum()) >= 8          # the cancellation the test is about
    o = nn.bnlin_fwd(a, W, gamma, beta, rm, rv)
    np.testing.assert_allclose(o["m2"].cpu().numpy(), a.double().sum(0).cpu().numpy(), rtol=1e-5)
    np.testing.assert_allclose(o["mean"].cpu().numpy(), mean.cpu().numpy(), rtol=1e-4, atol=1e-5)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    err = ((o["invstd"].double() - invstd) / invstd).abs()
    print("bnlin invstd at P=%d w=%d: max rel err %.3e (channels with mean^2/var >= 50: %.3e), worst ratio %.0f"
          % (P, w, float(err.max()), float(err[ratio >= 50].max()), float(ratio.max())))
    assert float(err.max()) < 1e-3, float(err.max())                               # a quarter of one bf16 ulp of the normalised output
    np.testing.assert_allclose(rv.cpu().numpy(), (0.9 + 0.1 * var * P / (P - 1)).cpu().numpy(), rtol=1e-3)


@pytest.mark.parametrize("P,C,w", [(524288, 256, 64), (131072, 512, 128), (32768, 2048, 512)])
def test_bnlin_backward_at_plan_sizes(nn, P, C, w):
    g, a, W, gamma, beta = hard_case(P, C, w, P + C + w + 7)
    dz = (torch.randn(P, C, generator=g) * (torch.rand(P, C, generator=g) > 0.5)).to(bf16).cuda()
    ad, Wd, gd, bd = (t.double().clone().requires_grad_(True) for t in (a, W, gamma, beta))
    out = F.batch_norm(ad @ Wd.T, None, None, gd, bd, True, 0.1, 1e-5)
    (out * dz.double()).sum().backward()
    del out
    fwd = nn.bnlin_fwd(a, W, gamma, beta)
    o = nn.bnlin_bwd(dz, a, W, fwd)
    rel = lambda x, r: float((x.double() - r).norm() / r.norm())
    e_b, e_g, e_w = rel(o["dbeta"], bd.grad), rel(o["dgamma"], gd.grad), rel(o["dW"], Wd.grad)
    print("bnlin backward at P=%d w=%d: dbeta %.2e dgamma %.2e dW %.2e" % (P, w, e_b, e_g, e_w))
    assert e_b < 1e-5 and e_g < 1e-3 and e_w < 2e-3, (e_b, e_g, e_w)
    d_a = nn.conv1x1_fused(dz, o["wd1"], bias=o["bvec"])
    d_a = nn.conv1x1_fused(a, o["wd2"], residual=d_a, inplace=True)
    e = rel(d_a, ad.grad)
    print("bnlin data gradient rel-L2 %.3e" % e)
    assert e < 1.5e-2, e


@pytest.mark.parametrize("P,C,w", [(2048, 128, 32), (8192, 256, 64), (32768, 1024, 256), (4100, 512, 128), (32768, 2048, 512)])
def test_bnlin_forward_statistics(nn, P, C, w):
    g = torch.Generator().manual_seed(P + C + w)
    a = torch.relu(torch.randn(P, w, generator=g) + 0.3).to(bf16).cuda()            # post-ReLU activations: non-zero channel means
    W = (torch.randn(C, w, generator=g) / w ** 0.5).to(bf16).cuda()
    gamma, beta = (0.5 + torch.rand(C, generator=g)).cuda(), (0.2 * torch.randn(C, generator=g)).cuda()
    rm, rv = torch.zeros(C).cuda(), torch.ones(C).cuda()
    raw = a.double() @ W.double().T
    mean, var = raw.mean(0), raw.var(0, unbiased=False)
    o = nn.bnlin_fwd(a, W, gamma, beta, rm, rv)
    np.testing.assert_allclose(o["m2"].cpu().numpy(), a.double().sum(0).cpu().numpy(), rtol=1e-5)
    np.testing.assert_allclose(o["gram"].cpu().numpy(), (a.double().T @ a.double()).cpu().numpy(), rtol=2e-5, atol=1e-3)
    np.testing.assert_allclose(o["mean"].cpu().numpy(), mean.cpu().numpy(), rtol=1e-4, atol=1e-5)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    np.testing.assert_allclose(o["invstd"].cpu().numpy(), invstd.cpu().numpy(), rtol=2e-4)
    np.testing.assert_allclose(o["scale"].cpu().numpy(), (gamma.double() * invstd).cpu().numpy(), rtol=2e-4)
    np.testing.assert_allclose(o["shift"].cpu().numpy(), (beta.double() - mean * gamma.double() * invstd).cpu().numpy(), rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(rm.cpu().numpy(), (0.1 * mean).cpu().numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(rv.cpu().numpy(), (0.9 + 0.1 * var * P / (P - 1)).cpu().numpy(), rtol=2e-4)


@pytest.mark.parametrize("P,C,w", [(2048, 128, 32), (8192, 256, 64), (32768, 1024, 256), (4100, 512, 128), (32768, 2048, 512)])
def test_bnlin_backward_matches_autograd(nn, P, C, w):
    """autograd through raw = a W^T -> training-mode batch_norm, in fp64, vs the moment form: dW, dgamma, dbeta and the data gradient
    assembled from the two weight images exactly as the net plan does (two 1x1 data-gradient GEMMs, the second accumulating in place)"""
    g = torch.Generator().manual_seed(P + C + w + 7)
    a = torch.relu(torch.randn(P, w, generator=g) + 0.3).to(bf16).cuda()
    W = (torch.randn(C, w, generator=g) / w ** 0.5).to(bf16).cuda()
    gamma, beta = (0.5 + torch.rand(C, generator=g)).cuda(), (0.2 * torch.randn(C, generator=g)).cuda()
    dz = (torch.randn(P, C, generator=g) * (torch.rand(P, C, generator=g) > 0.5)).to(bf16).cuda()      # a masked gradient
    ad, Wd, gd, bd = (t.double().clone().requires_grad_(True) for t in (a, W, gamma, beta))
    out = F.batch_norm(ad @ Wd.T, None, None, gd, bd, True, 0.1, 1e-5)
    (out * dz.double()).sum().backward()
    fwd = nn.bnlin_fwd(a, W, gamma, beta)
    o = nn.bnlin_bwd(dz, a, W, fwd)
    rel = lambda x, r: float((x.double() - r).norm() / r.norm())
    assert rel(o["dbeta"], bd.grad) < 1e-5
    assert rel(o["dgamma"], gd.grad) < 2e-4, rel(o["dgamma"], gd.grad)
    assert rel(o["dW"], Wd.grad) < 5e-4, rel(o["dW"], Wd.grad)
    d_a = nn.conv1x1_fused(dz, o["wd1"], bias=o["bvec"])
    d_a = nn.conv1x1_fused(a, o["wd2"], residual=d_a, inplace=True)
    e = rel(d_a, ad.grad)
    print("bnlin data gradient rel-L2 %.3e (bf16 weight images, two bf16 roundings of the result)" % e)
    assert e < 1e-2, e
