"""GPU parity of the HBM-bound trunk kernels (BatchNorm fwd/bwd, stem max-pool, head pooling, BN1d neck) against
torch CPU autograd on the SAME bf16 inputs.  fp32 math on both sides: tolerances are one bf16 ulp on bf16 outputs
and 1e-4-level on fp32 outputs."""
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


def close_bf16(got, ref, ulps=1.0, atol=1e-3):
    got, ref = got.float().cpu(), ref.float()
    err = (got - ref).abs()
    ok = err <= ulps * 2.0 ** -7 * ref.abs() + atol
    assert ok.all(), "max err %.4g at ref %.4g" % (float(err.max()), float(ref.flatten()[err.argmax()]))


def nchw(t):
    return t.permute(0, 3, 1, 2)


@pytest.mark.parametrize("shape", [(3, 5, 7, 64), (2, 8, 4, 256), (1, 3, 3, 2048), (2, 6, 5, 96)])
def test_bn_stats_finalize_act(nn, shape):
    g = torch.Generator().manual_seed(sum(shape))
    n, h, w, C = shape
    x = (torch.randn(n, h, w, 32, generator=g)).to(bf16)
    wt = (torch.randn(C, 1, 1, 32, generator=g) * 0.3).to(bf16)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    raw, stats = nn.conv2d_fwd(x.cuda(), wt.cuda(), want_stats=True)
    u = F.conv2d(nchw(x.float()), wt.float().permute(0, 3, 1, 2))          # fp32 result the statistics are taken from
    rm_g, rv_g = rm.clone().cuda(), rv.clone().cuda()
    scale, shift, mean, invstd = nn.bn_finalize(stats, n * h * w, gamma.cuda(), beta.cuda(), rm_g, rv_g)
    ref_mean, ref_var = u.mean((0, 2, 3)), u.var((0, 2, 3), unbiased=False)
    np.testing.assert_allclose(mean.cpu().numpy(), ref_mean.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(invstd.cpu().numpy(), (1 / torch.sqrt(ref_var + 1e-5)).numpy(), rtol=1e-4)
    cnt = n * h * w
    np.testing.assert_allclose(rm_g.cpu().numpy(), (0.9 * rm + 0.1 * ref_mean).numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(rv_g.cpu().numpy(), (0.9 * rv + 0.1 * ref_var * cnt / (cnt - 1)).numpy(), rtol=1e-4, atol=1e-5)
    rawf = raw.float().cpu()
    sc, sh = scale.cpu(), shift.cpu()
    idn = torch.randn(n, h, w, C, generator=g).to(bf16)
    close_bf16(nn.bn_act(raw, scale, shift, identity=idn.cuda(), relu=True), F.relu(rawf * sc + sh + idn.float()))
    close_bf16(nn.bn_act(raw, scale, shift, raw2=idn.cuda(), scale2=scale, shift2=shift, relu=True),
               F.relu(rawf * sc + sh + idn.float() * sc + sh))
    close_bf16(nn.bn_act(raw, scale, shift, relu=False), rawf * sc + sh)
    # the 1-bit ReLU mask written beside y (what the backward reads instead of y)
    yk, bits = nn.bn_act(raw, scale, shift, identity=idn.cuda(), relu=True, want_mask=True)
    expect = (yk.flatten().float() > 0).view(-1, 8).to(torch.int32)
    expect = (expect << torch.arange(8, device=expect.device, dtype=torch.int32)).sum(1).to(torch.uint8)
    assert torch.equal(bits, expect)


def _bn_ref(raw, gamma, beta, eps=1e-5):
    """train-mode BN on an NHWC fp32 tensor with autograd; also returns mean / invstd / scale / shift"""
    mean = raw.mean((0, 1, 2)); var = raw.var((0, 1, 2), unbiased=False)
    invstd = 1 / torch.sqrt(var + eps)
    return (raw - mean) * invstd * gamma + beta, mean.detach(), invstd.detach(), (gamma * invstd).detach(), (beta - mean * gamma * invstd).detach()


@pytest.mark.parametrize("shape", [(4, 6, 5, 64), (2, 8, 4, 512), (3, 3, 3, 2048), (2, 5, 5, 96)])
def test_bn_bwd_inner_and_block_output(nn, shape):
    g = torch.Generator().manual_seed(sum(shape) + 1)
    n, h, w, C = shape
    raw = torch.randn(n, h, w, C, generator=g).to(bf16)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    grad = torch.randn(n, h, w, C, generator=g).to(bf16)
    # (b) inner BN+ReLU: a = relu(bn(raw)); mask recomputed from raw*scale+shift
    r = raw.float().requires_grad_(True); gm = gamma.clone().requires_grad_(True); bt = beta.clone().requires_grad_(True)
    z, mean, invstd, scale, shift = _bn_ref(r, gm, bt)
    F.relu(z).backward(grad.float())
    draw, dg, db = nn.bn_bwd(grad.cuda(), raw.cuda(), mean.cuda(), invstd.cuda(), scale.cuda(), shift.cuda(), relu=True)
    close_bf16(draw, r.grad, ulps=1.5, atol=2e-3 * float(r.grad.abs().max()))
    np.testing.assert_allclose(dg.cpu().numpy(), gm.grad.numpy(), rtol=2e-3, atol=2e-3 * float(gm.grad.abs().max()))
    np.testing.assert_allclose(db.cpu().numpy(), bt.grad.numpy(), rtol=2e-3, atol=2e-3 * float(bt.grad.abs().max()))
    # (a) block output: y = relu(bn3(raw) + bn_d(raw_b)), mask from y; both BN backward in one pass + dz
    raw_b = torch.randn(n, h, w, C, generator=g).to(bf16)
    gamma_b, beta_b = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    r = raw.float().requires_grad_(True); rb = raw_b.float().requires_grad_(True)
    gm = gamma.clone().requires_grad_(True); bt = beta.clone().requires_grad_(True)
    gmb = gamma_b.clone().requires_grad_(True); btb = beta_b.clone().requires_grad_(True)
    z, mean, invstd, scale, shift = _bn_ref(r, gm, bt)
    zb, mean_b, invstd_b, scale_b, shift_b = _bn_ref(rb, gmb, btb)
    y = F.relu(z + zb)
    y.backward(grad.float())
    y16 = y.detach().to(bf16)
    out = nn.bn_bwd(grad.cuda(), raw.cuda(), mean.cuda(), invstd.cuda(), scale.cuda(), shift.cuda(), ymask=y16.cuda(), relu=True,
                    side_b=(raw_b.cuda(), mean_b.cuda(), invstd_b.cuda(), scale_b.cuda()), want_dz=True)
    draw_a, dga, dba, draw_b, dgb, dbb, dz = out
    mask = (y16.float() > 0).float()
    assert torch.equal(dz.float().cpu(), grad.float() * mask)
    # the oracle's mask is y>0 in fp32; y16>0 differs only where y underflows bf16 -> none here
    close_bf16(draw_a, r.grad, ulps=1.5, atol=2e-3 * float(r.grad.abs().max()))
    close_bf16(draw_b, rb.grad, ulps=1.5, atol=2e-3 * float(rb.grad.abs().max()))
    for got, ref in ((dga, gm.grad), (dba, bt.grad), (dgb, gmb.grad), (dbb, btb.grad)):
        np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=2e-3, atol=2e-3 * float(ref.abs().max()))
    # same thing with the mask bits instead of the y tensor: identical results
    bits = (y16.flatten().float() > 0).view(-1, 8).to(torch.int32)
    bits = (bits << torch.arange(8, dtype=torch.int32)).sum(1).to(torch.uint8).cuda()
    out2 = nn.bn_bwd(grad.cuda(), raw.cuda(), mean.cuda(), invstd.cuda(), scale.cuda(), shift.cuda(), ybits=bits, relu=True,
                     side_b=(raw_b.cuda(), mean_b.cuda(), invstd_b.cuda(), scale_b.cuda()), want_dz=True)
    for t1, t2 in zip(out, out2):
        assert torch.equal(t1, t2)


@pytest.mark.parametrize("shape", [(2, 8, 6, 64), (3, 16, 8, 32), (1, 6, 10, 64)])
def test_maxpool_bn_fwd_bwd(nn, shape):
    g = torch.Generator().manual_seed(sum(shape) + 2)
    n, h, w, C = shape
    raw = torch.randn(n, h, w, C, generator=g).to(bf16)
    gamma = (torch.rand(C, generator=g) + 0.5) * torch.where(torch.rand(C, generator=g) < 0.3, -1.0, 1.0)   # negative scales too
    beta = torch.randn(C, generator=g) * 0.3
    r = raw.float().requires_grad_(True); gm = gamma.clone().requires_grad_(True); bt = beta.clone().requires_grad_(True)
    z, mean, invstd, scale, shift = _bn_ref(r, gm, bt)
    p = F.max_pool2d(nchw(z), 3, 2, 1).permute(0, 2, 3, 1)
    dp = torch.randn(p.shape, generator=g).to(bf16)
    p.backward(dp.float())
    out, arg = nn.maxpool_bn_fwd(raw.cuda(), scale.cuda(), shift.cuda())
    close_bf16(out, p.detach(), ulps=1.0, atol=1e-6)
    draw, dg, db = nn.maxpool_bn_bwd(dp.cuda(), arg, raw.cuda(), mean.cuda(), invstd.cuda(), scale.cuda())
    close_bf16(draw, r.grad, ulps=1.5, atol=2e-3 * float(r.grad.abs().max()))
    np.testing.assert_allclose(dg.cpu().numpy(), gm.grad.numpy(), rtol=2e-3, atol=2e-3 * float(gm.grad.abs().max()))
    np.testing.assert_allclose(db.cpu().numpy(), bt.grad.numpy(), rtol=2e-3, atol=2e-3 * float(bt.grad.abs().max()))


@pytest.mark.parametrize("shape", [(4, 16, 8, 2048), (3, 5, 3, 64)])
def test_head_pool_and_bn1d(nn, shape):
    g = torch.Generator().manual_seed(sum(shape) + 3)
    n, h, w, C = shape
    # distinct values per (n, c) column so the arg-max is unique (ties would route the gradient differently)
    x = (torch.randn(n, h, w, C, generator=g) + torch.arange(h * w).view(1, h, w, 1) * 1e-2).to(bf16)
    xr = x.float().requires_grad_(True)
    # Encoders.py:341-345; AdaptiveMaxPool2d routes the gradient to ONE arg-max (first in scan order), as the kernel does
    f = xr.mean((1, 2)) + F.adaptive_max_pool2d(nchw(xr), 1).flatten(1)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    gm = gamma.clone().requires_grad_(True); bt = beta.clone().requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    emb = F.batch_norm(f, rm, rv, gm, bt, True, 0.1, 1e-5)
    d_emb = torch.randn(n, C, generator=g)
    f.retain_grad()
    emb.backward(d_emb)
    fk, arg = nn.head_pool_fwd(x.cuda())
    np.testing.assert_allclose(fk.cpu().numpy(), f.detach().numpy(), rtol=1e-5, atol=1e-5)
    rm_g, rv_g = torch.zeros(C).cuda(), torch.ones(C).cuda()
    y, mean, invstd = nn.bn1d_fwd(fk, gamma.cuda(), beta.cuda(), rm_g, rv_g, training=True)
    np.testing.assert_allclose(y.cpu().numpy(), emb.detach().numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(rm_g.cpu().numpy(), rm.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(rv_g.cpu().numpy(), rv.numpy(), rtol=1e-4, atol=1e-5)
    dfk, dg, db = nn.bn1d_bwd(fk, d_emb.cuda(), gamma.cuda(), mean, invstd)
    np.testing.assert_allclose(dfk.cpu().numpy(), f.grad.numpy(), rtol=2e-3, atol=2e-4 * float(f.grad.abs().max()))
    np.testing.assert_allclose(dg.cpu().numpy(), gm.grad.numpy(), rtol=2e-3, atol=1e-3 * float(gm.grad.abs().max()))
    np.testing.assert_allclose(db.cpu().numpy(), bt.grad.numpy(), rtol=1e-4, atol=1e-4)
    dx = nn.head_pool_bwd(f.grad.cuda(), arg, (h, w))
    close_bf16(dx, xr.grad, ulps=1.0, atol=1e-6)
    # eval mode
    y_eval, _, _ = nn.bn1d_fwd(fk, gamma.cuda(), beta.cuda(), rm_g, rv_g, training=False)
    ref_eval = F.batch_norm(f.detach(), rm, rv, gamma, beta, False, 0.1, 1e-5)
    np.testing.assert_allclose(y_eval.cpu().numpy(), ref_eval.numpy(), rtol=2e-4, atol=2e-4)
