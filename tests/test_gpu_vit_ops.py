"""GPU parity of the ViT kernels (linear + GELU epilogues, LayerNorm, attention, patchify, token assembly) against
torch CPU fp32 on the same bf16 inputs.  bf16 outputs: one-to-two bf16 ulps of the output scale; fp32 outputs: 1e-3."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
bf16 = torch.bfloat16


@pytest.fixture(scope="module")
def V():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from daliid_amd import ops_vit
    return ops_vit


def close(got, ref, rel=2.0 ** -7, abs_frac=4e-3):
    got, ref = got.float().cpu(), ref.float()
    err = (got - ref).abs()
    tol = rel * ref.abs() + abs_frac * ref.abs().max()
    assert (err <= tol).all(), "max err %.4g (ref scale %.4g)" % (float(err.max()), float(ref.abs().max()))


# the last two: >= 16384 rows, i.e. the tiles and kernels the full-size model runs (256 x 256 k-tile 64 with the bias / GELU /
# residual epilogue, wave-specialised 128 x 256, specialised weight gradient), ragged row count
# (25216, 1024, 768): ViT-B's row count with 768 outputs = the 256 x 320 tile (297 tiles of 256 x 256 would be 2 rounds of 256 CUs) and
# its column-block epilogue, ragged last tile (25216 = 78 * 320 + 256) through the general path
@pytest.mark.parametrize("rows,K,N", [(197 * 2, 768, 2304), (100, 3072, 768), (333, 768, 3072), (64, 64, 36), (16500, 3072, 768), (16500, 1024, 384),
                                      (25216, 1024, 768)])
def test_linear_fwd_dgrad_wgrad(V, rows, K, N):
    g = torch.Generator().manual_seed(rows + K + N)
    x = torch.randn(rows, K, generator=g).to(bf16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(bf16)
    b = torch.randn(N, generator=g) * 0.1
    res = torch.randn(rows, N, generator=g).to(bf16)
    xr, wr, br = x.float().requires_grad_(True), w.float().requires_grad_(True), b.clone().requires_grad_(True)
    pre = F.linear(xr, wr, br)
    y = F.gelu(pre) + res.float()
    dy = torch.randn(rows, N, generator=g).to(bf16)
    y.backward(dy.float())
    yk, prek = V.linear_fwd(x.cuda(), w.cuda(), b.cuda(), act=1, residual=res.cuda(), want_pre=True)
    close(prek, pre.detach())
    close(yk, y.detach())
    close(V.linear_fwd(x.cuda(), w.cuda()), F.linear(x.float(), w.float()))
    rs = (torch.rand(rows, generator=g) < 0.8).float() / 0.8          # DropPath row factors
    close(V.linear_fwd_scaled(x.cuda(), w.cuda(), rs.cuda(), bias=b.cuda(), residual=res.cuda()), rs[:, None] * pre.detach() + res.float())
    # backward through GELU: dpre = dy * gelu'(pre); dx = dpre @ w
    dpre_ref = torch.autograd.grad(F.gelu(pre.detach().to(bf16).float().requires_grad_(True)).sum(), [])  if False else None
    if N % 32 == 0:
        wt = w.t().contiguous().cuda()
        dx = V.linear_dgrad(dy.cuda(), wt)                       # plain dgrad
        close(dx, dy.float() @ w.float())
        # dgrad with the GELU derivative fused on the OUTPUT side (as fc1's input gradient is formed: d_h = (d_o @ W2) * gelu'(pre1))
        pre_k = torch.randn(rows, K, generator=g).to(bf16)
        pk = pre_k.float().requires_grad_(True)
        (F.gelu(pk) * (dy.float() @ w.float())).sum().backward()
        close(V.linear_dgrad(dy.cuda(), wt, gelu_pre=pre_k.cuda()), pk.grad)
    if K % 8 == 0 and N % 8 == 0:
        dpre = torch.randn(rows, N, generator=g).to(bf16)
        dw, db = V.linear_wgrad(x.cuda(), dpre.cuda())
        ref_dw = dpre.float().t() @ x.float()
        np.testing.assert_allclose(dw.cpu().numpy(), ref_dw.numpy(), rtol=2e-3, atol=2e-3 * float(ref_dw.abs().max()))
        np.testing.assert_allclose(db.cpu().numpy(), dpre.float().sum(0).numpy(), rtol=1e-4, atol=1e-3)


def test_linear_256x320_tile_bit_exact_on_integers(V):
    """Small integers: every product and fp32 partial sum is exact, so the 256 x 320 kernel (k-tile 64, ragged DMA pieces, transpose-staged
    store) must equal the fp32 matmul bit for bit, bias and residual included -- interior tiles and the ragged last one."""
    rows, K, N = 25216, 1024, 768
    g = torch.Generator().manual_seed(5)
    x = torch.randint(-3, 4, (rows, K), generator=g).float()
    w = torch.randint(-2, 3, (N, K), generator=g).float()
    b = torch.randint(-8, 9, (N,), generator=g).float()
    res = torch.randint(-16, 17, (rows, N), generator=g).float()
    ref = (x.cuda() @ w.cuda().t() + b.cuda() + res.cuda()).to(bf16)          # fp32 matmul of exactly representable values, one rounding
    got = V.linear_fwd(x.to(bf16).cuda(), w.to(bf16).cuda(), b.cuda(), residual=res.to(bf16).cuda())
    assert torch.equal(got, ref)


@pytest.mark.parametrize("rows,C", [(197 * 3, 768), (50, 64), (7, 2048)])
def test_layernorm_fwd_bwd(V, rows, C):
    g = torch.Generator().manual_seed(rows + C)
    x = (torch.randn(rows, C, generator=g) * 2 + 0.5).to(bf16)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    dy = torch.randn(rows, C, generator=g).to(bf16)
    add = torch.randn(rows, C, generator=g).to(bf16)
    xr, gr, br = x.float().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y = F.layer_norm(xr, (C,), gr, br, 1e-6)
    y.backward(dy.float())
    yk, mean, rstd = V.layernorm_fwd(x.cuda(), gamma.cuda(), beta.cuda(), 1e-6)
    close(yk, y.detach(), abs_frac=1e-3)
    np.testing.assert_allclose(mean.cpu().numpy(), x.float().mean(1).numpy(), rtol=1e-4, atol=1e-5)
    dx, dg, db = V.layernorm_bwd(dy.cuda(), x.cuda(), gamma.cuda(), mean, rstd, add=add.cuda())
    close(dx, xr.grad + add.float(), abs_frac=2e-3)
    np.testing.assert_allclose(dg.cpu().numpy(), gr.grad.numpy(), rtol=2e-3, atol=2e-3 * float(gr.grad.abs().max()))
    np.testing.assert_allclose(db.cpu().numpy(), br.grad.numpy(), rtol=2e-3, atol=2e-3 * float(br.grad.abs().max()))


# 211 tokens = TransReID at 256x128 / stride 12 (vit_pytorch.py:254-267); 209..224 run the 14-tile instance, 225..256 the 16-tile one
@pytest.mark.parametrize("B,T,H", [(2, 197, 12), (3, 50, 2), (1, 208, 1), (2, 129, 4), (2, 211, 12), (1, 224, 2), (2, 240, 1), (1, 256, 3)])
def test_attention_fwd_bwd(V, B, T, H):
    g = torch.Generator().manual_seed(B * 1000 + T + H)
    C = H * 64
    qkv = torch.randn(B * T, 3 * C, generator=g).to(bf16)
    d_out = torch.randn(B * T, C, generator=g).to(bf16)
    qr = qkv.float().requires_grad_(True)
    t = qr.reshape(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)                       # vit_pytorch.py:155
    q, k, v = t[0], t[1], t[2]
    attn = ((q @ k.transpose(-2, -1)) * 0.125).softmax(dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(B * T, C)
    out.backward(d_out.float())
    ok, lse = V.attention_fwd(qkv.cuda(), B, T, H)
    close(ok, out.detach(), abs_frac=4e-3)
    ref_lse = torch.logsumexp((q @ k.transpose(-2, -1)).detach() * 0.125, dim=-1).reshape(B * H, T)
    np.testing.assert_allclose(lse.cpu().numpy(), ref_lse.numpy(), rtol=1e-4, atol=1e-4)
    dqkv = V.attention_bwd(qkv.cuda(), ok, d_out.cuda(), lse, B, T, H)
    close(dqkv, qr.grad, rel=2.0 ** -6, abs_frac=8e-3)


def test_patchify_and_tokens(V):
    g = torch.Generator().manual_seed(3)
    B, Hh, Ww, C = 2, 64, 48, 64
    img = torch.randn(B, 3, Hh, Ww, generator=g)
    for patch, stride in ((16, 16), (16, 12)):
        p = V.patchify(img.cuda(), patch, stride).float().cpu()
        ref = F.unfold(img.to(bf16).float(), patch, stride=stride).transpose(1, 2).reshape(-1, 3 * patch * patch)
        assert torch.equal(p, ref)
    T = 1 + 12
    pe = torch.randn(B * 12, C, generator=g).to(bf16)
    cls, pos = torch.randn(C, generator=g), torch.randn(T, C, generator=g)
    x = V.assemble_tokens(pe.cuda(), cls.cuda(), pos.cuda(), B, T)
    ref = torch.cat((cls.expand(B, 1, C), pe.float().reshape(B, 12, C)), 1) + pos
    close(x, ref.reshape(B * T, C), abs_frac=1e-6)
    dx = torch.randn(B * T, C, generator=g).to(bf16)
    dpos, dcls, dpe = V.assemble_tokens_bwd(dx.cuda(), B, T)
    np.testing.assert_allclose(dpos.cpu().numpy(), dx.float().reshape(B, T, C).sum(0).numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(dcls.cpu().numpy(), dx.float().reshape(B, T, C)[:, 0].sum(0).numpy(), rtol=1e-5, atol=1e-5)
    assert torch.equal(dpe.cpu(), dx.reshape(B, T, C)[:, 1:].reshape(B * 12, C))
