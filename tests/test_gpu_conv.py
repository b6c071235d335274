"""GPU parity: implicit-GEMM convolution kernels (forward, data gradient, weight gradient) vs torch CPU fp32.

Small-integer operands are exact in bf16 and their products/sums exact in fp32, so forward/dgrad must equal the
fp32 oracle rounded once to bf16 BIT-EXACTLY and wgrad (fp32 out) must equal it exactly.  Random-data cases use a
1-bf16-ulp tolerance (accumulation order differs)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
bf16 = torch.bfloat16

# (n, h, w, cin, cout, r, s, stride, pad)
CASES = [
    (2, 8, 4, 64, 256, 1, 1, 1, 0),
    (3, 8, 8, 256, 64, 1, 1, 1, 0),       # narrow-M tile (64 x 256)
    (2, 8, 8, 64, 128, 1, 1, 2, 0),       # 1x1 stride-2 downsample
    (2, 9, 5, 32, 64, 3, 3, 1, 1),        # non power-of-two pixel grid
    (1, 16, 8, 128, 128, 3, 3, 1, 1),
    (2, 8, 8, 64, 96, 3, 3, 2, 1),        # 3x3 stride 2
    (5, 7, 3, 96, 160, 3, 3, 1, 1),       # ragged everything
    (4, 16, 8, 512, 512, 3, 3, 1, 1),     # layer4 conv2 shape (small batch)
    (2, 32, 16, 128, 128, 3, 3, 2, 1),    # layer2 conv2 (stride 2): dgrad split by output parity, 4 sub-problems
    (3, 6, 10, 32, 64, 3, 3, 2, 1),       # stride 2 on a grid whose halves are not powers of two: un-split dgrad
    (2, 16, 8, 256, 512, 1, 1, 2, 0),     # layer2 downsample: in-place dgrad touches the even-even quarter only
    (2, 16, 8, 64, 64, 3, 3, 1, 1),       # layer1 conv2: halo wgrad kernel with a half-empty 128-channel co tile
    (3, 32, 32, 64, 128, 3, 3, 1, 1),     # halo wgrad, W = 32 (one image row per k-step), 3 images
    (2, 8, 16, 128, 192, 3, 3, 1, 1),     # halo wgrad, W = 16, Cout = 192 (1.5 co tiles)
    # P >= 16384: the large-tile kernels (k-tile 64 / wave-specialised / 256 x 256) and the specialised weight gradient
    (33, 32, 16, 256, 512, 3, 3, 1, 1),   # fwd 256x256 k-tile 64 (K = 2304); dgrad 128x256 specialised (K = 4608); halo wgrad
    (33, 32, 16, 1024, 256, 1, 1, 1, 0),  # fwd 128x256 specialised (K = 1024); wgrad: specialised 128x256 kernel
    (17, 31, 33, 1088, 320, 1, 1, 1, 0),  # ragged P = 17391, Cm = 320 (2.5 tiles), K = 17 x 64: specialised fwd + wgrad, zero-filled tails
    (17, 31, 33, 1088, 576, 1, 1, 1, 0),  # ragged 256x256 k-tile 64 fwd (2.25 tiles); dgrad K = 576: k-tile 32 128x256 kernel
    (128, 32, 16, 256, 256, 3, 3, 2, 1),  # layer3 conv2 (stride 2) at full pixel count: parity-split dgrad, its 4-tap class on the specialised kernel
    (2, 16, 16, 64, 64, 3, 3, 1, 1),      # 64 -> 64 3x3 on the halo-patch kernel (igemm_conv_halo64_kernel): one 256-pixel tile per image, W = 16
    (3, 64, 32, 64, 64, 3, 3, 1, 1),      # layer1 conv2's geometry (64 x 32 images, 8 tiles each): forward and data gradient on the halo-patch kernel
]


@pytest.fixture(scope="module")
def nn():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from daliid_amd import ops_nn
    return ops_nn


def _ints(shape, gen, lo=-2, hi=3, density=1.0):
    t = torch.randint(lo, hi, shape, generator=gen).float()
    if density < 1.0:
        t = t * (torch.rand(shape, generator=gen) < density).float()
    return t


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("case", CASES)
def test_conv_fwd_dgrad_wgrad_exact_integers(nn, case):
    n, h, w, cin, cout, r, s, stride, pad = case
    gen = torch.Generator().manual_seed(sum(case) * 7 + cin)
    x = _ints((n, cin, h, w), gen).requires_grad_(True)
    wt = _ints((cout, cin, r, s), gen, density=0.5).requires_grad_(True)
    y = F.conv2d(x, wt, stride=stride, padding=pad)
    dy = _ints(tuple(y.shape), gen, density=0.5)
    y.backward(dy)
    xg = nhwc(x.detach()).to(bf16).cuda()
    w_fwd = wt.detach().permute(0, 2, 3, 1).contiguous().to(bf16).cuda()           # [cout][r][s][cin]
    w_dg = wt.detach().permute(1, 2, 3, 0).contiguous().to(bf16).cuda()            # [cin][r][s][cout]
    dyg = nhwc(dy).to(bf16).cuda()
    # forward (+ batch statistics)
    yk, stats = nn.conv2d_fwd(xg, w_fwd, stride, pad, want_stats=True)
    ref_y = nhwc(y.detach())
    assert torch.equal(yk.cpu(), ref_y.to(bf16)), (yk.cpu().float() - ref_y).abs().max()
    st = stats.double().sum(0).cpu()
    np.testing.assert_allclose(st[:, 0].numpy(), ref_y.double().sum((0, 1, 2)).numpy(), rtol=1e-6, atol=1e-3)
    np.testing.assert_allclose(st[:, 1].numpy(), (ref_y.double() ** 2).sum((0, 1, 2)).numpy(), rtol=1e-5, atol=1e-3)
    # data gradient (+ residual)
    if cout % 32 == 0:
        dxk = nn.conv2d_dgrad(dyg, w_dg, (h, w), stride, pad)
        ref_dx = nhwc(x.grad)
        assert torch.equal(dxk.cpu(), ref_dx.to(bf16)), (dxk.cpu().float() - ref_dx).abs().max()
        res = _ints((n, h, w, cin), gen)
        dxr = nn.conv2d_dgrad(dyg, w_dg, (h, w), stride, pad, residual=res.to(bf16).cuda())
        assert torch.equal(dxr.cpu(), (ref_dx + res).to(bf16))
        # residual gated by a 1-bit mask (the identity path of a bottleneck: dy * (y > 0) formed in the epilogue)
        bits = torch.randint(0, 2, (n, h, w, cin), generator=gen)
        packed = (bits.reshape(-1, 8) << torch.arange(8)).sum(1).to(torch.uint8).cuda()
        dxm = nn.conv2d_dgrad(dyg, w_dg, (h, w), stride, pad, residual=res.to(bf16).cuda(), residual_mask=packed)
        assert torch.equal(dxm.cpu(), (ref_dx + res * bits).to(bf16))
        buf = res.to(bf16).cuda()                                            # accumulate in place (residual == dx)
        dxi = nn.conv2d_dgrad(dyg, w_dg, (h, w), stride, pad, residual=buf, inplace=True)
        assert dxi.data_ptr() == buf.data_ptr() and torch.equal(dxi.cpu(), (ref_dx + res).to(bf16))
    # weight gradient (fp32, exact)
    dwk = nn.conv2d_wgrad(xg, dyg, (r, s), stride, pad)
    ref_dw = wt.grad.permute(0, 2, 3, 1).contiguous()
    assert torch.equal(dwk.cpu(), ref_dw), (dwk.cpu() - ref_dw).abs().max()
    acc = nn.conv2d_wgrad(xg, dyg, (r, s), stride, pad, out=dwk.clone(), accumulate=True)
    assert torch.equal(acc.cpu(), 2 * ref_dw)


# 1x1 weight gradients on the pipelined 128 x 256 kernel (igemm_wgrad_p_kernel) whose LAST split holds 1, 2 and 3 k-steps of 32 pixels (the
# ring's prologue / odd-count tail), the others 8: P = 16416, 16448, 16480 with 65 splits of 256 pixels
@pytest.mark.parametrize("n,h,w,cin,cout", [(27, 32, 19, 256, 128), (257, 8, 8, 256, 128), (103, 16, 10, 320, 192)])
def test_wgrad_pipelined_short_last_split_exact_integers(nn, n, h, w, cin, cout):
    gen = torch.Generator().manual_seed(n * 31 + cin)
    x = _ints((n, h, w, cin), gen)
    dy = _ints((n, h, w, cout), gen, density=0.5)
    ref = torch.einsum("nhwo,nhwi->oi", dy, x).reshape(cout, 1, 1, cin)
    dw = nn.conv2d_wgrad(x.to(bf16).cuda(), dy.to(bf16).cuda(), (1, 1), 1, 0)
    assert torch.equal(dw.cpu(), ref), (dw.cpu() - ref).abs().max()
    dw2 = nn.conv2d_wgrad(x.to(bf16).cuda(), dy.to(bf16).cuda(), (1, 1), 1, 0)
    assert torch.equal(dw, dw2)                                   # bit-reproducible


@pytest.mark.parametrize("case", [CASES[0], CASES[3], CASES[5], CASES[7]])
def test_conv_fused_bn_relu_operand(nn, case):
    """Operand transform relu(x*scale+shift) fused into the load (forward and wgrad), random data."""
    n, h, w, cin, cout, r, s, stride, pad = case
    gen = torch.Generator().manual_seed(77)
    x = torch.randn(n, h, w, cin, generator=gen).to(bf16)
    scale = torch.rand(cin, generator=gen) + 0.5
    shift = torch.randn(cin, generator=gen) * 0.3
    wt = (torch.randn(cout, r, s, cin, generator=gen) / (r * s * cin) ** 0.5).to(bf16)
    a = (x.float() * scale + shift).clamp(min=0).to(bf16)                 # what the kernel materialises in LDS
    ref = F.conv2d(a.float().permute(0, 3, 1, 2), wt.float().permute(0, 3, 1, 2), stride=stride, padding=pad)
    ref = nhwc(ref)
    y = nn.conv2d_fwd(x.cuda(), wt.cuda(), stride, pad, in_scale=scale.cuda(), in_shift=shift.cuda(), in_relu=True)
    err = (y.cpu().float() - ref).abs()
    assert (err <= 2.0 ** -7 * ref.abs() + 1e-3).all(), err.max()
    dy = torch.randn(ref.shape, generator=gen).to(bf16)
    af = a.float().permute(0, 3, 1, 2)
    wref = wt.float().permute(0, 3, 1, 2).clone().requires_grad_(True)
    F.conv2d(af, wref, stride=stride, padding=pad).backward(dy.float().permute(0, 3, 1, 2))
    dw = nn.conv2d_wgrad(x.cuda(), dy.cuda(), (r, s), stride, pad, in_scale=scale.cuda(), in_shift=shift.cuda(), in_relu=True)
    ref_dw = wref.grad.permute(0, 2, 3, 1)
    np.testing.assert_allclose(dw.cpu().numpy(), ref_dw.numpy(), rtol=2e-4, atol=2e-3 * float(ref_dw.abs().max()))


@pytest.mark.parametrize("elems,splits,accumulate", [(4096, 512, 0), (262144, 32, 0), (65536, 7, 1), (1024 * 9 + 4, 33, 0), (751, 16, 0), (751, 41, 1),
                                                     (2050, 3, 0), (5, 130, 1), (1048576, 8, 0)])
def test_splitk_reduce_alone(nn, elems, splits, accumulate):
    """The split-K reduce by itself (dali_debug_splitk_reduce), incl. element counts that are not a multiple of 4 (their own kernel): integer
    slabs make every order of additions exact, so the result must EQUAL the fp64 column sum."""
    import ctypes
    from daliid_amd import _lib
    lib = _lib.lib()
    lib.dali_debug_splitk_reduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int]
    gen = torch.Generator().manual_seed(elems + splits)
    slabs = torch.randint(-8, 9, (splits, elems), generator=gen).float().cuda()
    out = torch.randint(-8, 9, (elems + 8,), generator=gen).float().cuda()                # 8 guard elements behind the result
    before = out.clone()
    rc = lib.dali_debug_splitk_reduce(torch.cuda.current_stream().cuda_stream, slabs.data_ptr(), out.data_ptr(), elems, splits, accumulate)
    assert rc == 0
    ref = slabs.double().sum(0) + (before[:elems].double() if accumulate else 0)
    assert torch.equal(out[:elems].double(), ref)
    assert torch.equal(out[elems:], before[elems:])


# (pixels, c1, c2, cout): the kernels with a second operand tensor (IGemmArgs::X2, the SRC2 instantiations) -- 64 x 256 and 128 x 128 LDS-DMA,
# wave-specialised 128 x 256 k-tile 64 (K >= 1024, cout 128..256, P >= 16384), 256 x 256 k-tile 64 (cout >= 512); ragged pixel counts included
@pytest.mark.parametrize("P,c1,c2,cout", [(777, 256, 64, 64), (1000, 512, 128, 128), (300, 64, 32, 24), (16640, 1024, 256, 256), (16500, 2048, 512, 512),
                                           (16384, 1024, 64, 128),
                                           # 512 <= K < 1024 with a wide tile picked first: the K gate clears k-tile 64, the launch must still
                                           # land on a two-tensor kernel (round-4 advisor finding: it read x1 as [P][c1 + c2])
                                           (16384, 448, 64, 256), (20000, 512, 256, 512)])
def test_conv1x1_two_operand_tensors_exact_integers(nn, P, c1, c2, cout):
    """y = [x1 | x2] @ w^T + bias in ONE launch (dali_conv1x1_cat; the merged conv3 data gradient of the Gram-scheme blocks): small integers, so the
    result must equal the fp32 oracle rounded once to bf16 bit for bit."""
    g = torch.Generator().manual_seed(P + c1 + cout)
    x1 = torch.randint(-2, 3, (P, c1), generator=g).float()
    x2 = torch.randint(-2, 3, (P, c2), generator=g).float()
    w = torch.randint(-1, 2, (cout, c1 + c2), generator=g).float()
    w[:, c1:] *= 2                                              # the second tensor's weights are distinguishable from the first's
    bias = torch.randint(-3, 4, (cout,), generator=g).float()
    ref = (torch.cat((x1, x2), 1) @ w.t() + bias).to(bf16)
    y = nn.conv1x1_cat(x1.to(bf16).cuda(), x2.to(bf16).cuda(), w.to(bf16).cuda(), bias.cuda())
    assert torch.equal(y.cpu(), ref), (y.cpu().float() - ref.float()).abs().max()
    y0 = nn.conv1x1_cat(x1.to(bf16).cuda(), x2.to(bf16).cuda(), w.to(bf16).cuda())
    assert torch.equal(y0.cpu(), (torch.cat((x1, x2), 1) @ w.t()).to(bf16))


# the persistent streaming kernel (K = c1 + c2 <= 256) and the 256 x 256 k-tile-64 kernel (K >= 1024): the plan's layer1.0 / layer4.0 shapes at batch 256
# and an inference batch (500 images: ragged pixel tiles)
@pytest.mark.parametrize("P,c1,c2,cout", [(524288, 64, 64, 256), (1024000, 64, 64, 256), (40000, 128, 64, 256), (32768, 512, 1024, 2048), (64000, 512, 1024, 2048)])
def test_conv1x1_two_operand_tensors_with_output_stage_exact_integers(nn, P, c1, c2, cout):
    """y = relu(([x1 | x2] @ w^T) * scale + shift) in ONE launch (dali_conv1x1_cat_act: conv3 + a stride-1 downsample branch of the inference forward):
    small integers, power-of-two scales and integer shifts, so the result equals the fp32 reference rounded once to bf16 bit for bit."""
    g = torch.Generator().manual_seed(P + c1 + cout + 5)
    dev = "cuda"
    x1 = torch.randint(-2, 3, (P, c1), generator=g).to(dev)
    x2 = torch.randint(-2, 3, (P, c2), generator=g).to(dev)
    w = torch.randint(-1, 2, (cout, c1 + c2), generator=g).to(dev)
    w[:, c1:] *= 2
    shift = torch.randint(-3, 4, (cout,), generator=g).float().to(dev)
    scale = torch.tensor([0.5, 1.0, 2.0])[torch.randint(0, 3, (cout,), generator=g)].to(dev)
    acc = torch.cat((x1, x2), 1).float() @ w.float().T
    y = nn.conv1x1_cat_act(x1.to(bf16), x2.to(bf16), w.to(bf16), shift, relu=True)
    assert torch.equal(y, torch.relu(acc + shift).to(bf16))
    y2 = nn.conv1x1_cat_act(x1.to(bf16), x2.to(bf16), w.to(bf16), shift, out_scale=scale, relu=False)
    assert torch.equal(y2, (acc * scale + shift).to(bf16))
    if 2 * (c1 + c2) <= 256:
        # split weight image [W1 hi | W1 lo | W2 hi | W2 lo] (what the inference forward uses: BatchNorm scales folded into fp32-grade weights): weights with
        # a fractional part that bf16 cannot hold in one piece -- 1 + 2^-9 steps -- but hi + lo can; products and sums stay exact in fp32
        wf = w.float() * (1.0 + torch.randint(0, 4, w.shape, generator=g).to(dev) * 2.0 ** -9)
        hi = wf.to(bf16)
        lo = (wf - hi.float()).to(bf16)
        assert torch.equal(hi.float() + lo.float(), wf)
        w4 = torch.cat((hi[:, :c1], lo[:, :c1], hi[:, c1:], lo[:, c1:]), 1).contiguous()
        y3 = nn.conv1x1_cat_act(x1.to(bf16), x2.to(bf16), w4, shift, relu=True, parts=2)
        ref3 = torch.relu((torch.cat((x1, x2), 1).double() @ wf.double().T).float() + shift).to(bf16)
        assert torch.equal(y3, ref3)
