"""Python entry points of the trunk kernels (single ops; thin wrappers over the C ABI).

Tensors are torch CUDA tensors; activations NHWC bf16 ([N,H,W,C] contiguous), see include/daliid.h."""
import torch

from . import _lib

bf16 = torch.bfloat16


def _out_hw(h, w, r, s, stride, pad):
    return (h + 2 * pad - r) // stride + 1, (w + 2 * pad - s) // stride + 1


def conv2d_fwd(x, w, stride=1, pad=0, in_scale=None, in_shift=None, in_relu=False, want_stats=False):
    """x [N,H,W,Cin] bf16, w [Cout,R,S,Cin] bf16 -> y [N,Ho,Wo,Cout] bf16 (, stats [tiles,Cout,2] fp32)."""
    n, h, wd, cin = x.shape
    cout, r, s, _ = w.shape
    ho, wo = _out_hw(h, wd, r, s, stride, pad)
    y = torch.empty(n, ho, wo, cout, device=x.device, dtype=bf16)
    stats = None
    if want_stats:
        tiles = _lib.lib().dali_conv2d_stat_tiles(cout, cin, r, s, stride, pad, n, ho, wo, int(in_scale is not None))
        stats = torch.empty(tiles, cout, 2, device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().dali_conv2d_fwd(_lib.ctx(x.device), _lib.stream_ptr(), _lib.ptr(x, bf16, "x"), _lib.ptr(w, bf16, "w"),
                                           _lib.ptr(y), n, h, wd, cin, cout, r, s, stride, pad,
                                           _lib.ptr(in_scale), _lib.ptr(in_shift), int(in_relu), _lib.ptr(stats)),
               "dali_conv2d_fwd")
    return (y, stats) if want_stats else y


def conv2d_bn_act(x, w, scale, shift, stride=1, pad=0, relu=True):
    """y = relu?(conv(x) * scale[c] + shift[c]) in ONE launch: x [N,H,W,Cin] bf16, w [Cout,R,S,Cin] bf16, scale / shift fp32 [Cout] -> y bf16
    (dali_conv2d_bn_act: the inference forward of a bottleneck's conv + BatchNorm + ReLU, the affine applied to the fp32 accumulators)."""
    n, h, wd, cin = x.shape
    cout, r, s, _ = w.shape
    ho, wo = _out_hw(h, wd, r, s, stride, pad)
    y = torch.empty(n, ho, wo, cout, device=x.device, dtype=bf16)
    _lib.check(_lib.lib().dali_conv2d_bn_act(_lib.ctx(x.device), _lib.stream_ptr(), _lib.ptr(x, bf16, "x"), _lib.ptr(w, bf16, "w"), _lib.ptr(y),
                                              n, h, wd, cin, cout, r, s, stride, pad, _lib.ptr(scale, torch.float32, "scale"),
                                              _lib.ptr(shift, torch.float32, "shift"), int(relu)), "dali_conv2d_bn_act")
    return y


def stem_conv_bn_maxpool(images, weight, scale, shift):
    """maxpool3x3/2(conv7x7/2(images) * scale[c] + shift[c]) in ONE launch (dali_stem_conv_bn_maxpool: the inference stem, no ReLU before the pool):
    images fp32 [N,3,H,W], weight fp32 [64,7,7,3], scale / shift fp32 [64] -> bf16 [N,H/4,W/4,64]."""
    n, c, h, w = images.shape
    if c != 3 or tuple(weight.shape) != (64, 7, 7, 3):
        raise _lib.DaliError("stem_conv_bn_maxpool: images [N,3,H,W] and weight [64,7,7,3] expected, got %s / %s" % (tuple(images.shape), tuple(weight.shape)))
    y = torch.empty(n, h // 4, w // 4, 64, device=images.device, dtype=bf16)
    _lib.check(_lib.lib().dali_stem_conv_bn_maxpool(_lib.ctx(images.device), _lib.stream_ptr(), _lib.ptr(images, torch.float32, "images"), n, h, w,
                                                     _lib.ptr(weight, torch.float32, "weight"), _lib.ptr(scale, torch.float32, "scale"),
                                                     _lib.ptr(shift, torch.float32, "shift"), _lib.ptr(y)), "dali_stem_conv_bn_maxpool")
    return y


def conv2d_dgrad(dy, wt, x_hw, stride=1, pad=0, residual=None, inplace=False, residual_mask=None):
    """dy [N,Ho,Wo,Cout] bf16, wt [Cin,R,S,Cout] bf16 -> dx [N,H,W,Cin] bf16 (+ residual).  ``inplace``: accumulate into
    ``residual`` itself (residual == dx; what the net plan does for the downsample branch).  ``residual_mask``: uint8
    [N*H*W*Cin/8], 1 bit per residual element (``bn_act(..., want_mask=True)``): add the residual only where the bit is set."""
    n, ho, wo, cout = dy.shape
    cin, r, s, _ = wt.shape
    h, wd = x_hw
    assert _out_hw(h, wd, r, s, stride, pad) == (ho, wo)
    if inplace:
        assert residual is not None and tuple(residual.shape) == (n, h, wd, cin) and residual.is_contiguous()
    dx = residual if inplace else torch.empty(n, h, wd, cin, device=dy.device, dtype=bf16)
    _lib.check(_lib.lib().dali_conv2d_dgrad(_lib.ctx(dy.device), _lib.stream_ptr(), _lib.ptr(dy, bf16, "dy"), _lib.ptr(wt, bf16, "wt"),
                                             _lib.ptr(dx), _lib.ptr(residual), _lib.ptr(residual_mask, torch.uint8, "residual_mask"),
                                             n, h, wd, cin, cout, r, s, stride, pad),
               "dali_conv2d_dgrad")
    return dx


def conv2d_wgrad(x, dy, rs, stride=1, pad=0, in_scale=None, in_shift=None, in_relu=False, out=None, accumulate=False):
    """x [N,H,W,Cin] bf16, dy [N,Ho,Wo,Cout] bf16 -> dw [Cout,R,S,Cin] fp32."""
    n, h, wd, cin = x.shape
    cout = dy.shape[3]
    r, s = rs
    if out is None:
        out = torch.empty(cout, r, s, cin, device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().dali_conv2d_wgrad(_lib.ctx(x.device), _lib.stream_ptr(), _lib.ptr(x, bf16, "x"), _lib.ptr(dy, bf16, "dy"),
                                             _lib.ptr(out, torch.float32, "dw"), n, h, wd, cin, cout, r, s, stride, pad,
                                             _lib.ptr(in_scale), _lib.ptr(in_shift), int(in_relu), int(accumulate)),
               "dali_conv2d_wgrad")
    return out


def conv1x1_fused(x, w, out_scale=None, out_shift=None, bias=None, residual=None, relu=False, want_bits=False, out_mask=None, inplace=False,
                  res_scale=None):
    """x [P,cin] bf16, w [cout,cin] bf16 -> y [P,cout] = gate(relu?(acc*scale + shift + bias + residual)) (, bits uint8 [P*cout/8])."""
    P, cin = x.shape
    cout = w.shape[0]
    y = residual if inplace else torch.empty(P, cout, device=x.device, dtype=bf16)
    bits = torch.empty(P * cout // 8, device=x.device, dtype=torch.uint8) if want_bits else None
    _lib.check(_lib.lib().dali_conv1x1_fused(_lib.ctx(x.device), _lib.stream_ptr(), _lib.ptr(x, bf16, "x"), _lib.ptr(w, bf16, "w"), _lib.ptr(y),
                                              P, cin, cout, _lib.ptr(out_scale), _lib.ptr(out_shift), _lib.ptr(bias), _lib.ptr(residual), int(relu),
                                              _lib.ptr(bits), _lib.ptr(out_mask, torch.uint8, "out_mask"), _lib.ptr(res_scale)), "dali_conv1x1_fused")
    return (y, bits) if want_bits else y


def conv1x1_cat(x1, x2, w, bias=None):
    """x1 [P,c1], x2 [P,c2] bf16, w [cout, c1 + c2] bf16 -> y [P,cout] = [x1 | x2] @ w.T (+ bias): one GEMM over two operand tensors."""
    P, c1 = x1.shape
    c2 = x2.shape[1]
    cout = w.shape[0]
    assert x2.shape[0] == P and w.shape[1] == c1 + c2
    y = torch.empty(P, cout, device=x1.device, dtype=bf16)
    _lib.check(_lib.lib().dali_conv1x1_cat(_lib.ctx(x1.device), _lib.stream_ptr(), _lib.ptr(x1, bf16, "x1"), c1, _lib.ptr(x2, bf16, "x2"), c2,
                                            _lib.ptr(w, bf16, "w"), _lib.ptr(bias), _lib.ptr(y), P, cout), "dali_conv1x1_cat")
    return y


def conv1x1_cat_act(x1, x2, w, out_shift, out_scale=None, relu=True, parts=1):
    """y [P,cout] = relu?(([x1 | x2] @ w.T) * out_scale + out_shift) in one launch (dali_conv1x1_cat_act); parts = 2: w = [W1 hi | W1 lo | W2 hi | W2 lo]."""
    P, c1 = x1.shape
    c2 = x2.shape[1]
    cout = w.shape[0]
    assert x2.shape[0] == P and w.shape[1] == parts * (c1 + c2)
    y = torch.empty(P, cout, device=x1.device, dtype=bf16)
    _lib.check(_lib.lib().dali_conv1x1_cat_act(_lib.ctx(x1.device), _lib.stream_ptr(), _lib.ptr(x1, bf16, "x1"), c1, _lib.ptr(x2, bf16, "x2"), c2,
                                                _lib.ptr(w, bf16, "w"), int(parts), _lib.ptr(out_scale), _lib.ptr(out_shift, torch.float32, "out_shift"), int(relu),
                                                _lib.ptr(y), P, cout), "dali_conv1x1_cat_act")
    return y


def bnlin_fwd(a, w, gamma, beta, running_mean=None, running_var=None, momentum=0.1, eps=1e-5):
    """a [P,w] bf16, w [C,w] bf16 -> dict(gram, m2, scale, shift, mean, invstd): training-mode BatchNorm coefficients of a @ w.T"""
    P, wd = a.shape
    C = w.shape[0]
    dev = a.device
    o = dict(gram=torch.empty(wd, wd, device=dev), m2=_f32(wd, dev), ut=torch.empty(wd, C, device=dev), scale=_f32(C, dev), shift=_f32(C, dev),
             mean=_f32(C, dev), invstd=_f32(C, dev))
    _lib.check(_lib.lib().dali_bnlin_fwd(_lib.ctx(dev), _lib.stream_ptr(), _lib.ptr(a, bf16, "a"), _lib.ptr(w, bf16, "w"), P, C, wd, _lib.ptr(gamma),
                                          _lib.ptr(beta), _lib.ptr(running_mean), _lib.ptr(running_var), momentum, eps, _lib.ptr(o["gram"]),
                                          _lib.ptr(o["m2"]), _lib.ptr(o["ut"]), _lib.ptr(o["scale"]), _lib.ptr(o["shift"]), _lib.ptr(o["mean"]), _lib.ptr(o["invstd"])),
               "dali_bnlin_fwd")
    return o


def bnlin_bwd(dz, a, w, fwd):
    """-> dict(dW [C,w], dgamma, dbeta, wd1 [w,C] bf16, wd2 [w,w] bf16, bvec [w]); fwd = the dict bnlin_fwd returned"""
    P, wd = a.shape
    C = w.shape[0]
    dev = a.device
    o = dict(dW=torch.empty(C, wd, device=dev), dgamma=_f32(C, dev), dbeta=_f32(C, dev), wd1=torch.empty(wd, C, device=dev, dtype=bf16),
             wd2=torch.empty(wd, wd, device=dev, dtype=bf16), bvec=_f32(wd, dev))
    _lib.check(_lib.lib().dali_bnlin_bwd(_lib.ctx(dev), _lib.stream_ptr(), _lib.ptr(dz, bf16, "dz"), _lib.ptr(a, bf16, "a"), _lib.ptr(w, bf16, "w"), P, C, wd,
                                          _lib.ptr(fwd["ut"]), _lib.ptr(fwd["m2"]), _lib.ptr(fwd["scale"]), _lib.ptr(fwd["mean"]), _lib.ptr(fwd["invstd"]),
                                          _lib.ptr(o["dW"]), _lib.ptr(o["dgamma"]), _lib.ptr(o["dbeta"]), _lib.ptr(o["wd1"]), _lib.ptr(o["wd2"]),
                                          _lib.ptr(o["bvec"])), "dali_bnlin_bwd")
    return o


# ---- BatchNorm / pooling / head single ops -------------------------------------------------------------------
def _f32(n, dev):
    return torch.empty(n, device=dev, dtype=torch.float32)


def bn_finalize(partial, count, gamma, beta, running_mean=None, running_var=None, momentum=0.1, eps=1e-5):
    tiles, C, _ = partial.shape
    dev = partial.device
    scale, shift, mean, invstd = _f32(C, dev), _f32(C, dev), _f32(C, dev), _f32(C, dev)
    _lib.check(_lib.lib().dali_bn_finalize(_lib.ctx(dev), _lib.stream_ptr(), _lib.ptr(partial, torch.float32), tiles, C, float(count),
                                            _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(running_mean), _lib.ptr(running_var),
                                            momentum, eps, _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(mean), _lib.ptr(invstd)),
               "dali_bn_finalize")
    return scale, shift, mean, invstd


def bn_act(raw, scale, shift, identity=None, raw2=None, scale2=None, shift2=None, relu=True, want_mask=False):
    """-> y, or (y, mask bytes [numel/8]: bit t of byte i = y.flatten()[8i+t] > 0) with ``want_mask``."""
    C = raw.shape[-1]
    y = torch.empty_like(raw)
    mask = torch.empty(raw.numel() // 8, device=raw.device, dtype=torch.uint8) if want_mask else None
    _lib.check(_lib.lib().dali_bn_act(_lib.ctx(raw.device), _lib.stream_ptr(), _lib.ptr(raw, bf16), _lib.ptr(scale), _lib.ptr(shift),
                                       _lib.ptr(identity), _lib.ptr(raw2), _lib.ptr(scale2), _lib.ptr(shift2), int(relu),
                                       raw.numel() // C, C, _lib.ptr(y), _lib.ptr(mask)), "dali_bn_act")
    return (y, mask) if want_mask else y


def bn_bwd(g, raw_a, mean_a, invstd_a, scale_a, shift_a=None, ymask=None, relu=True, side_b=None, want_dz=False, ybits=None):
    """-> (draw_a, dgamma_a, dbeta_a[, draw_b, dgamma_b, dbeta_b][, dz])"""
    C = g.shape[-1]
    dev = g.device
    pixels = g.numel() // C
    draw_a, dga, dba = torch.empty_like(g), _f32(C, dev), _f32(C, dev)
    draw_b = dgb = dbb = None
    rb = mb = ib = sb = None
    if side_b is not None:
        rb, mb, ib, sb = side_b
        draw_b, dgb, dbb = torch.empty_like(g), _f32(C, dev), _f32(C, dev)
    dz = torch.empty_like(g) if want_dz else None
    _lib.check(_lib.lib().dali_bn_bwd(_lib.ctx(dev), _lib.stream_ptr(), _lib.ptr(g, bf16), _lib.ptr(ymask), _lib.ptr(ybits), int(relu), pixels, C,
                                       _lib.ptr(raw_a, bf16), _lib.ptr(mean_a), _lib.ptr(invstd_a), _lib.ptr(scale_a), _lib.ptr(shift_a),
                                       _lib.ptr(rb), _lib.ptr(mb), _lib.ptr(ib), _lib.ptr(sb), _lib.ptr(dga), _lib.ptr(dba),
                                       _lib.ptr(dgb), _lib.ptr(dbb), _lib.ptr(draw_a), _lib.ptr(draw_b), _lib.ptr(dz)), "dali_bn_bwd")
    out = (draw_a, dga, dba)
    if side_b is not None:
        out += (draw_b, dgb, dbb)
    if want_dz:
        out += (dz,)
    return out


def maxpool_bn_fwd(raw, scale, shift):
    n, h, w, C = raw.shape
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    out = torch.empty(n, ho, wo, C, device=raw.device, dtype=bf16)
    arg = torch.empty(n, ho, wo, C, device=raw.device, dtype=torch.uint8)
    _lib.check(_lib.lib().dali_maxpool_bn_fwd(_lib.ctx(raw.device), _lib.stream_ptr(), _lib.ptr(raw, bf16), _lib.ptr(scale), _lib.ptr(shift),
                                               n, h, w, C, _lib.ptr(out), _lib.ptr(arg)), "dali_maxpool_bn_fwd")
    return out, arg


def maxpool_bn_bwd(dpool, arg, raw, mean, invstd, scale):
    n, h, w, C = raw.shape
    dev = raw.device
    draw, dg, db = torch.empty_like(raw), _f32(C, dev), _f32(C, dev)
    _lib.check(_lib.lib().dali_maxpool_bn_bwd(_lib.ctx(dev), _lib.stream_ptr(), _lib.ptr(dpool, bf16), _lib.ptr(arg), _lib.ptr(raw, bf16),
                                               _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(scale), n, h, w, C, _lib.ptr(dg), _lib.ptr(db),
                                               _lib.ptr(draw)), "dali_maxpool_bn_bwd")
    return draw, dg, db


FEATURE_MODES = {"both": 0, "gap": 1, "gmp": 2}        # dali_feature (evaluateCleanATModels.py:335-340)


def head_pool_fwd(x, feature="both"):
    n, h, w, C = x.shape
    f = torch.empty(n, C, device=x.device, dtype=torch.float32)
    arg = torch.empty(n, C, device=x.device, dtype=torch.int16)
    _lib.check(_lib.lib().dali_head_pool_fwd(_lib.ctx(x.device), _lib.stream_ptr(), _lib.ptr(x, bf16), n, h * w, C, FEATURE_MODES[feature],
                                              _lib.ptr(f), _lib.ptr(arg)), "dali_head_pool_fwd")
    return f, arg


def head_pool_bwd(df, arg, hw_shape, feature="both"):
    n, C = df.shape
    h, w = hw_shape
    dx = torch.empty(n, h, w, C, device=df.device, dtype=bf16)
    _lib.check(_lib.lib().dali_head_pool_bwd(_lib.ctx(df.device), _lib.stream_ptr(), _lib.ptr(df, torch.float32), _lib.ptr(arg), n, h * w, C,
                                              FEATURE_MODES[feature], _lib.ptr(dx)), "dali_head_pool_bwd")
    return dx


def bn1d_fwd(x, gamma, beta, running_mean=None, running_var=None, training=True, momentum=0.1, eps=1e-5):
    n, C = x.shape
    y, mean, invstd = torch.empty_like(x), _f32(C, x.device), _f32(C, x.device)
    _lib.check(_lib.lib().dali_bn1d_fwd(_lib.ctx(x.device), _lib.stream_ptr(), _lib.ptr(x, torch.float32), n, C, _lib.ptr(gamma), _lib.ptr(beta),
                                         _lib.ptr(running_mean), _lib.ptr(running_var), int(training), momentum, eps, _lib.ptr(y),
                                         _lib.ptr(mean), _lib.ptr(invstd)), "dali_bn1d_fwd")
    return y, mean, invstd


def bn1d_bwd(x, dy, gamma, mean, invstd):
    n, C = x.shape
    dx, dg, db = torch.empty_like(x), _f32(C, x.device), _f32(C, x.device)
    _lib.check(_lib.lib().dali_bn1d_bwd(_lib.ctx(x.device), _lib.stream_ptr(), _lib.ptr(x, torch.float32), _lib.ptr(dy, torch.float32), n, C,
                                         _lib.ptr(gamma), _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(dx), _lib.ptr(dg), _lib.ptr(db)),
               "dali_bn1d_bwd")
    return dx, dg, db
