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
        tiles = _lib.lib().dali_conv2d_stat_tiles(cout, n, ho, wo)
        stats = torch.empty(tiles, cout, 2, device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().dali_conv2d_fwd(_lib.ctx(x.device), _lib.stream_ptr(), _lib.ptr(x, bf16, "x"), _lib.ptr(w, bf16, "w"),
                                           _lib.ptr(y), n, h, wd, cin, cout, r, s, stride, pad,
                                           _lib.ptr(in_scale), _lib.ptr(in_shift), int(in_relu), _lib.ptr(stats)),
               "dali_conv2d_fwd")
    return (y, stats) if want_stats else y


def conv2d_dgrad(dy, wt, x_hw, stride=1, pad=0, residual=None):
    """dy [N,Ho,Wo,Cout] bf16, wt [Cin,R,S,Cout] bf16 -> dx [N,H,W,Cin] bf16 (+ residual)."""
    n, ho, wo, cout = dy.shape
    cin, r, s, _ = wt.shape
    h, wd = x_hw
    assert _out_hw(h, wd, r, s, stride, pad) == (ho, wo)
    dx = torch.empty(n, h, wd, cin, device=dy.device, dtype=bf16)
    _lib.check(_lib.lib().dali_conv2d_dgrad(_lib.ctx(dy.device), _lib.stream_ptr(), _lib.ptr(dy, bf16, "dy"), _lib.ptr(wt, bf16, "wt"),
                                             _lib.ptr(dx), _lib.ptr(residual), n, h, wd, cin, cout, r, s, stride, pad),
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
