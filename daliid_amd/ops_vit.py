"""Python entry points of the ViT kernels (single ops; thin wrappers over the C ABI).  Tokens are [rows, C] bf16."""
import torch

from . import _lib

bf16 = torch.bfloat16


def linear_fwd(x, w, bias=None, act=0, residual=None, want_pre=False):
    rows, K = x.shape
    N = w.shape[0]
    y = torch.empty(rows, N, device=x.device, dtype=bf16)
    pre = torch.empty(rows, N, device=x.device, dtype=bf16) if want_pre else None
    _lib.check(_lib.lib().dali_linear_fwd(_lib.ctx(x.device), _lib.stream_ptr(), _lib.ptr(x, bf16, "x"), _lib.ptr(w, bf16, "w"), _lib.ptr(bias),
                                           int(act), _lib.ptr(residual), _lib.ptr(y), _lib.ptr(pre), rows, K, N), "dali_linear_fwd")
    return (y, pre) if want_pre else y


def linear_fwd_scaled(x, w, row_scale, bias=None, act=0, residual=None):
    """y = row_scale[:, None] * act(x @ w.T + bias) (+ residual): a residual branch under DropPath (vit_pytorch.py:45-62, :338)."""
    rows, K = x.shape
    N = w.shape[0]
    assert row_scale.shape == (rows,) and row_scale.dtype == torch.float32
    y = torch.empty(rows, N, device=x.device, dtype=bf16)
    _lib.check(_lib.lib().dali_linear_fwd_scaled(_lib.ctx(x.device), _lib.stream_ptr(), _lib.ptr(x, bf16, "x"), _lib.ptr(w, bf16, "w"), _lib.ptr(bias),
                                                  int(act), _lib.ptr(residual), _lib.ptr(row_scale), _lib.ptr(y), rows, K, N), "dali_linear_fwd_scaled")
    return y


def linear_dgrad(dy, wt, gelu_pre=None, residual=None):
    rows, N = dy.shape
    K = wt.shape[0]
    dx = torch.empty(rows, K, device=dy.device, dtype=bf16)
    _lib.check(_lib.lib().dali_linear_dgrad(_lib.ctx(dy.device), _lib.stream_ptr(), _lib.ptr(dy, bf16, "dy"), _lib.ptr(wt, bf16, "wt"),
                                             _lib.ptr(gelu_pre), _lib.ptr(residual), _lib.ptr(dx), rows, K, N), "dali_linear_dgrad")
    return dx


def linear_wgrad(x, dy, want_bias=True):
    rows, K = x.shape
    N = dy.shape[1]
    dw = torch.empty(N, K, device=x.device, dtype=torch.float32)
    db = torch.empty(N, device=x.device, dtype=torch.float32) if want_bias else None
    _lib.check(_lib.lib().dali_linear_wgrad(_lib.ctx(x.device), _lib.stream_ptr(), _lib.ptr(x, bf16, "x"), _lib.ptr(dy, bf16, "dy"), _lib.ptr(dw),
                                             _lib.ptr(db), rows, K, N), "dali_linear_wgrad")
    return (dw, db) if want_bias else dw


def patchify(img, patch=16, stride=16):
    B, _, H, W = img.shape
    ny, nx = (H - patch) // stride + 1, (W - patch) // stride + 1
    out = torch.empty(B * ny * nx, 3 * patch * patch, device=img.device, dtype=bf16)
    _lib.check(_lib.lib().dali_vit_patchify(_lib.ctx(img.device), _lib.stream_ptr(), _lib.ptr(img, torch.float32, "img"), B, H, W, patch, stride,
                                             _lib.ptr(out)), "dali_vit_patchify")
    return out


def assemble_tokens(pe, cls, pos, B, T):
    C = pe.shape[1]
    x = torch.empty(B * T, C, device=pe.device, dtype=bf16)
    _lib.check(_lib.lib().dali_vit_assemble_tokens(_lib.ctx(pe.device), _lib.stream_ptr(), _lib.ptr(pe, bf16), _lib.ptr(cls, torch.float32),
                                                    _lib.ptr(pos, torch.float32), B, T, C, _lib.ptr(x)), "dali_vit_assemble_tokens")
    return x


def assemble_tokens_bwd(dx, B, T):
    C = dx.shape[1]
    dpos = torch.empty(T, C, device=dx.device, dtype=torch.float32)
    dcls = torch.empty(C, device=dx.device, dtype=torch.float32)
    dpe = torch.empty(B * (T - 1), C, device=dx.device, dtype=bf16)
    _lib.check(_lib.lib().dali_vit_assemble_tokens_bwd(_lib.ctx(dx.device), _lib.stream_ptr(), _lib.ptr(dx, bf16), B, T, C, _lib.ptr(dpos),
                                                        _lib.ptr(dcls), _lib.ptr(dpe)), "dali_vit_assemble_tokens_bwd")
    return dpos, dcls, dpe


def layernorm_fwd(x, gamma, beta, eps=1e-6):
    rows, C = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(rows, device=x.device, dtype=torch.float32)
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().dali_layernorm_fwd(_lib.ctx(x.device), _lib.stream_ptr(), _lib.ptr(x, bf16, "x"), _lib.ptr(gamma, torch.float32),
                                              _lib.ptr(beta, torch.float32), rows, C, float(eps), _lib.ptr(y), _lib.ptr(mean), _lib.ptr(rstd)),
               "dali_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(g, x, gamma, mean, rstd, add=None):
    rows, C = x.shape
    dx = torch.empty_like(x)
    dg = torch.empty(C, device=x.device, dtype=torch.float32)
    db = torch.empty(C, device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().dali_layernorm_bwd(_lib.ctx(x.device), _lib.stream_ptr(), _lib.ptr(g, bf16, "g"), _lib.ptr(x, bf16, "x"),
                                              _lib.ptr(gamma, torch.float32), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(add), rows, C,
                                              _lib.ptr(dx), _lib.ptr(dg), _lib.ptr(db)), "dali_layernorm_bwd")
    return dx, dg, db


def attention_fwd(qkv, B, T, H, scale=None):
    hd = qkv.shape[1] // (3 * H)
    scale = hd ** -0.5 if scale is None else scale
    out = torch.empty(B * T, H * hd, device=qkv.device, dtype=bf16)
    lse = torch.empty(B * H, T, device=qkv.device, dtype=torch.float32)
    _lib.check(_lib.lib().dali_attention_fwd(_lib.ctx(qkv.device), _lib.stream_ptr(), _lib.ptr(qkv, bf16, "qkv"), B, T, H, hd, float(scale),
                                              _lib.ptr(out), _lib.ptr(lse)), "dali_attention_fwd")
    return out, lse


def attention_bwd(qkv, out, d_out, lse, B, T, H, scale=None):
    hd = qkv.shape[1] // (3 * H)
    scale = hd ** -0.5 if scale is None else scale
    dqkv = torch.empty_like(qkv)
    _lib.check(_lib.lib().dali_attention_bwd(_lib.ctx(qkv.device), _lib.stream_ptr(), _lib.ptr(qkv, bf16, "qkv"), _lib.ptr(out, bf16),
                                              _lib.ptr(d_out, bf16), _lib.ptr(lse, torch.float32), B, T, H, hd, float(scale), _lib.ptr(dqkv)),
               "dali_attention_bwd")
    return dqkv
