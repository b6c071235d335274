"""ctypes binding of libdaliid_hip.so (the C ABI declared in include/daliid.h).

Fails loudly: a missing library raises at first use, a non-zero status raises ``DaliError`` carrying
``dali_last_error()``.  PyTorch is used here only to obtain device pointers and the current HIP stream.
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DALIID_LIB") or os.path.join(_HERE, "libdaliid_hip.so")      # DALIID_LIB: an A/B build of the same ABI (development aid)

c_void_p, c_int, c_float, c_size_t = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t


class DaliError(RuntimeError):
    pass


# name -> argtypes; every function returns int status unless listed in _RESTYPES
_SIGNATURES = {
    "dali_version": [],
    "dali_last_error": [],
    "dali_ctx_create": [c_int, ctypes.POINTER(c_void_p)],
    "dali_ctx_destroy": [c_void_p],
    "dali_ctx_reserve": [c_void_p, c_size_t],
    "dali_comm_unique_id": [c_void_p],
    "dali_ctx_comm_init": [c_void_p, c_void_p, c_int, c_int],
    "dali_ctx_comm_destroy": [c_void_p],
    "dali_allreduce_bucket": [c_void_p, c_void_p, c_void_p, ctypes.c_int64],
    "dali_l2norm_rows": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p],
    "dali_l2norm_rows_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p],
    "dali_pairdist": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "dali_pairdist_operand_bytes": [c_int, c_int, c_int],
    "dali_pairdist_prepare": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "dali_pairdist_prepared": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                               c_void_p],
    "dali_rank_eval": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "dali_rank_shard_matches": [c_void_p] * 7 + [c_int] * 4 + [c_void_p] * 3,
    "dali_rank_shard_bins": [c_void_p] * 7 + [c_int] * 3 + [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p],
    "dali_rank_shard_finish": [c_void_p] * 4 + [c_int] * 4 + [c_void_p] * 6,
    "dali_conv2d_fwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 9 + [c_void_p, c_void_p, c_int, c_void_p],
    "dali_conv2d_bn_act": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 9 + [c_void_p, c_void_p, c_int],
    "dali_stem_fused_supported": [c_int, c_int, c_int],
    "dali_stem_conv_bn_maxpool": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "dali_conv2d_stat_tiles": [c_int] * 10,
    "dali_conv2d_dgrad": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 9,
    "dali_conv2d_wgrad": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 9 + [c_void_p, c_void_p, c_int, c_int],
    "dali_conv1x1_fused": [c_void_p] * 5 + [c_int] * 3 + [c_void_p] * 4 + [c_int, c_void_p, c_void_p, c_void_p],
    "dali_conv1x1_cat": [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int],
    "dali_conv1x1_cat_act": [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int],
    "dali_bnlin_fwd": [c_void_p] * 4 + [c_int] * 3 + [c_void_p] * 4 + [c_float, c_float] + [c_void_p] * 7,
    "dali_bnlin_bwd": [c_void_p] * 5 + [c_int] * 3 + [c_void_p] * 11,
    "dali_bn_finalize": [c_void_p, c_void_p, c_void_p, c_int, c_int, ctypes.c_double, c_void_p, c_void_p, c_void_p, c_void_p,
                         c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p],
    "dali_bn_act": [c_void_p] * 9 + [c_int, ctypes.c_int64, c_int, c_void_p, c_void_p],
    "dali_bn_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, ctypes.c_int64, c_int] + [c_void_p] * 16,
    "dali_maxpool_bn_fwd": [c_void_p] * 5 + [c_int] * 4 + [c_void_p, c_void_p],
    "dali_maxpool_bn_bwd": [c_void_p] * 8 + [c_int] * 4 + [c_void_p, c_void_p, c_void_p],
    "dali_head_pool_fwd": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "dali_head_pool_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "dali_bn1d_fwd": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_float,
                      c_void_p, c_void_p, c_void_p],
    "dali_bn1d_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int] + [c_void_p] * 6,
    "dali_linear_fwd": [c_void_p] * 5 + [c_int] + [c_void_p] * 3 + [c_int] * 3,
    "dali_linear_fwd_scaled": [c_void_p] * 5 + [c_int] + [c_void_p] * 3 + [c_int] * 3,
    "dali_linear_dgrad": [c_void_p] * 7 + [c_int] * 3,
    "dali_linear_wgrad": [c_void_p] * 6 + [c_int] * 3,
    "dali_vit_patchify": [c_void_p, c_void_p, c_void_p] + [c_int] * 5 + [c_void_p],
    "dali_vit_assemble_tokens": [c_void_p] * 5 + [c_int] * 3 + [c_void_p],
    "dali_vit_assemble_tokens_bwd": [c_void_p] * 3 + [c_int] * 3 + [c_void_p] * 3,
    "dali_layernorm_fwd": [c_void_p] * 5 + [c_int, c_int, c_float] + [c_void_p] * 3,
    "dali_layernorm_bwd": [c_void_p] * 8 + [c_int, c_int] + [c_void_p] * 3,
    "dali_attention_fwd": [c_void_p] * 3 + [c_int] * 4 + [c_float, c_void_p, c_void_p],
    "dali_attention_bwd": [c_void_p] * 6 + [c_int] * 4 + [c_float, c_void_p],
    "dali_center_loss_fwd": [c_void_p] * 6 + [c_float, c_int, c_int, c_void_p, c_void_p],
    "dali_center_loss_bwd": [c_void_p] * 6 + [c_float, c_int, c_int, c_void_p, c_float, c_void_p],
    "dali_proxy_loss_fwd": [c_void_p] * 6 + [c_float, c_int, c_int] + [c_void_p] * 5,
    "dali_proxy_loss_bwd": [c_void_p] * 5 + [c_int, c_int, c_void_p, c_float, c_int, c_void_p],
    "dali_proxy_kmax": [],
    "dali_adam_step": [c_void_p] * 6 + [ctypes.c_int64, c_float, c_float, c_float, c_float, c_float, c_int, c_float, c_void_p],
    "dali_ema_update": [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_int64, c_float],
    "dali_triplet_loss_fwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_void_p],
    "dali_triplet_loss_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_float, c_void_p],
    "dali_pairdist_blend": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                            c_void_p, c_void_p],
    "dali_class_targets": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p,
                           c_void_p, c_void_p],
    "dali_resize_bicubic_u8": [c_void_p] * 7 + [c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "dali_augment_batch": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, ctypes.POINTER(c_float), ctypes.POINTER(c_float), c_void_p],
    "dali_gemm_profile_begin": [c_void_p, c_int],
    "dali_gemm_profile_end": [c_void_p, c_void_p, c_void_p, c_void_p],
    "dali_resnet_create": [c_void_p, c_void_p, ctypes.POINTER(c_void_p)],
    "dali_resnet_destroy": [c_void_p],
    "dali_resnet_set_feature": [c_void_p, c_int],
    "dali_resnet_sizes": [c_void_p] + [c_void_p] * 6,
    "dali_resnet_tensor_info": [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "dali_resnet_stage_param_range": [c_void_p, c_int, c_void_p, c_void_p],
    "dali_resnet_bind": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t],
    "dali_resnet_refresh_weights": [c_void_p, c_void_p],
    "dali_resnet_forward": [c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "dali_resnet_backward": [c_void_p, c_void_p, c_void_p, c_int, c_int],
    "dali_resnet_debug_tensor": [c_void_p, ctypes.c_char_p, c_void_p, c_void_p],
    "dali_vit_create": [c_void_p, c_void_p, ctypes.POINTER(c_void_p)],
    "dali_vit_destroy": [c_void_p],
    "dali_vit_sizes": [c_void_p] + [c_void_p] * 6,
    "dali_vit_tensor_info": [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "dali_vit_bind": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t],
    "dali_vit_refresh_weights": [c_void_p, c_void_p],
    "dali_vit_forward": [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p],
    "dali_vit_backward": [c_void_p, c_void_p, c_void_p],
    "dali_vit_num_stages": [c_void_p],
    "dali_vit_stage_param_range": [c_void_p, c_int, c_void_p, c_void_p],
    "dali_vit_backward_stages": [c_void_p, c_void_p, c_void_p, c_int, c_int],
    "dali_vit_set_drop_path": [c_void_p, c_void_p],
}
_RESTYPES = {"dali_last_error": ctypes.c_char_p, "dali_pairdist_operand_bytes": ctypes.c_size_t}

_lock = threading.Lock()
_lib = None
_ctxs = {}


def exported_symbols():
    """Names include/daliid.h declares (used by the CPU symbol-export test)."""
    return sorted(_SIGNATURES)


def lib():
    """Load the shared library (no GPU needed for loading)."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise DaliError(
                        "libdaliid_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "or `make -C daliid_amd/csrc`. There is no CPU fallback." % LIB_PATH)
                L = ctypes.CDLL(LIB_PATH)
                for name, argtypes in _SIGNATURES.items():
                    fn = getattr(L, name)          # AttributeError if the library lacks a declared symbol
                    fn.argtypes = argtypes
                    fn.restype = _RESTYPES.get(name, c_int)
                _lib = L
    return _lib


def last_error():
    return lib().dali_last_error().decode("utf-8", "replace")


def check(status, what=""):
    if status != 0:
        raise DaliError("%s failed with status %d: %s" % (what or "libdaliid_hip call", status, last_error()))


def ctx(device=None, lane=0):
    """One dali_ctx per (device ordinal, lane), created on first use.  A context owns ONE grow-only workspace block that its calls use from
    offset 0, i.e. it serves one stream at a time (include/daliid.h, dali_ctx_create): work enqueued on a second stream that may run beside
    the main stream's (transforms.finish on its side stream) goes through its own context, ``lane="side"``."""
    if not torch.cuda.is_available():
        raise DaliError("daliid_amd needs a gfx950 GPU (torch.cuda.is_available() is False); there is no CPU path")
    if device is None:
        device = torch.cuda.current_device()
    device = torch.device(device).index if not isinstance(device, int) else device
    if device is None:
        device = torch.cuda.current_device()
    if device != torch.cuda.current_device():
        # kernels are enqueued on the CURRENT device's stream and the per-device kernel attributes are set for it: a tensor that
        # lives elsewhere must be handled under `with torch.cuda.device(...)`
        raise DaliError("tensors live on cuda:%d but the current device is cuda:%d; wrap the call in torch.cuda.device(%d)"
                        % (device, torch.cuda.current_device(), device))
    key = device if lane == 0 else (device, lane)
    c = _ctxs.get(key)
    if c is None:
        with _lock:
            c = _ctxs.get(key)
            if c is None:
                h = c_void_p()
                check(lib().dali_ctx_create(device, ctypes.byref(h)), "dali_ctx_create")
                c = _ctxs[key] = h
    return c


def stream_ptr():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, dtype=None, name="tensor"):
    """Device pointer of a contiguous CUDA(HIP) tensor, validated."""
    if t is None:
        return c_void_p(0)
    if not t.is_cuda:
        raise DaliError("%s must live on the GPU" % name)
    if not t.is_contiguous():
        raise DaliError("%s must be contiguous" % name)
    if dtype is not None and t.dtype != dtype:
        raise DaliError("%s must be %s, got %s" % (name, dtype, t.dtype))
    return c_void_p(t.data_ptr())


class GemmProfile:
    """Context manager over dali_gemm_profile_begin/_end (include/daliid.h): HIP-event timing of every MFMA GEMM
    kernel launch issued inside the block.  ``.ms/.flops/.launches`` are 2-lists (0 = conv fwd+dgrad / linears,
    1 = wgrad)."""

    def __init__(self, max_launches=8192, device=None):
        self._ctx, self._n = ctx(device), int(max_launches)
        self.ms = self.flops = self.launches = None

    def __enter__(self):
        check(lib().dali_gemm_profile_begin(self._ctx, self._n), "dali_gemm_profile_begin")
        return self

    def __exit__(self, *exc):
        ms, fl, ln = (ctypes.c_double * 2)(), (ctypes.c_double * 2)(), (ctypes.c_longlong * 2)()
        check(lib().dali_gemm_profile_end(self._ctx, ms, fl, ln), "dali_gemm_profile_end")
        self.ms, self.flops, self.launches = list(ms), list(fl), list(ln)
        return False
