"""Mirror of the reference's ``vit_pytorch.py`` for the path in scope: the ``TransReID`` encoder
(vit_pytorch.py:291-408) and its factory ``vit_base_patch16_224_TransReID`` (:453-459), on the native HIP ViT plan
(``dali_vit_*``).  ``make_models.build_transformer`` (the BN-neck wrapper the reference actually instantiates,
make_models.py:121-205) is the module that owns the plan; ``TransReID`` here is the same plan exposed at the pre-neck
feature so the reference's key names (``cls_token``, ``pos_embed``, ``patch_embed.proj.*``, ``blocks.N.*``, ``norm.*``,
``fc.*``) and call signature ``forward(x, cam_label=None, view_label=None)`` are kept.

Supported (everything the reference's callers use, evaluate.py:179-183): camera = view = 0 (no SIE embedding),
local_feature = False, drop / attn_drop = 0.  DropPath (vit_pytorch.py:45-62; per-block rate
``linspace(0, drop_path_rate, depth)``, :338) is applied in training mode: one uniform draw per (branch, sample) on the
device per step, scale = floor(keep + u) / keep; ``drop_path_uniform`` can be set to inject the draws (parity tests hand
the same ones to the oracle).  In eval mode it is the identity, as in the reference.
"""
import ctypes
import math

import torch
from torch import nn

from . import _lib


class _VitCfg(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("batch", "height", "width", "patch", "stride", "dim", "depth", "heads", "mlp_hidden", "num_classes")]


class _VitPlan:
    def __init__(self, device, cfg_tuple):
        cfg = _VitCfg(*cfg_tuple)
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().dali_vit_create(_lib.ctx(device), ctypes.byref(cfg), ctypes.byref(h)), "dali_vit_create")
        self.h = h
        pe, be, ab = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        fd, np_, nb = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _lib.check(_lib.lib().dali_vit_sizes(h, ctypes.byref(pe), ctypes.byref(be), ctypes.byref(ab), ctypes.byref(fd), ctypes.byref(np_),
                                              ctypes.byref(nb)), "dali_vit_sizes")
        self.param_elems, self.buffer_elems, self.arena_bytes = pe.value, be.value, ab.value
        self.feat_dim, self.n_params, self.n_buffers = fd.value, np_.value, nb.value

        self.n_stages = int(_lib.lib().dali_vit_num_stages(h))

    def stage_range(self, stage):
        b, e = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(_lib.lib().dali_vit_stage_param_range(self.h, stage, ctypes.byref(b), ctypes.byref(e)), "dali_vit_stage_param_range")
        return b.value, e.value

    def tensor_table(self, kind):
        out, name = [], ctypes.create_string_buffer(128)
        off, numel, ndim, shape = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int(), (ctypes.c_int * 4)()
        for i in range(self.n_params if kind == 0 else self.n_buffers):
            _lib.check(_lib.lib().dali_vit_tensor_info(self.h, kind, i, name, 128, ctypes.byref(off), ctypes.byref(numel), shape,
                                                        ctypes.byref(ndim)), "dali_vit_tensor_info")
            out.append((name.value.decode(), off.value, numel.value, tuple(shape[:ndim.value])))
        return out

    def __del__(self):
        try:
            _lib.lib().dali_vit_destroy(self.h)
        except Exception:
            pass


class _VitFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, net):
        ctx.net = net
        return net._run_forward(x, True)

    @staticmethod
    def backward(ctx, d_feat):
        ctx.net._run_backward(d_feat.contiguous())
        return None, torch.zeros_like(ctx.net._anchor), None


class ViTNeckNet(nn.Module):
    """TransReID encoder + BatchNorm1d neck on the HIP plan; state_dict keys as make_models.build_transformer
    (``base.*``, ``bottleneck.*``)."""

    def __init__(self, img_size=(224, 224), patch_size=16, stride_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0,
                 num_classes=1000, drop_path_rate=0.0, device=None, seed=None):
        super().__init__()
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
        if device is None:
            raise _lib.DaliError("the ViT path needs a gfx950 GPU; there is no CPU path")
        self._device = torch.device(device)
        self.img_size = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        self._geom = (patch_size, stride_size, embed_dim, depth, num_heads, int(embed_dim * mlp_ratio), num_classes)
        self.drop_path_rate = float(drop_path_rate)
        self.drop_path_uniform = None            # optional [2*depth, B] uniform draws in [0,1) used instead of torch.rand (tests)
        self._dp_scales = None
        self.in_planes = embed_dim
        self._plans, self._arena, self._last_plan, self._refreshed = {}, None, None, (None, -1)
        probe = self._plan(1)
        dev = self._device
        self.flat_params = torch.zeros(probe.param_elems, device=dev)
        self.flat_grads = torch.zeros(probe.param_elems, device=dev)
        self.flat_buffers = torch.zeros(probe.buffer_elems, device=dev)
        self.flat_nbt = torch.zeros(1, dtype=torch.long, device=dev)
        self._anchor = torch.zeros(1, device=dev, requires_grad=True)
        self._grad_views, self._param_names = {}, []
        for name, off, numel, shape in probe.tensor_table(0):
            leaf, attr = self._leaf(name)
            leaf.register_parameter(attr, nn.Parameter(self.flat_params[off:off + numel].view(*shape)))
            self._grad_views[name] = self.flat_grads[off:off + numel].view(*shape)
            self._param_names.append(name)
        for name, off, numel, shape in probe.tensor_table(1):
            leaf, attr = self._leaf(name)
            leaf.register_buffer(attr, self.flat_buffers[off:off + numel].view(*shape))
            if attr == "running_var":
                leaf.register_buffer("num_batches_tracked", self.flat_nbt[0])
        self.bottleneck.bias.requires_grad_(False)                                   # make_models.py:181
        # base.fc is built but never called (vit_pytorch.py:405-408 returns the cls feature): its grad stays None in the
        # reference, so torch.optim.Adam skips it (no weight decay either)
        self._no_grad_params = ("base.fc.weight", "base.fc.bias")
        self.reset_parameters(seed)
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.mark_weights_changed())

    def _leaf(self, dotted):
        parts, mod = dotted.split("."), self
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, nn.Module())
            mod = mod._modules[p]
        return mod, parts[-1]

    def reset_parameters(self, seed=None):
        """vit_pytorch.py:350-361 + PatchEmbed_overlap init (:268-271) + weights_init_kaiming on the neck (make_models.py:182)."""
        gen = torch.Generator().manual_seed(int(seed) if seed is not None else int(torch.initial_seed()) & 0x7fffffff)
        tn = lambda shape, std: torch.nn.init.trunc_normal_(torch.empty(shape), std=std, a=-2.0, b=2.0, generator=gen)
        with torch.no_grad():
            for name, p in self.named_parameters():
                if name.endswith("cls_token") or name.endswith("pos_embed"):
                    p.copy_(tn(p.shape, 0.02).to(p.device))
                elif "patch_embed.proj.weight" in name:
                    n = p.shape[2] * p.shape[3] * p.shape[0]
                    p.copy_((torch.randn(p.shape, generator=gen) * math.sqrt(2.0 / n)).to(p.device))
                elif name.endswith(".weight") and p.dim() == 2:
                    p.copy_(tn(p.shape, 0.02).to(p.device))
                elif name.endswith(".weight"):
                    p.fill_(1.0)                       # LayerNorm / BatchNorm weights
                else:
                    p.zero_()                          # every bias
            self.flat_buffers.zero_()
            for name, b in self.named_buffers():
                if name.endswith("running_var"):
                    b.fill_(1.0)
        self.mark_weights_changed()

    def mark_weights_changed(self):
        torch.autograd.graph.increment_version(self.flat_params)

    def _apply(self, fn, recurse=True):
        probe = fn(torch.zeros(1, device=self._device))
        if probe.device != self._device or probe.dtype != torch.float32:
            raise _lib.DaliError("this module lives on %s in fp32 storage; .to()/.half()/.cpu() are not supported" % self._device)
        return self

    def _plan(self, batch):
        p = self._plans.get(batch)
        if p is None:
            ps, st, dim, depth, heads, hidden, ncls = self._geom
            p = self._plans[batch] = _VitPlan(self._device, (batch, self.img_size[0], self.img_size[1], ps, st, dim, depth, heads, hidden, ncls))
        return p

    def _activate(self, plan):
        if self._arena is None or self._arena.numel() < plan.arena_bytes:
            self._arena = None
            torch.cuda.synchronize(self._device)
            self._arena = torch.empty(plan.arena_bytes + 256, device=self._device, dtype=torch.uint8)
            self._refreshed = (None, -1)
        L = _lib.lib()
        if self._last_plan is not plan:
            _lib.check(L.dali_vit_bind(plan.h, _lib.ptr(self.flat_params), _lib.ptr(self.flat_grads), _lib.ptr(self.flat_buffers),
                                       _lib.ptr(self._arena), self._arena.numel()), "dali_vit_bind")
            self._last_plan = plan
        key = (id(plan), self.flat_params._version)
        if self._refreshed != key:
            _lib.check(L.dali_vit_refresh_weights(plan.h, _lib.stream_ptr()), "dali_vit_refresh_weights")
            self._refreshed = key

    def _run_forward(self, x, training, want_global=False):
        if x.dim() != 4 or tuple(x.shape[1:]) != (3,) + self.img_size:
            raise _lib.DaliError("Input image size (%s) doesn't match model (%s)" % (tuple(x.shape[2:]), self.img_size))   # vit_pytorch.py:282-284
        x = x.to(device=self._device, dtype=torch.float32).contiguous()
        plan = self._plan(x.shape[0])
        self._activate(plan)
        self._dp_scales = self._draw_drop_path(x.shape[0]) if training else None
        _lib.check(_lib.lib().dali_vit_set_drop_path(plan.h, _lib.ptr(self._dp_scales)), "dali_vit_set_drop_path")
        feat = torch.empty(x.shape[0], plan.feat_dim, device=self._device)
        gf = torch.empty_like(feat) if want_global else None
        _lib.check(_lib.lib().dali_vit_forward(plan.h, _lib.stream_ptr(), _lib.ptr(x), int(training), _lib.ptr(feat), _lib.ptr(gf)), "dali_vit_forward")
        if training:
            self.flat_nbt += 1
            self._bwd_plan = plan
        return (feat, gf) if want_global else feat

    def _draw_drop_path(self, batch):
        """vit_pytorch.py:45-62 for every residual branch of every block at once: fp32 [2*depth, batch] of floor(keep + u)/keep,
        keep = 1 - linspace(0, rate, depth)[block]; blocks whose rate is 0 are nn.Identity in the reference (:171)."""
        if self.drop_path_rate <= 0:
            return None
        depth = self._geom[3]
        dpr = torch.linspace(0, self.drop_path_rate, depth).repeat_interleave(2).to(self._device).unsqueeze(1)      # :338
        u = self.drop_path_uniform
        if u is None:
            u = torch.rand(2 * depth, batch, device=self._device)
        elif tuple(u.shape) != (2 * depth, batch):
            raise _lib.DaliError("drop_path_uniform must be [2*depth, batch] = %s" % ((2 * depth, batch),))
        keep = 1.0 - dpr
        return ((keep + u.to(self._device, torch.float32)).floor() / keep).contiguous()

    # the trainer / data-parallel reducer drive the backward per stage (= gradient bucket): groups of blocks, last group first
    @property
    def n_bwd_stages(self):
        return self._bwd_plan.n_stages

    def _backward_stage(self, d_feat, stage):
        plan = self._bwd_plan
        if plan is not self._last_plan:
            raise _lib.DaliError("backward() after another forward of a different shape is not supported")
        _lib.check(_lib.lib().dali_vit_backward_stages(plan.h, _lib.stream_ptr(), _lib.ptr(d_feat, torch.float32, "d_feat"), stage, stage),
                   "dali_vit_backward_stages")

    def _run_backward(self, d_feat, attach=True):
        for stage in range(self.n_bwd_stages):
            self._backward_stage(d_feat, stage)
        if attach:
            for name, p in zip(self._param_names, self.parameters()):
                p.grad = self._grad_views[name] if p.requires_grad else None

    def forward(self, x, label=None, cam_label=None, view_label=None):
        """make_models.py:184-205: returns the post-neck ``feat``."""
        if self.training and torch.is_grad_enabled():
            return _VitFn.apply(x, self._anchor, self)
        return self._run_forward(x, self.training)

    def global_feat(self, x):
        """The pre-neck cls feature = ``TransReID.forward`` (vit_pytorch.py:405-408); eval-mode helper."""
        return self._run_forward(x, False, want_global=True)[1]


class TransReID(nn.Module):
    """vit_pytorch.TransReID (vit_pytorch.py:291-408) at the pre-neck feature; shares the plan of a ViTNeckNet."""

    def __init__(self, img_size=224, patch_size=16, stride_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4., qkv_bias=False, qk_scale=None, drop_rate=0., attn_drop_rate=0., camera=0, view=0, drop_path_rate=0.,
                 hybrid_backbone=None, norm_layer=None, local_feature=False, sie_xishu=1.0, device=None, seed=None):
        super().__init__()
        if in_chans != 3 or not qkv_bias or qk_scale is not None or hybrid_backbone is not None or local_feature:
            raise NotImplementedError("TransReID: only in_chans=3, qkv_bias=True, default qk_scale, no hybrid backbone, local_feature=False are in scope")
        if camera > 1 or view > 1:
            raise NotImplementedError("TransReID: SIE camera/view embeddings are out of scope (camera = view = 0 in every reference caller)")
        if drop_rate != 0. or attn_drop_rate != 0.:
            raise NotImplementedError("TransReID: dropout is not supported (the reference's callers use 0)")
        self.net = ViTNeckNet(img_size, patch_size, stride_size, embed_dim, depth, num_heads, mlp_ratio, num_classes, drop_path_rate, device, seed)
        self.num_features = self.embed_dim = embed_dim
        self.num_classes = num_classes

    def state_dict(self, *a, **k):
        return {key[len("base."):]: v for key, v in self.net.state_dict().items() if key.startswith("base.")}

    def load_state_dict(self, sd, strict=True):
        full = dict(self.net.state_dict())
        full.update({"base." + k: v for k, v in sd.items()})
        return self.net.load_state_dict(full, strict=strict)

    def forward(self, x, cam_label=None, view_label=None):
        return self.net.global_feat(x)


def vit_base_patch16_224_TransReID(img_size=(256, 128), stride_size=16, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.1, camera=0, view=0,
                                   local_feature=False, sie_xishu=1.5, **kwargs):
    """vit_pytorch.py:453-459."""
    return TransReID(img_size=img_size, patch_size=16, stride_size=stride_size, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True,
                     camera=camera, view=view, drop_path_rate=drop_path_rate, drop_rate=drop_rate, attn_drop_rate=attn_drop_rate,
                     sie_xishu=sie_xishu, local_feature=local_feature, **kwargs)
