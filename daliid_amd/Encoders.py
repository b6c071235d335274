"""Mirror of the reference's ``Encoders.py`` for the path in scope: ``getDCNN("resnet50")`` and
``ResNet50ReID`` (Encoders.py:25-48, 306-351), backed by the native HIP net plan (``dali_resnet_*`` in
include/daliid.h) instead of torchvision + cuDNN.

What is kept from the reference surface
  * ``getDCNN(gpu_indexes, model_name, embedding_size=None) -> (model_online, model_momentum)``; both come back
    wrapped (``.module`` attribute, ``module.``-prefixed state_dict keys like ``nn.DataParallel``), momentum
    initialised from online, both in eval mode (Encoders.py:39-48).
  * ``ResNet50ReID``: callable ``x[B,3,H,W] fp32 -> [B,2048] fp32``; ``.train()/.eval()``; ``.parameters()``;
    ``state_dict()/load_state_dict()`` with torchvision key names (``conv1.weight`` ... ``last_bn.bias``).
What differs by design
  * storage: all parameters are views into ONE flat fp32 buffer (conv weights in channels_last / OHWI storage),
    gradients into one flat fp32 buffer, BN running stats into a third: fused Adam/EMA and bucketed RCCL
    all-reduce run on the flat buffers.
  * ``torchvision.models.resnet50(pretrained=True)`` (Encoders.py:33,36) needs a network fetch; here the weights are
    torchvision's random init (kaiming fan_out) unless a state_dict is loaded.
  * one process drives ONE GPU (data parallelism = one process per GPU + RCCL), so ``gpu_indexes`` only selects
    the device of this process.
"""
import ctypes

import torch
from torch import nn

from . import _lib


class _Cfg(ctypes.Structure):
    _fields_ = [("batch", ctypes.c_int), ("height", ctypes.c_int), ("width", ctypes.c_int),
                ("layers", ctypes.c_int * 4), ("width_base", ctypes.c_int)]


class _Plan:
    """One dali_resnet plan for a fixed (batch, H, W)."""

    def __init__(self, device, batch, height, width, layers, width_base):
        self.key = (batch, height, width)
        cfg = _Cfg(batch, height, width, (ctypes.c_int * 4)(*layers), width_base)
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().dali_resnet_create(_lib.ctx(device), ctypes.byref(cfg), ctypes.byref(h)), "dali_resnet_create")
        self.h = h
        pe, be, ab = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        fd, np_, nb = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _lib.check(_lib.lib().dali_resnet_sizes(h, ctypes.byref(pe), ctypes.byref(be), ctypes.byref(ab), ctypes.byref(fd),
                                                 ctypes.byref(np_), ctypes.byref(nb)), "dali_resnet_sizes")
        self.param_elems, self.buffer_elems, self.arena_bytes = pe.value, be.value, ab.value
        self.feat_dim, self.n_params, self.n_buffers = fd.value, np_.value, nb.value

    def tensor_table(self, kind):
        out = []
        name = ctypes.create_string_buffer(128)
        off, numel, ndim = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
        shape = (ctypes.c_int * 4)()
        for i in range(self.n_params if kind == 0 else self.n_buffers):
            _lib.check(_lib.lib().dali_resnet_tensor_info(self.h, kind, i, name, 128, ctypes.byref(off), ctypes.byref(numel),
                                                           shape, ctypes.byref(ndim)), "dali_resnet_tensor_info")
            out.append((name.value.decode(), off.value, numel.value, tuple(shape[:ndim.value])))
        return out

    def stage_range(self, stage):
        b, e = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(_lib.lib().dali_resnet_stage_param_range(self.h, stage, ctypes.byref(b), ctypes.byref(e)), "stage_param_range")
        return b.value, e.value

    def __del__(self):
        try:
            _lib.lib().dali_resnet_destroy(self.h)
        except Exception:
            pass


class _NetFn(torch.autograd.Function):
    """autograd glue: forward = one C-ABI call, backward = one call per stage (+ optional gradient hook per stage)."""

    @staticmethod
    def forward(ctx, x, anchor, net):
        ctx.net = net
        return net._run_forward(x, training=True)

    @staticmethod
    def backward(ctx, d_emb):
        net = ctx.net
        net._run_backward(d_emb.contiguous())
        return None, torch.zeros_like(net._anchor), None


_FEATURE_MODES = {"both": 0, "gap": 1, "gmp": 2}        # dali_feature in include/daliid.h


class ResNet50ReID(nn.Module):
    """Encoders.ResNet50ReID (Encoders.py:306-351) on the HIP net plan."""

    def __init__(self, model_base=None, layers=(3, 4, 6, 3), width=64, device=None, seed=None):
        super().__init__()
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
        if device is None:
            raise _lib.DaliError("ResNet50ReID needs a gfx950 GPU; there is no CPU path")
        self._device = torch.device(device)
        self._layers, self._width = tuple(layers), int(width)
        self._plans = {}
        self._arena = None
        self._last_plan = None
        self._refreshed = (None, -1)
        self.grad_stage_hook = None          # callable(stage, begin, end) after each backward stage (DP all-reduce)
        self.feature = "both"               # head pooling: "both" | "gap" | "gmp" (evaluateCleanATModels.py:249-256, :335-340)
        probe = self._plan(1, 32, 32)
        self.feat_dim = probe.feat_dim
        dev = self._device
        self.flat_params = torch.zeros(probe.param_elems, device=dev, dtype=torch.float32)
        self.flat_grads = torch.zeros(probe.param_elems, device=dev, dtype=torch.float32)
        self.flat_buffers = torch.zeros(probe.buffer_elems, device=dev, dtype=torch.float32)
        self._anchor = torch.zeros(1, device=dev, requires_grad=True)
        self._grad_views = {}
        self._param_names = []
        self._bn_modules = []
        for name, off, numel, shape in probe.tensor_table(0):
            leaf, attr = self._leaf(name)
            view = self._view(self.flat_params, off, numel, shape)
            leaf.register_parameter(attr, nn.Parameter(view))
            self._grad_views[name] = self._view(self.flat_grads, off, numel, shape)
            self._param_names.append(name)
        table1 = probe.tensor_table(1)
        self.flat_nbt = torch.zeros(sum(1 for t in table1 if t[0].endswith("running_var")), dtype=torch.long, device=dev)
        for name, off, numel, shape in table1:
            leaf, attr = self._leaf(name)
            leaf.register_buffer(attr, self._view(self.flat_buffers, off, numel, shape))
            if attr == "running_var":                    # all num_batches_tracked counters are views of ONE int64 vector
                leaf.register_buffer("num_batches_tracked", self.flat_nbt[len(self._bn_modules)])
                self._bn_modules.append(leaf)
        self.reset_parameters(seed)
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.mark_weights_changed())
        if model_base is not None:
            self._load_base(model_base.state_dict() if isinstance(model_base, nn.Module) else model_base)

    def _load_base(self, sd):
        """``ResNet50ReID(model_base)`` as the reference calls it (Encoders.py:33-37, 306-328): model_base is a torchvision ``resnet50``,
        whose trunk (conv1 ... layer4) is taken over; its classifier ``fc.*`` is dropped (Encoders.py:312-328 never copies it) and
        ``last_bn`` is this module's own fresh BatchNorm1d (Encoders.py:350).  A state_dict of this class (with ``last_bn.*``) loads whole.
        Anything else missing or unexpected is an error, as in a strict load."""
        sd = {k: v for k, v in sd.items() if not k.startswith("fc.")}
        res = self.load_state_dict(sd, strict=False)
        missing = [k for k in res.missing_keys if not k.startswith("last_bn.")]
        if missing or res.unexpected_keys:
            raise _lib.DaliError("ResNet50ReID(model_base): model_base is not a ResNet-50 trunk (missing %s, unexpected %s)"
                                 % (missing[:4], list(res.unexpected_keys)[:4]))

    # ---- construction helpers -------------------------------------------------------------------
    @staticmethod
    def _view(flat, off, numel, shape):
        v = flat[off:off + numel]
        if len(shape) == 4:                       # logical OIHW over OHWI storage (channels_last)
            o, i, r, s = shape
            return v.view(o, r, s, i).permute(0, 3, 1, 2)
        return v.view(*shape)

    def _leaf(self, dotted):
        parts = dotted.split(".")
        mod = self
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, nn.Module())
            mod = mod._modules[p]
        return mod, parts[-1]

    def reset_parameters(self, seed=None):
        """torchvision's ResNet init: conv kaiming_normal_(fan_out, relu); BN weight 1 / bias 0; stats 0 / 1."""
        gen = torch.Generator(device="cpu")
        gen.manual_seed(int(seed) if seed is not None else int(torch.initial_seed()) & 0x7fffffff)
        with torch.no_grad():
            for name, p in self.named_parameters():
                if p.dim() == 4:
                    o, i, r, s = p.shape
                    std = (2.0 / (o * r * s)) ** 0.5
                    p.copy_((torch.randn(o, i, r, s, generator=gen) * std).to(p.device))
                elif name.endswith(".weight"):
                    p.fill_(1.0)
                else:
                    p.zero_()
            for name, b in self.named_buffers():
                if name.endswith("running_var"):
                    b.fill_(1.0)
                else:
                    b.zero_()
        self.mark_weights_changed()

    @property
    def _weights_version(self):
        # every in-place write through any parameter view (optimizer step, copy_, load_state_dict) bumps the version
        # counter the views share with the flat buffer; kernels that write it behind torch's back call
        # mark_weights_changed()
        return self.flat_params._version

    def mark_weights_changed(self):
        torch.autograd.graph.increment_version(self.flat_params)

    def _apply(self, fn, recurse=True):
        # parameters are views into flat storages owned by this module: device / dtype moves are not supported
        probe = fn(torch.zeros(1, device=self._device))
        if probe.device != self._device or probe.dtype != torch.float32:
            raise _lib.DaliError("ResNet50ReID lives on %s in fp32 storage; .to()/.half()/.cpu() are not supported" % self._device)
        return self

    # ---- plans ---------------------------------------------------------------------------------------
    def _plan(self, batch, height, width):
        key = (batch, height, width)
        p = self._plans.get(key)
        if p is None:
            p = self._plans[key] = _Plan(self._device, batch, height, width, self._layers, self._width)
        return p

    def _activate(self, plan):
        if self._arena is None or self._arena.numel() < plan.arena_bytes:
            self._arena = None
            torch.cuda.synchronize(self._device)
            self._arena = torch.empty(plan.arena_bytes + 256, device=self._device, dtype=torch.uint8)
            self._refreshed = (None, -1)
        L = _lib.lib()
        if self._last_plan is not plan:
            _lib.check(L.dali_resnet_bind(plan.h, _lib.ptr(self.flat_params), _lib.ptr(self.flat_grads), _lib.ptr(self.flat_buffers),
                                          _lib.ptr(self._arena), self._arena.numel()), "dali_resnet_bind")
            self._last_plan = plan
        if self._refreshed != (plan.key, self._weights_version):
            _lib.check(L.dali_resnet_refresh_weights(plan.h, _lib.stream_ptr()), "dali_resnet_refresh_weights")
            self._refreshed = (plan.key, self._weights_version)

    def _run_forward(self, x, training):
        if x.dim() != 4 or x.shape[1] != 3:
            raise _lib.DaliError("expected images [B,3,H,W], got %s" % (tuple(x.shape),))
        x = x.to(device=self._device, dtype=torch.float32).contiguous()
        plan = self._plan(x.shape[0], x.shape[2], x.shape[3])
        self._activate(plan)
        emb = torch.empty(x.shape[0], plan.feat_dim, device=self._device, dtype=torch.float32)
        if self.feature not in _FEATURE_MODES:
            raise _lib.DaliError("feature must be 'both', 'gap' or 'gmp' (got %r)" % (self.feature,))
        _lib.check(_lib.lib().dali_resnet_set_feature(plan.h, _FEATURE_MODES[self.feature]), "dali_resnet_set_feature")
        _lib.check(_lib.lib().dali_resnet_forward(plan.h, _lib.stream_ptr(), _lib.ptr(x), int(training), _lib.ptr(emb)),
                   "dali_resnet_forward")
        if training:
            self.flat_nbt += 1
            self._bwd_plan = plan
        return emb

    n_bwd_stages = 4          # 0: neck + head + layer4, 1: layer3, 2: layer2, 3: layer1 + stem (dali_resnet_backward)

    def _run_backward(self, d_emb):
        for stage in range(self.n_bwd_stages):
            self._backward_stage(d_emb, stage)
        self.attach_grads()

    def _backward_stage(self, d_emb, stage):
        plan = self._bwd_plan
        if plan is not self._last_plan:
            raise _lib.DaliError("backward() after another forward of a different shape is not supported")
        _lib.check(_lib.lib().dali_resnet_backward(plan.h, _lib.stream_ptr(), _lib.ptr(d_emb, torch.float32, "d_emb"), stage, stage),
                   "dali_resnet_backward")
        if self.grad_stage_hook is not None:
            b, e = plan.stage_range(stage)
            self.grad_stage_hook(stage, b, e)

    def attach_grads(self):
        """Point every parameter's .grad at its view of the flat gradient buffer (torch optimizers read .grad)."""
        for name, p in zip(self._param_names, self.parameters()):
            p.grad = self._grad_views[name]

    def stage_ranges(self, batch, height, width):
        plan = self._plan(batch, height, width)
        return [plan.stage_range(s) for s in range(4)]

    def debug_tensor(self, name, dtype, shape):
        ptr, nbytes = ctypes.c_void_p(), ctypes.c_int64()
        _lib.check(_lib.lib().dali_resnet_debug_tensor(self._last_plan.h, name.encode(), ctypes.byref(ptr), ctypes.byref(nbytes)),
                   "dali_resnet_debug_tensor")
        off = ptr.value - self._arena.data_ptr()
        return self._arena[off:off + nbytes.value].view(dtype).view(*shape).clone()

    # ---- nn.Module protocol -----------------------------------------------------------------------------
    def forward(self, x):
        if self.training and torch.is_grad_enabled():
            return _NetFn.apply(x, self._anchor, self)
        return self._run_forward(x, training=self.training)


class _DataParallelShim(nn.Module):
    """Keeps the reference's ``nn.DataParallel`` object protocol (``.module``, ``module.``-prefixed keys) for a
    one-GPU-per-process design."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *a, **k):
        return self.module(*a, **k)

    def cuda(self, device=None):
        return self


def getEnsembles(gpu_indexes):
    """Encoders.getEnsembles (Encoders.py:245-303) builds ResNet-50 + OSNet + DenseNet-121 pairs; OSNet and DenseNet are outside the
    scope table.  The name exists so that ``from Encoders import getDCNN, getEnsembles`` (mainKIT.py:27,
    train_encodersKIT.py:25, evaluateCleanATModels.py:13) keeps importing; mainKIT.main never calls it."""
    raise NotImplementedError("getEnsembles: the OSNet / DenseNet-121 ensemble members are out of scope of this build (SURVEY.md 2.1); "
                              "use getDCNN(gpu_indexes, 'resnet50')")


def getDCNN(gpu_indexes, model_name, embedding_size=None):
    """Encoders.getDCNN (Encoders.py:25-48, 241): -> (model_online, model_momentum), both eval()."""
    if model_name != "resnet50":
        raise NotImplementedError("getDCNN: only model_name='resnet50' is in scope of this build (got %r)" % model_name)
    if embedding_size not in (None, 2048):
        raise NotImplementedError("getDCNN: embedding_size is fixed at 2048 for resnet50")
    dev = torch.device("cuda", gpu_indexes[0] if len(gpu_indexes) else 0)
    model_source = _DataParallelShim(ResNet50ReID(device=dev))
    model_momentum = _DataParallelShim(ResNet50ReID(device=dev))
    model_momentum.load_state_dict(model_source.state_dict())
    return model_source.eval(), model_momentum.eval()
