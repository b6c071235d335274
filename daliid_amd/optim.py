"""Optimizer side of the trainer hot loop on the model's flat storages: fused Adam(L2) and the EMA momentum update.

``FusedAdam`` reproduces ``torch.optim.Adam(params, lr, weight_decay=wd)`` as mainKIT.py:99 configures it and reads
``lr`` / ``weight_decay`` from a torch optimizer's ``param_groups`` each step, so the reference driver's per-epoch
``lambda_lr_warmup`` (mainKIT.py:144, 204-208) keeps working."""
import torch

from . import _lib


def _param_table(net):
    """[(name, parameter, flat offset, numel)] of a net on flat storage, by address."""
    base, out = net.flat_params.data_ptr(), []
    for name, p in net.named_parameters():
        out.append((name, p, (p.data_ptr() - base) // 4, p.numel()))
    return out


def active_ranges(net, params=None):
    """Element ranges of the flat parameter buffer that the optimizer updates, as torch.optim.Adam would: the parameters it
    was given (``params``; default: those with requires_grad) minus the ones that never receive a gradient (torch skips
    ``grad is None``: the ViT's unused ``base.fc.*``, make_models.py never calls it; the net lists them in
    ``_no_grad_params``).  Neighbouring segments are merged across the zero padding between them (padding has p = g = 0,
    so Adam leaves it at 0)."""
    table = _param_table(net)
    no_grad = set(getattr(net, "_no_grad_params", ()))
    if params is None:
        chosen = {id(p) for _, p, _, _ in table if p.requires_grad}
    else:
        by_ptr = {p.data_ptr(): p for _, p, _, _ in table}
        chosen = set()
        for q in params:
            m = by_ptr.get(q.data_ptr())
            if m is None:
                raise _lib.DaliError("FusedAdam: an optimizer parameter is not a view of this net's flat parameter buffer")
            chosen.add(id(m))
    # requires_grad=False parameters never get a .grad, so torch.optim.Adam skips them even when they are in its list (the reference
    # builds Adam(model.parameters()), mainKIT.py:99: the ViT's frozen bottleneck.bias is in that list, make_models.py:180-182)
    segs = sorted((off, off + n, id(p) in chosen and p.requires_grad and name not in no_grad) for name, p, off, n in table)
    total = net.flat_params.numel()
    ranges, cur = [], None
    for i, (b, e, on) in enumerate(segs):
        nxt = segs[i + 1][0] if i + 1 < len(segs) else total          # the padding behind a segment belongs to it
        if on:
            if cur is not None and cur[1] == b:
                cur[1] = nxt
            else:
                cur = [b, nxt]
                ranges.append(cur)
        else:
            cur = None
    return [(b, e) for b, e in ranges]


class FusedAdam:
    """One fused Adam launch per ACTIVE range of the flat buffers (one range for ResNet-50; two for the ViT, which skips
    the unused ``base.fc.*`` and the frozen ``bottleneck.bias`` exactly as torch.optim.Adam does for grad-less or
    excluded parameters)."""

    def __init__(self, net, lr=3.5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, torch_optimizer=None, params=None):
        self.net = net
        self.torch_optimizer = torch_optimizer
        self.defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.exp_avg = torch.zeros_like(net.flat_params)
        self.exp_avg_sq = torch.zeros_like(net.flat_params)
        self.step_count = 0
        self.ranges = active_ranges(net, params)
        if not self.ranges:
            raise _lib.DaliError("FusedAdam: no trainable parameter")
        for b, e in self.ranges:
            if b % 4 or e % 4:
                raise _lib.DaliError("FusedAdam: active range [%d, %d) is not 16-byte aligned" % (b, e))
        dev = net.flat_params.device
        self._sq = torch.zeros(len(self.ranges), device=dev, dtype=torch.float32)
        self.weights_sqsum = torch.zeros(1, device=dev, dtype=torch.float32)
        self._frozen_key, self._frozen_sq = None, None

    @classmethod
    def from_torch(cls, optimizer, net):
        if not isinstance(optimizer, torch.optim.Adam) or len(optimizer.param_groups) != 1:
            raise _lib.DaliError("FusedAdam.from_torch expects a single-group torch.optim.Adam (mainKIT.py:99)")
        g = optimizer.param_groups[0]
        if g.get("amsgrad", False) or g.get("maximize", False):
            raise _lib.DaliError("amsgrad / maximize are not supported")
        return cls(net, lr=g["lr"], betas=tuple(g["betas"]), eps=g["eps"], weight_decay=g["weight_decay"], torch_optimizer=optimizer,
                   params=list(g["params"]))

    def hyper(self):
        if self.torch_optimizer is not None:
            g = self.torch_optimizer.param_groups[0]
            return g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"]
        d = self.defaults
        return d["lr"], d["betas"], d["eps"], d["weight_decay"]

    def zero_grad(self, set_to_none=True):
        pass                               # the backward overwrites the flat gradient buffer

    def _frozen_sqsum(self):
        """sum p^2 over the parameters Adam does not touch (the trainer's weights_sum runs over ALL parameters,
        train_encodersKIT.py:229-231); they only change through load_state_dict / copy_, which bump the version counter."""
        net = self.net
        key = net.flat_params._version
        if self._frozen_key != key or self._frozen_sq is None:
            tot = torch.zeros(1, device=net.flat_params.device)
            pos = 0
            for b, e in self.ranges + [(net.flat_params.numel(), net.flat_params.numel())]:
                if b > pos:
                    tot += net.flat_params[pos:b].square().sum()
                pos = e
            self._frozen_sq = tot
        return self._frozen_sq

    def step(self, grad_scale=1.0):
        lr, (b1, b2), eps, wd = self.hyper()
        self.step_count += 1
        net = self.net
        L, c, st = _lib.lib(), _lib.ctx(net.flat_params.device), _lib.stream_ptr()
        frozen = self._frozen_sqsum() if len(self.ranges) > 1 or self.ranges[0] != (0, net.flat_params.numel()) else None
        for i, (b, e) in enumerate(self.ranges):
            _lib.check(L.dali_adam_step(c, st, _lib.ptr(net.flat_params[b:e]), _lib.ptr(net.flat_grads[b:e]), _lib.ptr(self.exp_avg[b:e]),
                                        _lib.ptr(self.exp_avg_sq[b:e]), e - b, float(lr), float(b1), float(b2), float(eps), float(wd),
                                        self.step_count, float(grad_scale), _lib.ptr(self._sq[i:i + 1])), "dali_adam_step")
        if frozen is None:
            self.weights_sqsum = self._sq[0:1]
        else:
            self.weights_sqsum = self._sq.sum().reshape(1) + frozen
        net.mark_weights_changed()
        if frozen is not None:
            self._frozen_key = net.flat_params._version            # our own bump is not a change of the frozen parameters


def ema_update(momentum_net, online_net, beta):
    """train_encodersKIT.py:218-226 over every state_dict entry: parameters, BN running statistics (flat fp32) and the
    int64 num_batches_tracked counters (float result truncated back to int64, as load_state_dict does)."""
    L = _lib.lib()
    dev = online_net.flat_params.device
    for m, o in ((momentum_net.flat_params, online_net.flat_params), (momentum_net.flat_buffers, online_net.flat_buffers)):
        _lib.check(L.dali_ema_update(_lib.ctx(dev), _lib.stream_ptr(), _lib.ptr(m), _lib.ptr(o), m.numel(), float(beta)), "dali_ema_update")
    momentum_net.mark_weights_changed()
    momentum_net.flat_nbt.copy_((beta * momentum_net.flat_nbt.double() + (1 - beta) * online_net.flat_nbt.double()).to(torch.long))
