"""Optimizer side of the trainer hot loop on the model's flat storages: fused Adam(L2) and the EMA momentum update.

``FusedAdam`` reproduces ``torch.optim.Adam(params, lr, weight_decay=wd)`` as mainKIT.py:99 configures it and reads
``lr`` / ``weight_decay`` from a torch optimizer's ``param_groups`` each step, so the reference driver's per-epoch
``lambda_lr_warmup`` (mainKIT.py:144, 204-208) keeps working."""
import torch

from . import _lib


class FusedAdam:
    def __init__(self, net, lr=3.5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, torch_optimizer=None):
        self.net = net
        self.torch_optimizer = torch_optimizer
        self.defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.exp_avg = torch.zeros_like(net.flat_params)
        self.exp_avg_sq = torch.zeros_like(net.flat_params)
        self.step_count = 0
        self.weights_sqsum = torch.zeros(1, device=net.flat_params.device, dtype=torch.float32)

    @classmethod
    def from_torch(cls, optimizer, net):
        if not isinstance(optimizer, torch.optim.Adam) or len(optimizer.param_groups) != 1:
            raise _lib.DaliError("FusedAdam.from_torch expects a single-group torch.optim.Adam (mainKIT.py:99)")
        g = optimizer.param_groups[0]
        if g.get("amsgrad", False) or g.get("maximize", False):
            raise _lib.DaliError("amsgrad / maximize are not supported")
        return cls(net, lr=g["lr"], betas=tuple(g["betas"]), eps=g["eps"], weight_decay=g["weight_decay"], torch_optimizer=optimizer)

    def hyper(self):
        if self.torch_optimizer is not None:
            g = self.torch_optimizer.param_groups[0]
            return g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"]
        d = self.defaults
        return d["lr"], d["betas"], d["eps"], d["weight_decay"]

    def zero_grad(self, set_to_none=True):
        pass                               # the backward overwrites the flat gradient buffer

    def step(self, grad_scale=1.0):
        lr, (b1, b2), eps, wd = self.hyper()
        self.step_count += 1
        net = self.net
        _lib.check(_lib.lib().dali_adam_step(_lib.ctx(net.flat_params.device), _lib.stream_ptr(), _lib.ptr(net.flat_params), _lib.ptr(net.flat_grads),
                                              _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq), net.flat_params.numel(), float(lr), float(b1),
                                              float(b2), float(eps), float(wd), self.step_count, float(grad_scale), _lib.ptr(self.weights_sqsum)),
                   "dali_adam_step")
        net.mark_weights_changed()


def ema_update(momentum_net, online_net, beta):
    """train_encodersKIT.py:218-226 over every state_dict entry: parameters, BN running statistics (flat fp32) and the
    int64 num_batches_tracked counters (float result truncated back to int64, as load_state_dict does)."""
    L = _lib.lib()
    dev = online_net.flat_params.device
    for m, o in ((momentum_net.flat_params, online_net.flat_params), (momentum_net.flat_buffers, online_net.flat_buffers)):
        _lib.check(L.dali_ema_update(_lib.ctx(dev), _lib.stream_ptr(), _lib.ptr(m), _lib.ptr(o), m.numel(), float(beta)), "dali_ema_update")
    momentum_net.mark_weights_changed()
    momentum_net.flat_nbt.copy_((beta * momentum_net.flat_nbt.double() + (1 - beta) * online_net.flat_nbt.double()).to(torch.long))
