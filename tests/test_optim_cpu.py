"""CPU: which element ranges of the flat parameter buffer the fused Adam updates (daliid_amd.optim.active_ranges).
torch.optim.Adam skips parameters that are not in its list or whose grad is None (the ViT's frozen ``bottleneck.bias``,
make_models.py:181, and its never-called ``base.fc``); the fused step must skip exactly those."""
import pytest
import torch
from torch import nn

from daliid_amd import optim


class _FlatNet(nn.Module):
    """parameters as views into one flat buffer, 64-element aligned segments (the layout of the HIP plans)"""

    def __init__(self, sizes):
        super().__init__()
        offs, total = [], 0
        for n in sizes.values():
            offs.append(total); total += (n + 63) // 64 * 64
        self.flat_params = torch.zeros(total)
        for (name, n), off in zip(sizes.items(), offs):
            self.register_parameter(name, nn.Parameter(self.flat_params[off:off + n]))


def test_all_trainable_is_one_range():
    net = _FlatNet({"a": 100, "b": 64, "c": 7})
    assert optim.active_ranges(net) == [(0, net.flat_params.numel())]


def test_frozen_and_gradless_parameters_are_skipped():
    net = _FlatNet({"w0": 130, "fc_w": 200, "fc_b": 10, "neck_w": 64, "neck_b": 64})
    net.neck_b.requires_grad_(False)
    net._no_grad_params = ("fc_w", "fc_b")
    # w0 occupies [0,192) with padding, fc_w [192,448), fc_b [448,512), neck_w [512,576), neck_b [576,640)
    assert optim.active_ranges(net) == [(0, 192), (512, 576)]
    # an explicit optimizer list (bench.py passes the requires_grad parameters): same result
    assert optim.active_ranges(net, [p for p in net.parameters() if p.requires_grad]) == [(0, 192), (512, 576)]
    # the reference's own construction, torch.optim.Adam(model.parameters()) (mainKIT.py:99): the frozen parameter IS in the list, and torch
    # still skips it because its grad stays None
    assert optim.active_ranges(net, list(net.parameters())) == [(0, 192), (512, 576)]
    # a list that leaves out w0
    assert optim.active_ranges(net, [net.neck_w]) == [(512, 576)]


def test_foreign_parameter_is_rejected():
    net = _FlatNet({"a": 64})
    from daliid_amd._lib import DaliError
    with pytest.raises(DaliError):
        optim.active_ranges(net, [nn.Parameter(torch.zeros(3))])
