"""GPU composition parity of the net plan, block by block, against the rounding-matched oracle.

A randomly initialised BatchNorm ResNet is chaotic (a 2^-9 perturbation grows by ~5 %/layer; see DESIGN.md
"Numerics"), so end-to-end comparisons cannot separate rounding noise from wiring bugs.  Here every bottleneck is
checked in isolation: the oracle re-runs ONE block (bf16 rounding at the kernel's rounding points,
oracle/resnet50_bf16.py) from the plan's own block input and back-propagates the plan's own incoming gradient;
block output, data gradient and every parameter gradient of that block must agree to bf16-ulp level."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.resnet50_reid import ResNet50ReID as OracleNet
from oracle import resnet50_bf16 as M

pytestmark = pytest.mark.gpu
bf16 = torch.bfloat16


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


def block_forward_matched(blk, x):
    u1 = M._conv(x, blk.conv1); a1 = M.Q(F.relu(M._bn_train(u1, M.Q(u1), blk.bn1)))
    u2 = M._conv(a1, blk.conv2); a2 = M.Q(F.relu(M._bn_train(u2, M.Q(u2), blk.bn2)))
    u3 = M._conv(a2, blk.conv3)
    # narrow blocks without a downsample branch never store conv3's output (bn3 through the moments of a2, csrc/bnlin.hip): bn3 acts on the
    # fp32 accumulators there; elsewhere raw3 is stored in bf16 as before
    out = M._bn_train(u3, M.Q(u3) if M.stores_raw3(blk) else u3, blk.bn3)
    if blk.downsample is not None:
        ud = M._conv(x, blk.downsample[0]); idn = M._bn_train(ud, M.Q(ud), blk.downsample[1])
    else:
        idn = x
    return M.Q(F.relu(out + idn))


@pytest.mark.parametrize("layers,width,shape", [((1, 1, 1, 1), 32, (8, 3, 64, 32)), ((1, 2, 1, 2), 64, (4, 3, 128, 64)),
                                                ((3, 4, 6, 3), 64, (8, 3, 256, 128))])
def test_plan_block_by_block(layers, width, shape):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from daliid_amd import Encoders
    torch.manual_seed(4)
    ref = OracleNet(layers=layers, width=width)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
                m.weight.copy_(0.5 + torch.rand(m.weight.shape, generator=g))
                m.bias.copy_(0.2 * torch.randn(m.bias.shape, generator=g))
    net = Encoders.ResNet50ReID(layers=layers, width=width)
    net.load_state_dict(ref.state_dict())
    ref.train(); net.train()
    n = shape[0]
    x = torch.randn(*shape, generator=g)
    d_emb = torch.randn(n, width * 32, generator=g)
    emb = net._run_forward(x.cuda(), training=True)

    blocks = [b for l in (ref.layer1, ref.layer2, ref.layer3, ref.layer4) for b in l]
    names = ["layer%d.%d" % (li + 1, bi) for li, l in enumerate((ref.layer1, ref.layer2, ref.layer3, ref.layer4)) for bi in range(len(l))]

    def nhwc_dbg(name, c, h, w):
        return net.debug_tensor(name, bf16, (n, h, w, c)).float().cpu().permute(0, 3, 1, 2).contiguous()

    # geometry of every block output
    hw = [(shape[2] // 4, shape[3] // 4)]
    for li, l in enumerate(layers):
        for bi in range(l):
            h, w = hw[-1]
            if bi == 0 and li in (1, 2):
                h, w = h // 2, w // 2
            hw.append((h, w))
    pool0 = nhwc_dbg("pool0", width, *hw[0])
    ys = [nhwc_dbg("block%d.y" % i, blocks[i].conv3.out_channels, *hw[i + 1]) for i in range(len(blocks))]

    # ---- forward, block by block: oracle(block)(plan's input) == plan's output ----
    for i, blk in enumerate(blocks):
        xin = pool0 if i == 0 else ys[i - 1]
        with torch.no_grad():
            yo = block_forward_matched(blk, xin)
        e = rel_l2(ys[i], yo)
        assert e < 6e-3, ("forward", names[i], e)
    # stem
    with torch.no_grad():
        u = M._conv(M.Q(x), ref.conv1)
        p0 = M.Q(F.max_pool2d(M._bn_train(u, M.Q(u), ref.bn1), 3, 2, 1))
    assert rel_l2(pool0, p0) < 3e-3
    # head: pooled feature + BN1d on the plan's last block output
    y_last = ys[-1].clone().requires_grad_(True)
    f = y_last.mean((2, 3)) + F.adaptive_max_pool2d(y_last, 1).flatten(1)
    emb_o = M._bn_train(f, f, ref.last_bn)
    assert rel_l2(emb.cpu(), emb_o.detach()) < 1e-3
    emb_o.backward(d_emb)
    dy = y_last.grad.to(bf16).float()                   # what head_pool_bwd stores

    # ---- backward, stage by stage (a stage = one layer); inside a stage the blocks chain on the plan's gradients ----
    grads = {k: v for k, v in net._grad_views.items()}
    d_emb_g = d_emb.cuda()
    bi_hi = len(blocks)
    worst = 0.0
    for stage in range(4):
        li = 3 - stage
        nblk = layers[li]
        net._backward_stage(d_emb_g, stage)
        if stage == 0:
            for k in ("weight", "bias"):
                e = rel_l2(grads["last_bn." + k].cpu(), getattr(ref.last_bn, k).grad)
                assert e < 1e-3, ("last_bn." + k, e)
        # oracle: chain this layer's blocks backwards from the gradient entering the layer
        for bi in range(bi_hi - 1, bi_hi - nblk - 1, -1):
            blk = blocks[bi]
            xin = (pool0 if bi == 0 else ys[bi - 1]).clone().requires_grad_(True)
            for p_ in blk.parameters():
                p_.grad = None
            yo = block_forward_matched(blk, xin)
            yo.backward(dy)
            dy = xin.grad.to(bf16).float()
            for pname, p_ in blk.named_parameters():
                got = grads[names[bi] + "." + pname].cpu()
                e = rel_l2(got, p_.grad)
                scale_free = float((got - p_.grad).abs().max() / p_.grad.abs().max().clamp(min=1e-30))
                worst = max(worst, min(e, scale_free))
                assert e < 4e-2 or scale_free < 2e-2, (names[bi], pname, e, scale_free)
        bi_hi -= nblk
        # the gradient the plan hands to the next stage vs the oracle chain
        c_in = blocks[bi_hi].conv1.in_channels
        h, w = hw[bi_hi]
        got_dx = net.debug_tensor("grad_cur", bf16, (n, h, w, c_in)).float().cpu().permute(0, 3, 1, 2)
        if bi_hi > 0:
            dy = dy * (ys[bi_hi - 1] > 0)               # the plan hands on the MASKED gradient dz = dy * (y > 0) of the previous block's output
        e = rel_l2(got_dx, dy)
        assert e < 4e-2, ("dx after stage", stage, e)
        dy = got_dx.contiguous()                        # continue from the plan's own gradient: errors do not chain
    # stem parameters (stage 3 ran them): oracle stem forward/backward from the image with the plan's d(pool0)
    for p_ in list(ref.conv1.parameters()) + list(ref.bn1.parameters()):
        p_.grad = None
    u = M._conv(M.Q(x), ref.conv1)
    p0 = M.Q(F.max_pool2d(M._bn_train(u, M.Q(u), ref.bn1), 3, 2, 1))
    p0.backward(dy)
    for name, p_ in (("conv1.weight", ref.conv1.weight), ("bn1.weight", ref.bn1.weight)):
        e = rel_l2(grads[name].cpu(), p_.grad)
        assert e < 4e-2, (name, e)
    print("worst per-block parameter-gradient error %.3e" % worst)
