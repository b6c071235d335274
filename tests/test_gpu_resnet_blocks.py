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
    # blocks whose bn3 runs through the moments of a2 (csrc/bnlin.hip) never store conv3's output: bn3 acts on the fp32 accumulators and the
    # backward uses the merged bf16 weight images (oracle _Conv3Bn3Moments); elsewhere raw3 is stored in bf16 as before
    if not M.stores_raw3(blk):
        out = M._Conv3Bn3Moments.apply(a2, blk.conv3.weight, blk.bn3.weight, blk.bn3.bias, 1e-5)
    else:
        u3 = M._conv(a2, blk.conv3)
        out = M._bn_train(u3, M.Q(u3), blk.bn3)
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


def test_plan_first_blocks_at_the_benchmarked_batch():
    """The same block-by-block check INSIDE the benchmarked plan (configs[1]: batch 256 of 256 x 128, ResNet-50 (3, 4, 6, 3)), restricted to
    the stem and the first bottleneck of every layer -- the four blocks with a downsample branch, both strides, the moments scheme of layer1's
    branch -- so that the CPU twin stays cheap.  At this batch the plan takes its own tile shapes, statistic-tile counts, split-K counts, the
    Gram path at P = 524288 and the persistent streaming kernel for conv3 / the masked conv1 gradients; the batch-8 case above takes none of
    them.  The plan's own block input and the plan's own incoming gradient (read block by block through dali_debug_resnet_backward_block) go
    through the rounding-matched twin of ONE block: output, data gradient and every parameter gradient, same bounds as above."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import ctypes
    from daliid_amd import Encoders, _lib
    torch.set_num_threads(max(torch.get_num_threads(), 16))
    layers, width, n, H, W = (3, 4, 6, 3), 64, 256, 256, 128
    torch.manual_seed(4)
    ref = OracleNet(layers=layers, width=width)
    g = torch.Generator().manual_seed(6)
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
                m.weight.copy_(0.5 + torch.rand(m.weight.shape, generator=g))
                m.bias.copy_(0.2 * torch.randn(m.bias.shape, generator=g))
    net = Encoders.ResNet50ReID(layers=layers, width=width)
    net.load_state_dict(ref.state_dict())
    ref.train(); net.train()
    x = torch.randn(n, 3, H, W, generator=g)
    d_emb = torch.randn(n, width * 32, generator=g).cuda()
    net._run_forward(x.cuda(), training=True)
    blocks = [b for l in (ref.layer1, ref.layer2, ref.layer3, ref.layer4) for b in l]
    names = ["layer%d.%d" % (li + 1, bi) for li, l in enumerate((ref.layer1, ref.layer2, ref.layer3, ref.layer4)) for bi in range(len(l))]
    first = [0, 3, 7, 13]                                          # first bottleneck of layer1 .. layer4
    hw = [(H // 4, W // 4)]
    for li, l in enumerate(layers):
        for bi in range(l):
            h, w = hw[-1]
            if bi == 0 and li in (1, 2):
                h, w = h // 2, w // 2
            hw.append((h, w))

    def nchw(name, c, h, w):
        return net.debug_tensor(name, bf16, (n, h, w, c)).float().cpu().permute(0, 3, 1, 2).contiguous()
    L = _lib.lib()
    L.dali_debug_resnet_backward_block.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    plan = net._last_plan.h
    # backward block by block; keep the gradient ENTERING each first block (= grad_cur after the block behind it) and the one it hands on
    dy_in, dx_out, d_raw = {}, {}, {}
    for bi in range(len(blocks) - 1, -1, -1):
        if bi in first:
            c, (h, w) = blocks[bi].conv3.out_channels, hw[bi + 1]
            if bi == len(blocks) - 1:
                raise AssertionError("the last block is never a first block of ResNet-50")
            dy_in[bi] = nchw("grad_cur", c, h, w)
        _lib.check(L.dali_debug_resnet_backward_block(plan, _lib.stream_ptr(), _lib.ptr(d_emb), bi), "dali_debug_resnet_backward_block")
        if bi in first:
            dx_out[bi] = nchw("grad_cur", blocks[bi].conv1.in_channels, *hw[bi])
            wdt = blocks[bi].conv1.out_channels
            d_raw[bi] = (nchw("block%d.d_raw1" % bi, wdt, *hw[bi]),)
    _lib.check(L.dali_debug_resnet_backward_block(plan, _lib.stream_ptr(), _lib.ptr(d_emb), -1), "dali_debug_resnet_backward_block")
    torch.cuda.synchronize()
    grads = {k: v.cpu() for k, v in net._grad_views.items()}
    worst, report, bad = 0.0, [], []
    for bi in first:
        blk = blocks[bi]
        cin, cout = blk.conv1.in_channels, blk.conv3.out_channels
        xin = (nchw("pool0", width, *hw[0]) if bi == 0 else nchw("block%d.y" % (bi - 1), cin, *hw[bi])).requires_grad_(True)
        y_plan = nchw("block%d.y" % bi, cout, *hw[bi + 1])
        for p_ in blk.parameters():
            p_.grad = None
        keep = {}
        yo = block_forward_matched_first(blk, xin, bi == 0, keep)
        e = rel_l2(y_plan, yo.detach())
        report.append("%s forward %.2e" % (names[bi], e))
        if not e < 6e-3: bad.append(("forward", names[bi], e))
        yo.backward(dy_in[bi])                                     # the plan's masked gradient dz of this block's output
        want_dx = xin.grad.to(bf16).float()
        if bi > 0:
            want_dx = want_dx * (xin.detach() > 0)                 # the plan hands on dz of the PREVIOUS block's output
        e = rel_l2(dx_out[bi], want_dx)
        report.append("%s data gradient %.2e" % (names[bi], e))
        if not e < 4e-2: bad.append(("data gradient", names[bi], e))
        report.append("%s d_raw1 (the stored gradient of conv1's raw output) vs the twin's %.2e" % (names[bi], rel_l2(d_raw[bi][0], keep["conv1"][1].grad)))
        for pname, p_ in blk.named_parameters():
            got = grads[names[bi] + "." + pname]
            e = rel_l2(got, p_.grad)
            scale_free = float((got - p_.grad).abs().max() / p_.grad.abs().max().clamp(min=1e-30))
            worst = max(worst, min(e, scale_free))
            report.append("%s %s rel-L2 %.2e max-abs/max %.2e" % (names[bi], pname, e, scale_free))
            if not (e < 4e-2 or scale_free < 2e-2):
                # Found at batch 256 on layer1.0's conv1.weight (plan vs twin 5.7e-2).  Traced: the plan's stored d_raw1 agrees with the twin's to 5e-3
                # element by element; the weight-gradient kernel reproduces its sum from that stored operand to 1e-5 (and is accurate to 7e-8 under
                # cancellation, scripts/wgrad_cancel_probe.py); shifting the twin's BatchNorm gradient sums to the plan's values, or giving the twin
                # the bf16 weight images of the moments scheme, changes nothing.  What remains is the quantity itself: against the UN-ROUNDED fp32
                # block (plain autograd on the same input and incoming gradient) the twin -- a CPU program -- is 8.9e-2 off and the plan 1.07e-1:
                # two stacked BatchNorm backwards leave conv1's weight gradient as a small residual of sums over 524288 pixels, and bf16 storage of
                # the gradients above it moves it by ~10 % in ANY implementation.  Bound such a parameter the way the end-to-end tests do: the plan
                # must be no further from the twin than the twin is from fp32, and at most 1.5 x as far from fp32 as the twin.
                saved = {n_: q_.grad.clone() for n_, q_ in blk.named_parameters()}
                for q_ in blk.parameters():
                    q_.grad = None
                x3 = xin.detach().clone().requires_grad_(True)
                bnf = lambda t_, m_: F.batch_norm(t_, None, None, m_.weight, m_.bias, True, 0.0, 1e-5)
                a_ = F.relu(bnf(F.conv2d(x3, blk.conv1.weight), blk.bn1))
                a_ = F.relu(bnf(F.conv2d(a_, blk.conv2.weight, stride=blk.conv2.stride, padding=1), blk.bn2))
                o_ = bnf(F.conv2d(a_, blk.conv3.weight), blk.bn3) + bnf(F.conv2d(x3, blk.downsample[0].weight, stride=blk.downsample[0].stride), blk.downsample[1])
                F.relu(o_).backward(dy_in[bi])
                e_plan32, e_twin32 = rel_l2(got, p_.grad), rel_l2(saved[pname], p_.grad)
                report.append("%s %s against the un-rounded fp32 block: plan %.2e, twin %.2e" % (names[bi], pname, e_plan32, e_twin32))
                for n_, q_ in blk.named_parameters():
                    q_.grad = saved[n_]
                del x3, a_, o_
                if e <= e_twin32 and e_plan32 <= 1.5 * e_twin32:
                    continue
                bad.append((names[bi], pname, e, scale_free))
        del xin, yo, y_plan
    # stem: oracle stem forward / backward from the image with the plan's d(pool0)
    pool0 = nchw("pool0", width, *hw[0])
    for p_ in list(ref.conv1.parameters()) + list(ref.bn1.parameters()):
        p_.grad = None
    u = M._conv(M.Q(x), ref.conv1)
    p0 = M.Q(F.max_pool2d(M._bn_train(u, M.Q(u), ref.bn1), 3, 2, 1))
    assert rel_l2(pool0, p0.detach()) < 3e-3
    p0.backward(dx_out[0])
    for name, p_ in (("conv1.weight", ref.conv1.weight), ("bn1.weight", ref.bn1.weight)):
        e = rel_l2(grads[name], p_.grad)
        assert e < 4e-2, (name, e)
    print("\n".join(report))
    print("batch 256: worst first-block parameter-gradient error %.3e" % worst)
    assert not bad, bad


def block_forward_matched_first(blk, x, first_of_net, keep=None):
    """block_forward_matched with the downsample branch's rounding point of the net's first block (its BatchNorm backward runs through the moments
    of the block input: no d_rawd tensor, resnet_plan.hip lin_ds / oracle ds_through_moments).  keep (dict): receives conv1's input and raw output (whose gradient the test reads)."""
    u1 = M._conv(x, blk.conv1); a1 = M.Q(F.relu(M._bn_train(u1, M.Q(u1), blk.bn1)))
    u2 = M._conv(a1, blk.conv2); a2 = M.Q(F.relu(M._bn_train(u2, M.Q(u2), blk.bn2)))
    if not M.stores_raw3(blk):
        out = M._Conv3Bn3Moments.apply(a2, blk.conv3.weight, blk.bn3.weight, blk.bn3.bias, 1e-5)
        u3 = None
    else:
        u3 = M._conv(a2, blk.conv3)
        out = M._bn_train(u3, M.Q(u3), blk.bn3)
    ud = M._conv(x, blk.downsample[0])
    idn = M._bn_train(ud, (M.QF if M.ds_through_moments(blk, first_of_net) else M.Q)(ud), blk.downsample[1])
    if keep is not None:
        u1.retain_grad()
        keep.update({"conv1": (x, u1)})
    return M.Q(F.relu(out + idn))
