"""Oracle (rounding-matched): the ResNet-50-ReID train-mode forward/backward in fp32 torch on CPU with a
bf16 round-trip inserted at exactly the points where the HIP pipeline stores a tensor in bf16.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Why it exists: a randomly initialised BatchNorm ResNet amplifies perturbations layer over layer, so the
bf16-storage HIP path drifts from the pure-fp32 oracle (oracle/resnet50_reid.py) by rounding noise alone;
that drift says nothing about correctness.  This twin rounds where the kernels round, so what is left is fp32
summation order -- and any real indexing / fusion bug.

Rounding points mirrored (daliid_amd/csrc/resnet_plan.hip):
  images and conv weights -> bf16 operands; every raw conv output stored bf16 (except conv3 of the blocks without a downsample
  branch, whose BatchNorm is applied to the fp32 accumulators inside the GEMM: csrc/bnlin.hip); relu(bn(raw)) rounded to bf16 when
  it is formed in the consumer's operand load; block outputs y and the pooled stem output stored bf16; in the
  backward pass the gradients of those same tensors are stored bf16 (except the downsample output's where its BatchNorm backward runs
  through the moments of the block input: ds_through_moments).  BatchNorm statistics come from the fp32
  accumulators (the un-rounded conv result) and are applied to the rounded tensor; weight gradients stay fp32.
"""
import torch
import torch.nn.functional as F


class _Q(torch.autograd.Function):
    """bf16 round-trip in the forward AND on the gradient in the backward."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).float()

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).float()


class _QW(torch.autograd.Function):
    """bf16 round-trip of a weight operand; its gradient stays fp32."""

    @staticmethod
    def forward(ctx, w):
        return w.to(torch.bfloat16).float()

    @staticmethod
    def backward(ctx, g):
        return g


class _QF(torch.autograd.Function):
    """bf16 round-trip in the forward only (a stored tensor whose gradient the HIP plan never forms as a tensor of its own)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).float()

    @staticmethod
    def backward(ctx, g):
        return g


Q = _Q.apply
QW = _QW.apply
QF = _QF.apply


def _bn_train(u, uq, bn, eps=1e-5):
    """statistics from the un-rounded conv result u, affine applied to the stored (rounded) tensor uq."""
    dims = (0, 2, 3) if u.dim() == 4 else (0,)
    mean = u.mean(dims)
    var = u.var(dims, unbiased=False)
    scale = bn.weight / torch.sqrt(var + eps)
    shift = bn.bias - mean * scale
    shp = (1, -1, 1, 1) if u.dim() == 4 else (1, -1)
    return uq * scale.view(shp) + shift.view(shp)


def stores_raw3(blk):
    """Does the HIP plan store conv3's output of this bottleneck in bf16?  Not where bn3 runs through the moments of a2 (csrc/bnlin.hip):
    blocks whose width is a multiple of 32 and at most DALI_BNLIN_MAXW (default 512 = every block of ResNet-50; resnet_plan.hip)."""
    import os
    w = blk.conv3.in_channels
    return not (w % 32 == 0 and w <= int(os.environ.get("DALI_BNLIN_MAXW", "512")))


def ds_through_moments(blk, first_of_net):
    """Does the HIP plan run the downsample BatchNorm's backward through the moments of the block input (no d_rawd tensor, resnet_plan.hip
    `lin_ds`)?  The net's first block, when its branch is a stride-1 convolution of at most 128 channels and bn3 goes through moments too."""
    d = blk.downsample
    return (first_of_net and d is not None and d[0].stride == (1, 1) and d[0].in_channels % 32 == 0 and d[0].in_channels <= 128 and not stores_raw3(blk))


class _Conv3Bn3Moments(torch.autograd.Function):
    """conv3 (1x1) + training-mode bn3 of a bottleneck whose BatchNorm runs through the moments of conv3's input (csrc/bnlin.hip; every block of
    ResNet-50 by default).  Forward: bn3 on the fp32 accumulators (raw3 is never stored).  Backward, as the plan forms it (resnet_plan.hip
    block_backward, launch_bnlin_bwd): with dz the masked gradient of the block output, m_b = mean(dz), m_g = mean(dz xhat),
        d_raw3 = A dz - Kc - Qc raw3,   A = scale,  Qc = scale invstd m_g,  Kc = scale (m_b - mean invstd m_g)
        d_a2   = dz (A.W3) - Kc W3 - a2 (W3^T diag(Qc) W3)
    where the two matrices (A.W3)^T and W3^T diag(Qc) W3 are WEIGHT IMAGES OF THE DATA-GRADIENT GEMM, i.e. rounded to bf16 after the fp32
    coefficients were folded in.  That second rounding is a coherent 2^-9 perturbation of the backward map (every pixel sees the same perturbed
    matrix), which plain autograd through bf16(W3) does not have; the weight gradients of the convolutions below -- small residuals of large sums --
    pick it up in full (layer1.0 conv1.weight at batch 256: 5.7e-2 against a twin without it).  dW3, gamma', beta' stay fp32 as in the plan."""

    @staticmethod
    def forward(ctx, a2, w, gamma, beta, eps):
        wq = w.to(torch.bfloat16).float()
        u = F.conv2d(a2, wq)
        mean = u.mean((0, 2, 3))
        var = u.var((0, 2, 3), unbiased=False)
        invstd = 1.0 / torch.sqrt(var + eps)
        ctx.save_for_backward(a2, wq, gamma, u, mean, invstd)
        sh = (1, -1, 1, 1)
        return (u - mean.view(sh)) * (gamma * invstd).view(sh) + beta.view(sh)

    @staticmethod
    def backward(ctx, dz):
        a2, wq, gamma, u, mean, invstd = ctx.saved_tensors
        sh = (1, -1, 1, 1)
        P = u.numel() / u.shape[1]
        xhat = (u - mean.view(sh)) * invstd.view(sh)
        d_beta = dz.sum((0, 2, 3))
        d_gamma = (dz * xhat).sum((0, 2, 3))
        scale = gamma * invstd
        d_raw = scale.view(sh) * (dz - (d_beta / P).view(sh) - xhat * (d_gamma / P).view(sh))
        dw = torch.nn.grad.conv2d_weight(a2, wq.shape, d_raw)                       # fp32, as the plan's G0-based form up to summation order
        w2 = wq.flatten(1)                                                          # [C, width]
        Qc = scale * invstd * d_gamma / P
        Kc = scale * (d_beta / P - mean * invstd * d_gamma / P)
        m1 = (scale.view(-1, 1) * w2).to(torch.bfloat16).float()                    # (A.W3): rounded once more as a GEMM operand image
        m2 = (w2.t() @ (Qc.view(-1, 1) * w2)).to(torch.bfloat16).float()            # W3^T diag(Qc) W3 [width, width], likewise
        bvec = w2.t() @ Kc                                                          # fp32 bias of the merged GEMM
        d_a2 = torch.einsum("nchw,ck->nkhw", dz, m1) - torch.einsum("nkhw,kj->njhw", a2, m2) - bvec.view(sh)
        return d_a2, dw, d_gamma, d_beta, None


def _conv(x, conv):
    return F.conv2d(x, QW(conv.weight), stride=conv.stride, padding=conv.padding)


def _bn_eval(u, uq, bn, eps=1e-5):
    """running statistics folded into scale / shift (dali_resnet_forward, training = 0), applied to the stored (rounded) tensor"""
    scale = bn.weight / torch.sqrt(bn.running_var + eps)
    shift = bn.bias - bn.running_mean * scale
    shp = (1, -1, 1, 1) if uq.dim() == 4 else (1, -1)
    return uq * scale.view(shp) + shift.view(shp)


def cat_eval_supported(blk, n_pixels, n_cus=256):
    """Does the HIP plan run this block's conv3 and downsample convolution as ONE GEMM over [a2 | x] in the inference forward (resnet_plan.hip
    `cat_eval`, conv.hip conv_cat_act_supported)?  Stride-1 branch, channel counts multiples of 64, and a size one of the two kernels with that
    size the persistent streaming kernel takes with a split weight image: K = 2 (w + cin) <= 256 and at least two 128 x 128 tiles per CU.
    ResNet-50 at 256 x 128: layer1's first block."""
    import os
    d = blk.downsample
    if d is None or d[0].stride != (1, 1) or stores_raw3(blk) or os.environ.get("DALI_EVAL_FUSED", "1") == "0" or os.environ.get("DALI_EVAL_CAT", "1") == "0":
        return False
    w, cin, C = blk.conv3.in_channels, d[0].in_channels, blk.conv3.out_channels
    if w % 64 or cin % 64 or C % 128:
        return False
    K = 2 * (w + cin)                                   # split (hi + lo) weight images: every channel twice
    if K > 256:
        return False
    tiles_m, tiles_n = C // 128, (n_pixels + 127) // 128
    return tiles_m * tiles_n >= 2 * n_cus and (n_cus // 8) % tiles_m == 0


def _bn_eval_coeffs(bn, eps=1e-5):
    scale = bn.weight / torch.sqrt(bn.running_var + eps)
    return scale, bn.bias - bn.running_mean * scale


def forward_matched(model, x, training=True):
    """model: oracle.resnet50_reid.ResNet50ReID (its parameters receive the gradients).  training=True: batch statistics (the train
    step); False: running statistics (extractFeatures), same rounding points."""
    _bn = _bn_train if training else _bn_eval
    # inference (dali_resnet_forward, training = 0): bn1 / bn2 + ReLU ride in their convolution's output stage, i.e. they act on the fp32
    # accumulators and a1 / a2 are the only tensors stored (conv_bn_relu_eval, resnet_plan.hip); training stores the raw outputs in bf16
    Qr = Q if training else (lambda t: t)
    x = Q(x)
    u = _conv(x, model.conv1)
    # (inference runs the stem in one launch, stem.hip, with these rounding points: bf16 of the convolution's output, fp32 affine, bf16 of the pooled value)
    z = _bn(u, Q(u), model.bn1)                   # no ReLU after the stem BN (Encoders.py:334)
    x = Q(F.max_pool2d(z, 3, 2, 1))
    first = True
    for layer in (model.layer1, model.layer2, model.layer3, model.layer4):
        for blk in layer:
            u1 = _conv(x, blk.conv1)
            a1 = Q(F.relu(_bn(u1, Qr(u1), blk.bn1)))
            u2 = _conv(a1, blk.conv2)
            a2 = Q(F.relu(_bn(u2, Qr(u2), blk.bn2)))
            if not training and cat_eval_supported(blk, a2.shape[0] * a2.shape[2] * a2.shape[3]):
                # inference, stride-1 downsample branch: one GEMM over [a2 | x] against [s3.W3 | sd.Wd], the BatchNorm scales folded into the weight
                # image (split in bf16 hi + lo parts, from the fp32 weights) and the shifts summed (resnet_plan.hip cat_eval, fold_cat_weights_kernel)
                s3, h3 = _bn_eval_coeffs(blk.bn3)
                sd, hd = _bn_eval_coeffs(blk.downsample[1])
                def hi_lo(v):                        # the split weight image: hi = bf16(v), lo = bf16(v - hi); the GEMM sees hi + lo
                    hi = v.to(torch.bfloat16).float()
                    return hi + (v - hi).to(torch.bfloat16).float()
                f3 = hi_lo(s3.view(-1, 1, 1, 1) * blk.conv3.weight)
                fd = hi_lo(sd.view(-1, 1, 1, 1) * blk.downsample[0].weight)
                x = Q(F.relu(F.conv2d(a2, f3) + F.conv2d(x, fd) + (h3 + hd).view(1, -1, 1, 1)))
                first = False
                continue
            if training and not stores_raw3(blk):
                out = _Conv3Bn3Moments.apply(a2, blk.conv3.weight, blk.bn3.weight, blk.bn3.bias, 1e-5)
            else:
                u3 = _conv(a2, blk.conv3)
                # where conv3's output is never stored (csrc/bnlin.hip) bn3 acts on the fp32 accumulators
                out = _bn(u3, Q(u3) if stores_raw3(blk) else u3, blk.bn3)
            if blk.downsample is not None:
                ud = _conv(x, blk.downsample[0])
                idn = _bn(ud, (QF if ds_through_moments(blk, first) else Q)(ud), blk.downsample[1])
            else:
                idn = x
            first = False
            x = Q(F.relu(out + idn))
    f = x.mean((2, 3)) + F.adaptive_max_pool2d(x, 1).flatten(1)     # Encoders.py:341-345
    return _bn(f, f, model.last_bn)               # BatchNorm1d neck in fp32
