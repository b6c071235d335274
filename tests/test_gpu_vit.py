"""GPU parity of the TransReID ViT plan: (1) full ViT-B/16 + BN neck, eval mode, against outputs of the REFERENCE's own
make_models.make_model / vit_pytorch.TransReID (tests/golden/vit.npz: weights are re-seeded per key so only outputs are
stored); (2) forward + backward of a small head_dim-64 config against the CPU oracle (oracle/vit.py, itself pinned to
the reference by the tiny golden)."""
import types

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import vit as OV

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


def _cfg(size=(224, 224)):
    return types.SimpleNamespace(
        MODEL=types.SimpleNamespace(NAME="transformer", JPM=False, LAST_STRIDE=1, PRETRAIN_PATH="", PRETRAIN_CHOICE="none", COS_LAYER=False,
                                    NECK="bnneck", TRANSFORMER_TYPE="vit_base_patch16_224_TransReID", SIE_CAMERA=False, SIE_VIEW=False,
                                    SIE_COE=3.0, STRIDE_SIZE=16, DROP_PATH=0.0, DROP_OUT=0.0, ATT_DROP_RATE=0.0, ID_LOSS_TYPE="softmax",
                                    RE_ARRANGE=False),
        TEST=types.SimpleNamespace(NECK_FEAT="after"), INPUT=types.SimpleNamespace(SIZE_TRAIN=size))


def _golden_state(keys, shapes):
    """the per-key seeded weights make_golden.gen_vit loaded into the reference model"""
    sd = {}
    for i, (k, shp) in enumerate(zip(keys, shapes)):
        shape = tuple(int(v) for v in shp.strip("()").split(",") if v.strip())
        gg = torch.Generator().manual_seed(1000 + i)
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith("running_var"):
            sd[k] = 0.5 + torch.rand(shape, generator=gg)
        elif k.endswith("norm1.weight") or k.endswith("norm2.weight") or k.endswith("norm.weight") or k == "bottleneck.weight":
            sd[k] = 1.0 + 0.1 * torch.randn(shape, generator=gg)
        else:
            sd[k] = 0.02 * torch.randn(shape, generator=gg)
    return sd


def test_full_vit_b16_matches_reference_outputs():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from daliid_amd import make_models
    z = load_golden("vit.npz")
    keys, shapes = [str(k) for k in z["full/keys"]], [str(s) for s in z["full/shapes"]]
    model = make_models.make_model(_cfg(), 10, 0, 0)
    assert list(model.state_dict().keys()) == keys                                   # 157 reference keys, same order
    for (k, v), shp in zip(model.state_dict().items(), shapes):
        assert str(tuple(v.shape)) == shp, k
    model.load_state_dict(_golden_state(keys, shapes))
    model.eval()
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(77))
    with torch.no_grad():
        y = model(x.cuda()).cpu()
        gf = model.global_feat(x.cuda()).cpu()
    e_gf, e_y = rel_l2(gf, torch.from_numpy(z["full/global_feat"])), rel_l2(y, torch.from_numpy(z["full/y_eval"]))
    print("ViT-B/16 eval: global_feat rel-L2 %.3e, post-neck feat rel-L2 %.3e (bf16 activations vs the fp32 reference)" % (e_gf, e_y))
    assert e_gf < 2e-2 and e_y < 2e-2


def test_small_vit_forward_backward_vs_oracle():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from daliid_amd import vit_pytorch as V
    net = V.ViTNeckNet(img_size=(64, 32), patch_size=16, stride_size=16, embed_dim=128, depth=2, num_heads=2, mlp_ratio=4.0, num_classes=10, seed=3)
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for p in net.parameters():
            p.add_((0.05 * torch.randn(p.shape, generator=g)).to(p.device))
    B = 6
    x = torch.randn(B, 3, 64, 32, generator=g)
    w = torch.randn(B, 128, generator=g)
    pnames = {n for n, _ in net.named_parameters()}
    sd = {k: v.detach().cpu().clone().requires_grad_(k in pnames) for k, v in net.state_dict().items()}
    ref = OV.build_transformer_forward(sd, x, num_heads=2, patch=16, stride=16, training=True)
    (ref * w).sum().backward()
    net.train()
    y = net(x.cuda())
    (y * w.cuda()).sum().backward()
    e = rel_l2(y.detach().cpu(), ref.detach())
    print("small ViT train-mode feat rel-L2 %.3e" % e)
    assert e < 3e-2
    worst, bad = ("", 0.0), []
    for name, p in net.named_parameters():
        if name.startswith("base.fc.") or name == "bottleneck.bias":
            continue                                   # unused head (zero gradient) / frozen neck bias
        r = rel_l2(p.grad.cpu(), sd[name].grad)
        if r > worst[1]:
            worst = (name, r)
        # base.norm.{weight,bias}: their gradient is ~0 in exact arithmetic (a train-mode BatchNorm follows: sum_b d = 0,
        # sum_b d*xhat = 0); compare on the scale of the other gradients instead of relatively
        if name.startswith("base.norm."):
            assert float((p.grad.cpu() - sd[name].grad).abs().max()) < 2e-3, name
            continue
        if r >= 8e-2:
            bad.append((name, r, float(p.grad.abs().max()), float(sd[name].grad.abs().max())))
    print("worst gradient rel-L2 %.3e at %s" % (worst[1], worst[0]))
    assert not bad, bad
    assert float(net.base.fc.weight.grad.abs().max()) == 0.0
    # eval mode through the running statistics
    net.eval()
    with torch.no_grad():
        e2 = rel_l2(net(x.cuda()).cpu(), OV.build_transformer_forward({k: v.cpu() for k, v in net.state_dict().items()}, x, 2, 16, 16, training=False))
    assert e2 < 3e-2, e2
