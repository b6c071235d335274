"""Oracle vs the reference trainer (proxy picker, full trainer.train epochs) and ViT goldens."""
import numpy as np
import torch

from conftest import load_golden
from oracle import trainstep as TS
from oracle.resnet50_reid import ResNet50ReID, conv_flops_per_image
from oracle import vit as OV


def test_proxy_selection_matches_reference():
    z = load_golden("proxies.npz")
    idx, md = TS.select_proxies_farthest_point(torch.from_numpy(z["X40"]), 5, first=int(z["X40_first"]))
    assert idx.tolist() == z["X40_idx"].tolist() and abs(md - float(z["X40_maxdist"])) < 1e-4
    idx, md = TS.select_proxies_farthest_point(torch.from_numpy(z["X3"]), 5, first=int(z["X3_first"]))
    assert idx.tolist() == z["X3_idx"].tolist() and abs(md - float(z["X3_maxdist"])) < 1e-5
    idx, md = TS.select_proxies_farthest_point(torch.from_numpy(z["X1"]), 5, first=0)
    assert idx.tolist() == z["X1_idx"].tolist() and md == float(z["X1_maxdist"])


def test_resnet50_reid_shape_params_flops():
    m = ResNet50ReID()
    assert sum(p.numel() for p in m.parameters()) == 23512128      # SURVEY: 23.512 M (incl. last_bn)
    assert abs(conv_flops_per_image() / 1e9 - 8.107) < 0.01
    keys = list(m.state_dict().keys())
    assert "conv1.weight" in keys and "layer4.0.downsample.1.running_var" in keys and "last_bn.bias" in keys
    m = ResNet50ReID(layers=(1, 1, 1, 1), width=8).eval()
    assert m(torch.randn(2, 3, 64, 32)).shape == (2, 256)


def test_trainer_two_epochs_match_reference():
    """Replays train_encodersKIT.trainer.train (2 epochs x 2 PK batches) with the oracle pieces and
    compares every state_dict entry of the online and momentum models."""
    z = load_golden("trainer_epoch.npz")
    lr, wd, P, tau, beta, lam, n_epochs, per_id, n_ids = z["hyper"]
    P, per_id, n_ids, n_epochs = int(P), int(per_id), int(n_ids), int(n_epochs)
    images = torch.from_numpy(z["images"])
    labels = z["labels"]
    distort = z["distort"]
    online = ResNet50ReID(layers=(1, 1, 1, 1), width=8)
    momentum = ResNet50ReID(layers=(1, 1, 1, 1), width=8)
    init = {k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("init/")}
    online.load_state_dict(init)
    momentum.load_state_dict(init)
    opt = torch.optim.Adam(online.parameters(), lr=lr, weight_decay=wd)
    ids = np.unique(labels)
    for e, epoch in enumerate((1, 2)):
        online.eval()
        with torch.no_grad():
            fvs = online(images)
        centers, clabels, proxies, plabels = TS.build_centers_and_proxies(fvs, labels, 5, z["first_picks"][e])
        online.train(); momentum.eval()
        for b in range(len(ids) // P):
            sel = np.concatenate([np.where(labels == pid)[0] for pid in ids[b * P:(b + 1) * P]])
            TS.train_step(online, momentum, opt, images[sel], torch.from_numpy(labels[sel].astype(np.float32)),
                          torch.from_numpy(distort[sel]).long(), centers, clabels, proxies, plabels,
                          epoch, n_epochs, tau, beta, lam)
    bad = []
    for name, model in (("online", online), ("momentum", momentum)):
        for k, v in model.state_dict().items():
            ref = z["%s/%s" % (name, k)]
            if k == "bn1.bias":
                # The stem has no ReLU (Encoders.py:334) and feeds only 1x1 convs followed by train-mode BN,
                # so d loss / d bn1.bias is exactly zero in exact arithmetic: Adam normalises pure rounding
                # noise to +-lr per step.  Only its bound is comparable.
                assert np.abs(v.numpy()).max() <= 4 * lr * 1.01 and np.abs(ref).max() <= 4 * lr * 1.01
                continue
            # Adam turns rounding-level gradient differences into O(lr/10) parameter differences on
            # near-zero-gradient elements; running means downstream of bn1.bias inherit its +-lr noise.
            atol = 3e-4 if "running_mean" in k else 3e-5
            if not np.allclose(v.numpy(), ref, rtol=1e-3, atol=atol):
                bad.append((name, k, float(np.abs(v.numpy() - ref).max())))
    assert not bad, bad


def test_vit_tiny_forward_backward_matches_reference():
    z = load_golden("vit.npz")
    sd = {k[len("tiny/sd/"):]: torch.from_numpy(z[k]).requires_grad_(True) for k in z.files if k.startswith("tiny/sd/")}
    x = torch.from_numpy(z["tiny/x"]).requires_grad_(True)
    y = OV.transreid_forward(sd, x, num_heads=4, patch=8, stride=8)
    np.testing.assert_allclose(y.detach().numpy(), z["tiny/y"], rtol=1e-5, atol=1e-6)
    (y * torch.from_numpy(z["tiny/w"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), z["tiny/dx"], rtol=1e-4, atol=1e-6)
    for k in z.files:
        if k.startswith("tiny/grad/"):
            np.testing.assert_allclose(sd[k[len("tiny/grad/"):]].grad.numpy(), z[k], rtol=2e-4, atol=2e-6, err_msg=k)


def test_vit_droppath_matches_reference_train_mode():
    """DropPath (vit_pytorch.py:45-62): the oracle fed the uniform draws the reference's forward consumed reproduces the
    reference's train-mode output and every parameter gradient; with other draws it does not (the draws matter)."""
    z = load_golden("vit_droppath.npz")
    sd = {k[len("sd/"):]: torch.from_numpy(z[k]).requires_grad_(True) for k in z.files if k.startswith("sd/")}
    x, u, rate = torch.from_numpy(z["x"]), torch.from_numpy(z["u"]), float(z["rate"])
    y = OV.transreid_forward(sd, x, num_heads=1, patch=8, stride=8, drop_path=(rate, u))
    np.testing.assert_allclose(y.detach().numpy(), z["y"], rtol=1e-5, atol=1e-6)
    (y * torch.from_numpy(z["w"])).sum().backward()
    n = 0
    for k in z.files:
        if k.startswith("grad/"):
            np.testing.assert_allclose(sd[k[len("grad/"):]].grad.numpy(), z[k], rtol=2e-4, atol=2e-6, err_msg=k); n += 1
    assert n > 30
    y0 = OV.transreid_forward(sd, x, num_heads=1, patch=8, stride=8)            # no DropPath: a different function
    assert float((y0.detach() - torch.from_numpy(z["y"])).abs().max()) > 1e-3
