#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING the reference's own Python files.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden.py

What is imported from /root/reference/Person-ReID and how:
  * losses.py            -- as-is.  Harness shims: a no-op ``termcolor`` module (imported at
                            losses.py:3, never called on the live path) and
                            ``torch.Tensor.cuda = identity`` (the file hard-codes ``.cuda(gpu_index)``,
                            losses.py:52,76,292; there is no GPU here).
  * vit_pytorch.py, make_models.py -- as-is (torch only).
  * train_encodersKIT.py -- with inert placeholder modules for torchreid / torchvision /
                            matplotlib (they compute nothing); used for
                            ``selectProxiesByTriagulation`` and to drive ``trainer.train`` with
                            ``extractFeatures`` / ``samplePKBatches`` / ``DataLoader`` patched to serve
                            in-memory synthetic tensors in a fixed order.
Only data (inputs + the reference's outputs) is written; no reference source text is copied.
"""
import os
import sys
import types
import io
import contextlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/Person-ReID"
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)


class _Inert(types.ModuleType):
    """Placeholder module: any attribute is another placeholder; calling it returns None."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        m = _Inert(self.__name__ + "." + name)
        setattr(self, name, m)
        sys.modules[m.__name__] = m
        return m

    def __call__(self, *a, **k):
        return None


def install_shims():
    for n in ["termcolor", "torchreid", "torchvision", "torchvision.models", "torchvision.transforms",
              "torchvision.utils", "torchreid.metrics", "matplotlib", "matplotlib.pyplot"]:
        if n not in sys.modules:
            try:
                __import__(n)
            except Exception:
                sys.modules[n] = _Inert(n)
    sys.modules["termcolor"].colored = lambda s, *a, **k: s
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self


def unit_rows(n, d, gen):
    x = torch.randn(n, d, generator=gen)
    return x / x.norm(dim=1, keepdim=True)


def case_inputs(nb, D, NC, ppc, seed, unknown_labels=0, ragged=False):
    """Seeded synthetic inputs of one loss case (also imported by the tests to regenerate the
    D=2048 case instead of storing 50 MB of centers/proxies)."""
    g = torch.Generator().manual_seed(seed)
    centers = unit_rows(NC, D, g)
    centers_labels = (np.arange(NC) * 3 + 7).astype(np.int64)          # non-contiguous ids
    counts = np.full(NC, ppc)
    if ragged:
        counts = 1 + (np.arange(NC) % ppc)                               # 1..ppc proxies/class
    proxies_labels = np.repeat(centers_labels, counts)
    proxies = unit_rows(int(counts.sum()), D, g)
    lab_idx = torch.randint(0, NC, (nb,), generator=g).numpy()
    labels = centers_labels[lab_idx].astype(np.float32)
    if unknown_labels:
        labels[:unknown_labels] = 5.0                                    # id with no center / proxy
    distortion = torch.randint(0, 6, (nb,), generator=g)
    # embeddings correlated with their class so the softmax is not uniform
    fv = unit_rows(nb, D, g) + 0.7 * centers[lab_idx]
    fv = fv / fv.norm(dim=1, keepdim=True)
    return dict(fv=fv, labels=labels, distortion=distortion, centers=centers, centers_labels=centers_labels,
                proxies=proxies, proxies_labels=proxies_labels)


# --------------------------------------------------------------------------- losses
def gen_losses(ref_losses):
    out = {}
    cases = []

    def run_case(name, nb, D, NC, ppc, epoch, n_epochs, tau, seed, unknown_labels=0, ragged=False,
                 store_full_grad=True):
        ci = case_inputs(nb, D, NC, ppc, seed, unknown_labels, ragged)
        fv, labels, distortion = ci["fv"], ci["labels"], ci["distortion"]
        centers, centers_labels, proxies, proxies_labels = (ci["centers"], ci["centers_labels"], ci["proxies"],
                                                            ci["proxies_labels"])

        fn = fv.clone().requires_grad_(True)
        bl = torch.from_numpy(labels)
        with contextlib.redirect_stdout(io.StringIO()):
            lc, acc, amp = ref_losses.BatchWeightedCenterLoss(fn, bl, distortion, centers, centers_labels,
                                                              epoch, n_epochs, 0, tau=tau, gpu_index=0)
        (gc,) = torch.autograd.grad(lc, fn)
        fn2 = fv.clone().requires_grad_(True)
        lp = ref_losses.BatchWeightedProxyLoss(fn2, bl, distortion, proxies, proxies_labels, epoch, n_epochs,
                                               top_negs=50, tau=tau, gpu_index=0)
        (gp,) = torch.autograd.grad(lp, fn2)
        pre = name + "/"
        if store_full_grad:
            out[pre + "fv"] = fv.numpy()
            out[pre + "labels"] = labels
            out[pre + "distortion"] = distortion.numpy().astype(np.int64)
            out[pre + "centers"] = centers.numpy()
            out[pre + "centers_labels"] = centers_labels
            out[pre + "proxies"] = proxies.numpy()
            out[pre + "proxies_labels"] = proxies_labels
        else:   # regenerated by the test from the seed; checksums guard against RNG drift
            out[pre + "gen_args"] = np.array([nb, D, NC, ppc, seed, unknown_labels, int(ragged)])
            out[pre + "input_checksums"] = np.array([fv.double().sum().item(), centers.double().abs().sum().item(),
                                                     proxies.double().abs().sum().item(), float(labels.sum()),
                                                     float(distortion.sum())])
        out[pre + "hyper"] = np.array([epoch, n_epochs, tau], dtype=np.float64)
        out[pre + "center_loss"] = np.float32(lc.item())
        out[pre + "center_acc"] = np.float64(acc)
        out[pre + "center_avg_max_prob"] = np.float64(amp)
        out[pre + "proxy_loss"] = np.float32(lp.item())
        if store_full_grad:
            out[pre + "center_grad"] = gc.numpy()
            out[pre + "proxy_grad"] = gp.numpy()
        else:
            out[pre + "center_grad_head"] = gc[:8].numpy()
            out[pre + "proxy_grad_head"] = gp[:8].numpy()
            out[pre + "center_grad_abs_sum"] = np.float64(gc.double().abs().sum())
            out[pre + "proxy_grad_abs_sum"] = np.float64(gp.double().abs().sum())
            out[pre + "center_grad_rowsum"] = gc.double().sum(1).numpy()
            out[pre + "proxy_grad_rowsum"] = gp.double().sum(1).numpy()
        cases.append(name)

    i = 0
    for epoch in (1, 10, 250):
        for tau in (0.05, 0.1):
            run_case("small_e%d_t%s" % (epoch, str(tau).replace(".", "p")), 32, 64, 16, 5, epoch, 250, tau, 100 + i)
            i += 1
    run_case("unknown_ids", 32, 64, 16, 5, 10, 250, 0.05, 200, unknown_labels=3)
    run_case("ragged_proxies", 48, 64, 12, 5, 40, 250, 0.05, 201, ragged=True)
    run_case("tiny_batch", 3, 32, 4, 2, 7, 20, 0.1, 202)
    run_case("full_d2048", 256, 2048, 1024, 5, 10, 250, 0.05, 203, store_full_grad=False)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "losses.npz"), **out)
    print("losses.npz:", len(cases), "cases")

    # schedule table (losses.py:5-7)
    t = [(1, 250), (10, 250), (125, 250), (250, 250), (3, 7)]
    mins = [0.8, 0.6, 0.4, 0.2, 0.1, 0.0]
    vals = np.array([[ref_losses.getValueFromCosineSchedule(a, b, n_min=m, n_max=1.0) for m in mins] for a, b in t])
    acc_pred = np.array([1, 2, 2, 3, 3, 3, 9])
    acc_gt = np.array([1, 2, 3, 3, 3, 4, 9])
    np.savez(os.path.join(HERE, "schedule.npz"), t=np.array(t), mins=np.array(mins), values=vals,
             acc_pred=acc_pred, acc_gt=acc_gt, acc_bal=np.float64(ref_losses.getACCBal(acc_pred, acc_gt)))
    print("schedule.npz ok")


# --------------------------------------------------------------------------- optional triplet head
def gen_triplet(ref_losses):
    """BatchWeightedSoftmaxTripletLoss (losses.py:607-654): in-batch hardest positive / hardest negative per row,
    13-entry distortion weight table."""
    out, cases = {}, []
    for name, nb, D, n_ids, epoch, n_epochs, tau, seed in (("pk_small", 32, 64, 8, 10, 250, 0.05, 300),
                                                            ("pk_small_t0p1", 32, 64, 8, 1, 250, 0.1, 301),
                                                            ("pk_256", 256, 256, 16, 120, 250, 0.05, 302),
                                                            ("two_ids", 6, 32, 2, 250, 250, 0.1, 303)):
        g = torch.Generator().manual_seed(seed)
        ids = torch.randint(0, n_ids, (nb,), generator=g)
        ids[0], ids[1] = 0, 1                                            # at least two identities
        proto = unit_rows(n_ids, D, g)
        fv = unit_rows(nb, D, g) + 0.8 * proto[ids]
        fv = fv / fv.norm(dim=1, keepdim=True)
        labels = (ids.numpy() * 5 + 2).astype(np.float32)
        distortion = torch.randint(0, 13, (nb,), generator=g)
        fn = fv.clone().requires_grad_(True)
        loss = ref_losses.BatchWeightedSoftmaxTripletLoss(fn, torch.from_numpy(labels), distortion, epoch, n_epochs, tau=tau, gpu_index=0)
        (grad,) = torch.autograd.grad(loss, fn)
        pre = name + "/"
        out[pre + "fv"], out[pre + "labels"], out[pre + "distortion"] = fv.numpy(), labels, distortion.numpy().astype(np.int64)
        out[pre + "hyper"] = np.array([epoch, n_epochs, tau], dtype=np.float64)
        out[pre + "loss"], out[pre + "grad"] = np.float32(loss.item()), grad.numpy()
        cases.append(name)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "triplet.npz"), **out)
    print("triplet.npz:", len(cases), "cases")


# --------------------------------------------------------------------------- proxies
def gen_proxies(T):
    out = {}
    g = torch.Generator().manual_seed(7)
    X = torch.randn(40, 2048, generator=g)
    np.random.seed(12)
    state_first = int(np.random.choice(40))
    np.random.seed(12)
    idx, md = T.selectProxiesByTriagulation(X, num_proxies=5)
    out["X40"] = X.numpy(); out["X40_first"] = state_first
    out["X40_idx"] = idx.numpy(); out["X40_maxdist"] = np.float64(md)
    X3 = torch.randn(3, 64, generator=g)
    np.random.seed(3)
    f3 = int(np.random.choice(3))
    np.random.seed(3)
    idx3, md3 = T.selectProxiesByTriagulation(X3, num_proxies=5)
    out["X3"] = X3.numpy(); out["X3_first"] = f3; out["X3_idx"] = idx3.numpy(); out["X3_maxdist"] = np.float64(md3)
    X1 = torch.randn(1, 16, generator=g)
    np.random.seed(0)
    idx1, md1 = T.selectProxiesByTriagulation(X1, num_proxies=5)
    out["X1"] = X1.numpy(); out["X1_idx"] = idx1.numpy(); out["X1_maxdist"] = np.float64(md1)
    np.savez_compressed(os.path.join(HERE, "proxies.npz"), **out)
    print("proxies.npz: idx40", idx.tolist(), "idx3", idx3.tolist(), "idx1", idx1.tolist())


# --------------------------------------------------------------------------- trainer epoch
def gen_trainer(T):
    """Drive the reference trainer.train for two epochs on a tiny ResNet (oracle topology) and
    record the resulting online / momentum state_dicts."""
    from oracle.resnet50_reid import ResNet50ReID
    from torch.utils.data import DataLoader as RealLoader, Dataset

    torch.manual_seed(5)
    n_ids, per_id, H, W = 6, 5, 32, 16
    g = torch.Generator().manual_seed(11)
    images = torch.randn(n_ids * per_id, 3, H, W, generator=g)
    labels = np.repeat(np.arange(n_ids) * 2 + 1, per_id).astype(np.int32)
    distort = torch.randint(0, 6, (n_ids * per_id,), generator=g).numpy().astype(np.int32)
    records = np.array([["img%03d" % i, str(labels[i]), "0", "person"] for i in range(len(labels))])

    online = ResNet50ReID(layers=(1, 1, 1, 1), width=8)
    momentum = ResNet50ReID(layers=(1, 1, 1, 1), width=8)
    momentum.load_state_dict(online.state_dict())
    init_sd = {k: v.clone() for k, v in online.state_dict().items()}
    online.eval(); momentum.eval()

    def fake_extract(subset, h, w, model, bs, gpu_index=0, **kw):
        model.eval()
        ids = [int(r[0][3:]) for r in subset]
        with torch.no_grad():
            return model(images[ids]).data.cpu()

    class FakePK(Dataset):
        def __init__(self, dataset, imgs, lbls, h, w, tdir, kind, K=4, turb_strength=0):
            self.lbls = lbls
            self.set = np.unique(lbls)                       # fixed order (no shuffle)

        def __getitem__(self, i):
            pid = self.set[i]
            sel = np.where(self.lbls == pid)[0]
            return images[sel], torch.ones(len(sel)) * pid, distort[sel]

        def __len__(self):
            return len(self.set)

    T.extractFeatures = fake_extract
    T.samplePKBatches = FakePK
    T.DataLoader = lambda ds, batch_size, collate_fn, **kw: RealLoader(ds, batch_size=batch_size, shuffle=False,
                                                                       drop_last=True, collate_fn=collate_fn)
    T.tqdm = lambda x: x
    lr, wd, P, tau, beta, lam, n_epochs = 3.5e-4, 5e-4, 3, 0.05, 0.9, 0.4, 250
    opt = torch.optim.Adam(online.parameters(), lr=lr, weight_decay=wd)
    tr = T.trainer("Synthetic", records, "resnet50", {}, H, W, None, False, 0, opt, P, per_id, tau, beta, lam,
                   n_epochs, online, momentum, [0], "v0")
    firsts = []
    np.random.seed(21)
    buf = io.StringIO()
    for epoch in (1, 2):
        # record the np.random draws the proxy picker will make (one per class, in class order)
        st = np.random.get_state()
        firsts.append([int(np.random.choice(per_id)) for _ in range(n_ids)])
        np.random.set_state(st)
        with contextlib.redirect_stdout(buf):
            tr.train(records, labels, 1, epoch)
    log = buf.getvalue()
    out = {"images": images.numpy(), "labels": labels, "distort": distort, "first_picks": np.array(firsts),
           "hyper": np.array([lr, wd, P, tau, beta, lam, n_epochs, per_id, n_ids], dtype=np.float64),
           "log": np.array(log)}
    for k, v in init_sd.items():
        out["init/" + k] = v.numpy()
    for k, v in online.state_dict().items():
        out["online/" + k] = v.numpy()
    for k, v in momentum.state_dict().items():
        out["momentum/" + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "trainer_epoch.npz"), **out)
    print("trainer_epoch.npz ok;", [l for l in log.splitlines() if "Mean" in l][-3:])


# --------------------------------------------------------------------------- ViT
def gen_vit():
    import vit_pytorch as V
    import make_models as M
    out = {}
    with contextlib.redirect_stdout(io.StringIO()):
        torch.manual_seed(3)
        tiny = V.TransReID(img_size=(32, 32), patch_size=8, stride_size=8, embed_dim=64, depth=2, num_heads=4,
                           mlp_ratio=4, qkv_bias=True, drop_path_rate=0.0, num_classes=10,
                           norm_layer=__import__("functools").partial(torch.nn.LayerNorm, eps=1e-6))
    tiny.eval()
    # the reference initialises cls/pos with tiny std; perturb all params so every term matters
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for p in tiny.parameters():
            p.add_(0.05 * torch.randn(p.shape, generator=g))
    x = torch.randn(3, 3, 32, 32, generator=g)
    xg = x.clone().requires_grad_(True)
    y = tiny(xg)
    w = torch.randn(y.shape, generator=g)
    (y * w).sum().backward()
    for k, v in tiny.state_dict().items():
        out["tiny/sd/" + k] = v.numpy()
    out["tiny/x"] = x.numpy(); out["tiny/y"] = y.detach().numpy(); out["tiny/w"] = w.numpy()
    out["tiny/dx"] = xg.grad.numpy()
    for k, p in tiny.named_parameters():
        if p.grad is not None:
            out["tiny/grad/" + k] = p.grad.numpy()

    # full ViT-B/16 through make_model (BN neck): weights are seeded here, only outputs stored
    cfg = types.SimpleNamespace(
        MODEL=types.SimpleNamespace(NAME="transformer", JPM=False, LAST_STRIDE=1, PRETRAIN_PATH="", PRETRAIN_CHOICE="none",
                                    COS_LAYER=False, NECK="bnneck", TRANSFORMER_TYPE="vit_base_patch16_224_TransReID",
                                    SIE_CAMERA=False, SIE_VIEW=False, SIE_COE=3.0, STRIDE_SIZE=16, DROP_PATH=0.0,
                                    DROP_OUT=0.0, ATT_DROP_RATE=0.0, ID_LOSS_TYPE="softmax", RE_ARRANGE=False),
        TEST=types.SimpleNamespace(NECK_FEAT="after"),
        INPUT=types.SimpleNamespace(SIZE_TRAIN=(224, 224)))
    with contextlib.redirect_stdout(io.StringIO()):
        torch.manual_seed(8)
        full = M.make_model(cfg, 10, 0, 0)
    keys = list(full.state_dict().keys())
    out["full/keys"] = np.array(keys)
    out["full/shapes"] = np.array([str(tuple(v.shape)) for v in full.state_dict().values()])
    # deterministic re-init that the test can reproduce without the reference: per-key seeded normal
    sd = {}
    for i, (k, v) in enumerate(full.state_dict().items()):
        gg = torch.Generator().manual_seed(1000 + i)
        if v.dtype.is_floating_point:
            if k.endswith("running_var"):
                sd[k] = 0.5 + torch.rand(v.shape, generator=gg)
            elif k.endswith("norm1.weight") or k.endswith("norm2.weight") or k.endswith("norm.weight") or k == "bottleneck.weight":
                sd[k] = 1.0 + 0.1 * torch.randn(v.shape, generator=gg)
            else:
                sd[k] = 0.02 * torch.randn(v.shape, generator=gg)
        else:
            sd[k] = v.clone()
    full.load_state_dict(sd)
    full.eval()
    xf = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(77))
    with torch.no_grad():
        yf = full(xf)
        gf = full.base(xf)
    out["full/y_eval"] = yf.numpy(); out["full/global_feat"] = gf.numpy()
    np.savez_compressed(os.path.join(HERE, "vit.npz"), **out)
    print("vit.npz ok: tiny y", tuple(y.shape), "full y", tuple(yf.shape), "keys", len(keys))


def gen_vit_droppath():
    """DropPath in training mode (vit_pytorch.py:45-62, wired :178-179 with the per-block rates of :338): the reference's
    TransReID, tiny config, rate 0.5 over depth 3 (block rates 0, 0.25, 0.5), forward + backward under a fixed CPU seed.
    The uniform draws the forward consumed are replayed from the same seed (one torch.rand((B,1,1)) per DropPath module in
    execution order; block 0 is nn.Identity) and stored, so that the oracle / the HIP plan can be fed the same draws."""
    import vit_pytorch as V
    out = {}
    depth, rate, B = 3, 0.5, 6
    with contextlib.redirect_stdout(io.StringIO()):
        torch.manual_seed(5)
        tiny = V.TransReID(img_size=(32, 32), patch_size=8, stride_size=8, embed_dim=64, depth=depth, num_heads=1,
                           mlp_ratio=4, qkv_bias=True, drop_path_rate=rate, num_classes=10,
                           norm_layer=__import__("functools").partial(torch.nn.LayerNorm, eps=1e-6))
    g = torch.Generator().manual_seed(6)
    with torch.no_grad():
        for p in tiny.parameters():
            p.add_(0.05 * torch.randn(p.shape, generator=g))
    x = torch.randn(B, 3, 32, 32, generator=g)
    w = torch.randn(B, 64, generator=g)
    tiny.train()
    torch.manual_seed(77)
    y = tiny(x)
    (y * w).sum().backward()
    torch.manual_seed(77)                                 # replay the draws of the forward above
    u = torch.full((2 * depth, B), 0.5)
    for i in range(depth):
        if isinstance(tiny.blocks[i].drop_path, torch.nn.Identity):
            continue
        for br in range(2):
            u[2 * i + br] = torch.rand((B, 1, 1)).reshape(B)
    # sanity: some sample must actually be dropped somewhere, some kept
    dpr = torch.linspace(0, rate, depth).repeat_interleave(2).unsqueeze(1)
    kept = (1 - dpr + u).floor()
    assert 0 < int((kept == 0).sum()) < kept.numel(), kept
    for k, v in tiny.state_dict().items():
        out["sd/" + k] = v.numpy()
    out["x"] = x.numpy(); out["w"] = w.numpy(); out["y"] = y.detach().numpy(); out["u"] = u.numpy()
    out["rate"] = np.float64(rate); out["depth"] = np.int64(depth)
    for k, p in tiny.named_parameters():
        if p.grad is not None:
            out["grad/" + k] = p.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "vit_droppath.npz"), **out)
    print("vit_droppath.npz ok: y", tuple(y.shape), "dropped", int((kept == 0).sum()), "of", kept.numel())


def main():
    install_shims()
    if sys.argv[1:] == ["--only", "vit_droppath"]:                       # added in round 2
        gen_vit_droppath()
        return
    import losses as ref_losses
    if sys.argv[1:] == ["--only", "triplet"]:                            # added after the first fixtures were frozen
        gen_triplet(ref_losses)
        return
    gen_losses(ref_losses)
    gen_triplet(ref_losses)
    import train_encodersKIT as T
    gen_proxies(T)
    gen_trainer(T)
    gen_vit()
    gen_vit_droppath()


if __name__ == "__main__":
    main()
