"""GPU: the trainer / validator mirrors end to end on synthetic identities (small net so the CPU oracle is quick)."""
import numpy as np
import pytest
import torch

from oracle import evalrank as E
from oracle import losses as OL
from oracle.resnet50_bf16 import forward_matched
from oracle.resnet50_reid import ResNet50ReID as OracleNet
from oracle.trainstep import l2norm_train

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from daliid_amd import Encoders, synthetic, train_encodersKIT, validateModels, getFeatures
    data = synthetic.SyntheticImages(n_ids=8, per_id=6, n_cams=3, seed=5, noise=0.4).install()
    yield Encoders, data, train_encodersKIT, validateModels, getFeatures
    synthetic.SyntheticImages.uninstall()


def _models(Encoders, seed=7):
    online = Encoders._DataParallelShim(Encoders.ResNet50ReID(layers=(1, 1, 1, 1), width=32, seed=seed))
    momentum = Encoders._DataParallelShim(Encoders.ResNet50ReID(layers=(1, 1, 1, 1), width=32, seed=seed))
    momentum.load_state_dict(online.state_dict())
    return online.eval(), momentum.eval()


def test_first_step_losses_match_oracle_and_training_reduces_loss(env):
    Encoders, data, T, V, G = env
    online, momentum = _models(Encoders)
    train, gallery, query = data.split(1)
    labels = np.int32(train[:, 1])
    H, W = 64, 32
    opt = torch.optim.Adam(online.parameters(), lr=3.5e-4, weight_decay=5e-4)                 # mainKIT.py:99
    tr = T.trainer("Synthetic", train, "resnet50", {}, H, W, None, False, 0, opt, 4, 4, 0.05, 0.9, 0.4, 250, online, momentum, [0], "t")
    # --- epoch-level targets + one explicit step, compared with the oracle on identical weights / batch ---
    ref = OracleNet(layers=(1, 1, 1, 1), width=32)
    ref.load_state_dict({k[len("module."):]: v.cpu() for k, v in online.state_dict().items()})
    np.random.seed(3)
    heads = tr.build_targets(train, labels)
    imgs = data.loader(list(train[:16, 0]), H, W)
    blabels = torch.from_numpy(labels[:16].astype(np.float32))
    dist = torch.randint(0, 6, (16,), generator=torch.Generator().manual_seed(1))
    from daliid_amd.losses import _codes, _sample_weights
    acc = torch.zeros(6, device="cuda")
    online.train()
    params_before = online.module.flat_params.clone()
    mom_before = momentum.module.flat_params.clone()
    stats = tr.train_step(heads, imgs, _codes(blabels, imgs.device), _sample_weights(dist, 1, 250, imgs.device), acc)
    c_hip, p_hip = float(stats[0] / stats[1]), float(stats[2] / stats[3])
    ref.train()
    fn = l2norm_train(ref(imgs.cpu()))
    c_ref = OL.center_loss(fn, blabels, dist, heads.centers.cpu(), heads.clabels.cpu().numpy(), 1, 250, 0.05)[0].item()
    p_ref = OL.proxy_loss(fn, blabels, dist, heads.proxies.cpu(), heads.plabels.cpu().numpy(), 1, 250, 0.05).item()
    # the same step through the rounding-matched twin (fp32 arithmetic, bf16 round trips where the kernels store bf16): the bound on
    # HIP-vs-fp32 is the twin's own distance from fp32 (x 1.5) plus a fixed 0.5 % -- not a percentage picked to pass
    fn_t = l2norm_train(forward_matched(ref, imgs.cpu()))
    c_twin = OL.center_loss(fn_t, blabels, dist, heads.centers.cpu(), heads.clabels.cpu().numpy(), 1, 250, 0.05)[0].item()
    p_twin = OL.proxy_loss(fn_t, blabels, dist, heads.proxies.cpu(), heads.plabels.cpu().numpy(), 1, 250, 0.05).item()
    print("first step: center %.4f (oracle %.4f, twin %.4f)  proxy %.4f (oracle %.4f, twin %.4f)" % (c_hip, c_ref, c_twin, p_hip, p_ref, p_twin))
    assert abs(c_hip - c_ref) < 1.5 * abs(c_twin - c_ref) + 5e-3 * abs(c_ref), (c_hip, c_ref, c_twin)
    assert abs(p_hip - p_ref) < 1.5 * abs(p_twin - p_ref) + 5e-3 * abs(p_ref), (p_hip, p_ref, p_twin)
    # Adam moved the online weights by ~lr, the momentum model by (1-beta) of that
    d_on = (online.module.flat_params - params_before).abs().max().item()
    assert 1e-4 < d_on < 1e-3
    ema_expect = 0.9 * mom_before + 0.1 * online.module.flat_params
    assert torch.allclose(momentum.module.flat_params, ema_expect, rtol=1e-5, atol=1e-7)
    assert float(acc[4]) == 1 and abs(float(acc[3]) - float((online.module.flat_params.double() ** 2).sum())) < 1e-3 * float(acc[3])
    # --- full epochs through the reference-shaped entry point: the loss must go down ---
    losses = []
    for epoch in range(1, 7):
        tr.train(train, labels, 1, epoch)
        losses.append(tr.last_epoch_stats["loss"])
        assert tr.last_epoch_stats["steps"] == 2 and np.isfinite(losses[-1])
    print("epoch losses", ["%.3f" % l for l in losses])
    assert losses[-1] < losses[0] - 0.05
    assert not online.training and not momentum.training               # train_encodersKIT.py:248-249


def test_validate_matches_oracle_on_same_features(env):
    Encoders, data, T, V, G = env
    online, _ = _models(Encoders, seed=9)
    train, gallery, query = data.split(1)
    validator = V.validationManager.getValidator("Market")
    validator.setParameters(64, 32, False, 0)
    cmc, mAP, distmat = validator.validate(query, gallery, online)
    q = G.extractFeatures(query, 64, 32, online, 500, 0)            # CPU fp32 like the reference
    g = G.extractFeatures(gallery, 64, 32, online, 500, 0)
    assert q.device.type == "cpu" and q.shape == (len(query), 1024)
    ref_d = E.validate_features(q, g)
    ref_cmc, ref_map = E.eval_market1501(ref_d.numpy(), query[:, 1], gallery[:, 1], query[:, 2], gallery[:, 2])
    assert (distmat.cpu() - ref_d).abs().max().item() < 5e-6
    assert abs(mAP - ref_map) < 1e-4
    np.testing.assert_allclose(cmc, ref_cmc, atol=1.0 / len(query) + 1e-6)


def test_getdcnn_full_model_forward_shapes(env):
    Encoders = env[0]
    online, momentum = Encoders.getDCNN([0], "resnet50")
    x = torch.randn(3, 3, 256, 128, device="cuda")
    with torch.no_grad():
        y = online(x)
    assert y.shape == (3, 2048) and torch.isfinite(y).all()
    with pytest.raises(NotImplementedError):
        Encoders.getDCNN([0], "osnet")


def test_full_size_step_is_finite_and_bit_reproducible(env):
    """configs[1] size (ResNet-50, 256 x 3 x 256 x 128, NC = 1024): properties that do not need the oracle at this size --
    finite loss / gradients, an Adam-sized weight change, and bit-identical results when the same step is run on two
    identically seeded replicas (every reduction in the path is fixed-order: no float atomics anywhere)."""
    Encoders, _, T, _, _ = env
    from daliid_amd.losses import LossHeads, _sample_weights
    from daliid_amd.ops_eval import l2norm_rows
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(3)
    imgs = torch.randn(256, 3, 256, 128, device=dev, generator=gen)
    labels = torch.arange(16, device=dev).repeat_interleave(16).to(torch.int32)
    centers = l2norm_rows(torch.randn(1024, 2048, device=dev, generator=gen))
    proxies = l2norm_rows(torch.randn(5120, 2048, device=dev, generator=gen))
    w = _sample_weights(torch.randint(0, 6, (256,), generator=torch.Generator().manual_seed(1)), 10, 250, dev)
    results = []
    for _ in range(2):
        online, momentum = Encoders.ResNet50ReID(device=dev, seed=12), Encoders.ResNet50ReID(device=dev, seed=12)
        opt = torch.optim.Adam(online.parameters(), lr=3.5e-4, weight_decay=5e-4)
        tr = T.trainer("Synthetic", None, "resnet50", {}, 256, 128, None, False, 1, opt, 16, 16, 0.05, 0.999, 0.4, 250, online, momentum, [0], "t")
        heads = LossHeads(centers, np.arange(1024), proxies, np.repeat(np.arange(1024), 5), 0.05, 0.4, None)
        online.train(); momentum.eval()
        before = online.flat_params.clone()
        acc = torch.zeros(6, device=dev)
        stats = tr.train_step(heads, imgs, labels, w, acc)
        results.append((online.flat_params.clone(), online.flat_grads.clone(), momentum.flat_params.clone(), acc.clone(), stats.clone()))
        step = (online.flat_params - before).abs().max().item()
        assert torch.isfinite(acc).all() and torch.isfinite(online.flat_grads).all() and torch.isfinite(online.flat_params).all()
        assert 1e-4 < step < 1.1e-3, step                                     # Adam's first step moves every weight by ~lr
        assert float(acc[2]) > 0 and float(acc[4]) == 1
        del online, momentum, tr, opt
    for a, b in zip(results[0], results[1]):
        assert torch.equal(a, b)
