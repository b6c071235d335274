"""GPU: two-model distance fusion (evaluateCleanATModels.py:103-160) -- blend epilogue, pooling switch, mirror."""
import numpy as np
import pytest
import torch

from oracle import evalrank as E
from oracle.resnet50_reid import ResNet50ReID as OracleNet

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from daliid_amd import ops_eval, ops_nn, Encoders, evaluateCleanATModels, getFeatures, synthetic
    return ops_eval, ops_nn, Encoders, evaluateCleanATModels, getFeatures, synthetic


@pytest.mark.parametrize("nq,ng,d", [(37, 301, 2048), (130, 1024, 768), (5, 7, 64)])
@pytest.mark.parametrize("weighted", [True, False])
def test_blend_epilogue_matches_oracle(mods, nq, ng, d, weighted):
    ops_eval = mods[0]
    g = torch.Generator().manual_seed(nq + ng + d)
    q1, g1, q2, g2 = (torch.randn(n, d, generator=g) * 3 for n in (nq, ng, nq, ng))
    m1 = (torch.rand(nq, 1, generator=g) * 20 + 5, torch.rand(ng, 1, generator=g) * 20 + 5)
    m2 = (torch.rand(nq, 1, generator=g) * 20 + 5, torch.rand(ng, 1, generator=g) * 20 + 5)
    ref = E.fused_distmat(q1, g1, q2, g2, m1 if weighted else None, m2 if weighted else None)
    dm = ops_eval.pairdist(q1.cuda(), g1.cuda(), normalize=True)
    cu = lambda m: tuple(t.cuda() for t in m)
    out = ops_eval.pairdist_blend(dm, q2.cuda(), g2.cuda(), cu(m1) if weighted else None, cu(m2) if weighted else None)
    assert out.data_ptr() == dm.data_ptr()
    # fp32-grade: bf16x3 distances are within ~2e-6 (8e-6 at tiny d) of fp32 and the blend is a convex combination
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=0, atol=1e-5 if d < 128 else 4e-6)


def test_blend_argument_errors(mods):
    ops_eval = mods[0]
    from daliid_amd import _lib
    q, g = torch.randn(4, 32, device="cuda"), torch.randn(6, 32, device="cuda")
    dm = ops_eval.pairdist(q, g, normalize=True)
    L = _lib.lib()
    rc = L.dali_pairdist_blend(_lib.ctx(q.device), _lib.stream_ptr(), _lib.ptr(q), _lib.ptr(g), 4, 6, 32, 0, 1,
                               _lib.ptr(torch.ones(4, device="cuda")), None, None, None, _lib.ptr(dm))
    assert rc != 0 and "all set or all null" in _lib.last_error()


@pytest.mark.parametrize("feature", ["both", "gap", "gmp"])
def test_head_pool_feature_modes(mods, feature):
    ops_nn = mods[1]
    g = torch.Generator().manual_seed(3)
    n, h, w, C = 3, 4, 2, 64
    x = (torch.randn(n, h, w, C, generator=g) + torch.arange(h * w).view(1, h, w, 1) * 1e-2).to(torch.bfloat16)
    xr = x.float().requires_grad_(True)
    avg, mx = xr.mean((1, 2)), torch.nn.functional.adaptive_max_pool2d(xr.permute(0, 3, 1, 2), 1).flatten(1)
    f = {"both": avg + mx, "gap": avg, "gmp": mx}[feature]
    df = torch.randn(n, C, generator=g)
    f.backward(df)
    fk, arg = ops_nn.head_pool_fwd(x.cuda(), feature)
    np.testing.assert_allclose(fk.cpu().numpy(), f.detach().numpy(), rtol=1e-5, atol=1e-5)
    dx = ops_nn.head_pool_bwd(df.cuda(), arg, (h, w), feature)
    ref = xr.grad.to(torch.bfloat16).float()
    np.testing.assert_allclose(dx.float().cpu().numpy(), ref.numpy(), rtol=1e-2, atol=1e-6)


def test_fusion_mirror_end_to_end(mods):
    """Two small nets, synthetic identities: every distmat of the mirror against the oracle restatement fed with the
    mirror's own embeddings (the nets themselves are covered by test_gpu_resnet*), and the pooling switch against the
    oracle net (eval mode, running statistics)."""
    ops_eval, ops_nn, Encoders, FUS, getFeatures, synthetic = mods
    data = synthetic.SyntheticImages(n_ids=6, per_id=5, n_cams=3, seed=11, noise=0.4).install()
    try:
        nets = [Encoders._DataParallelShim(Encoders.ResNet50ReID(layers=(1, 1, 1, 1), width=32, seed=s)).eval() for s in (1, 2)]
        H, W = 64, 32
        _, gallery, queries = data.split(1)
        # pooling switch vs the oracle net on the same weights
        onet = OracleNet(layers=(1, 1, 1, 1), width=32).eval()
        onet.load_state_dict({k[len("module."):]: v.float().cpu().contiguous() for k, v in nets[0].state_dict().items()})
        imgs = data.loader(list(queries[:, 0]), H, W)
        for pooling in ("gap", "gmp", "both"):
            mag, unit = FUS.getWeightsByMagnitude(queries, pooling, H, W, nets[0], [0])
            onet.feature = pooling
            with torch.no_grad():
                ref = onet(imgs.float().cpu())
            assert nets[0].module.feature == "both"
            rel = (mag[:, 0].cpu() - ref.norm(dim=1)).abs().max() / ref.norm(dim=1).max()
            assert rel < 3e-2, (pooling, float(rel))                                   # bf16 trunk vs fp32
            assert torch.allclose(unit.norm(dim=1), torch.ones_like(mag[:, 0]), atol=1e-5)
        res = FUS.validate(queries, gallery, nets[0], nets[1], H, W, [0], verbose=False)
        assert set(res) == {"concatenation", "clean", "distortion", "simple_ensemble", "ensemble_gap", "ensemble_gmp", "ensemble_both"}
        # oracle on the mirror's embeddings
        ex = lambda s, m: getFeatures.extractFeatures(s, H, W, m, 500, 0, keep_on_device=True, verbose=False).cpu()
        q_c, q_d, g_c, g_d = ex(queries, nets[0]), ex(queries, nets[1]), ex(gallery, nets[0]), ex(gallery, nets[1])
        qp, gp, qc, gc = queries[:, 1], gallery[:, 1], queries[:, 2], gallery[:, 2]
        def ref_metrics(dm):
            return E.eval_market1501(dm.numpy(), qp, gp, qc, gc)
        checks = {"clean": E.validate_features(q_c, g_c), "distortion": E.validate_features(q_d, g_d),
                  "concatenation": E.validate_features(torch.cat((q_c, q_d), 1), torch.cat((g_c, g_d), 1)),
                  "simple_ensemble": E.fused_distmat(q_c, g_c, q_d, g_d)}
        for pooling in ("gap", "gmp", "both"):
            mags = []
            for net in nets:
                mags.append(tuple(FUS.getWeightsByMagnitude(s, pooling, H, W, net, [0])[0].cpu() for s in (queries, gallery)))
            checks["ensemble_" + pooling] = E.fused_distmat(q_c, g_c, q_d, g_d, mags[0], mags[1])
        for name, dm in checks.items():
            cmc_ref, map_ref = ref_metrics(dm)
            cmc, mAP = res[name]
            assert abs(mAP - map_ref) < 2e-3, (name, mAP, map_ref)
            np.testing.assert_allclose(cmc[:5], cmc_ref[:5], atol=0.05)
    finally:
        synthetic.SyntheticImages.uninstall()
