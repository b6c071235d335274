"""GPU parity: loss heads, Adam and EMA kernels through the C ABI against the golden vectors produced by the
reference's own losses.py (tests/golden/losses.npz) and against torch.optim.Adam."""
import numpy as np
import pytest
import torch

from conftest import load_golden, loss_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from daliid_amd import losses
    return losses


def _cases():
    return [str(c) for c in load_golden("losses.npz")["cases"]]


@pytest.mark.parametrize("name", _cases())
def test_losses_match_reference_goldens(L, name, golden_losses):
    c = loss_case(golden_losses, name)
    epoch, n_epochs, tau = c["hyper"]
    fv = torch.from_numpy(c["fv"]).cuda().requires_grad_(True)
    labels = torch.from_numpy(c["labels"])
    dist = torch.from_numpy(c["distortion"])
    lc, acc, amp = L.BatchWeightedCenterLoss(fv, labels, dist, torch.from_numpy(c["centers"]).cuda(), c["centers_labels"],
                                             int(epoch), int(n_epochs), 0, tau=tau, gpu_index=0)
    (gc,) = torch.autograd.grad(lc, fv)
    fv2 = torch.from_numpy(c["fv"]).cuda().requires_grad_(True)
    lp = L.BatchWeightedProxyLoss(fv2, labels, dist, torch.from_numpy(c["proxies"]).cuda(), c["proxies_labels"], int(epoch),
                                  int(n_epochs), top_negs=50, tau=tau, gpu_index=0)
    (gp,) = torch.autograd.grad(lp, fv2)
    # fp32 tolerance: similarities come from the split-bf16 MFMA GEMM (~1e-6 abs), then /tau (x10..20) into exp
    assert np.isclose(lc.item(), c["center_loss"], rtol=3e-5, atol=3e-6), (lc.item(), c["center_loss"])
    assert np.isclose(lp.item(), c["proxy_loss"], rtol=3e-5, atol=3e-6), (lp.item(), c["proxy_loss"])
    assert np.isclose(acc, c["center_acc"], atol=1e-9)
    assert np.isclose(amp, c["center_avg_max_prob"], rtol=1e-4)
    if "center_grad" in c:
        gsc, gsp = np.abs(c["center_grad"]).max(), np.abs(c["proxy_grad"]).max()
        np.testing.assert_allclose(gc.cpu().numpy(), c["center_grad"], rtol=2e-4, atol=2e-5 * gsc)
        np.testing.assert_allclose(gp.cpu().numpy(), c["proxy_grad"], rtol=2e-4, atol=2e-5 * gsp)
    else:
        np.testing.assert_allclose(gc[:8].cpu().numpy(), c["center_grad_head"], rtol=2e-4, atol=2e-5 * np.abs(c["center_grad_head"]).max())
        np.testing.assert_allclose(gp[:8].cpu().numpy(), c["proxy_grad_head"], rtol=2e-4, atol=2e-5 * np.abs(c["proxy_grad_head"]).max())
        assert np.isclose(gc.double().abs().sum().item(), c["center_grad_abs_sum"], rtol=1e-4)
        assert np.isclose(gp.double().abs().sum().item(), c["proxy_grad_abs_sum"], rtol=1e-4)


def test_fused_heads_equal_separate_heads(L, golden_losses):
    c = loss_case(golden_losses, "ragged_proxies")
    epoch, n_epochs, tau = c["hyper"]
    lam = 0.4
    fv = torch.from_numpy(c["fv"]).cuda()
    labels = L._codes(torch.from_numpy(c["labels"]), fv.device)
    w = L._sample_weights(torch.from_numpy(c["distortion"]), int(epoch), int(n_epochs), fv.device)
    heads = L.LossHeads(torch.from_numpy(c["centers"]).cuda(), c["centers_labels"], torch.from_numpy(c["proxies"]).cuda(),
                        c["proxies_labels"], tau, lam)
    stats, dfn = heads(fv, labels, w)
    total, lc, lp = L.LossHeads.losses_from_stats(stats, lam)
    assert np.isclose(lc.item(), c["center_loss"], rtol=3e-5) and np.isclose(lp.item(), c["proxy_loss"], rtol=3e-5)
    ref = c["center_grad"] + lam * c["proxy_grad"]
    np.testing.assert_allclose(dfn.cpu().numpy(), ref, rtol=2e-4, atol=2e-5 * np.abs(ref).max())


def test_cosine_schedule_table(L):
    z = load_golden("schedule.npz")
    for (a, b), row in zip(z["t"], z["values"]):
        got = [L.getValueFromCosineSchedule(int(a), int(b), n_min=m, n_max=1.0) for m in z["mins"]]
        np.testing.assert_allclose(got, row, rtol=0, atol=1e-15)
    assert np.isclose(L.getACCBal(z["acc_pred"], z["acc_gt"]), z["acc_bal"], atol=1e-12)


def test_fused_adam_and_ema_match_torch():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from daliid_amd import Encoders, optim
    net = Encoders.ResNet50ReID(layers=(1, 1, 1, 1), width=32, seed=3)
    mom = Encoders.ResNet50ReID(layers=(1, 1, 1, 1), width=32, seed=3)
    ref_p = net.flat_params.detach().cpu().clone().requires_grad_(True)
    topt = torch.optim.Adam([ref_p], lr=3.5e-4, weight_decay=5e-4)
    drv = torch.optim.Adam(net.parameters(), lr=3.5e-4, weight_decay=5e-4)       # what mainKIT.py:99 builds
    fused = optim.FusedAdam.from_torch(drv, net)
    g = torch.Generator().manual_seed(0)
    mom_ref = mom.flat_params.detach().cpu().clone()
    for step in range(1, 6):
        grad = torch.randn(ref_p.shape, generator=g) * 1e-2
        grad[ref_p.detach() == 0] = 0            # alignment padding of the flat buffer carries no gradient
        if step == 3:                            # the driver changes lr / wd per epoch through param_groups
            for opt in (topt, drv):
                opt.param_groups[0]["lr"] = 3.5e-5
        ref_p.grad = grad.clone()
        topt.step()
        net.flat_grads.copy_(grad)
        fused.step()
        optim.ema_update(mom, net, 0.999)
        mom_ref = 0.999 * mom_ref + 0.001 * ref_p.detach()
        np.testing.assert_allclose(net.flat_params.cpu().numpy(), ref_p.detach().numpy(), rtol=2e-6, atol=1e-8)
        assert np.isclose(fused.weights_sqsum.item(), float(ref_p.detach().double().pow(2).sum()), rtol=1e-5)
    np.testing.assert_allclose(mom.flat_params.cpu().numpy(), mom_ref.numpy(), rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("name", [str(c) for c in load_golden("triplet.npz")["cases"]])
def test_triplet_head_matches_reference_golden(L, name):
    """Optional head BatchWeightedSoftmaxTripletLoss (losses.py:607-654) vs the reference's own outputs."""
    z = load_golden("triplet.npz")
    epoch, n_epochs, tau = z[name + "/hyper"]
    fv = torch.from_numpy(z[name + "/fv"]).cuda().requires_grad_(True)
    loss = L.BatchWeightedSoftmaxTripletLoss(fv, torch.from_numpy(z[name + "/labels"]), torch.from_numpy(z[name + "/distortion"]),
                                             int(epoch), int(n_epochs), tau=tau, gpu_index=0)
    loss.backward()
    g_ref = z[name + "/grad"]
    assert np.isclose(loss.item(), float(z[name + "/loss"]), rtol=2e-5, atol=2e-6)           # similarities by split-bf16 MFMA (~1e-6 abs)
    np.testing.assert_allclose(fv.grad.cpu().numpy(), g_ref, rtol=2e-3, atol=2e-4 * float(np.abs(g_ref).max()))


def test_triplet_head_single_identity_raises(L):
    from daliid_amd._lib import DaliError
    fv = torch.nn.functional.normalize(torch.randn(8, 32, device="cuda"))
    with pytest.raises(DaliError):
        L.BatchWeightedSoftmaxTripletLoss(fv, torch.ones(8), torch.zeros(8, dtype=torch.long), 1, 10)
