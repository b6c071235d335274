"""Oracle vs the reference's own losses.py outputs (tests/golden/losses.npz, schedule.npz)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, loss_case
from oracle import losses as O


def _cases():
    return [str(c) for c in load_golden("losses.npz")["cases"]]


@pytest.mark.parametrize("name", _cases())
def test_losses_match_reference(name, golden_losses):
    c = loss_case(golden_losses, name)
    epoch, n_epochs, tau = c["hyper"]
    fv = torch.from_numpy(c["fv"]).requires_grad_(True)
    labels = torch.from_numpy(c["labels"])
    dist = torch.from_numpy(c["distortion"])
    lc, acc, amp = O.center_loss(fv, labels, dist, torch.from_numpy(c["centers"]), c["centers_labels"],
                                 int(epoch), int(n_epochs), tau)
    (gc,) = torch.autograd.grad(lc, fv)
    fv2 = torch.from_numpy(c["fv"]).requires_grad_(True)
    lp = O.proxy_loss(fv2, labels, dist, torch.from_numpy(c["proxies"]), c["proxies_labels"],
                      int(epoch), int(n_epochs), tau)
    (gp,) = torch.autograd.grad(lp, fv2)
    assert np.isclose(lc.item(), c["center_loss"], rtol=2e-6, atol=1e-6)
    assert np.isclose(lp.item(), c["proxy_loss"], rtol=2e-6, atol=1e-6)
    assert np.isclose(acc, c["center_acc"], atol=1e-9)
    assert np.isclose(amp, c["center_avg_max_prob"], rtol=1e-5)
    if "center_grad" in c:
        np.testing.assert_allclose(gc.numpy(), c["center_grad"], rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(gp.numpy(), c["proxy_grad"], rtol=1e-4, atol=2e-6)
    else:
        np.testing.assert_allclose(gc[:8].numpy(), c["center_grad_head"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(gp[:8].numpy(), c["proxy_grad_head"], rtol=1e-4, atol=1e-6)
        assert np.isclose(gc.double().abs().sum().item(), c["center_grad_abs_sum"], rtol=1e-5)
        assert np.isclose(gp.double().abs().sum().item(), c["proxy_grad_abs_sum"], rtol=1e-5)


def test_schedule_and_accbal():
    z = load_golden("schedule.npz")
    for (a, b), row in zip(z["t"], z["values"]):
        got = [O.cosine_schedule(int(a), int(b), n_min=m, n_max=1.0) for m in z["mins"]]
        np.testing.assert_allclose(got, row, rtol=0, atol=1e-15)
    assert np.isclose(O.acc_balanced(z["acc_pred"], z["acc_gt"]), z["acc_bal"], atol=1e-12)


@pytest.mark.parametrize("name", [str(c) for c in load_golden("triplet.npz")["cases"]])
def test_triplet_head_matches_reference(name):
    z = load_golden("triplet.npz")
    epoch, n_epochs, tau = z[name + "/hyper"]
    fv = torch.from_numpy(z[name + "/fv"]).requires_grad_(True)
    loss = O.softmax_triplet_loss(fv, torch.from_numpy(z[name + "/labels"]), torch.from_numpy(z[name + "/distortion"]),
                                  int(epoch), int(n_epochs), tau)
    (g,) = torch.autograd.grad(loss, fv)
    assert np.isclose(loss.item(), float(z[name + "/loss"]), rtol=3e-6, atol=1e-6)
    np.testing.assert_allclose(g.numpy(), z[name + "/grad"], rtol=1e-4, atol=2e-6 * float(np.abs(z[name + "/grad"]).max()) + 1e-9)
