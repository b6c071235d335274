"""Oracle: centers/proxies builder, Adam(L2) step, EMA and one full trainer step on CPU fp32.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import numpy as np
import torch

from . import losses as olosses


def select_proxies_farthest_point(X, num_proxies=5, first=None):
    """train_encodersKIT.py:252-284 selectProxiesByTriagulation.

    Farthest-point sampling on euclidean distance: first index random
    (``np.random.choice(n)``, pinned here through ``first``), then repeatedly the point whose
    minimum distance to the chosen set is largest (``argsort(...)[-1]``: on ties the LAST index
    in stable-sort order, i.e. the highest index among the maxima).  Returns
    (indices int64 [min(num_proxies,n)], max pairwise distance among the chosen).
    Pinned by tests/golden/proxies.npz.
    """
    dist = torch.cdist(X, X, p=2.0)
    n = dist.shape[0]
    if first is None:
        first = int(np.random.choice(n))
    chosen = [first]
    running = torch.ones(n) * dist.max()
    for j in range(min(num_proxies, n) - 1):
        running = torch.minimum(running, dist[chosen[j]])
        chosen.append(int(torch.argsort(running)[-1]))
    idx = torch.tensor(chosen, dtype=torch.long)
    return idx, float(dist[idx][:, idx].max())


def build_centers_and_proxies(fvs, labels, num_proxies=5, first_picks=None):
    """train_encodersKIT.py:113-143: per class, proxies by farthest-point sampling and the center as
    the mean of the UN-normalised embeddings; both L2-normalised afterwards (no epsilon).

    fvs [N,D] fp32 CPU, labels numpy [N].  Returns (centers [NC,D], centers_labels [NC],
    proxies [<=5NC,D], proxies_labels)."""
    labels = np.asarray(labels)
    centers_labels = np.unique(labels)
    centers, proxies, plabels = [], [], []
    for ci, lab in enumerate(centers_labels):
        rows = fvs[torch.from_numpy(labels == lab)]
        first = None if first_picks is None else int(first_picks[ci])
        idx, _ = select_proxies_farthest_point(rows, num_proxies, first)
        proxies.append(rows[idx])
        plabels.append(np.array([lab] * len(idx)))
        centers.append(rows.mean(dim=0, keepdim=True))
    centers = torch.cat(centers, 0)
    centers = centers / torch.norm(centers, dim=1, keepdim=True)
    proxies = torch.cat(proxies, 0)
    proxies = proxies / torch.norm(proxies, dim=1, keepdim=True)
    return centers, centers_labels, proxies, np.concatenate(plabels)


def adam_l2_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """torch.optim.Adam (NOT AdamW) single-tensor arithmetic as the reference configures it
    (mainKIT.py:99): g += wd*p; m = b1*m+(1-b1)g; v = b2*v+(1-b2)g^2;
    p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).  In-place on p, m, v."""
    g = g + weight_decay * p
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))
    return p


def ema_update(momentum_sd, online_sd, beta):
    """train_encodersKIT.py:218-226: m = beta*m + (1-beta)*theta for EVERY state_dict key
    (BN running stats and the int64 ``num_batches_tracked`` included: the float result is cast
    back to int64 by ``load_state_dict``, i.e. truncated)."""
    out = {}
    for k, mv in momentum_sd.items():
        new = beta * mv + (1 - beta) * online_sd[k].detach()
        out[k] = new.to(mv.dtype) if new.dtype != mv.dtype else new
    return out


def l2norm_train(x):
    """train_encodersKIT.py:198: x / (|x| + 1e-9)."""
    return x / (torch.norm(x, dim=1, keepdim=True) + 1e-9)


def train_step(model_online, model_momentum, optimizer, imgs, labels, distortions, centers,
               centers_labels, proxies, proxies_labels, epoch, n_epochs, tau, beta, lambda_proxy):
    """One pass of the hot loop train_encodersKIT.py:197-231 (model in train mode).
    Returns dict(loss, center, proxy, weights_sum)."""
    feats = model_online(imgs)
    fn = l2norm_train(feats)
    loss, lc, lp, acc, amp = olosses.total_loss(fn, labels, distortions, centers, centers_labels,
                                                proxies, proxies_labels, epoch, n_epochs, tau, lambda_proxy)
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    model_online.eval()
    model_momentum.load_state_dict(ema_update(model_momentum.state_dict(), model_online.state_dict(), beta))
    model_online.train()
    wsum = sum(float(p.detach().pow(2).sum()) for p in model_online.parameters())
    return dict(loss=float(loss.detach()), center=float(lc.detach()), proxy=float(lp.detach()), weights_sum=wsum,
                acc_bal=acc, avg_max_prob=amp)
