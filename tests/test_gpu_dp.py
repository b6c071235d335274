"""GPU, 2 processes sharing the card, gloo backend (RCCL refuses two ranks on one device; the driver exercises RCCL on
the 8-GPU node): the data-parallel train step of the trainer mirror -- stage-bucketed gradient all-reduce on a side
stream, global loss normalisers, redundant Adam/EMA -- against a single-process emulation of the same two shards."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _setup(rank_seed=0):
    from daliid_amd import Encoders
    from daliid_amd.losses import LossHeads, _sample_weights
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    online = Encoders._DataParallelShim(Encoders.ResNet50ReID(layers=(1, 1, 1, 1), width=32, seed=7, device=dev))
    momentum = Encoders._DataParallelShim(Encoders.ResNet50ReID(layers=(1, 1, 1, 1), width=32, seed=7, device=dev))
    NC, D, P, K = 8, 1024, 4, 4
    centers = torch.nn.functional.normalize(torch.randn(NC, D, generator=g)).to(dev)
    proxies = torch.nn.functional.normalize(torch.randn(3 * NC, D, generator=g)).to(dev)
    imgs = torch.randn(P * K, 3, 64, 32, generator=g)
    ids = np.arange(P)
    labels = torch.arange(P).repeat_interleave(K).float()
    dist_lv = torch.randint(0, 6, (P * K,), generator=g)
    return dev, online, momentum, centers, proxies, NC, imgs, ids, labels, dist_lv, _sample_weights, LossHeads


def _emulated_step(net, mom, heads, adam, shards, beta):
    """One data-parallel step in ONE process: every shard through the same net (local BatchNorm statistics), loss normalisers summed over
    the shards first, gradients of every shard with the GLOBAL denominators summed, one Adam step, one EMA update.  The net keeps the
    running statistics of shard 0's forward only (replica 0's survive under nn.DataParallel, Encoders.py:39-40).
    shards: [(images, label codes, sample weights)] -> (global statistics [4], summed gradient)"""
    from daliid_amd import losses as L
    from daliid_amd import ops_eval, optim
    tau, lam = heads.tau, heads.lam
    nbt_before, buf_before = net.flat_nbt.clone(), net.flat_buffers.clone()
    fwd = []
    for x, lab, w in shards:                                                      # pass 1: local numerators / denominators
        emb = net._run_forward(x, True)
        fn = ops_eval.l2norm_rows(emb, 1e-9)
        Sc = ops_eval.pairdist(fn, heads.centers, metric="dot"); Sp = ops_eval.pairdist(fn, heads.proxies, metric="dot")
        _, sc = L.center_fwd(Sc, lab, heads.clabels, w, tau)
        _, sp, _, _, _ = L.proxy_fwd(Sp, lab, heads.plabels, w, tau)
        fwd.append(torch.cat((sc, sp)))
    total = sum(fwd[1:], fwd[0])
    net.flat_nbt.copy_(nbt_before); net.flat_buffers.copy_(buf_before)            # rank 0's BN buffers see one forward per step
    gsum = torch.zeros_like(net.flat_grads)
    keep = None
    for i, (x, lab, w) in enumerate(shards):                                      # pass 2: gradients with the GLOBAL denominators
        emb = net._run_forward(x, True)
        if i == 0:
            keep = (net.flat_nbt.clone(), net.flat_buffers.clone())
        fn = ops_eval.l2norm_rows(emb, 1e-9)
        Sc = ops_eval.pairdist(fn, heads.centers, metric="dot"); Sp = ops_eval.pairdist(fn, heads.proxies, metric="dot")
        dS = L.center_bwd(Sc, lab, heads.clabels, w, tau, total[1:2])
        dfn = ops_eval.pairdist(dS, heads.centers_t, metric="dot")
        _, _, sel_idx, sel_coef, _ = L.proxy_fwd(Sp, lab, heads.plabels, w, tau)
        L.proxy_bwd(sel_idx, sel_coef, heads.proxies, total[3:4], gscale=lam, out=dfn, accumulate=True)
        d_emb = ops_eval.l2norm_rows_bwd(emb, dfn, 1e-9)
        for s in range(4):
            net._backward_stage(d_emb, s)
        gsum += net.flat_grads
    net.flat_nbt.copy_(keep[0]); net.flat_buffers.copy_(keep[1])                  # shard 0's running statistics
    net.flat_grads.copy_(gsum)
    adam.step()
    optim.ema_update(mom, net, beta)
    return total, gsum


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    from daliid_amd import parallel
    from daliid_amd.losses import _codes
    from daliid_amd.train_encodersKIT import trainer
    torch.cuda.set_device(0)
    parallel.init_from_env("gloo")
    dev, online, momentum, centers, proxies, NC, imgs, ids, labels, dist_lv, sample_w, LossHeads = _setup()
    pg = dist.group.WORLD
    opt = torch.optim.Adam(online.parameters(), lr=3.5e-4, weight_decay=5e-4)
    tr = trainer("Synthetic", None, "resnet50", {}, 64, 32, None, False, 0, opt, 4, 4, 0.05, 0.9, 0.4, 250, online, momentum, [0], "t", process_group=pg)
    heads = LossHeads(centers, np.arange(NC), proxies, np.repeat(np.arange(NC), 3), 0.05, 0.4, pg)
    mine = parallel.shard_identities(ids, rank, world)
    sel = torch.from_numpy(np.isin(labels.numpy(), mine))
    online.train()
    acc = torch.zeros(6, device=dev)
    stats = None
    for step in range(2):
        stats = tr.train_step(heads, imgs[sel].to(dev), _codes(labels[sel], dev), sample_w(dist_lv[sel], 10, 250, dev), acc)
    torch.cuda.synchronize()
    torch.save({"params": online.module.flat_params.cpu(), "mom": momentum.module.flat_params.cpu(), "stats": stats.cpu(),
                "grads": online.module.flat_grads.cpu()}, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_train_steps_match_sharded_emulation(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r)) for r in range(world)]
    # every rank ends with identical weights / momentum weights / summed gradients / global loss statistics
    for k in ("params", "mom", "stats", "grads"):
        assert torch.equal(outs[0][k], outs[1][k]), k
    # single-process emulation: both shards through the same net, global normalisers, gradients summed, one Adam step
    from daliid_amd import optim, ops_eval, parallel
    from daliid_amd.losses import _codes
    dev, online, momentum, centers, proxies, NC, imgs, ids, labels, dist_lv, sample_w, LossHeads = _setup()
    net, mom = online.module, momentum.module
    heads = LossHeads(centers, np.arange(NC), proxies, np.repeat(np.arange(NC), 3), 0.05, 0.4, None)
    adam = optim.FusedAdam(net, lr=3.5e-4, weight_decay=5e-4)
    net.train()
    for step in range(2):
        shards = []
        for r in range(world):
            sel = torch.from_numpy(np.isin(labels.numpy(), parallel.shard_identities(ids, r, world)))
            shards.append((imgs[sel].to(dev), _codes(labels[sel], dev), sample_w(dist_lv[sel], 10, 250, dev)))
        total, gsum = _emulated_step(net, mom, heads, adam, shards, 0.9)
    np.testing.assert_allclose(outs[0]["stats"].numpy(), total.cpu().numpy(), rtol=1e-5)
    np.testing.assert_allclose(outs[0]["grads"].numpy(), gsum.cpu().numpy(), rtol=2e-3, atol=2e-5 * float(gsum.abs().max()))
    d = (outs[0]["params"] - net.flat_params.cpu()).abs().max().item()
    assert d < 2e-4, d            # Adam turns rounding-level gradient differences into <= lr-sized parameter differences


def _rccl_worker(rank, world, port, out_dir, comm="torch"):
    """backend "nccl" (= RCCL): one rank per GPU.  A one-GPU box can only host world size 1, which still runs every RCCL call
    of the step (communicator init, the 4-float loss-statistics all-reduce, the per-stage gradient all-reduces on the side
    stream with async work handles)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY="0", DALIID_COMM=comm)
    import torch.distributed as dist
    from daliid_amd.losses import _codes
    from daliid_amd.train_encodersKIT import trainer
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", device_id=torch.device("cuda", rank))
    dev, online, momentum, centers, proxies, NC, imgs, ids, labels, dist_lv, sample_w, LossHeads = _setup()
    pg = dist.group.WORLD
    opt = torch.optim.Adam(online.parameters(), lr=3.5e-4, weight_decay=5e-4)
    tr = trainer("Synthetic", None, "resnet50", {}, 64, 32, None, False, 0, opt, 4, 4, 0.05, 0.9, 0.4, 250, online, momentum, [rank], "t", process_group=pg)
    heads = LossHeads(centers, np.arange(NC), proxies, np.repeat(np.arange(NC), 3), 0.05, 0.4, pg)
    online.train()
    acc = torch.zeros(6, device=dev)
    for step in range(2):
        stats = tr.train_step(heads, imgs.to(dev), _codes(labels, dev), sample_w(dist_lv, 10, 250, dev), acc)
    torch.cuda.synchronize()
    assert tr._dp is not None and dist.get_backend() == "nccl"
    assert (tr._dp.abi_ctx is not None) == (comm == "abi")
    torch.save({"params": online.module.flat_params.cpu(), "stats": stats.cpu(), "grads": online.module.flat_grads.cpu()},
               os.path.join(out_dir, "rccl_rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("comm", ["torch", "abi"])
def test_rccl_backend_runs_the_step_and_changes_nothing_at_world_size_1(tmp_path, comm):
    """comm = "torch": gradient buckets through torch.distributed (backend nccl = RCCL); comm = "abi": through the library's own RCCL
    communicator (dali_ctx_comm_init / dali_allreduce_bucket, include/daliid.h), torch.distributed only carrying the unique id."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    world = min(torch.cuda.device_count(), 2)
    mp.spawn(_rccl_worker, args=(world, _free_port(), str(tmp_path), comm), nprocs=world, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), "rccl_rank%d.pt" % r)) for r in range(world)]
    for o in outs:
        assert torch.isfinite(o["params"]).all() and torch.isfinite(o["grads"]).all()
    if world == 1:
        # a 1-rank all-reduce is the identity: same two steps without a process group give the same bits
        from daliid_amd.losses import _codes
        from daliid_amd.train_encodersKIT import trainer
        dev, online, momentum, centers, proxies, NC, imgs, ids, labels, dist_lv, sample_w, LossHeads = _setup()
        opt = torch.optim.Adam(online.parameters(), lr=3.5e-4, weight_decay=5e-4)
        tr = trainer("Synthetic", None, "resnet50", {}, 64, 32, None, False, 0, opt, 4, 4, 0.05, 0.9, 0.4, 250, online, momentum, [0], "t")
        heads = LossHeads(centers, np.arange(NC), proxies, np.repeat(np.arange(NC), 3), 0.05, 0.4, None)
        online.train()
        acc = torch.zeros(6, device=dev)
        for step in range(2):
            stats = tr.train_step(heads, imgs.to(dev), _codes(labels, dev), sample_w(dist_lv, 10, 250, dev), acc)
        assert torch.equal(outs[0]["params"], online.module.flat_params.cpu())
        assert torch.equal(outs[0]["grads"], online.module.flat_grads.cpu())
    else:
        for k in ("params", "grads", "stats"):
            assert torch.equal(outs[0][k], outs[1][k]), k


# ---- one full trainer.train epoch under data parallelism: rank-0 running statistics, sharded epoch inference ----
_EPOCH = dict(n_ids=8, per_id=6, P=4, K=4, H=64, W=32, beta=0.9, seed=31, epochs=2)


def _epoch_setup():
    from daliid_amd import Encoders, synthetic
    c = _EPOCH
    data = synthetic.SyntheticImages(n_ids=c["n_ids"], per_id=c["per_id"], n_cams=3, seed=5, noise=0.4).install()
    online = Encoders._DataParallelShim(Encoders.ResNet50ReID(layers=(1, 1, 1, 1), width=32, seed=7))
    momentum = Encoders._DataParallelShim(Encoders.ResNet50ReID(layers=(1, 1, 1, 1), width=32, seed=7))
    momentum.load_state_dict(online.state_dict())
    train = data.records
    return data, online.eval(), momentum.eval(), train, np.int32(train[:, 1])


def _epoch_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import contextlib, io
    import torch.distributed as dist
    from daliid_amd import parallel
    from daliid_amd.train_encodersKIT import trainer
    torch.cuda.set_device(0)
    parallel.init_from_env("gloo")
    c = _EPOCH
    data, online, momentum, train, labels = _epoch_setup()
    pg = dist.group.WORLD
    opt = torch.optim.Adam(online.parameters(), lr=3.5e-4, weight_decay=5e-4)
    tr = trainer("Synthetic", train, "resnet50", {}, c["H"], c["W"], None, False, 0, opt, c["P"], c["K"], 0.05, c["beta"], 0.4, 250, online,
                 momentum, [0], "t", process_group=pg)
    np.random.seed(c["seed"])
    out = {}
    with contextlib.redirect_stdout(io.StringIO()):
        for epoch in range(1, c["epochs"] + 1):
            tr.train(train, labels, 1, epoch)
            out["centers%d" % epoch] = tr.last_targets.centers.cpu()
            out["proxies%d" % epoch] = tr.last_targets.proxies.cpu()
            out["rows%d" % epoch] = tr.last_inference_rows
        feats = tr.extract_train_features(train)                      # eval-mode features after the last epoch's buffer hand-over
    assert parallel.buffers_in_sync((online, momentum), pg)
    torch.cuda.synchronize()
    on, mo = online.module, momentum.module
    out.update(params=on.flat_params.cpu(), mom=mo.flat_params.cpu(), buf=on.flat_buffers.cpu(), nbt=on.flat_nbt.cpu(),
               mom_buf=mo.flat_buffers.cpu(), mom_nbt=mo.flat_nbt.cpu(), feats=feats.cpu(), steps=tr.last_epoch_stats["steps"])
    # a model whose statistics differ between ranks is refused, on every rank, before any collective of the data path
    on.flat_buffers[:4] += float(rank)
    ok = parallel.buffers_in_sync((online,), pg)
    out["diverged_detected"] = not ok
    torch.save(out, os.path.join(out_dir, "epoch_rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_epoch_keeps_rank0_statistics_and_shards_the_inference(tmp_path):
    """trainer.train under data parallelism, two epochs: (1) all ranks end every epoch with bit-identical BatchNorm running statistics and
    counters (online and momentum net) -- rank 0's, as under nn.DataParallel (Encoders.py:39-40); (2) the epoch inference is sharded
    (each rank forwards half of the train set, getFeatures.py:56-67 / train_encodersKIT.py:104-110) and every rank builds bit-identical
    centers and proxies, in the second epoch too (i.e. from the synchronised statistics, with rank 0's first picks); (3) all of it matches
    a single-process emulation that runs both shards through one net and keeps shard 0's statistics."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    world, c = 2, _EPOCH
    mp.spawn(_epoch_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), "epoch_rank%d.pt" % r), weights_only=False) for r in range(world)]
    keys = ["params", "mom", "buf", "nbt", "mom_buf", "mom_nbt", "feats"] + ["%s%d" % (k, e) for k in ("centers", "proxies") for e in (1, 2)]
    for k in keys:
        assert torch.equal(outs[0][k], outs[1][k]), k
    n_train = c["n_ids"] * c["per_id"]
    assert outs[0]["rows1"] + outs[1]["rows1"] == n_train and abs(outs[0]["rows1"] - outs[1]["rows1"]) <= 1      # sharded, not replicated
    assert outs[0]["steps"] == c["n_ids"] // c["P"] and int(outs[0]["nbt"][0]) == c["epochs"] * outs[0]["steps"]
    assert all(o["diverged_detected"] for o in outs)
    # ---- single-process emulation: virtual ranks with their own numpy streams, one net, shard 0's statistics kept ----
    import contextlib, io
    from daliid_amd import optim, parallel
    from daliid_amd import train_encodersKIT as T
    from daliid_amd.getFeatures import extractFeatures
    from daliid_amd.losses import LossHeads, _codes, _sample_weights
    data, online, momentum, train, labels = _epoch_setup()
    try:
        dev = torch.device("cuda", 0)
        net, mom = online.module, momentum.module
        adam = optim.FusedAdam(net, lr=3.5e-4, weight_decay=5e-4)
        np.random.seed(c["seed"])
        states = [np.random.get_state() for _ in range(world)]

        def per_rank(fn):
            """fn() under every virtual rank's numpy stream; -> the results, rank 0's first"""
            res = []
            for r in range(world):
                np.random.set_state(states[r])
                res.append(fn(r))
                states[r] = np.random.get_state()
            return res
        emu = {}
        with contextlib.redirect_stdout(io.StringIO()):
            for epoch in range(1, c["epochs"] + 1):
                dsets = per_rank(lambda r: T.samplePKBatches("Synthetic", train, labels, c["H"], c["W"], None, 0, K=c["K"]))
                for d in dsets:
                    d.labels_set = dsets[0].labels_set                                        # broadcast from rank 0
                online.eval()
                b = parallel.slice_bounds(len(train), world)
                fvs = torch.cat([extractFeatures(train[b[r]:b[r + 1]], c["H"], c["W"], online, 500, 0, keep_on_device=True) for r in range(world)], 0)
                targets = per_rank(lambda r: T.build_centers_and_proxies(fvs, labels, 5))[0]  # rank 0's first picks
                heads = LossHeads(targets[0], targets[1], targets[2], targets[3], 0.05, 0.4, None)
                emu["centers%d" % epoch], emu["proxies%d" % epoch] = heads.centers.cpu(), heads.proxies.cpu()
                online.train()
                order = per_rank(lambda r: np.random.permutation(len(dsets[0])))[0]
                for bi in range(len(order) // c["P"]):
                    ids = order[bi * c["P"]:(bi + 1) * c["P"]]
                    parts = per_rank(lambda r: [dsets[r][i] for i in parallel.shard_identities(ids, r, world)])
                    shards = []
                    for pr in parts:
                        x = torch.cat([q[0] for q in pr], 0).to(dev)
                        lab = _codes(torch.cat([q[1] for q in pr], 0), dev)
                        w = _sample_weights(torch.from_numpy(np.concatenate([q[2] for q in pr])), epoch, 250, dev)
                        shards.append((x, lab, w))
                    _emulated_step(net, mom, heads, adam, shards, c["beta"])
                online.eval()
            feats = torch.cat([extractFeatures(train[b[r]:b[r + 1]], c["H"], c["W"], online, 500, 0, keep_on_device=True) for r in range(world)], 0)
    finally:
        from daliid_amd import synthetic
        synthetic.SyntheticImages.uninstall()
    o = outs[0]
    # epoch 1's targets come from untouched, identical models: the same bits; later quantities inherit the step test's tolerances
    assert torch.equal(o["centers1"], emu["centers1"]) and torch.equal(o["proxies1"], emu["proxies1"])
    assert torch.equal(o["nbt"], net.flat_nbt.cpu()) and torch.equal(o["mom_nbt"], mom.flat_nbt.cpu())
    assert (o["params"] - net.flat_params.cpu()).abs().max().item() < 4e-4
    np.testing.assert_allclose(o["buf"].numpy(), net.flat_buffers.cpu().numpy(), rtol=2e-2, atol=2e-4)
    np.testing.assert_allclose(o["mom_buf"].numpy(), mom.flat_buffers.cpu().numpy(), rtol=2e-2, atol=2e-4)
    cos = torch.nn.functional.cosine_similarity(o["centers2"], emu["centers2"], dim=1)
    assert cos.min().item() > 0.999, cos.min().item()
    rel = (o["feats"] - feats.cpu()).norm() / feats.cpu().norm()
    assert rel.item() < 2e-2, rel.item()
