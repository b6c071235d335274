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
    from daliid_amd import losses as L
    for step in range(2):
        shards = []
        for r in range(world):
            sel = torch.from_numpy(np.isin(labels.numpy(), parallel.shard_identities(ids, r, world)))
            shards.append((imgs[sel].to(dev), _codes(labels[sel], dev), sample_w(dist_lv[sel], 10, 250, dev)))
        # pass 1: local numerators / denominators of both shards -> global statistics
        fwd = []
        nbt_before = net.flat_nbt.clone(); buf_before = net.flat_buffers.clone()
        for x, lab, w in shards:
            emb = net._run_forward(x, True)
            fn = ops_eval.l2norm_rows(emb, 1e-9)
            Sc = ops_eval.pairdist(fn, heads.centers, metric="dot"); Sp = ops_eval.pairdist(fn, heads.proxies, metric="dot")
            _, sc = L.center_fwd(Sc, lab, heads.clabels, w, 0.05)
            _, sp, _, _, _ = L.proxy_fwd(Sp, lab, heads.plabels, w, 0.05)
            fwd.append(torch.cat((sc, sp)))
        total = fwd[0] + fwd[1]
        net.flat_nbt.copy_(nbt_before); net.flat_buffers.copy_(buf_before)       # rank 0's BN buffers see one forward per step
        # pass 2: gradients of each shard with the GLOBAL denominators, summed
        gsum = torch.zeros_like(net.flat_grads)
        for i, (x, lab, w) in enumerate(shards):
            if i == 1:
                keep_nbt, keep_buf = net.flat_nbt.clone(), net.flat_buffers.clone()
            emb = net._run_forward(x, True)
            fn = ops_eval.l2norm_rows(emb, 1e-9)
            Sc = ops_eval.pairdist(fn, heads.centers, metric="dot"); Sp = ops_eval.pairdist(fn, heads.proxies, metric="dot")
            dS = L.center_bwd(Sc, lab, heads.clabels, w, 0.05, total[1:2])
            dfn = ops_eval.pairdist(dS, heads.centers_t, metric="dot")
            _, _, sel_idx, sel_coef, _ = L.proxy_fwd(Sp, lab, heads.plabels, w, 0.05)
            L.proxy_bwd(sel_idx, sel_coef, heads.proxies, total[3:4], gscale=0.4, out=dfn, accumulate=True)
            d_emb = ops_eval.l2norm_rows_bwd(emb, dfn, 1e-9)
            for s in range(4):
                net._backward_stage(d_emb, s)
            gsum += net.flat_grads
            if i == 1:
                net.flat_nbt.copy_(keep_nbt); net.flat_buffers.copy_(keep_buf)   # keep rank 0's running statistics
        net.flat_grads.copy_(gsum)
        adam.step()
        optim.ema_update(mom, net, 0.9)
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
