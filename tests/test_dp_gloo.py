"""CPU, world_size 2, gloo: the data-parallel decomposition of the train step (daliid_amd/parallel.py) against the
single-process N-shard oracle of SURVEY 8(e): the CPU restatement run on each shard separately with shared weights
(local BatchNorm statistics), loss normalisers global, gradients summed."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _problem():
    from oracle.resnet50_reid import ResNet50ReID
    torch.manual_seed(3)
    model = ResNet50ReID(layers=(1, 1, 1, 1), width=8)
    g = torch.Generator().manual_seed(4)
    P, K, NC, D = 4, 3, 6, 256
    ids = np.array([7, 10, 13, 16])
    imgs = torch.randn(P * K, 3, 32, 16, generator=g)
    labels = torch.from_numpy(np.repeat(ids, K).astype(np.float32))
    distortion = torch.randint(0, 6, (P * K,), generator=g)
    centers = torch.nn.functional.normalize(torch.randn(NC, D, generator=g))
    clabels = np.arange(NC) * 3 + 7
    proxies = torch.nn.functional.normalize(torch.randn(3 * NC, D, generator=g))
    plabels = np.repeat(clabels, 3)
    return model, ids, K, imgs, labels, distortion, centers, clabels, proxies, plabels


def _local_pass(model, imgs, labels, distortion, centers, clabels, proxies, plabels, reduce_stats):
    """what one rank does: forward on its shard, local numerators/denominators, global normalisers, backward."""
    from oracle import losses as OL
    from oracle.trainstep import l2norm_train
    model.train()
    model.zero_grad()
    fn = l2norm_train(model(imgs))
    w = OL.distortion_weight_table(10, 250)[distortion]
    # numerators / denominators exactly as the heads define them (losses.py:77, :338)
    lc, _, _ = OL.center_loss(fn, labels, distortion, centers, clabels, 10, 250, 0.05)
    mask = (labels.reshape(-1, 1) == torch.as_tensor(clabels, dtype=torch.float32).reshape(1, -1)).float()
    c_den = (w.reshape(-1, 1) * mask.sum(1, keepdim=True)).sum()
    lp = OL.proxy_loss(fn, labels, distortion, proxies, plabels, 10, 250, 0.05)
    has_pos = (labels.reshape(-1, 1) == torch.as_tensor(plabels, dtype=torch.float32).reshape(1, -1)).any(1)
    p_den = (w * has_pos).sum()
    stats = torch.stack((lc.detach() * c_den, c_den, lp.detach() * p_den, p_den))
    stats = reduce_stats(stats.clone())
    loss = lc * c_den / stats[1] + 0.4 * lp * p_den / stats[3]          # local numerator / GLOBAL denominator
    loss.backward()
    flat = torch.cat([p.grad.flatten() for p in model.parameters()])
    return stats, flat


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from daliid_amd import parallel
    w, r, _ = parallel.init_from_env("gloo")
    assert (w, r) == (world, rank)
    model, ids, K, imgs, labels, distortion, centers, clabels, proxies, plabels = _problem()
    mine = parallel.shard_identities(ids, rank, world)
    sel = np.isin(labels.numpy(), mine)
    stats, flat = _local_pass(model, imgs[sel], labels[sel], distortion[sel], centers, clabels, proxies, plabels,
                              lambda s: parallel.allreduce_loss_stats(s))
    n = flat.numel()
    red = parallel.GradReducer(flat, [(0, n // 3), (n // 3, n // 2), (n // 2, n // 2), (n // 2, n)])
    for stage in range(4):
        red.reduce_stage(stage)
    red.finish()
    # the PK sampler's identity order is rank 0's on every rank, whatever the local RNG stream did (trainer.train under DP)
    order = parallel.broadcast_from_rank0(np.random.RandomState(100 + rank).permutation(16))
    torch.save({"stats": stats, "grads": flat, "order": order, "mine": parallel.shard_identities(order[:4], rank, world)},
               os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_equals_sharded_oracle(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r), weights_only=False) for r in range(world)]
    want = np.random.RandomState(100).permutation(16)
    for r, o in enumerate(outs):
        assert np.array_equal(o["order"], want) and np.array_equal(o["mine"], want[2 * r:2 * r + 2])
    # single-process reference: shards run one after the other with shared weights; normalisers summed first
    from daliid_amd import parallel
    model, ids, K, imgs, labels, distortion, centers, clabels, proxies, plabels = _problem()
    shard_sel = [np.isin(labels.numpy(), parallel.shard_identities(ids, r, world)) for r in range(world)]
    local_stats = [_local_pass(model, imgs[s], labels[s], distortion[s], centers, clabels, proxies, plabels, lambda t: t)[0] for s in shard_sel]
    total = sum(local_stats)
    grads = sum(_local_pass(model, imgs[s], labels[s], distortion[s], centers, clabels, proxies, plabels, lambda t: total.clone())[1] for s in shard_sel)
    for o in outs:
        np.testing.assert_allclose(o["stats"].numpy(), total.numpy(), rtol=1e-6)
        np.testing.assert_allclose(o["grads"].numpy(), grads.numpy(), rtol=1e-4, atol=1e-5 * float(grads.abs().max()))  # thread count differs: fp32 summation order
    assert torch.equal(outs[0]["grads"], outs[1]["grads"])          # every rank ends with the identical summed gradient
    # and the global loss equals the full-batch heads on the concatenated (per-shard-BN) embeddings
    c, p = total[0] / total[1], total[2] / total[3]
    assert np.isfinite(float(c)) and np.isfinite(float(p))


def test_shard_identities_validation():
    from daliid_amd import parallel
    assert list(parallel.shard_identities(np.arange(16), 3, 8)) == [6, 7]
    with pytest.raises(ValueError):
        parallel.shard_identities(np.arange(10), 0, 4)


def test_gallery_shard_bounds_cover_the_gallery_once():
    """ops_eval.shard_bounds (the gallery split of validate_sharded): contiguous, disjoint, complete, 128-row aligned starts; trailing
    shards may be empty."""
    from daliid_amd import ops_eval
    for n, world in [(100000, 8), (15913, 8), (300, 8), (168, 2), (1, 4), (128, 1)]:
        b = ops_eval.shard_bounds(n, world)
        assert len(b) == world + 1 and b[0] == 0 and b[-1] == n
        assert all(b[i] <= b[i + 1] for i in range(world)) and all(x % 128 == 0 or x == n for x in b)


# ---- epoch-level data parallelism: rank-0 running statistics, sharded epoch inference ----
def _epoch_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import types
    from daliid_amd import parallel
    from oracle.resnet50_reid import ResNet50ReID
    parallel.init_from_env("gloo")
    torch.manual_seed(3)
    model = ResNet50ReID(layers=(1, 1, 1, 1), width=8)
    g = torch.Generator().manual_seed(9)
    imgs = torch.randn(11, 3, 32, 16, generator=g)                    # 11 rows over 2 ranks: ragged slices (6 + 5)
    # the ranks' running statistics diverge in the PK loop (each forwards its own shard in train mode) ...
    model.train()
    with torch.no_grad():
        model(imgs[rank * 4:rank * 4 + 4])
    bufs = [b for n, b in model.named_buffers() if not n.endswith("num_batches_tracked")]
    nbts = [b for n, b in model.named_buffers() if n.endswith("num_batches_tracked")]
    net = types.SimpleNamespace(flat_buffers=torch.cat([b.flatten() for b in bufs]), flat_nbt=torch.stack(nbts) + rank)
    before = parallel.buffers_in_sync((net,))
    mine = net.flat_buffers.clone()
    # ... and every rank takes rank 0's before anything runs in eval mode (Encoders.py:39-40)
    parallel.sync_buffers_from_rank0((net,))
    after = parallel.buffers_in_sync((net,))
    o = 0
    for b in bufs:
        b.copy_(net.flat_buffers[o:o + b.numel()].view_as(b)); o += b.numel()
    # sharded epoch inference: every rank forwards its contiguous slice, one all-gather hands everyone all rows in dataset order
    model.eval()
    bounds = parallel.slice_bounds(imgs.shape[0], world)
    with torch.no_grad():
        local = model(imgs[bounds[rank]:bounds[rank + 1]])
        full = model(imgs)
    gathered = parallel.all_gather_rows(local, bounds)
    empty = parallel.all_gather_rows(local[:0] if rank == 1 else local[:3], [0, 3, 3])          # a rank without rows still takes part
    torch.save({"before": before, "after": after, "mine": mine, "buf": net.flat_buffers, "nbt": net.flat_nbt, "gathered": gathered,
                "full": full, "empty": empty, "local3": local[:3]}, os.path.join(out_dir, "epoch_rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_rank0_buffers_and_sharded_inference(tmp_path):
    world = 2
    mp.spawn(_epoch_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), "epoch_rank%d.pt" % r), weights_only=False) for r in range(world)]
    assert not outs[0]["before"] and not outs[1]["before"] and outs[0]["after"] and outs[1]["after"]
    assert not torch.equal(outs[0]["mine"], outs[1]["mine"])                       # they really had diverged
    for o in outs:                                                                 # rank 0's statistics and counters everywhere
        assert torch.equal(o["buf"], outs[0]["mine"]) and torch.equal(o["nbt"], outs[0]["nbt"])
    assert torch.equal(outs[0]["gathered"], outs[1]["gathered"]) and outs[0]["gathered"].shape[0] == 11
    np.testing.assert_allclose(outs[0]["gathered"].numpy(), outs[0]["full"].numpy(), rtol=1e-4, atol=1e-5)   # eval mode: split-independent
    assert torch.equal(outs[0]["empty"], outs[0]["local3"]) and torch.equal(outs[1]["empty"], outs[0]["local3"])


def test_slice_bounds():
    from daliid_amd import parallel
    assert parallel.slice_bounds(12936, 8) == [0, 1617, 3234, 4851, 6468, 8085, 9702, 11319, 12936]
    assert parallel.slice_bounds(11, 2) == [0, 6, 11] and parallel.slice_bounds(3, 4) == [0, 1, 2, 3, 3]
