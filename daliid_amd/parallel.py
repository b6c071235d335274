"""Data parallelism for the train step: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" on CPU for the tests).  The reference uses single-process ``nn.DataParallel`` (Encoders.py:39-40): a
per-step weight broadcast, a gather of the outputs and a reduce of all gradients to GPU 0.  Here every rank holds
the weights, runs the step on its shard of the PK batch (identities split across ranks, local BatchNorm statistics =
DataParallel semantics, Encoders.py:88-89) and exchanges exactly

  1. one all-reduce(SUM) of 4 floats: numerator / denominator of the two loss heads (their normalisers are global over
     the batch: losses.py:77, :338), issued between the heads' forward and backward;
  2. one all-reduce(SUM, not mean) of the flat gradient buffer per backward stage (4 buckets: layer4+neck 15.0 M,
     layer3 7.1 M, layer2 1.2 M, layer1+stem 0.2 M floats), each launched on a side stream as soon as its stage is
     enqueued so it overlaps the remaining backward.  xGMI is a point-to-point mesh: few large buckets beat many small.
Adam and the EMA then run redundantly on identical data on every rank: WEIGHTS are never broadcast.

BatchNorm running statistics are the one piece of state that is rank-local during the PK loop (each rank's forward folds in
its own shard's batch statistics).  Under ``nn.DataParallel`` only replica 0's buffers survive a forward (replica 0 shares
storage with the wrapped module, Encoders.py:39-40), so after the loop -- before anything runs in eval mode -- every rank
takes rank 0's ``flat_buffers`` / ``flat_nbt`` of the online AND the momentum net (``sync_buffers_from_rank0``; rank 0's
momentum buffers are the EMA of rank 0's online buffers, which is what the reference's EMA over state_dict sees).  Train mode
never reads the running statistics, so one broadcast per epoch gives the same bits as one per step.

Epoch inference (``extractFeatures`` over the whole train set, train_encodersKIT.py:104-110, which DataParallel splits over the
GPUs batch by batch, getFeatures.py:56-67) is sharded: every rank extracts one contiguous slice and ONE all-gather hands every rank
the full ``[N, D]`` fp32 matrix (``extract_features_sharded``), so that all ranks derive identical centers / proxies.

The gradient buckets go either through ``torch.distributed`` (default) or, with ``DALIID_COMM=abi``, through the library's own RCCL
communicator (``dali_allreduce_bucket``, include/daliid.h): torch.distributed then only carries the 128-byte unique id to the ranks.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run sets them)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return world, rank, local


def shard_identities(ids, rank, world):
    """Split the P identities of a PK batch across ranks (contiguous blocks; every rank gets P/world ids x K images).
    Any row split is valid because both heads are sums of per-row terms against fixed centers/proxies."""
    n = len(ids)
    if n % world != 0:
        raise ValueError("P=%d identities do not divide over %d ranks" % (n, world))
    per = n // world
    return ids[rank * per:(rank + 1) * per]


def broadcast_from_rank0(obj, group=None):
    """Rank 0's Python object on every rank (the shuffled identity list and the per-iteration batch order of the PK sampler:
    every rank must walk the same identities, whatever its own RNG stream did while loading images)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return obj
    box = [obj if dist.get_rank(group) == 0 else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return box[0]


def allreduce_loss_stats(stats, group=None):
    """stats [4] = local (center_num, center_den, proxy_num, proxy_den) -> global sums, in place."""
    if group is not None or (dist.is_initialized() and dist.get_world_size() > 1):
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
    return stats


def use_abi_comm():
    return os.environ.get("DALIID_COMM", "torch") == "abi"


def init_abi_comm(device, group=None):
    """Create this rank's RCCL communicator inside the library's context (once per process): rank 0 draws the unique id, the process
    group's store carries it to the others.  -> the ctx handle."""
    import ctypes
    from . import _lib
    c = _lib.ctx(device)
    # one communicator per (device, process group): after destroy_process_group() + init_process_group() in the same process the group
    # object is a new one and the old communicator (other peers, maybe another world size) must not be reused
    key = (device.index, id(group if group is not None else dist.group.WORLD), dist.get_world_size(group), dist.get_rank(group))
    if getattr(init_abi_comm, "_done", None) == key:
        return c
    if getattr(init_abi_comm, "_done", None) is not None and init_abi_comm._done[0] == device.index:
        _lib.check(_lib.lib().dali_ctx_comm_destroy(c), "dali_ctx_comm_destroy")
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    box = [None]
    if rank == 0:
        buf = ctypes.create_string_buffer(128)
        _lib.check(_lib.lib().dali_comm_unique_id(buf), "dali_comm_unique_id")
        box[0] = bytes(buf.raw)
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    idbuf = ctypes.create_string_buffer(box[0], 128)
    _lib.check(_lib.lib().dali_ctx_comm_init(c, idbuf, rank, world), "dali_ctx_comm_init")
    init_abi_comm._done = key
    return c


class GradReducer:
    """SUM all-reduce of the flat gradient buffer in per-stage buckets.  On CUDA the collectives run on a side stream
    behind an event recorded right after the stage's kernels were enqueued (overlap with the next stage's backward);
    on CPU tensors (gloo tests) they run synchronously."""

    def __init__(self, flat_grads, ranges, group=None):
        self.flat, self.ranges, self.group = flat_grads, list(ranges), group
        self.cuda = flat_grads.is_cuda
        self.works = []
        self.abi_ctx = None
        if self.cuda:
            self.stream = torch.cuda.Stream(device=flat_grads.device)
            self.ready = torch.cuda.Event()
            if use_abi_comm():
                self.abi_ctx = init_abi_comm(flat_grads.device, group)

    def reduce_stage(self, stage):
        b, e = self.ranges[stage]
        if e <= b:
            return
        if not self.cuda:
            dist.all_reduce(self.flat[b:e], op=dist.ReduceOp.SUM, group=self.group)
            return
        self.ready.record(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(self.ready)
            if self.abi_ctx is not None:
                from . import _lib
                _lib.check(_lib.lib().dali_allreduce_bucket(self.abi_ctx, _lib.c_void_p(self.stream.cuda_stream), _lib.ptr(self.flat[b:e]), e - b),
                           "dali_allreduce_bucket")
            else:
                self.works.append(dist.all_reduce(self.flat[b:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        for w in self.works:
            w.wait()
        self.works = []
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.stream)


def slice_bounds(n, world):
    """Contiguous, near-equal slices of ``n`` rows over ``world`` ranks: [0, ..., n] with world + 1 entries."""
    per, extra = divmod(n, world)
    b = [0]
    for r in range(world):
        b.append(b[-1] + per + (1 if r < extra else 0))
    return b


def sync_buffers_from_rank0(nets, group=None):
    """Every rank takes rank 0's BatchNorm running statistics and ``num_batches_tracked`` counters (DataParallel keeps replica 0's,
    Encoders.py:39-40).  ``nets``: modules on flat storage (``flat_buffers`` fp32, ``flat_nbt`` int64)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    src = dist.get_global_rank(group, 0) if group is not None else 0
    for net in nets:
        net = getattr(net, "module", net)
        dist.broadcast(net.flat_buffers, src=src, group=group)
        dist.broadcast(net.flat_nbt, src=src, group=group)


def buffers_in_sync(nets, group=None):
    """True on every rank iff all ranks hold bit-identical running statistics (one all-reduce of 2 x len(nets) checksum words:
    MIN and MAX of an order-independent integer checksum agree only for identical bits)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return True
    sums = []
    for net in nets:
        net = getattr(net, "module", net)
        bits = net.flat_buffers.view(torch.int32).to(torch.int64)
        idx = torch.arange(1, bits.numel() + 1, device=bits.device, dtype=torch.int64)
        sums.append((bits * idx).sum() + net.flat_nbt.sum() * 1000003)           # int64 wrap-around is fine: equal bits, equal sums
    lo = torch.stack(sums)
    hi = lo.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    return bool(torch.equal(lo, hi))


def all_gather_rows(local, bounds, group=None):
    """local [bounds[r+1] - bounds[r], D] on rank r -> the full [bounds[-1], D] matrix on every rank (one all-gather of equal-sized,
    zero-padded blocks; xGMI is point to point, so one large collective beats one per inference batch)."""
    world = dist.get_world_size(group)
    n = bounds[-1]
    d = local.shape[1]
    per = max(bounds[r + 1] - bounds[r] for r in range(world))
    send = torch.zeros(per, d, device=local.device, dtype=local.dtype)
    send[:local.shape[0]] = local
    recv = torch.empty(world * per, d, device=local.device, dtype=local.dtype)
    dist.all_gather_into_tensor(recv, send, group=group) if local.is_cuda and dist.get_backend(group) == "nccl" else \
        dist.all_gather(list(recv.view(world, per, d).unbind(0)), send, group=group)
    if all(bounds[r + 1] - bounds[r] == per for r in range(world)):
        return recv[:n]
    return torch.cat([recv[r * per:r * per + bounds[r + 1] - bounds[r]] for r in range(world)], 0)
