#!/usr/bin/env python3
"""bench.py -- headline benchmark of the DaliID Person-ReID hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload train|distance]

Prints ONE JSON line (rank 0).  Workloads (BASELINE.json):
  * train    (configs[1]): ResNet-50 ReID bf16, PK batch 16x16=256 per GPU, center + proxy heads, Adam, EMA;
               metric images/sec; data-parallel over N GPUs with RCCL gradient all-reduce (weak scaling).
  * distance (configs[4]): 10k x 100k x 2048 cosine distmat (+ CMC/mAP timed separately); metric Gpairs/sec.
Inputs are synthetic and resident in HBM before the timed region.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA


def dist_setup(n_gpus):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    assert world == n_gpus, "launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (n_gpus, world)
    return world, rank, local


def barrier_sync(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x, world):
    if world == 1:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], device="cuda", dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# --------------------------------------------------------------------------------------------------
def bench_distance(args, world, rank):
    from daliid_amd import ops_eval
    nq, ng, d = 10000, 100000, 2048
    gen = torch.Generator(device="cuda").manual_seed(12 + rank)
    q = torch.randn(nq, d, device="cuda", generator=gen)
    g = torch.randn(ng, d, device="cuda", generator=gen)
    out = torch.empty(nq, ng, device="cuda", dtype=torch.float32)
    prec = args.precision
    for _ in range(args.warmup):
        ops_eval.pairdist(q, g, precision=prec, normalize=True, out=out)
    barrier_sync(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ops_eval.pairdist(q, g, precision=prec, normalize=True, out=out)
    barrier_sync(world)
    dt = max_over_ranks(time.perf_counter() - t0, world)
    ms_step = dt / args.steps * 1e3
    gpairs = world * nq * ng / 1e9 / (dt / args.steps)

    # dominant kernel alone (pairdist MFMA kernel on prepared operands), HIP events on the launch stream
    qp, gp = ops_eval.PreparedRows(q, True, prec), ops_eval.PreparedRows(g, True, prec)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ops_eval.pairdist_prepared(qp, gp, out=out)
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(args.steps):
        ops_eval.pairdist_prepared(qp, gp, out=out)
    ev[1].record()
    torch.cuda.synchronize()
    k_ms = ev[0].elapsed_time(ev[1]) / args.steps
    flops = 2.0 * d * nq * ng
    achieved = flops / (k_ms * 1e-3) / 1e12
    roofline = {"kernel": "pairdist_kernel<%d>" % (3 if prec == "bf16x3" else 1), "bound": "mfma",
                "achieved": round(achieved, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None,
                "kernel_ms": round(k_ms, 4), "mfma_issue_multiplier": 3 if prec == "bf16x3" else 1}

    # ranking (CMC/mAP) timed separately
    import numpy as np
    rng = np.random.default_rng(12)
    g_pids = np.repeat(np.arange(1000), 100); q_pids = np.repeat(np.arange(1000), 10)
    g_cams = rng.integers(0, 6, ng); q_cams = rng.integers(0, 6, nq)
    ops_eval.rank_eval(out, q_pids, g_pids, q_cams, g_cams)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ops_eval.rank_eval(out, q_pids, g_pids, q_cams, g_cams)
    torch.cuda.synchronize()
    rank_ms = (time.perf_counter() - t0) * 1e3

    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline_distance()
    return {"metric": "gallery-distance Gpairs/sec", "value": round(gpairs, 3), "unit": "Gpairs/s",
            "ms_per_step": round(ms_step, 4), "dtype": "bf16" if prec == "bf16" else "bf16x3(fp32-grade)",
            "config": {"workload": "configs[4]: 10k x 100k x 2048 cosine distmat, normalise fused; per GPU",
                       "nq": nq, "ng": ng, "d": d, "precision": prec, "rank_eval_ms": round(rank_ms, 3)},
            "roofline": roofline, "cpu_baseline": cpu}


def best_cpu_threads(fn, candidates, budget_s=3.0):
    """The host box may have far more hardware threads than a small CPU GEMM can use; time `fn` briefly at a few
    thread counts and keep the fastest (the count actually used is what gets reported as `cores`)."""
    best = (1e30, candidates[0])
    for nt in candidates:
        torch.set_num_threads(nt)
        fn()
        t_end, n, b = time.perf_counter() + budget_s, 0, 1e30
        while time.perf_counter() < t_end and n < 5:
            t0 = time.perf_counter(); fn(); b = min(b, time.perf_counter() - t0); n += 1
        if b < best[0]:
            best = (b, nt)
    torch.set_num_threads(best[1])
    return best[1]


def cpu_thread_candidates():
    hw = os.cpu_count() or 1
    return sorted({min(hw, c) for c in (hw, 128, 64, 32, 16)}, reverse=True)


def cpu_baseline_distance():
    """The CPU restatement of validateModels.py:41-47 (oracle) on the host cores: 2k x 20k x 2048 sample."""
    from oracle import evalrank as E
    g = torch.Generator().manual_seed(12)
    q = torch.randn(2000, 2048, generator=g)
    gal = torch.randn(20000, 2048, generator=g)
    cores = best_cpu_threads(lambda: E.validate_features(q, gal), cpu_thread_candidates())
    best, t_end, n = 1e9, time.perf_counter() + 10.0, 0
    while time.perf_counter() < t_end and n < 20:
        t0 = time.perf_counter()
        E.validate_features(q, gal)
        best = min(best, time.perf_counter() - t0)
        n += 1
    return {"value": round(2000 * 20000 / 1e9 / best, 4), "unit": "Gpairs/s", "cores": cores, "kind": "port",
            "sample": "oracle.evalrank.validate_features (normalise + 1 - q@g.T, fp32 torch CPU) on 2k x 20k x 2048, "
                      "best of %d at the fastest of %s threads" % (n, cpu_thread_candidates())}


# --------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="distance", choices=["distance"])
    ap.add_argument("--precision", default="bf16x3", choices=["bf16x3", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    world, rank, local = dist_setup(args.gpus)
    res = bench_distance(args, world, rank)
    res.update({"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "data": "synthetic"})
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
