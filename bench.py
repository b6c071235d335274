#!/usr/bin/env python3
"""bench.py -- headline benchmark of the DaliID Person-ReID hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload train|vit|distance|epoch]

Prints ONE JSON line (rank 0).  Workloads (BASELINE.json):
  * train    (configs[1]): ResNet-50 ReID bf16, PK batch 16x16=256 per GPU, center + proxy heads, Adam, EMA;
               metric images/sec; data-parallel over N GPUs with RCCL gradient all-reduce (weak scaling).
               The default run also carries the other half of BASELINE.json's metric as a ``"distance"`` sub-record
               (configs[4], below) unless --no-distance is given.
  * distance (configs[4]): 10k x 100k x 2048 cosine distmat (+ CMC/mAP timed separately); metric Gpairs/sec.
  * vit      (configs[3]): TransReID ViT-B/16 train step, batch 128 per GPU (also a ``"vit"`` sub-record of the default run).
  * epoch    one Market-1501-shaped epoch of trainer.train (12,936-image inference + 751-id targets + 46 PK steps of 384), the
               only thing the reference's logs time; also an ``"epoch"`` sub-record of the default run.
Inputs are synthetic and resident in HBM before the timed region.

``--gpus N`` with N > 1: when no launcher has set WORLD_SIZE, this process starts N ranks itself
(``python -m torch.distributed.run --nproc-per-node N bench.py ...`` as a CHILD process, before anything here touches a
GPU), relays rank 0's JSON line and exits with the launcher's status.  Under an outer ``torch.distributed.run`` it is
a rank and reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA


def self_launch(args, argv):
    """--gpus N > 1 without a launcher: start N ranks as a child process group and relay rank 0's line.  Nothing in this
    process has touched a GPU yet (``torch.cuda.device_count()`` does not initialise HIP), and the ranks are CHILD
    processes, never an exec of this one."""
    n_dev = torch.cuda.device_count()
    if 0 < n_dev < args.gpus:
        print(json.dumps({"error": "--gpus %d but only %d GPUs are visible" % (args.gpus, n_dev), "n_gpus": args.gpus}))
        return 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC (RCCL across processes)
    env["DALIID_BENCH_CHILD"] = "1"
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    log("self-launch: %s" % " ".join(cmd))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:
        out = out.rstrip("\n")
        if out.startswith("{") and '"n_gpus"' in out:
            line = out
        elif out:
            print(out, file=sys.stderr, flush=True)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
        if rc == 0 and '"error"' in line:
            rc = 3
    elif rc == 0:
        rc = 1
    return rc


def dist_setup(n_gpus):
    """-> (world, rank, local, backend).  world > 1 on a GPU box: backend "nccl" (= RCCL over xGMI), one rank per GPU.
    Without a GPU (the build container) the ranks still form a gloo group so that the launch path itself is testable;
    the workloads then refuse to run (there is no CPU path)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != n_gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d, or leave "
                         "WORLD_SIZE unset and bench.py starts the ranks itself)" % (n_gpus, world, n_gpus))
    have_gpu = torch.cuda.device_count() > 0
    backend = None
    if world > 1:
        import torch.distributed as dist
        if have_gpu:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            backend = "nccl"
        else:
            dist.init_process_group("gloo")
            backend = "gloo"
    elif have_gpu:
        torch.cuda.set_device(0)
    return world, rank, local, backend


def cpu_guard_line(args, world, rank, backend):
    """No GPU here: prove the launch path (N ranks, process group, one all-reduce) and say so in the line."""
    seen = world
    if world > 1:
        import torch.distributed as dist
        t = torch.ones(1)
        dist.all_reduce(t)
        seen = int(t.item())
        dist.barrier()
    return {"error": "no GPU visible: daliid_amd has no CPU path; launch path only", "n_gpus": world, "world_size_seen": seen,
            "backend": backend, "steps": args.steps, "warmup": args.warmup}


def barrier_sync(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def ranks_seen(world):
    """number of ranks that took part in a SUM all-reduce of ones on the job's process group (RCCL on a GPU node): the line's
    world_size_seen is what the collective saw, not what the environment said"""
    if world == 1:
        return 1
    import torch.distributed as dist
    t = torch.ones(1, device="cuda", dtype=torch.float32)
    dist.all_reduce(t)
    return int(round(float(t.item())))


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
    nprod = 3 if prec == "bf16x3" else 1
    d_traffic, d_traffic_source = committed_traffic("distance", "pairdist_bf16x3_hbm_bytes_per_launch") if nprod == 3 else (None, "no PMC pass for this precision")
    roofline = {"kernel": "pairdist_dma_kernel<%d>" % nprod, "bound": "mfma",
                "achieved": round(achieved, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": d_traffic, "traffic_source": d_traffic_source,
                "kernel_ms": round(k_ms, 4), "mfma_issue_multiplier": nprod,
                "issue_frac": round(nprod * achieved / MFMA_BF16_PEAK_TFLOPS, 4),
                "note": "achieved = 2*D FLOP per pair (algorithmic) / kernel time; bf16x3 issues 3 bf16 MFMA products per algorithmic "
                        "product (hi*hi + hi*lo + lo*hi) for fp32-grade results, issue_frac = MFMA issue rate / dense bf16 peak"}

    # ranking (CMC/mAP) timed separately
    import numpy as np
    rng = np.random.default_rng(12)
    g_pids = np.repeat(np.arange(1000), 100); q_pids = np.repeat(np.arange(1000), 10)
    g_cams = rng.integers(0, 6, ng); q_cams = rng.integers(0, 6, nq)
    cmc, mAP = ops_eval.rank_eval(out, q_pids, g_pids, q_cams, g_cams)           # end to end incl. host id factorisation
    qp_, gp_ = ops_eval.factorize_ids(q_pids, g_pids)
    qc_, gc_ = ops_eval.factorize_ids(q_cams, g_cams)
    codes = [torch.from_numpy(a).cuda() for a in (qp_, gp_, qc_, gc_)]
    ops_eval.rank_eval_codes(out, *codes)
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(args.steps):
        ops_eval.rank_eval_codes(out, *codes)
    ev[1].record()
    torch.cuda.synchronize()
    rank_ms = ev[0].elapsed_time(ev[1]) / args.steps

    del out, q, g, qp, gp
    torch.cuda.empty_cache()
    cpu = None
    parity = None
    if rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline_distance()
        parity = distance_parity_vs_oracle(prec)
    return {"metric": "gallery-distance Gpairs/sec", "value": round(gpairs, 3), "unit": "Gpairs/s",
            "ms_per_step": round(ms_step, 4), "dtype": "bf16" if prec == "bf16" else "bf16x3(fp32-grade)",
            "config": {"workload": "configs[4]: 10k x 100k x 2048 cosine distmat, normalise fused; per GPU; RANDOM features (randn rows, 1000 ids x 100 "
                                   "gallery / 10 query entries): the ranking kernel's worst case (every gallery entry is binned), so mAP is chance level "
                                   "by construction and says nothing about accuracy (tests/test_gpu_eval.py covers SURVEY 8d's structured gallery)",
                       "nq": nq, "ng": ng, "d": d, "precision": prec, "mAP_on_random_features": round(float(mAP), 6)},
            "rank_eval_ms": round(rank_ms, 3), "rank_eval_GBps": round(nq * ng * 4 / 1e9 / (rank_ms * 1e-3), 1),
            "rank_eval_hbm_frac": round(nq * ng * 4 / 1e9 / (rank_ms * 1e-3) / HBM_PEAK_GBS, 4),
            "roofline": roofline, "cpu_baseline": cpu, "map_vs_oracle": parity}


def distance_parity_vs_oracle(prec):
    """The accuracy half of configs[4] in the driver's line: SURVEY 8(d)'s structured gallery (rows = normalize(id_centroid + noise * randn),
    queries 10 / id, gallery 100 / id, camids uniform{0..5}, seed 12) at 2 k x 20 k x 2048 through the HIP distance + ranking kernels and
    through the CPU oracle (validateModels.py:41-47 + torchreid's market1501 protocol restated in oracle/evalrank.py).  noise = 4.0 instead
    of 8(d)'s 0.5: at 0.5 the identities are separable (mAP = 1.000 on both sides, which a ranking error could hide behind)."""
    import numpy as np
    from daliid_amd import ops_eval
    from oracle import evalrank as E
    noise = 4.0
    q, gal, qp, gp, qc, gc = E.synthetic_reid_set(200, 100, 10, 2048, noise=noise, seed=12)
    d_ref = E.validate_features(q, gal)
    cmc_ref, map_ref = E.eval_market1501(d_ref.numpy(), qp, gp, qc, gc)
    d = ops_eval.pairdist(q.cuda(), gal.cuda(), precision=prec, normalize=True)
    cmc, mAP = ops_eval.rank_eval(d, qp, gp, qc, gc)
    return {"workload": "structured gallery 2k x 20k x 2048 (200 ids, noise %.1f, seed 12)" % noise, "mAP": round(float(mAP), 6),
            "mAP_oracle": round(float(map_ref), 6), "abs_diff": float(abs(mAP - map_ref)), "rank1": round(float(cmc[0]), 6),
            "rank1_oracle": round(float(cmc_ref[0]), 6), "cmc_max_abs_diff": float(np.abs(np.asarray(cmc) - np.asarray(cmc_ref)).max()),
            "distmat_max_abs_diff": float((d.cpu() - d_ref).abs().max())}


def kernel_source_hash():
    """sha1 over the HIP kernel sources: ties a committed PMC profile to the code it was measured on (.git does not travel
    to the GPU box, so a commit id cannot be read there)."""
    h = hashlib.sha1()
    d = os.path.join(ROOT, "daliid_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def committed_traffic(which, key="gemm_kernels_hbm_bytes_per_step"):
    """HBM bytes per step of the GEMM kernels from the committed PMC pass (profiles/*_pmc_traffic.json: rocprofv3 --pmc in
    its own run, corrected as MI355X_MICROARCH.md prescribes).  -> (bytes or None, provenance string).  The number is NOT
    measured in this run; it is reported only while the kernel sources still hash to what the profile was taken on."""
    for rnd in ("r05", "r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", "%s_%s_pmc_traffic.json" % (rnd, which))
        try:
            with open(path) as f:
                j = json.load(f)
            val = j[key]
        except (OSError, KeyError, ValueError):
            continue
        want, have = j.get("kernel_source_hash"), kernel_source_hash()
        rel = os.path.relpath(path, ROOT)
        if want is None:
            return None, "%s carries no kernel_source_hash (taken before the kernels were last changed): not reported" % rel
        if want != have:
            return None, "%s was taken on kernel sources %s, this tree is %s: not reported" % (rel, want, have)
        return val, "committed profile %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in their own runs; kernel sources %s)" % (rel, have)
    return None, "no committed PMC profile"


def log(msg):
    print("[bench] " + msg, file=sys.stderr, flush=True)


def best_cpu_threads(fn, candidates, budget_s=3.0):
    """A GPU box gives this job a CPU share (16 cores per GPU here) far below os.cpu_count(); oversubscribing torch's
    thread pool is catastrophically slow.  Time `fn` once or twice at a few thread counts, smallest first, stop as
    soon as more threads stop helping, and report the count actually used as `cores`."""
    best = (1e30, candidates[0])
    for nt in candidates:
        torch.set_num_threads(nt)
        t0 = time.perf_counter(); fn(); first = time.perf_counter() - t0
        b = first
        if first < budget_s:
            t0 = time.perf_counter(); fn(); b = min(b, time.perf_counter() - t0)
        log("cpu baseline probe: %d threads -> %.3f s" % (nt, b))
        if b < best[0]:
            best = (b, nt)
        elif b > 1.3 * best[0]:
            break
    torch.set_num_threads(best[1])
    return best[1]


def cpu_thread_candidates():
    hw = os.cpu_count() or 1
    try:
        hw = min(hw, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    return sorted({min(hw, c) for c in (8, 16, 32, 64)})


def cpu_baseline_distance():
    """The CPU restatement of validateModels.py:41-47 (oracle) on the host cores: 2k x 20k x 2048 sample."""
    from oracle import evalrank as E
    g = torch.Generator().manual_seed(12)
    q = torch.randn(2000, 2048, generator=g)
    gal = torch.randn(20000, 2048, generator=g)
    cores = best_cpu_threads(lambda: E.validate_features(q, gal), cpu_thread_candidates())
    log("cpu baseline: distance with %d threads" % cores)
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
RESNET50_GFLOP_PER_IMAGE = 24.320      # conv/GEMM FLOPs fwd+bwd per 256x128 image


def make_train_state(args, world, rank, batch, device):
    """configs[1] / configs[2]: synthetic PK batch (16 ids x 16) resident on the device, NC = 1024 identities,
    5 proxies per identity, epoch 10 of 250, tau 0.05, lambda 0.4, lr 3.5e-4, wd 5e-4, beta 0.999."""
    from daliid_amd import Encoders, optim
    from daliid_amd.losses import LossHeads, _sample_weights
    from daliid_amd.train_encodersKIT import trainer
    from daliid_amd.ops_eval import l2norm_rows
    gen = torch.Generator(device=device).manual_seed(12 + rank)
    online = Encoders.ResNet50ReID(device=device, seed=12)
    momentum = Encoders.ResNet50ReID(device=device, seed=12)
    NC, D = 1024, 2048
    centers = l2norm_rows(torch.randn(NC, D, device=device, generator=torch.Generator(device=device).manual_seed(1)))
    proxies = l2norm_rows(torch.randn(5 * NC, D, device=device, generator=torch.Generator(device=device).manual_seed(2)))
    pg = None
    if world > 1:
        import torch.distributed as dist
        pg = dist.group.WORLD
    drv = torch.optim.Adam(online.parameters(), lr=3.5e-4, weight_decay=5e-4)             # mainKIT.py:99
    tr = trainer("Synthetic", None, "resnet50", {}, 256, 128, None, False, 1, drv, 16, 16, 0.05, 0.999, 0.4, 250, online, momentum,
                 [device.index], "bench", process_group=pg)
    heads = LossHeads(centers, torch.arange(NC).numpy(), proxies, torch.arange(NC).repeat_interleave(5).numpy(), 0.05, 0.4, pg)
    imgs = torch.randn(batch, 3, 256, 128, device=device, generator=gen)
    P = batch // 16
    ids = (torch.arange(P, device=device) + rank * P) % NC
    labels = ids.repeat_interleave(16).to(torch.int32)
    distortion = torch.stack((torch.zeros(batch // 2, dtype=torch.long), torch.randint(1, 6, (batch // 2,))), 1).reshape(-1)   # AT pairing
    w = _sample_weights(distortion, 10, 250, device)
    online.train(); momentum.eval()
    acc = torch.zeros(6, device=device)
    return tr, heads, imgs, labels, w, acc


VIT_B16_GFLOP_PER_IMAGE = 105.378      # fwd+bwd GEMM FLOPs per 224x224 image


def make_vit_state(args, world, rank, batch, device):
    """configs[3]: TransReID ViT-B/16 (vit_pytorch.py) + BN neck, 224x224, bf16, same heads / Adam / EMA as configs[1]
    (the reference has no training route for it; the build wires it into the same trainer)."""
    from daliid_amd import vit_pytorch as V
    from daliid_amd.losses import LossHeads, _sample_weights
    from daliid_amd.train_encodersKIT import trainer
    from daliid_amd.ops_eval import l2norm_rows
    gen = torch.Generator(device=device).manual_seed(12 + rank)
    mk = lambda: V.ViTNeckNet(img_size=(224, 224), patch_size=16, stride_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0,
                              drop_path_rate=float(os.environ.get("DALIID_BENCH_DROP_PATH", "0.1")), device=device, seed=12)   # 0.1: the reference factory's default (vit_pytorch.py:453)
    online, momentum = mk(), mk()
    NC, D = 1024, 768
    centers = l2norm_rows(torch.randn(NC, D, device=device, generator=torch.Generator(device=device).manual_seed(1)))
    proxies = l2norm_rows(torch.randn(5 * NC, D, device=device, generator=torch.Generator(device=device).manual_seed(2)))
    pg = None
    if world > 1:
        import torch.distributed as dist
        pg = dist.group.WORLD
    drv = torch.optim.Adam([p for p in online.parameters() if p.requires_grad], lr=3.5e-4, weight_decay=5e-4)
    tr = trainer("Synthetic", None, "vit_base", {}, 224, 224, None, False, 1, drv, batch // 16, 16, 0.05, 0.999, 0.4, 250, online, momentum,
                 [device.index], "bench", process_group=pg)
    heads = LossHeads(centers, torch.arange(NC).numpy(), proxies, torch.arange(NC).repeat_interleave(5).numpy(), 0.05, 0.4, pg)
    imgs = torch.randn(batch, 3, 224, 224, device=device, generator=gen)
    P = batch // 16
    labels = ((torch.arange(P, device=device) + rank * P) % NC).repeat_interleave(16).to(torch.int32)
    distortion = torch.stack((torch.zeros(batch // 2, dtype=torch.long), torch.randint(1, 6, (batch // 2,))), 1).reshape(-1)
    w = _sample_weights(distortion, 10, 250, device)
    online.train(); momentum.eval()
    return tr, heads, imgs, labels, w, torch.zeros(6, device=device)


def bench_train(args, world, rank):
    device = torch.device("cuda", torch.cuda.current_device())
    vit = args.workload == "vit"
    batch = args.batch if args.batch else (128 if vit else 256)
    gflop_img = VIT_B16_GFLOP_PER_IMAGE if vit else RESNET50_GFLOP_PER_IMAGE
    tr, heads, imgs, labels, w, acc = (make_vit_state if vit else make_train_state)(args, world, rank, batch, device)
    step = lambda: tr.train_step(heads, imgs, labels, w, acc)
    log("train state built (batch %d per GPU); warmup x%d" % (batch, args.warmup))
    for _ in range(args.warmup):
        step()
    barrier_sync(world)
    log("warmup done; timing %d steps" % args.steps)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    barrier_sync(world)
    dt = max_over_ranks(time.perf_counter() - t0, world)
    gpu_ms = ev0.elapsed_time(ev1) / args.steps
    ms_step = dt / args.steps * 1e3
    ips = world * batch / (dt / args.steps)
    step_tflops = batch * gflop_img / 1e3 / (gpu_ms * 1e-3)
    # Dominant kernel family = the implicit-GEMM MFMA kernels (igemm_conv_* forward/dgrad + igemm_wgrad_*).  Their
    # launch durations are measured live: the same steps again with every such launch bracketed by HIP events on the
    # launch stream (dali_gemm_profile_*).  achieved = algorithmic GEMM FLOPs of those launches / their summed duration.
    from daliid_amd._lib import GemmProfile
    psteps = min(args.steps, 3)
    with GemmProfile(max_launches=4096 * psteps) as gp:
        for _ in range(psteps):
            step()
    barrier_sync(world)
    k_ms = sum(gp.ms) / psteps
    k_launches = sum(gp.launches) // psteps
    alg_flops = batch * gflop_img * 1e9
    tflops = alg_flops / (k_ms * 1e-3) / 1e12
    traffic, traffic_source = committed_traffic("vit" if vit else "train")
    step_bytes, _ = committed_traffic("vit" if vit else "train", "all_kernels_hbm_bytes_per_step")     # the step is priced by both roofs
    roofline = {"kernel": "igemm_conv_* + fused1x1_persist + igemm_wgrad_* (implicit-GEMM MFMA kernels%s)" % ("; attention kernels not included" if vit else ""),
                "bound": "mfma", "achieved": round(tflops, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(tflops / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                "launches_per_step": int(k_launches), "kernel_ms_per_step": round(k_ms, 3),
                "avg_launch_us": round(k_ms * 1e3 / max(k_launches, 1), 2),
                "by_class_ms_per_step": {"conv_fwd_dgrad": round(gp.ms[0] / psteps, 3), "wgrad": round(gp.ms[1] / psteps, 3)},
                "launched_gemm_tflop_per_step": round(sum(gp.flops) / psteps / 1e12, 4),
                "whole_step": {"achieved": round(step_tflops, 2), "frac": round(step_tflops / MFMA_BF16_PEAK_TFLOPS, 4),
                               "device_ms": round(gpu_ms, 3)},
                "hbm_frac": None if step_bytes is None else round(step_bytes / 1e9 / (gpu_ms * 1e-3) / HBM_PEAK_GBS, 4),
                "all_kernels_hbm_bytes_per_step": step_bytes,
                "note": "achieved = algorithmic GEMM FLOPs per step (%.3f GFLOP/img x %d) / summed duration of the GEMM kernel launches of "
                        "one step, HIP events per launch on the launch stream; whole_step divides the same FLOPs by the device time of the "
                        "entire step (BatchNorm, pools, heads, Adam, EMA included).  The GEMM launches also carry fused non-GEMM work (every bn3 + residual + ReLU + "
                        "mask output stage, the Gram-scheme products and column sums: launched_gemm_tflop_per_step > the algorithmic figure), so moving "
                        "work into them lowers `frac` while the step gets faster: read whole_step beside it.  hbm_frac = HBM bytes of ALL kernels of "
                        "one step (committed PMC pass) / device time of the step / 8 TB/s" % (gflop_img, batch)}
    final = acc.cpu().numpy()
    log("GPU: %.3f ms/step (device %.3f ms), %.1f images/s" % (ms_step, gpu_ms, ips))
    comm = None
    if world > 1:
        comm = allreduce_probe(tr, world, args.steps)
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        if vit:
            sd = {k: v.detach().cpu().clone() for k, v in tr.model_online.state_dict().items()}
            del tr, heads, imgs
            torch.cuda.empty_cache()
            cpu = cpu_baseline_vit(sd)
        else:
            cpu = cpu_baseline_train()
    wl = ("configs[3]: TransReID ViT-B/16 bf16 224x224, drop_path 0.1, batch %d per GPU, center+proxy heads, Adam, EMA" % batch) if vit else \
         "configs[1]: ResNet-50 ReID bf16 256x128, PK batch 16x16=256 per GPU, center+proxy heads, Adam, EMA"
    return {"metric": "images/sec (train step)", "value": round(ips, 2), "unit": "images/s", "ms_per_step": round(ms_step, 3),
            "dtype": "bf16", "config": {"workload": wl + ("; data-parallel, RCCL all-reduce of gradients" if world > 1 else ""),
                                        "global_batch": world * batch, "per_gpu_batch": batch, "parallelism": "dp%d" % world,
                                        "mean_loss": float(final[2] / max(final[4], 1))},
            "roofline": roofline, "cpu_baseline": cpu, "comm": comm}


def bench_epoch(args, world, rank):
    """One Market-1501-shaped epoch of ``trainer.train`` (train_encodersKIT.py:104-156,176-235) on device-resident synthetic
    images: eval-mode inference of the 12,936 train images at batch 500 (getFeatures.py:47-71) -> class centers + 5
    farthest-point proxies for 751 identities -> 46 PK steps of 16 identities x 12 images x (clean, distorted) = 384 images
    (mainKIT.py:326-327,340 defaults, AT pairing).  The product's own entry points run (extractFeatures,
    build_centers_and_proxies, samplePKBatches, trainer.train); only decode / augmentation are replaced by a gather from an
    image pool in HBM (no dataloader in the timed region).  The reference's log lines for the same epoch
    (log_AT_training_Market.txt:14,19 and :7269: 9.49 s inference, 70.43 s per epoch, 3 unnamed GPUs, JPEG decode included)
    are context, not a target."""
    import numpy as np
    from daliid_amd import Encoders, getFeatures, train_encodersKIT as T
    device = torch.device("cuda", torch.cuda.current_device())
    N, NID, P, K, H, W = 12936, 751, 16, 12, 256, 128
    gen = torch.Generator(device=device).manual_seed(12 + rank)
    pool = torch.empty(N, 3, H, W, device=device)
    for b in range(0, N, 1024):
        pool[b:b + 1024] = torch.randn(min(1024, N - b), 3, H, W, device=device, generator=gen)
    labels = (np.arange(N) % NID).astype(np.int32)
    records = np.array([["pool://%d" % i, str(labels[i]), str(i % 6), "person"] for i in range(N)])

    def index_of(paths, turb):
        idx = np.fromiter((int(p[7:]) for p in paths), dtype=np.int64, count=len(paths))
        return (idx + 97 * int(turb[1])) % N if turb is not None else idx     # the distorted partner: another image of the pool

    def take(idx):
        # a run of consecutive pool rows (the epoch inference walks the train set in order) is a view, anything else one gather
        if len(idx) and int(idx[-1]) - int(idx[0]) == len(idx) - 1 and bool(np.all(np.diff(idx) == 1)):
            return pool[int(idx[0]):int(idx[-1]) + 1]
        return pool[torch.from_numpy(idx).to(device)]

    def loader(paths, img_height, img_width, turb=None):
        return take(index_of(paths, turb))

    # the batched loader protocol of daliid_amd.transforms (plan -> submit -> finish): a PK batch is ONE gather from the pool instead of
    # one per identity and distorted image, as the real-data loader's one resize + one augment launch per batch
    class _Plan:
        def __init__(self, idx):
            self.files, self.idx = idx, idx

        @staticmethod
        def concat(plans, order=None):
            idx = np.concatenate([p.idx for p in plans])
            return _Plan(idx[np.asarray(order)] if order is not None else idx)
    loader.plan = lambda paths, h, w, turb=None: _Plan(index_of(paths, turb))
    loader.submit = lambda plan: plan
    loader.finish = lambda plan, dev=None, side_stream=True: take(plan.idx)
    getFeatures.set_image_loader(loader); T.set_train_loader(loader)
    try:
        online = Encoders._DataParallelShim(Encoders.ResNet50ReID(device=device, seed=12))
        momentum = Encoders._DataParallelShim(Encoders.ResNet50ReID(device=device, seed=12))
        pg = None
        if world > 1:
            import torch.distributed as dist
            pg = dist.group.WORLD
        drv = torch.optim.Adam(online.parameters(), lr=3.5e-4, weight_decay=5e-4)
        tr = T.trainer("Synthetic", records, "resnet50", {}, H, W, "pool", False, 1, drv, P, K, 0.05, 0.999, 0.4, 250, online, momentum,
                       [device.index], "bench", process_group=pg)
        import contextlib, io
        quiet = contextlib.redirect_stdout(io.StringIO())
        with quiet:
            tr.train(records, labels, 1, 10)                  # warm-up epoch (plan construction for batch 500 / 436 / 384, allocator)
        barrier_sync(world)
        # (1) inference alone
        t0 = time.perf_counter()
        with quiet:
            fvs = tr.extract_train_features(records)          # world > 1: this rank's slice + one all-gather (train_encodersKIT.extract_train_features)
        torch.cuda.synchronize()
        t_inf = max_over_ranks(time.perf_counter() - t0, world)
        rows_this_rank = getattr(tr, "last_inference_rows", N) if world > 1 else N
        # (1b) the forward alone on one resident batch of 500 (no loader, no gather from the pool, no concatenation): what the kernels do
        xb = pool[:500].contiguous()
        online.eval()
        with torch.no_grad():
            for _ in range(2):
                online(xb)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                online(xb)
            torch.cuda.synchronize()
        t_fwd = (time.perf_counter() - t0) / 10
        del xb
        # (2) centers + proxies alone
        np.random.seed(12)
        T.build_centers_and_proxies(fvs, labels, 5); torch.cuda.synchronize()
        t0 = time.perf_counter()
        T.build_centers_and_proxies(fvs, labels, 5)
        torch.cuda.synchronize()
        t_tgt = time.perf_counter() - t0
        del fvs
        # (3) the whole epoch through trainer.train
        barrier_sync(world)
        t0 = time.perf_counter()
        with quiet:
            tr.train(records, labels, 1, 10)
        barrier_sync(world)
        t_epoch = max_over_ranks(time.perf_counter() - t0, world)
        st = tr.last_epoch_stats
    finally:
        getFeatures.set_image_loader(None); T.set_train_loader(None)
    steps = int(st["steps"])
    t_loop = t_epoch - t_inf - t_tgt
    return {"metric": "epoch seconds (trainer.train: inference + targets + PK loop)", "value": round(t_epoch, 3), "unit": "s",
            "higher_is_better": False,
            "config": {"workload": "Market-1501-shaped synthetic epoch, ResNet-50 bf16 256x128: %d images eval-mode at batch 500, %d ids, "
                                   "%d PK steps of %d images (P=%d, K=%d, clean + distorted); device-resident image pool, per GPU"
                                   % (N, NID, steps, 2 * P * K, P, K)},
            "inference_images_per_s": round(N / t_inf, 1), "inference_s": round(t_inf, 3), "inference_rows_per_rank": int(rows_this_rank),
            "inference_images_per_s_per_rank": round(rows_this_rank / t_inf, 1),
            "inference_forward_only": {"images_per_s": round(500 / t_fwd, 1), "ms_per_batch": round(t_fwd * 1e3, 3),
                                       "note": "the eval-mode forward on one resident batch of 500; inference_images_per_s above adds the "
                                               "ragged last batch, the plan switches and the concatenation of extractFeatures"},
            "targets_ms": round(t_tgt * 1e3, 2),
            "pk_steps": steps, "pk_loop_s": round(t_loop, 3), "pk_images_per_s": round(world * steps * 2 * P * K / max(t_loop, 1e-9), 1),
            "mean_loss": float(st["loss"]),
            "reference_context": "log_AT_training_Market.txt:14,19 / :7269: 9.49 s inference, 70.43 s per epoch on 3 unnamed GPUs, JPEG decode and "
                                 "PIL transforms included (here: device-resident pool) -- context, not a target"}


def allreduce_probe(tr, world, steps):
    """The gradient all-reduce of one step alone (all stage buckets back to back on the reducer's stream, nothing to overlap
    with): ms per step and bus bandwidth, so that the scaling curve can be read against the collective's own cost."""
    import torch.distributed as dist
    dp = tr._dp
    if dp is None:
        return None
    n_bytes = sum(max(e - b, 0) for b, e in dp.ranges) * 4
    def one():
        for s in range(len(dp.ranges)):
            dp.reduce_stage(s)
        dp.finish()
    one(); barrier_sync(world)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(steps):
        one()
    ev1.record()
    barrier_sync(world)
    ms = max_over_ranks(ev0.elapsed_time(ev1) / steps, world)
    return {"backend": dist.get_backend(), "world_size_seen": ranks_seen(world), "buckets": len(dp.ranges), "bytes_per_step": n_bytes,
            "allreduce_ms_per_step": round(ms, 3),
            "bus_GBps": round(2.0 * (world - 1) / world * n_bytes / 1e9 / (ms * 1e-3), 1),
            "note": "standalone (un-overlapped) SUM all-reduce of the flat fp32 gradient buffer in its per-stage buckets; in the timed "
                    "step the buckets run on a side stream under the remaining backward"}


def cpu_baseline_train():
    """configs[0]: the CPU restatement (oracle) of one full train step -- ResNet-50 fp32, 32 x 3 x 256 x 128, both heads,
    Adam, EMA -- on the host cores."""
    from oracle.resnet50_reid import ResNet50ReID
    from oracle import trainstep as TS
    torch.manual_seed(12)
    online, momentum = ResNet50ReID(), ResNet50ReID()
    momentum.load_state_dict(online.state_dict())
    online.train(); momentum.eval()
    opt = torch.optim.Adam(online.parameters(), lr=3.5e-4, weight_decay=5e-4)
    g = torch.Generator().manual_seed(12)
    nb, NC = 32, 1024
    imgs = torch.randn(nb, 3, 256, 128, generator=g)
    centers = torch.nn.functional.normalize(torch.randn(NC, 2048, generator=g))
    proxies = torch.nn.functional.normalize(torch.randn(5 * NC, 2048, generator=g))
    clabels = torch.arange(NC).numpy(); plabels = torch.arange(NC).repeat_interleave(5).numpy()
    labels = torch.arange(2).repeat_interleave(16).float()
    dist = torch.randint(0, 6, (nb,), generator=g)
    def one():
        TS.train_step(online, momentum, opt, imgs, labels, dist, centers, clabels, proxies, plabels, 10, 250, 0.05, 0.999, 0.4)
    log("cpu baseline: oracle ResNet-50 train step at batch 32")
    cores = best_cpu_threads(one, cpu_thread_candidates(), budget_s=8.0)
    best, t_end, n = 1e9, time.perf_counter() + 15.0, 0
    while time.perf_counter() < t_end and n < 8:
        t0 = time.perf_counter(); one(); best = min(best, time.perf_counter() - t0); n += 1
        log("cpu baseline step %d: %.3f s" % (n, best))
    rec = {"value": round(nb / best, 2), "unit": "images/s", "cores": cores, "kind": "port",
           "sample": "oracle train step (oracle.trainstep.train_step: ResNet-50 fp32 fwd+bwd, center+proxy heads, Adam, EMA) at batch 32 "
                     "(configs[0]), best of %d" % n}
    # configs[1]'s own batch on the same cores (1 warm-up + 2 timed steps, for the speed-up ratio only)
    if os.environ.get("DALIID_BENCH_CPU256", "1") != "0":
        try:
            nb2 = 256
            imgs = torch.randn(nb2, 3, 256, 128, generator=g)
            labels = torch.arange(16).repeat_interleave(16).float()
            dist = torch.randint(0, 6, (nb2,), generator=g)
            times = []
            for i in range(3):
                t0 = time.perf_counter(); one(); times.append(time.perf_counter() - t0)
                log("cpu baseline batch 256 step %d: %.2f s" % (i, times[-1]))
                if times[-1] > 45.0:                                 # a slow host: keep the default run within minutes
                    break
            timed = times[1:] if len(times) > 1 else times
            rec["batch256"] = {"value": round(nb2 / min(timed), 2), "unit": "images/s", "steps_timed": len(timed),
                               "sample": "the same oracle step at batch 256 (configs[1]'s batch), %d warm-up + %d timed" % (len(times) - len(timed), len(timed))}
        except (RuntimeError, MemoryError) as e:                      # host memory: the fp32 activations of 256 images are ~30 GB
            rec["batch256"] = {"error": str(e)[:200]}
    return rec


def cpu_baseline_vit(sd):
    """configs[3] on the host cores: the CPU restatement of TransReID ViT-B/16 + BN neck (oracle/vit.py, pinned to
    the reference's own vit_pytorch.py outputs by tests/golden/vit.npz), forward + backward at batch 8, fp32, train mode without DropPath."""
    from oracle import vit as OV
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd.items()}
    g = torch.Generator().manual_seed(12)
    nb = 8
    x = torch.randn(nb, 3, 224, 224, generator=g)
    w = torch.randn(nb, 768, generator=g)

    def one():
        for v in sd.values():
            if v.requires_grad:
                v.grad = None
        (OV.build_transformer_forward(sd, x, num_heads=12, patch=16, stride=16, training=True) * w).sum().backward()
    log("cpu baseline: oracle ViT-B/16 forward + backward at batch %d" % nb)
    cores = best_cpu_threads(one, cpu_thread_candidates(), budget_s=6.0)
    best, t_end, n = 1e9, time.perf_counter() + 10.0, 0
    while time.perf_counter() < t_end and n < 6:
        t0 = time.perf_counter(); one(); best = min(best, time.perf_counter() - t0); n += 1
    return {"value": round(nb / best, 2), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "oracle ViT-B/16 + BN neck (oracle.vit.build_transformer_forward, fp32 torch CPU) forward + backward at batch %d, 224x224, "
                      "best of %d (no loss heads / optimizer: the GEMM work of the step)" % (nb, n)}


# --------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="train", choices=["train", "vit", "distance", "epoch"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default 256 for train, 128 for vit)")
    ap.add_argument("--precision", default="bf16x3", choices=["bf16x3", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-distance", action="store_true", help="train workload: skip the configs[4] distance sub-record")
    ap.add_argument("--no-vit", action="store_true", help="train workload: skip the configs[3] ViT-B/16 sub-record")
    ap.add_argument("--no-epoch", action="store_true", help="train workload: skip the Market-shaped epoch sub-record")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))
    world, rank, local, backend = dist_setup(args.gpus)
    if torch.cuda.device_count() == 0:
        res = cpu_guard_line(args, world, rank, backend)
        if rank == 0:
            print(json.dumps(res), flush=True)
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()
        sys.exit(0 if os.environ.get("DALIID_BENCH_CHILD") else 3)      # the self-launching parent turns the error line into status 3
    if args.workload == "distance":
        res = bench_distance(args, world, rank)
    elif args.workload == "epoch":
        res = bench_epoch(args, world, rank)
    else:
        res = bench_train(args, world, rank)
        if args.workload == "train" and not args.no_distance:
            # the other half of BASELINE.json's metric in the same line: configs[4] on every rank's own synthetic gallery
            # (weak scaling: the gallery shards over ranks with no exchange)
            d = bench_distance(args, world, rank)
            d.pop("metric")
            res["distance"] = d
        # (the vit / epoch sub-records belong to the single-GPU line: the scaling runs time the train step and the distance leg only)
        if args.workload == "train" and not args.no_vit and world == 1:
            # configs[3] in the driver's line as well: 5 warm-up + 20 timed ViT-B/16 steps (~0.6 s)
            torch.cuda.empty_cache()
            va = argparse.Namespace(**vars(args))
            va.workload, va.batch, va.steps, va.warmup = "vit", 0, 20, 5
            v = bench_train(va, world, rank)
            v.pop("metric"); v["steps"], v["warmup"] = va.steps, va.warmup
            res["vit"] = v
        if args.workload == "train" and not args.no_epoch and world == 1:
            torch.cuda.empty_cache()
            e = bench_epoch(args, world, rank)
            res["epoch"] = e
    res.update({"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": args.workload != "epoch",
                "scaling": "weak", "vs_baseline": None, "data": "synthetic"})
    if world > 1:
        res["world_size_seen"] = ranks_seen(world)          # from an all-reduce of ones on the NCCL (= RCCL) group
        res["backend"] = backend
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
