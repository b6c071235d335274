"""The short-K 1x1 launches with the fused output stage (conv3 + bn3 + identity + ReLU forward; masked conv1 data gradient), per layer shape,
with the persistent streaming kernel (csrc/fused1x1.h, DALI_CONV_PERSIST=1) and with the tile-per-workgroup kernels (0), alternating in one
process; three operand sets rotate so that nothing is served from the Infinity Cache.   python scripts/bench_fused1x1.py [batch=256]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daliid_amd import _lib, ops_nn as nn
bf16 = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = _lib.lib()
SHAPES = [("layer1", B * 64 * 32, 64, 256), ("layer2", B * 32 * 16, 128, 512), ("layer3", B * 16 * 8, 256, 1024), ("layer4", B * 16 * 8, 512, 2048)]
NSET = 3


def timeit(fn, n=30):
    for i in range(NSET):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i % NSET)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, P, K, Cm in SHAPES:
    xs = [torch.randn(P, K, device="cuda").to(bf16) for _ in range(NSET)]
    rs = [torch.randn(P, Cm, device="cuda").to(bf16) for _ in range(NSET)]
    ys = [torch.empty(P, Cm, device="cuda", dtype=bf16) for _ in range(NSET)]
    w = (torch.randn(Cm, K, device="cuda") / K ** 0.5).to(bf16)
    sc, sh = torch.rand(Cm, device="cuda") + 0.5, torch.randn(Cm, device="cuda")
    mask = torch.randint(0, 256, (P * Cm // 8,), device="cuda", dtype=torch.uint8)
    bits = torch.empty(P * Cm // 8, device="cuda", dtype=torch.uint8)
    L = _lib

    def fwd(i):
        _lib.check(lib.dali_conv1x1_fused(L.ctx(xs[i].device), L.stream_ptr(), L.ptr(xs[i]), L.ptr(w), L.ptr(ys[i]), P, K, Cm, L.ptr(sc), L.ptr(sh), None,
                                          L.ptr(rs[i]), 1, L.ptr(bits), None, None), "fused fwd")

    def dgrad(i):
        _lib.check(lib.dali_conv1x1_fused(L.ctx(xs[i].device), L.stream_ptr(), L.ptr(xs[i]), L.ptr(w), L.ptr(ys[i]), P, K, Cm, None, None, None,
                                          L.ptr(rs[i]), 0, None, L.ptr(mask), None), "fused dgrad")

    def evalf(i):
        _lib.check(lib.dali_conv1x1_fused(L.ctx(xs[i].device), L.stream_ptr(), L.ptr(xs[i]), L.ptr(w), L.ptr(ys[i]), P, K, Cm, L.ptr(sc), L.ptr(sh), None,
                                          L.ptr(rs[i]), 1, None, None, None), "fused eval")
    byts = P * (K + 2 * Cm) * 2 + P * Cm // 8
    out = []
    for rep in range(2):
        for mode in ("0", "1"):
            os.environ["DALI_CONV_PERSIST"] = mode
            lib.dali_debug_reload_env()
            out.append((mode, timeit(fwd), timeit(dgrad), timeit(evalf)))
    for mode, tf, td, te in out:
        print("%s P=%d K=%d Cm=%d persist=%s: fwd %.1f us (%.2f TB/s)  masked dgrad %.1f us  eval fwd %.1f us | byte roof at 6 TB/s %.1f us"
              % (name, P, K, Cm, mode, tf, byts / tf / 1e6, td, te, byts / 6e6))
    del xs, rs, ys
