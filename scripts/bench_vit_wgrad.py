"""ViT-B/16 linear weight gradients at batch 128 (25216 rows), with and without the bias gradient riding on the GEMM, variants interleaved in ONE process:
    python scripts/bench_vit_wgrad.py "DALI_WGRAD_CFG=0" "" ..."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from daliid_amd import ops_vit as V, _lib
bf16 = torch.bfloat16
rows = 128 * 197
variants = sys.argv[1:] or [""]
lib = _lib.lib()
def select(v):
    for kv in (v.split() if v else []):
        k, val = kv.split("="); os.environ[k] = val
    lib.dali_debug_reload_env()
def unselect(v):
    for kv in (v.split() if v else []): os.environ.pop(kv.split("=")[0], None)
def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
xs = {k: torch.randn(rows, k, device="cuda").to(bf16) for k in (768, 2304, 3072)}
print("%-22s | " % "layer (K x N)" + " | ".join("%-30s" % (v or "default") for v in variants))
tot = np.zeros((len(variants), 2))
for name, K, N, cnt in (("qkv 768x2304", 768, 2304, 12), ("proj 768x768", 768, 768, 12), ("fc1 768x3072", 768, 3072, 12), ("fc2 3072x768", 3072, 768, 12)):
    t = np.zeros((3, len(variants), 2))
    for r in range(3):
        for i, v in enumerate(variants):
            select(v)
            t[r, i, 0] = timeit(lambda: V.linear_wgrad(xs[K], xs[N], want_bias=True))
            t[r, i, 1] = timeit(lambda: V.linear_wgrad(xs[K], xs[N], want_bias=False))
            unselect(v)
    med = np.median(t, 0); tot += med * cnt
    fl = 2.0 * rows * K * N
    print("%-22s | " % name + " | ".join("bias %6.1f us %4.0f TF, none %6.1f us" % (m[0], fl / m[0] / 1e6, m[1]) for m in med), flush=True)
print("x12 layers (ms): " + " | ".join("bias %.3f, none %.3f" % (a / 1e3, b / 1e3) for a, b in tot))
