"""Weight-gradient launches (incl. the split-K reduce) at the ResNet-50-ReID shapes, per layer, variants interleaved in ONE process:
    python scripts/bench_wgrad.py "DALI_WGRAD_CFG=0" "" ... [--filter l4] [--reps 5]
Each variant is a space-separated list of NAME=VALUE switches (re-read through dali_debug_reload_env)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from daliid_amd import ops_nn as nn, _lib
from bench_convs_shapes import L
bf16 = torch.bfloat16
B = int(os.environ.get("B", "256"))
args = sys.argv[1:]
flt, reps, variants = "", 5, []
while args:
    a = args.pop(0)
    if a == "--filter": flt = args.pop(0)
    elif a == "--reps": reps = int(args.pop(0))
    else: variants.append(a)
variants = variants or [""]
lib = _lib.lib()
def select(v):
    for kv in (v.split() if v else []):
        k, val = kv.split("="); os.environ[k] = val
    lib.dali_debug_reload_env()
def unselect(v):
    for kv in (v.split() if v else []):
        os.environ.pop(kv.split("=")[0], None)
def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tot = np.zeros(len(variants))
print("%-14s %3s %5s %6s %8s | " % ("layer", "x", "Cm", "Ntot", "P") + " | ".join("%-22s" % (v or "default") for v in variants))
for name, H, W, cin, cout, k, st, cnt in L:
    if flt and flt not in name: continue
    pad = k // 2
    x = torch.randn(B, H, W, cin, device="cuda").to(bf16)
    ho, wo = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
    dy = torch.randn(B, ho, wo, cout, device="cuda").to(bf16)
    fl = 2.0 * B * ho * wo * cout * cin * k * k
    t = np.zeros((reps, len(variants)))
    for r in range(reps):
        for i, v in enumerate(variants):
            select(v)
            t[r, i] = timeit(lambda: nn.conv2d_wgrad(x, dy, (k, k), st, pad))
            unselect(v)
    med = np.median(t, 0)
    tot += med * cnt
    print("%-14s x%d %5d %6d %8d | " % (name, cnt, cout, cin * k * k, B * ho * wo) + " | ".join("%7.1f us %5.0f TF (%4.1f)" % (m, fl / m / 1e6, t[:, i].min()) for i, m in enumerate(med)), flush=True)
print("total x count (ms): " + " | ".join("%.3f" % (x / 1e3) for x in tot))
