"""Stride-2 data gradients of the net, merged launch vs four launches, interleaved in one process."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from daliid_amd import ops_nn as nn, _lib
bf16 = torch.bfloat16
lib = _lib.lib()
def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, H, W, cin, cout in [("l2.c2(s2)", 64, 32, 128, 128), ("l3.c2(s2)", 32, 16, 256, 256)]:
    dy = torch.randn(256, H // 2, W // 2, cout, device="cuda").to(bf16)
    wt = torch.randn(cin, 3, 3, cout, device="cuda").to(bf16)
    x = torch.randn(256, H, W, cin, device="cuda").to(bf16)
    w = torch.randn(cout, 3, 3, cin, device="cuda").to(bf16)
    t = np.zeros((5, 3))
    for r in range(5):
        for i, v in enumerate(["0", "1"]):
            os.environ["DALI_DGRAD_MERGE"] = v; lib.dali_debug_reload_env()
            t[r, i] = timeit(lambda: nn.conv2d_dgrad(dy, wt, (H, W), 2, 1))
        t[r, 2] = timeit(lambda: nn.conv2d_fwd(x, w, 2, 1, want_stats=True))
    m = np.median(t, 0)
    print("%-10s dgrad 4 launches %.1f us | merged %.1f us | forward %.1f us" % (name, m[0], m[1], m[2]))
