"""layer1 conv2 (64 -> 64, 3x3) forward / data gradient: halo-patch kernel vs the tap-gather kernel, interleaved in one process."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from daliid_amd import ops_nn as nn, _lib
bf16 = torch.bfloat16
lib = _lib.lib()
B, H, W, C = 256, 64, 32, 64
x = torch.randn(B, H, W, C, device="cuda").to(bf16)
w = torch.randn(C, 3, 3, C, device="cuda").to(bf16)
dy = torch.randn(B, H, W, C, device="cuda").to(bf16)
def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
t = np.zeros((5, 2, 2))
for r in range(5):
    for i, v in enumerate(["0", "1"]):
        os.environ["DALI_CONV_HALO64"] = v; lib.dali_debug_reload_env()
        t[r, i, 0] = timeit(lambda: nn.conv2d_fwd(x, w, 1, 1, want_stats=True))
        t[r, i, 1] = timeit(lambda: nn.conv2d_dgrad(dy, w, (H, W), 1, 1))
m = np.median(t, 0)
print("l1.c2 forward: tap gather %.1f us | halo patch %.1f us ; data gradient: %.1f us | %.1f us" % (m[0, 0], m[1, 0], m[0, 1], m[1, 1]))
