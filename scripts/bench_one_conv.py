"""One conv shape, fwd / dgrad / wgrad timings (us): python scripts/bench_one_conv.py H W cin cout k stride [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daliid_amd import ops_nn as nn
bf16 = torch.bfloat16
H, W, cin, cout, k, st = [int(v) for v in sys.argv[1:7]]
B = int(sys.argv[7]) if len(sys.argv) > 7 else 256
pad = k // 2
x = torch.randn(B, H, W, cin, device="cuda").to(bf16)
w = torch.randn(cout, k, k, cin, device="cuda").to(bf16)
wt = torch.randn(cin, k, k, cout, device="cuda").to(bf16)
ho, wo = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
dy = torch.randn(B, ho, wo, cout, device="cuda").to(bf16)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
fl = 2.0 * B * ho * wo * cout * cin * k * k
tf = timeit(lambda: nn.conv2d_fwd(x, w, st, pad, want_stats=True))
td = timeit(lambda: nn.conv2d_dgrad(dy, wt, (H, W), st, pad))
tw = timeit(lambda: nn.conv2d_wgrad(x, dy, (k, k), st, pad))
print("%s env NP=%s: fwd %.1f us (%.0f TF/s)  dgrad %.1f us  wgrad %.1f us" % (sys.argv[1:7], "-", tf, fl / tf / 1e6, td, tw))
