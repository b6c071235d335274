"""Time conv2d_fwd with and without the BN statistics epilogue for a few layer shapes (HIP events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daliid_amd import ops_nn as nn
bf16 = torch.bfloat16
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, (B, H, W, cin, cout, k) in {"l1.c3": (256, 64, 32, 64, 256, 1), "l1.c1": (256, 64, 32, 256, 64, 1), "l2.c3": (256, 32, 16, 128, 512, 1),
                                      "l3.c3": (256, 16, 8, 256, 1024, 1), "l4.c3": (256, 16, 8, 512, 2048, 1), "l2.c2": (256, 32, 16, 128, 128, 3)}.items():
    x = torch.randn(B, H, W, cin, device="cuda").to(bf16)
    w = torch.randn(cout, k, k, cin, device="cuda").to(bf16)
    a = t(lambda: nn.conv2d_fwd(x, w, 1, k // 2, want_stats=True))
    b = t(lambda: nn.conv2d_fwd(x, w, 1, k // 2, want_stats=False))
    print("%-6s with stats %7.1f us   without %7.1f us" % (name, a, b))
