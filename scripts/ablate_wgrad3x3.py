"""Development aid: 3x3 weight-gradient time of a few layers, run per library variant (normal / MFMAs compiled out / DMA issue
compiled out: temporary builds, the macro is not kept in the source)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daliid_amd import ops_nn as nn
bf16 = torch.bfloat16
for name, H, W, c in [("l1.c2", 64, 32, 64), ("l2.c2", 32, 16, 128), ("l3.c2", 16, 8, 256), ("l4.c2", 16, 8, 512)]:
    x = torch.randn(256, H, W, c, device="cuda").to(bf16); dy = torch.randn(256, H, W, c, device="cuda").to(bf16)
    f = lambda: nn.conv2d_wgrad(x, dy, (3, 3), 1, 1)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print("%s %-6s wgrad %7.1f us" % (sys.argv[1] if len(sys.argv) > 1 else "", name, e0.elapsed_time(e1) * 100))
