"""Development aid: forward time of a few large conv layers (the wave-grid kernels), run once per library variant to see which
resource bounds the main loop.  The variants were temporary builds of conv.hip with the MFMAs / the fragment reads / the DMA issue
of the k-loop compiled out under a macro (not kept in the source); results in DESIGN.md section 4."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daliid_amd import ops_nn as nn
bf16 = torch.bfloat16
L = [("l4.c2", 16, 8, 512, 512, 3), ("l3.c2", 16, 8, 256, 256, 3), ("l4.c1", 16, 8, 2048, 512, 1), ("l4.c3", 16, 8, 512, 2048, 1), ("l2.c2", 32, 16, 128, 128, 3)]
for name, H, W, cin, cout, k in L:
    x = torch.randn(256, H, W, cin, device="cuda").to(bf16)
    w = torch.randn(cout, k, k, cin, device="cuda").to(bf16)
    f = lambda: nn.conv2d_fwd(x, w, 1, k // 2, want_stats=True)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    ksteps = k * k * cin // 32
    print("%s %-6s fwd %7.1f us  = %.3f us per k-step of 32 (%d k-steps)" % (sys.argv[1] if len(sys.argv) > 1 else "", name, us, us / ksteps, ksteps))
