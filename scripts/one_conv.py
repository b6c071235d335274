"""Run fwd/dgrad/wgrad of one conv shape a few times (for rocprofv3 --pmc runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daliid_amd import ops_nn as nn
bf16 = torch.bfloat16
B, H, W, cin, cout, k, st = 256, 16, 8, 512, 512, 3, 1
if len(sys.argv) > 1: B, H, W, cin, cout, k, st = [int(v) for v in sys.argv[1:8]]
pad = k // 2
x = torch.randn(B, H, W, cin, device="cuda").to(bf16)
w = torch.randn(cout, k, k, cin, device="cuda").to(bf16)
wt = torch.randn(cin, k, k, cout, device="cuda").to(bf16)
ho, wo = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
dy = torch.randn(B, ho, wo, cout, device="cuda").to(bf16)
for _ in range(3):
    nn.conv2d_fwd(x, w, st, pad, want_stats=True)
    nn.conv2d_dgrad(dy, wt, (H, W), st, pad)
    nn.conv2d_wgrad(x, dy, (k, k), st, pad)
torch.cuda.synchronize()
