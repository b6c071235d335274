"""Development aid: run one conv layer's forward / dgrad / wgrad a few times (for rocprofv3 --kernel-trace --stats).
usage: one_layer.py H W cin cout k stride [fwd|dgrad|wgrad]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daliid_amd import ops_nn as nn
H, W, cin, cout, k, st = map(int, sys.argv[1:7]); what = sys.argv[7]
bf16 = torch.bfloat16; pad = k // 2; B = 256
x = torch.randn(B, H, W, cin, device="cuda").to(bf16)
w = torch.randn(cout, k, k, cin, device="cuda").to(bf16)
wt = torch.randn(cin, k, k, cout, device="cuda").to(bf16)
ho, wo = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
dy = torch.randn(B, ho, wo, cout, device="cuda").to(bf16)
f = {"fwd": lambda: nn.conv2d_fwd(x, w, st, pad, want_stats=True), "dgrad": lambda: nn.conv2d_dgrad(dy, wt, (H, W), st, pad),
     "wgrad": lambda: nn.conv2d_wgrad(x, dy, (k, k), st, pad)}[what]
for _ in range(10): f()
torch.cuda.synchronize()
