"""Per-phase shader cycles of igemm_wgrad3x3_pp_kernel (waves 0 and 4), from its diagnostic stamps: python scripts/pp_phases.py B H W cin cout"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from daliid_amd import ops_nn as nn, _lib
bf16 = torch.bfloat16
B, H, W, cin, cout = [int(v) for v in sys.argv[1:6]]
x = torch.randn(B, H, W, cin, device="cuda").to(bf16)
dy = torch.randn(B, H, W, cout, device="cuda").to(bf16)
run = lambda: nn.conv2d_wgrad(x, dy, (3, 3), 1, 1)
for _ in range(3): run()
stamps = torch.zeros(1 << 14, 12, device="cuda", dtype=torch.int64)
L = _lib.lib(); L.dali_debug_set_conv_stamps.argtypes = [ctypes.c_void_p]
L.dali_debug_set_conv_stamps(ctypes.c_void_p(stamps.data_ptr()))
run(); torch.cuda.synchronize()
L.dali_debug_set_conv_stamps(None)
s = stamps.cpu().numpy()
s = s[s[:, 3] > 0]
P = B * H * W
print("blocks %d; lifetime mean %.1f us" % (len(s), ((s[:, 3] - s[:, 0]) * 10.0 / 1e3).mean()))
names = ["DMA issue", "fragment reads + settle", "DMA wait (grp 1) / lgkm", "barrier 1", "multiply", "DMA wait (grp 0) + barrier 2"]
for w, base in (("wave 0 (group 0)", 5), ("wave 4 (group 1)", 8)):
    v = s[:, base:base + 3].astype(np.uint64)
    parts = np.stack([v[:, 0] & 0xffffffff, v[:, 0] >> 32, v[:, 1] & 0xffffffff, v[:, 1] >> 32, v[:, 2] & 0xffffffff, v[:, 2] >> 32], 1).astype(np.float64)
    tot = parts.sum(1).mean()
    print(w + ": total %.0f cycles per block" % tot)
    for n, c in zip(names, parts.mean(0)):
        print("    %-30s %9.0f cycles  %5.1f %%" % (n, c, 100 * c / tot))
