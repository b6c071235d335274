"""Per-block timeline of the LDS-DMA conv kernel from in-kernel s_memrealtime stamps (diagnostic)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from daliid_amd import ops_nn as nn, _lib
bf16 = torch.bfloat16
B, H, W, cin, cout, k = [int(v) for v in sys.argv[1:7]] if len(sys.argv) > 6 else (256, 64, 32, 64, 256, 1)
x = torch.randn(B, H, W, cin, device="cuda").to(bf16)
w = torch.randn(cout, k, k, cin, device="cuda").to(bf16)
for _ in range(3): nn.conv2d_fwd(x, w, 1, k // 2, want_stats=True)
nblk = 1 << 16
stamps = torch.zeros(nblk, 4, device="cuda", dtype=torch.int64)
L = _lib.lib(); L.dali_debug_set_conv_stamps.argtypes = [ctypes.c_void_p]
L.dali_debug_set_conv_stamps(ctypes.c_void_p(stamps.data_ptr()))
nn.conv2d_fwd(x, w, 1, k // 2, want_stats=True)
torch.cuda.synchronize()
L.dali_debug_set_conv_stamps(None)
s = stamps.cpu().numpy()
s = s[s[:, 0] > 0]
t0 = s[:, 0].min()
s = (s - t0) * 10.0 / 1e3           # 100 MHz ticks -> us
print("blocks stamped: %d ; kernel span %.1f us" % (len(s), s[:, 3].max()))
d = s[:, 3] - s[:, 0]
print("block lifetime us: mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f  max %.2f" % (d.mean(), *np.percentile(d, [10, 50, 90]), d.max()))
print("  start->first tile landed: mean %.2f us ; mainloop: mean %.2f ; epilogue+stores: mean %.2f" %
      ((s[:, 1] - s[:, 0]).mean(), (s[:, 2] - s[:, 1]).mean(), (s[:, 3] - s[:, 2]).mean()))
print("  mean concurrency = sum(lifetime)/span = %.1f blocks (256 CUs)" % (d.sum() / s[:, 3].max()))
edges = np.linspace(0, s[:, 3].max(), 11)
for a, b in zip(edges[:-1], edges[1:]):
    alive = ((s[:, 0] < b) & (s[:, 3] > a)).sum()
    started = ((s[:, 0] >= a) & (s[:, 0] < b)).sum()
    print("  t %6.1f-%6.1f us: %5d blocks started, %5d alive at some point" % (a, b, started, alive))
