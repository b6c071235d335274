"""Per-block timeline of the LDS-DMA conv kernel from in-kernel s_memrealtime stamps (diagnostic)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from daliid_amd import ops_nn as nn, _lib
bf16 = torch.bfloat16
B, H, W, cin, cout, k = [int(v) for v in sys.argv[1:7]] if len(sys.argv) > 6 else (256, 64, 32, 64, 256, 1)
MODE = sys.argv[7] if len(sys.argv) > 7 else "fwd"
x = torch.randn(B, H, W, cin, device="cuda").to(bf16)
w = torch.randn(cout, k, k, cin, device="cuda").to(bf16)
dy = torch.randn(B, H, W, cout, device="cuda").to(bf16)
STATS = os.environ.get("STATS", "1") != "0"          # STATS=0: a linear layer's launch (no BatchNorm statistics in the epilogue)
run = (lambda: nn.conv2d_fwd(x, w, 1, k // 2, want_stats=STATS)) if MODE == "fwd" else (lambda: nn.conv2d_wgrad(x, dy, (k, k), 1, k // 2))
if MODE in ("lin", "lin_gelu", "lin_dgelu"):              # a ViT linear layer of B*H*W rows: plain / fc1 forward (bias + GELU + pre-activation copy) / fc2 data gradient (GELU')
    from daliid_amd import ops_vit as V
    xr, wr = x.reshape(-1, cin), w.reshape(cout, cin)
    bias = torch.randn(cout, device="cuda")
    pre = torch.randn(B * H * W, cout, device="cuda").to(bf16)
    run = {"lin": lambda: V.linear_fwd(xr, wr, bias), "lin_gelu": lambda: V.linear_fwd(xr, wr, bias, act=1, want_pre=True),
           "lin_dgelu": lambda: V.linear_dgrad(xr, wr, gelu_pre=pre)}[MODE]
if MODE in ("fused", "fused_dgrad"):                  # conv3 with the fused output stage: y = relu(acc * scale + shift + residual) + mask bits (forward) /
    xr, wr = x.reshape(-1, cin), w.reshape(cout, cin)  # a data gradient with residual + output mask
    sc, sh = torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda")
    res = torch.randn(B * H * W, cout, device="cuda").to(bf16)
    bits = torch.randint(0, 256, (B * H * W * cout // 8,), device="cuda", dtype=torch.uint8)
    run = {"fused": lambda: nn.conv1x1_fused(xr, wr, sc, sh, None, res, relu=True, want_bits=True),
           "fused_dgrad": lambda: nn.conv1x1_fused(xr, wr, None, None, None, res, relu=False, out_mask=bits)}[MODE]
for _ in range(3): run()
nblk = 1 << 16
stamps = torch.zeros(nblk, 12, device="cuda", dtype=torch.int64)
L = _lib.lib(); L.dali_debug_set_conv_stamps.argtypes = [ctypes.c_void_p]
L.dali_debug_set_conv_stamps(ctypes.c_void_p(stamps.data_ptr()))
run()
torch.cuda.synchronize()
L.dali_debug_set_conv_stamps(None)
s = stamps.cpu().numpy()
hw = s[s[:, 3] > 0][:, 10]
s = s[s[:, 3] > 0].astype(np.float64)
t0 = s[:, 0].min()
s = np.where(s > 0, (s - t0) * 10.0 / 1e3, np.nan)           # 100 MHz ticks -> us
span = np.nanmax(s[:, 3])
print("blocks stamped: %d ; kernel span %.1f us" % (len(s), span))
print("blocks started within 3 us of the first: %d ; start-time deciles (us): %s" % (int((s[:, 0] < 3.0).sum()), np.round(np.percentile(s[:, 0], [10, 30, 50, 70, 90, 100]), 1)))
if (hw != 0).any():                                  # kernels that stamp HW_ID: where did the first round sit?
    first = s[:, 0] < 3.0
    xcc, se, cu = (hw >> 32) & 0xf, (hw >> 13) & 0x7, (hw >> 8) & 0xf
    import collections
    cnt = collections.Counter(zip(xcc[first].tolist(), se[first].tolist()))
    print("first-round blocks per (XCC, SE):", dict(sorted(cnt.items())))
    print("distinct (XCC, SE, CU) in the first round: %d ; over all blocks: %d" % (len(set(zip(xcc[first].tolist(), se[first].tolist(), cu[first].tolist()))),
                                                                                   len(set(zip(xcc.tolist(), se.tolist(), cu.tolist())))))
d = s[:, 3] - s[:, 0]
print("block lifetime us: mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f  max %.2f" % (d.mean(), *np.percentile(d, [10, 50, 90]), d.max()))
names = ["start", "first tile landed", "mainloop done", "end (stores acked)", "all stores issued", "stats done", "half0 staged", "half0 stores issued",
         "half1 staged", "half1 stores issued"]
order = [0, 1, 2, 5, 6, 7, 8, 9, 4, 3]
prev = None
for k in order:
    col = s[:, k]
    if np.isnan(col).all():
        continue
    if prev is not None:
        print("  %-22s -> %-22s mean %6.2f us" % (names[prev], names[k], np.nanmean(col - s[:, prev])))
    prev = k
print("  mean concurrency = sum(lifetime)/span = %.1f blocks (256 CUs)" % (d.sum() / span))
