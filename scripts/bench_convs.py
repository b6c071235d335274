"""Per-layer conv kernel timings at the ResNet-50-ReID shapes (B=256, 256x128) vs a simple roofline."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daliid_amd import ops_nn as nn
bf16 = torch.bfloat16
B = int(os.environ.get("B", "256"))
from bench_convs_shapes import L          # (name, H, W, cin, cout, k, stride, count)
def timeit(fn, n=24):
    """fn(i): launch on operand set i.  The sets rotate so that no launch finds its operands in the 256 MB Infinity Cache (a standalone loop on ONE
    set re-launches on cache-warm tensors and reads above the HBM roof: rows at 1.10-1.18 'of roof' in the round-3 table)."""
    fn(0); fn(1); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us
tot = {"fwd": 0, "dgrad": 0, "wgrad": 0, "roof": 0}
MD = os.environ.get("MD")          # MD=<path>: also write the per-layer roofline table (profiles/rNN_conv_layers.md)
md = ["# Per-layer implicit-GEMM kernels at the ResNet-50-ReID shapes (batch %d, 256x128), standalone launches\n" % B,
      "`python scripts/bench_convs.py` on MI355X: every conv of the net as forward / data-gradient / weight-gradient (incl. its split-K reduce), HIP-event "
      "time per launch over >= 3 rotating operand sets (together larger than the 256 MB Infinity Cache), against `max(FLOP / 2.5 PFLOP/s, (input + output bytes) / 6 TB/s)`; GB/s = those algorithmic bytes / time.\n",
      "| layer | x | M=pixels | N=Cout | K | GFLOP | pass | us | TFLOP/s | GB/s | binding roof | frac of roof |", "|---|---:|---:|---:|---:|---:|---|---:|---:|---:|---|---:|"]
print("%-14s %8s | %9s %9s %9s | %8s %8s  (us; roof = max(flops/2.5PF, bytes/6TB/s) per pass)" % ("layer", "GFLOP", "fwd", "dgrad", "wgrad", "roof", "x cnt"))
for name, H, W, cin, cout, k, st, cnt in L:
    pad = k // 2
    ho_, wo_ = (H + 2 * (k // 2) - k) // st + 1, (W + 2 * (k // 2) - k) // st + 1
    nset = max(3, min(16, -(-(320 << 20) // ((B * H * W * cin + B * ho_ * wo_ * cout) * 2))))      # >= 3 sets, together beyond the Infinity Cache
    xs = [torch.randn(B, H, W, cin, device="cuda").to(bf16) for _ in range(nset)]
    x = xs[0]
    w = torch.randn(cout, k, k, cin, device="cuda").to(bf16)
    wt = torch.randn(cin, k, k, cout, device="cuda").to(bf16)
    sc = torch.rand(cin, device="cuda") + 0.5; sh = torch.randn(cin, device="cuda")
    ho, wo = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
    dys = [torch.randn(B, ho, wo, cout, device="cuda").to(bf16) for _ in range(nset)]
    dy = dys[0]
    fl = 2.0 * B * ho * wo * cout * cin * k * k
    byt = ((x.numel() // (st * st) if k == 1 else x.numel()) + dy.numel()) * 2      # a strided 1x1 touches a quarter of the input pixels
    roof = max(fl / 2.5e15, byt / 6e12) * 1e6
    tf = timeit(lambda i: nn.conv2d_fwd(xs[i % nset], w, st, pad, want_stats=True))
    if k == 1 and st == 2:
        # the plan accumulates a stride-2 1x1 (downsample) data gradient in place onto conv1's (resnet_plan.hip: block_backward): only the
        # even-even quarter of the positions is touched.  Without a residual the same call has to write zeros to the other three quarters.
        td = timeit(lambda i: nn.conv2d_dgrad(dys[i % nset], wt, (H, W), st, pad, residual=xs[i % nset], inplace=True))
    else:
        td = timeit(lambda i: nn.conv2d_dgrad(dys[i % nset], wt, (H, W), st, pad))
    tw = timeit(lambda i: nn.conv2d_wgrad(xs[i % nset], dys[i % nset], (k, k), st, pad))
    print("%-14s %8.1f | %9.1f %9.1f %9.1f | %8.1f x%d   eff fwd %.0f%% dg %.0f%% wg %.0f%%" % (name, fl / 1e9, tf, td, tw, roof, cnt, 100 * roof / tf, 100 * roof / td, 100 * roof / tw))
    tot["fwd"] += tf * cnt; tot["dgrad"] += td * cnt; tot["wgrad"] += tw * cnt; tot["roof"] += roof * cnt
    bound = "MFMA" if fl / 2.5e15 >= byt / 6e12 else "HBM"
    for pname, t in (("fwd", tf), ("dgrad", td), ("wgrad", tw)):
        md.append("| %s | %d | %d | %d | %d | %.1f | %s | %.1f | %.0f | %.0f | %s | %.2f |" % (name, cnt, B * ho * wo, cout, cin * k * k, fl / 1e9, pname, t, fl / t / 1e6,
                                                                                          byt / t / 1e3, bound, roof / t))
print("totals (ms): fwd %.2f dgrad %.2f wgrad %.2f ; roof per pass %.2f" % (tot["fwd"] / 1e3, tot["dgrad"] / 1e3, tot["wgrad"] / 1e3, tot["roof"] / 1e3))
if MD:
    md.append("\nTotals over the net (x count): forward %.2f ms, data gradient %.2f ms, weight gradient %.2f ms; sum of the per-layer roofs %.2f ms per pass."
              % (tot["fwd"] / 1e3, tot["dgrad"] / 1e3, tot["wgrad"] / 1e3, tot["roof"] / 1e3))
    open(MD, "w").write("\n".join(md) + "\n")
