"""Yardstick, not product: MIOpen (torch.nn.functional.conv2d, bf16, channels_last) on the 3x3 layers of ResNet-50-ReID at batch 256: forward, and
input / weight gradients through torch.autograd.grad.  Printed beside nothing: compare with profiles/r03_conv_layers.md."""
import torch, torch.nn.functional as F
bf16 = torch.bfloat16
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
import os
shapes = [("l1.c2", 64, 32, 64, 64, 1), ("l2.c2", 32, 16, 128, 128, 1), ("l3.c2", 16, 8, 256, 256, 1), ("l4.c2", 16, 8, 512, 512, 1), ("l3.c2(s2)", 32, 16, 256, 256, 2)]
if os.environ.get("ONLY"): shapes = [s for s in shapes if s[0] in os.environ["ONLY"].split(",")]
print("%-10s %9s %9s %9s   (us; GFLOP %s)" % ("layer", "fwd", "dgrad", "wgrad", "per pass"))
for name, H, W, cin, cout, st in shapes:
    x = torch.randn(256, cin, H, W, device="cuda", dtype=bf16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = torch.randn(cout, cin, 3, 3, device="cuda", dtype=bf16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = F.conv2d(x, w, stride=st, padding=1)
    dy = torch.randn_like(y)
    tf = timeit(lambda: F.conv2d(x, w, stride=st, padding=1))
    td = timeit(lambda: torch.autograd.grad(F.conv2d(x, w, stride=st, padding=1), x, dy)) - tf
    tw = timeit(lambda: torch.autograd.grad(F.conv2d(x, w, stride=st, padding=1), w, dy)) - tf
    gf = 2.0 * 256 * y.shape[2] * y.shape[3] * cout * cin * 9 / 1e9
    print("%-10s %9.1f %9.1f %9.1f   (%.1f)" % (name, tf, td, tw, gf), flush=True)
