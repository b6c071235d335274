"""HBM probes: write-only (fill), read-only (sum), copy; sizes beyond the 256 MiB Infinity Cache."""
import torch
def t(fn, n=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (268, 1024, 4096):
    n = mb * 1024 * 1024 // 2
    a = torch.empty(n, device="cuda", dtype=torch.bfloat16); b = torch.empty_like(a)
    a.normal_()
    w = t(lambda: a.fill_(1.0)); r = t(lambda: a.view(torch.int16).max()); c = t(lambda: b.copy_(a))
    print("%5d MiB: fill %.2f TB/s   read(max) %.2f TB/s   copy %.2f TB/s (r+w)" % (mb, n * 2 / w / 1e12, n * 2 / r / 1e12, 2 * n * 2 / c / 1e12))
