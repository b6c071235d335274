"""Yardstick, not product: the configs[4] cosine distance matrix (10k x 100k x 2048) the way the reference computes it on a GPU -- fp32 features,
F.normalize, 1 - q @ g.T (torchreid's cosine metric) -- through PyTorch-ROCm's own GEMM, fp32 and (not fp32-grade) bf16."""
import torch, torch.nn.functional as F
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
q = torch.randn(10000, 2048, device="cuda"); g = torch.randn(100000, 2048, device="cuda")
out = torch.empty(10000, 100000, device="cuda")
def fp32():
    qn, gn = F.normalize(q, dim=1), F.normalize(g, dim=1)
    torch.mm(qn, gn.t(), out=out); out.neg_().add_(1.0)
def bf16():
    qn, gn = F.normalize(q, dim=1).bfloat16(), F.normalize(g, dim=1).bfloat16()
    out.copy_(torch.mm(qn, gn.t())); out.neg_().add_(1.0)
for name, fn in (("fp32 (the reference's precision)", fp32), ("bf16 operands (1e-3 errors: not what the reference computes)", bf16)):
    ms = timeit(fn)
    print("torch cosine distmat 10k x 100k x 2048, %s: %.2f ms = %.1f Gpairs/s" % (name, ms, 1e9 / ms / 1e6), flush=True)
