"""Yardstick, not product: what the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) reaches on the plain-GEMM shapes of the plan (1x1 convolutions as
[pixels x Cin] x [Cin x Cout], their weight gradients as [Cout x pixels] x [pixels x Cin], the ViT linears), bf16 in, bf16 out, no epilogue.  The library's
own kernels do more per launch (BatchNorm statistics, residual, masks, split-K slabs); the table says how far the bare main loops are from a tuned one."""
import torch
bf16 = torch.bfloat16
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [("l4.c3 fwd", 32768, 2048, 512), ("l4.c1 fwd", 32768, 512, 2048), ("l4.ds fwd", 32768, 2048, 1024), ("l3.c3 fwd", 32768, 1024, 256),
          ("l3.c1 fwd", 32768, 256, 1024), ("l2.c3 fwd", 131072, 512, 128), ("l2.c1 fwd", 131072, 128, 512), ("l1.c3 fwd", 524288, 256, 64),
          ("l4.c3 wgrad", 2048, 512, 32768), ("l4.c1 wgrad", 512, 2048, 32768), ("l3.c3 wgrad", 1024, 256, 32768), ("l2.c3 wgrad", 512, 128, 131072),
          ("vit fc1", 25216, 3072, 768), ("vit fc2", 25216, 768, 3072), ("vit qkv", 25216, 2304, 768), ("vit proj", 25216, 768, 768),
          ("vit fc1 wgrad", 3072, 768, 25216)]
print("%-14s %8s %6s %6s %9s %9s" % ("shape", "M", "N", "K", "us", "TFLOP/s"))
for name, M, N, K in shapes:
    a = torch.randn(M, K, device="cuda").to(bf16); b = torch.randn(K, N, device="cuda").to(bf16)
    if "wgrad" in name:                       # A^T B with the long dimension contracted: operands as the plan holds them ([pixels][channels])
        at = torch.randn(K, M, device="cuda").to(bf16)
        t = timeit(lambda: torch.matmul(at.t(), b))
    else:
        bt = torch.randn(N, K, device="cuda").to(bf16)           # weights [Cout][Cin] as the plan holds them
        t = timeit(lambda: torch.matmul(a, bt.t()))
    print("%-14s %8d %6d %6d %9.1f %9.0f" % (name, M, N, K, t, 2.0 * M * N * K / t / 1e6), flush=True)
