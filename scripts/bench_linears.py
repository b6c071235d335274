"""Per-GEMM timing of one ViT-B/16 block's linear layers at configs[3] (batch 128 x 197 tokens = 25216 rows): the 4 forward GEMMs, the 4
data gradients and the 4 weight gradients with the epilogues the plan gives them.  HIP-event time per launch (20 launches after 3 warm-ups).
The kernel choice follows the library's environment switches (DALI_CONV_CFG, DALI_CONV_K64), read once per process: run it once
per setting.  MD=path appends a markdown table."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daliid_amd import ops_vit as V

bf16 = torch.bfloat16
rows = int(os.environ.get("ROWS", 25216))
D = 768
g = torch.Generator(device="cuda").manual_seed(1)
rn = lambda *s: torch.randn(*s, device="cuda", generator=g).to(bf16)
x768, x2304, x3072 = rn(rows, D), rn(rows, 3 * D), rn(rows, 4 * D)
w = {(n, k): (torch.randn(n, k, device="cuda", generator=g) / k ** 0.5).to(bf16) for n, k in [(2304, 768), (768, 768), (3072, 768), (768, 3072), (768, 2304)]}
b = {n: torch.randn(n, device="cuda", generator=g) for n in (768, 2304, 3072)}
rs = torch.ones(rows, device="cuda")

cases = [
    ("qkv fwd    768->2304 bias", 768, 2304, lambda: V.linear_fwd(x768, w[(2304, 768)], b[2304])),
    ("proj fwd   768->768  bias+rowscale+res", 768, 768, lambda: V.linear_fwd_scaled(x768, w[(768, 768)], rs, bias=b[768], residual=x768)),
    ("fc1 fwd    768->3072 bias+gelu+pre", 768, 3072, lambda: V.linear_fwd(x768, w[(3072, 768)], b[3072], act=1, want_pre=True)),
    ("fc2 fwd    3072->768 bias+rowscale+res", 3072, 768, lambda: V.linear_fwd_scaled(x3072, w[(768, 3072)], rs, bias=b[768], residual=x768)),
    ("d_fc2      768->3072 gelu'", 768, 3072, lambda: V.linear_dgrad(x768, w[(3072, 768)], gelu_pre=x3072)),       # wt = [K_out][N_in]
    ("d_fc1      3072->768", 3072, 768, lambda: V.linear_dgrad(x3072, w[(768, 3072)])),
    ("d_proj     768->768", 768, 768, lambda: V.linear_dgrad(x768, w[(768, 768)])),
    ("d_qkv      2304->768", 2304, 768, lambda: V.linear_dgrad(x2304, w[(768, 2304)])),
    ("wgrad qkv  768x2304", 768, 2304, lambda: V.linear_wgrad(x768, x2304)),
    ("wgrad proj 768x768", 768, 768, lambda: V.linear_wgrad(x768, x768)),
    ("wgrad fc1  768x3072", 768, 3072, lambda: V.linear_wgrad(x768, x3072)),
    ("wgrad fc2  3072x768", 3072, 768, lambda: V.linear_wgrad(x3072, x768)),
]
out = []
tot = 0.0
for name, K, N, fn in cases:
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    tf = 2.0 * rows * K * N / us / 1e6
    tot += us
    out.append((name, us, tf))
    print("%-42s %8.1f us  %7.1f TFLOP/s" % (name, us, tf))
print("one block: %.1f us ; x12 = %.2f ms" % (tot, tot * 12 / 1e3))
if os.environ.get("MD"):
    with open(os.environ["MD"], "a") as f:
        f.write("\n`%s` (rows %d)\n\n| GEMM | us | TFLOP/s | frac of 2.5 PF |\n|---|---:|---:|---:|\n" % (os.environ.get("TAG", "default"), rows))
        for name, us, tf in out: f.write("| %s | %.1f | %.0f | %.2f |\n" % (name, us, tf, tf / 2500.0))
        f.write("\none block %.1f us, x12 = %.2f ms\n" % (tot, tot * 12 / 1e3))
