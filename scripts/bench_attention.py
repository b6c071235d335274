"""Attention forward / backward at the ViT-B/16 shape (batch 128, 197 tokens, 12 heads), first form vs the register-resident form, interleaved."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from daliid_amd import ops_vit as V, _lib
lib = _lib.lib()
B, T, H = int(os.environ.get("B", 128)), int(os.environ.get("T", 197)), 12
C = H * 64
qkv = torch.randn(B * T, 3 * C, device="cuda").to(torch.bfloat16)
d_out = torch.randn(B * T, C, device="cuda").to(torch.bfloat16)
def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
res = np.zeros((5, 2, 2))
for r in range(5):
    for i, v in enumerate(["1", "0"]):
        os.environ["DALI_ATT_V1"] = v; lib.dali_debug_reload_env()
        o, lse = V.attention_fwd(qkv, B, T, H)
        res[r, i, 0] = timeit(lambda: V.attention_fwd(qkv, B, T, H))
        res[r, i, 1] = timeit(lambda: V.attention_bwd(qkv, o, d_out, lse, B, T, H))
m = np.median(res, 0)
fl = 4.0 * B * H * T * T * 64
print("B=%d T=%d: forward  first form %.1f us (%.0f TFLOP/s) | register form %.1f us (%.0f TFLOP/s)" % (B, T, m[0, 0], fl / m[0, 0] / 1e6, m[1, 0], fl / m[1, 0] / 1e6))
print("              backward first form %.1f us (%.0f TFLOP/s) | register form %.1f us (%.0f TFLOP/s)" % (m[0, 1], 2.5 * fl / m[0, 1] / 1e6, m[1, 1], 2.5 * fl / m[1, 1] / 1e6))
