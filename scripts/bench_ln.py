"""LayerNorm backward / forward alone at the ViT-B/16 shape (batch 128 x 197 tokens x 768), variants interleaved in ONE process:
    python scripts/bench_ln.py "" ...      (each argument: a space-separated list of NAME=VALUE switches, may be empty)
GB/s = algorithmic bytes (g, x, [add] read + dx written, bf16) / time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from daliid_amd import ops_vit as V, _lib
bf16 = torch.bfloat16
rows, C = 128 * 197, 768
variants = sys.argv[1:] or [""]
lib = _lib.lib()
def select(v):
    for kv in (v.split() if v else []):
        k, val = kv.split("="); os.environ[k] = val
    lib.dali_debug_reload_env()
def unselect(v):
    for kv in (v.split() if v else []): os.environ.pop(kv.split("=")[0], None)
def timeit(fn, n=30):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
x = torch.randn(rows, C, device="cuda").to(bf16); g = torch.randn(rows, C, device="cuda").to(bf16); add = torch.randn(rows, C, device="cuda").to(bf16)
gamma = torch.rand(C, device="cuda") + 0.5; beta = torch.randn(C, device="cuda")
y, mean, rstd = V.layernorm_fwd(x, gamma, beta)
ref = None
for name, fn, nbytes in (("bwd", lambda: V.layernorm_bwd(g, x, gamma, mean, rstd), 3 * rows * C * 2),
                         ("bwd+add", lambda: V.layernorm_bwd(g, x, gamma, mean, rstd, add), 4 * rows * C * 2),
                         ("fwd", lambda: V.layernorm_fwd(x, gamma, beta), 2 * rows * C * 2)):
    t = np.zeros((5, len(variants)))
    outs = []
    for r in range(5):
        for i, v in enumerate(variants):
            select(v); t[r, i] = timeit(fn)
            if r == 0: outs.append(fn()[0].float())
            unselect(v)
    med = np.median(t, 0)
    same = all(torch.equal(o, outs[0]) for o in outs)
    print("%-8s " % name + " | ".join("%-34s %6.1f us %5.0f GB/s" % (v or "default", m, nbytes / m / 1e3) for v, m in zip(variants, med)) + ("  [outputs identical]" if same else "  [OUTPUTS DIFFER]"), flush=True)
