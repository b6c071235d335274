"""One-launch inference stem (dali_stem_conv_bn_maxpool: pack + conv1 + bn1 + max-pool) at batch B, with the kernel's diagnostic ablations:
    python scripts/bench_stem.py [B=500]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daliid_amd import ops_nn as nn, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 500
img = torch.randn(B, 3, 256, 128, device="cuda")
w = torch.randn(64, 7, 7, 3, device="cuda") * 0.05
sc, sh = torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda")
def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for ab in (0, 1, 2, 4, 8, 3, 7, 15):
    os.environ["DALI_STEM_ABLATE"] = str(ab); _lib.lib().dali_debug_reload_env()
    print("ablate %2d (1 no pool, 2 no output stage, 4 no MFMA, 8 no requests): %7.1f us (pack + weights + stem kernel)" % (ab, timeit(lambda: nn.stem_conv_bn_maxpool(img, w, sc, sh))), flush=True)
