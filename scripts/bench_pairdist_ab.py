"""Distance kernel at configs[4] (10k x 100k x 2048): environment-switch variants interleaved in one process.
usage: python scripts/bench_pairdist_ab.py "DALI_PAIRDIST_LINES=0" "DALI_PAIRDIST_LINES=1" "DALI_PAIRDIST_XSTAGGER=2" ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from daliid_amd import ops_eval, _lib
lib = _lib.lib()
nq, ng, d = 10000, 100000, 2048
variants = sys.argv[1:] or ["DALI_PAIRDIST_LINES=0", "DALI_PAIRDIST_LINES=1"]
keys = sorted({kv.split("=")[0] for v in variants for kv in v.split()})
out = torch.empty(nq, ng, device="cuda")
q = torch.randn(nq, d, device="cuda"); g = torch.randn(ng, d, device="cuda")
for prec in ("bf16x3", "bf16"):
    qp, gp = ops_eval.PreparedRows(q, True, prec), ops_eval.PreparedRows(g, True, prec)
    t = np.zeros((5, len(variants)))
    for r in range(5):
        for i, v in enumerate(variants):
            for k in keys: os.environ.pop(k, None)
            for kv in v.split(): os.environ[kv.split("=")[0]] = kv.split("=")[1]
            lib.dali_debug_reload_env()
            ops_eval.pairdist_prepared(qp, gp, out=out); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): ops_eval.pairdist_prepared(qp, gp, out=out)
            e1.record(); torch.cuda.synchronize()
            t[r, i] = e0.elapsed_time(e1) / 3
    m = np.median(t, 0)
    for v, x in zip(variants, m):
        print("%s  %-40s %.3f ms  %.1f Gpairs/s" % (prec, v, x, nq * ng / x / 1e6), flush=True)
