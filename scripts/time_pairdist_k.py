"""Development aid: distance kernel time vs feature width -> per-k-step cost (slope) and per-tile epilogue cost (intercept)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daliid_amd import ops_eval
nq, ng = 10000, 100000
out = torch.empty(nq, ng, device="cuda")
for prec in ("bf16x3", "bf16"):
    res = []
    for d in (512, 1024, 2048, 4096):
        q = torch.randn(nq, d, device="cuda"); g = torch.randn(ng, d, device="cuda")
        qp, gp = ops_eval.PreparedRows(q, True, prec), ops_eval.PreparedRows(g, True, prec)
        ops_eval.pairdist_prepared(qp, gp, out=out); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops_eval.pairdist_prepared(qp, gp, out=out)
        e1.record(); torch.cuda.synchronize()
        res.append((d, e0.elapsed_time(e1) / 5))
        del q, g, qp, gp
    tiles_per_cu = ((ng + 127) // 128) * ((nq + 255) // 256) / 256
    slope = (res[-1][1] - res[0][1]) / (res[-1][0] - res[0][0])          # ms per unit of d
    icpt = res[0][1] - slope * res[0][0]
    print(prec, ["d=%d: %.3f ms" % r for r in res], "| per 32-deep k-step per tile %.3f us, per-tile fixed cost %.2f us" % (slope * 32 / tiles_per_cu * 1e3, icpt / tiles_per_cu * 1e3))
