"""rank_eval kernel time: worst case (random distances) vs well-separated identities."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from daliid_amd import ops_eval
nq, ng = 10000, 100000
rng = np.random.default_rng(12)
g_pids = np.repeat(np.arange(1000), 100); q_pids = np.repeat(np.arange(1000), 10)
g_cams = rng.integers(0, 6, ng); q_cams = rng.integers(0, 6, nq)
qp_, gp_ = ops_eval.factorize_ids(q_pids, g_pids); qc_, gc_ = ops_eval.factorize_ids(q_cams, g_cams)
codes = [torch.from_numpy(a).cuda() for a in (qp_, gp_, qc_, gc_)]
def t(dm):
    ops_eval.rank_eval_codes(dm, *codes); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): r = ops_eval.rank_eval_codes(dm, *codes)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 3, r
dm = torch.rand(nq, ng, device="cuda") * 2
ms, r = t(dm); print("random distances      : %.2f ms  mAP %s" % (ms, str(r)[:60]))
same = torch.from_numpy(q_pids).cuda()[:, None] == torch.from_numpy(g_pids).cuda()[None, :]
dm2 = torch.where(same, dm * 0.2, 0.5 + dm * 0.75)
ms, r = t(dm2); print("separated identities  : %.2f ms  mAP %s" % (ms, str(r)[:60]))
