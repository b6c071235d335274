"""What the split-K reduce's ~32 us is made of: the same launch behind a slab fill, behind a fill + a tiny kernel, replayed on clean slabs,
behind a 600 MB flush, and torch's own column sum of the same bytes (scripts/bench_reduce.py is the per-shape table)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from daliid_amd import _lib
lib = _lib.lib()
lib.dali_debug_splitk_reduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int]
st = torch.cuda.current_stream().cuda_stream
flush = torch.empty(150 * 1024 * 1024, device="cuda", dtype=torch.float32)
tiny = torch.zeros(256, device="cuda")
def ev(): return torch.cuda.Event(enable_timing=True)
for elems, sp in ((262144, 32), (65536, 128), (1048576, 8), (589824, 32), (16384, 256)):
    slabs = torch.empty(sp * elems, device="cuda", dtype=torch.float32)
    out = torch.empty(elems, device="cuda", dtype=torch.float32)
    red = lambda: lib.dali_debug_splitk_reduce(st, slabs.data_ptr(), out.data_ptr(), elems, sp, 0)
    res = {}
    def run(name, pre, n=7):
        ts = []
        for _ in range(n):
            pre(); e0, e1 = ev(), ev(); e0.record(); red(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
        res[name] = np.median(ts)
    run("fill", lambda: slabs.fill_(1.0))
    run("fill+sync", lambda: (slabs.fill_(1.0), torch.cuda.synchronize()))
    run("fill+tiny", lambda: (slabs.fill_(1.0), tiny.add_(1.0)))
    run("replay", lambda: None)
    run("flush", lambda: flush.fill_(0.0))
    run("flush+tiny", lambda: (flush.fill_(0.0), tiny.add_(1.0)))
    # tiny kernel behind the fill: what a kernel boundary after 34 MB of fresh stores costs by itself
    ts = []
    for _ in range(7):
        slabs.fill_(1.0); e0, e1 = ev(), ev(); e0.record(); tiny.add_(1.0); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    res["tiny after fill"] = np.median(ts)
    ts = []
    for _ in range(7):
        torch.cuda.synchronize(); e0, e1 = ev(), ev(); e0.record(); tiny.add_(1.0); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    res["tiny idle"] = np.median(ts)
    v = slabs.view(sp, elems)
    ts = []
    for _ in range(7):
        slabs.fill_(1.0); e0, e1 = ev(), ev(); e0.record(); o2 = v.sum(0); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    res["torch sum(0) after fill"] = np.median(ts)
    # back-to-back launches: 10 reduces between one pair of events
    slabs.fill_(1.0); e0, e1 = ev(), ev(); e0.record()
    for _ in range(10): red()
    e1.record(); torch.cuda.synchronize(); res["10 back to back / 10"] = e0.elapsed_time(e1) * 100
    print("elems %8d splits %4d (%.1f MB): " % (elems, sp, (sp + 1) * elems * 4 / 1e6) + "; ".join("%s %.1f" % kv for kv in res.items()), flush=True)
