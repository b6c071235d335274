"""The split-K reduce alone at the slab shapes of the ResNet-50-ReID weight gradients (batch 256):
    python scripts/bench_reduce.py [--reps 5]
The slabs are rewritten by a fill kernel before every timed launch (as the weight-gradient kernel leaves them: freshly written, not read),
each launch is timed by its own pair of events.  GB/s = (splits + 1) x elems x 4 bytes / time."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from daliid_amd import _lib
from bench_convs_shapes import L
B = int(os.environ.get("B", "256"))
args = sys.argv[1:]
reps, waves = 5, []
while args:
    a = args.pop(0)
    if a == "--reps": reps = int(args.pop(0))
    else: waves.append(int(a))
waves = waves or [0]
lib = _lib.lib()
lib.dali_debug_wgrad_splits.restype = ctypes.c_int
lib.dali_debug_splitk_reduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int]
def halo_w(k, st, wo):
    return wo if (k == 3 and st == 1 and wo in (8, 16, 32)) else 0
tot, totb = np.zeros(len(waves)), 0.0
print("%-14s %3s %9s %4s %7s | " % ("layer", "x", "elems", "spl", "MB") + " | ".join("W=%-16d" % w for w in waves))
for name, H, W, cin, cout, k, st, cnt in L:
    pad = k // 2
    ho, wo = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
    P, elems = B * ho * wo, cout * cin * k * k
    sp = lib.dali_debug_wgrad_splits(cout, cin * k * k, P, k * k, halo_w(k, st, wo))
    if sp < 2:
        print("%-14s x%d %9d %4d   (no reduce)" % (name, cnt, elems, sp)); continue
    slabs = torch.empty(sp * elems, device="cuda", dtype=torch.float32)
    out = torch.empty(elems, device="cuda", dtype=torch.float32)
    mb = (sp + 1) * elems * 4 / 1e6
    t = np.zeros((reps, len(waves)))
    st_ptr = torch.cuda.current_stream().cuda_stream
    for r in range(reps):
        for i, w in enumerate(waves):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            slabs.fill_(1.0)
            e0.record()
            rc = lib.dali_debug_splitk_reduce(st_ptr, slabs.data_ptr(), out.data_ptr(), elems, sp, 0)
            e1.record(); torch.cuda.synchronize()
            assert rc == 0
            t[r, i] = e0.elapsed_time(e1) * 1e3
        if r == 0: assert float(out.min()) == sp and float(out.max()) == sp
    med = np.median(t, 0)
    tot += med * cnt; totb += mb * cnt
    print("%-14s x%d %9d %4d %7.1f | " % (name, cnt, elems, sp, mb) + " | ".join("%6.1f us %5.0f GB/s" % (m, mb / m * 1e3) for m in med), flush=True)
print("total x count: %.0f MB; ms: " % totb + " | ".join("%.3f" % (x / 1e3) for x in tot))
