#!/usr/bin/env python3
"""Per-kernel ms/step table from a rocprofv3 --kernel-trace --stats run: python scripts/kstats.py <dir> <steps_in_process> [top]"""
import csv, glob, sys
d, steps = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
f = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
print("sum over kernels: %.2f ms/step" % (tot / steps))
fam = {}
for r in rows:
    n = r["Name"]
    k = ("gemm" if "igemm" in n else "bn" if ("bn_" in n or "maxpool" in n or "head_pool" in n or "bn1d" in n) else
         "reduce" if ("splitk" in n or "reduce_partials" in n or "colsum" in n or "finish_sum" in n) else "bnlin" if "bnlin" in n else "other")
    fam[k] = fam.get(k, 0.0) + float(r["TotalDurationNs"]) / 1e6 / steps
print("families (ms/step):", {k: round(v, 3) for k, v in sorted(fam.items())})
for r in rows[:top]:
    t = float(r["TotalDurationNs"]) / 1e6
    print("%-100s calls/step %6.1f  ms/step %7.3f  avg us %7.1f" % (r["Name"][:100], int(r["Calls"]) / steps, t / steps, float(r["AverageNs"]) / 1e3))
