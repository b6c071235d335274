"""Per-launch table of the GEMM kernels INSIDE the train step (bench.py's event-bracketed profile steps):
    DALI_GEMM_PROFILE_DUMP=gpurun_out/gemm_launches.csv python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-distance
    python scripts/gemm_launch_table.py gpurun_out/gemm_launches.csv [steps=3]
Groups launches by signature, prints count per step, mean us, TFLOP/s and the share of the GEMM time, sorted by total time."""
import sys, collections
path = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
groups = collections.OrderedDict()
for line in open(path):
    parts = line.strip().split(",")
    if len(parts) < 5: continue
    sig = ",".join(parts[1:-2]); us = float(parts[-2].split("=")[1]); gf = float(parts[-1].split("=")[1])
    g = groups.setdefault(sig, [0, 0.0, 0.0]); g[0] += 1; g[1] += us; g[2] += gf
tot = sum(g[1] for g in groups.values())
def roof_us(sig, gf):
    """max(FLOPs / 2.5 PFLOP/s, algorithmic bytes / 6 TB/s): operand + output (+ residual, + mask bits) bytes, weights left out"""
    f = dict(kv.split("=") for kv in sig.split(",")[1:]); kind = sig.split(",")[0]
    Cm, K, P, taps, stride = (int(f[k]) for k in ("Cm", "K", "P", "taps", "stride"))
    cin = K // taps
    if kind == "wgrad": b = P * (cin * (stride * stride if taps == 1 else 1) + Cm) * 2      # a strided 1x1 reads a quarter of x: counted whole (lines)
    else: b = P * cin * 2 * (stride * stride if kind == "fwd" and stride > 1 else 1) + P * Cm * 2 * (1 + int(f["res"])) + P * Cm // 8 * int(f["mask"])
    return max(gf * 1e9 / 2.5e15, b / 6e12) * 1e6
print("%-106s %5s %8s %8s %7s %9s %6s" % ("signature", "x/stp", "us", "TFLOP/s", "roof us", "over roof", "%"))
over_tot = 0.0
for sig, (n, us, gf) in sorted(groups.items(), key=lambda kv: -(kv[1][1] - kv[1][0] * roof_us(kv[0], kv[1][2] / kv[1][0]))):
    r = roof_us(sig, gf / n); over_tot += (us / n - r) * n / steps
    print("%-106s %5.1f %8.1f %8.0f %7.1f %9.1f %6.1f" % (sig, n / steps, us / n, gf / us * 1e3 if us else 0, r, (us / n - r) * n / steps, 100 * us / tot))
print("total %.3f ms per step over %d signatures; %.3f ms over the per-launch roofs (column 'over roof' = us per step above the roof, sorted by it)" %
      (tot / steps / 1e3, len(groups), over_tot / 1e3))
