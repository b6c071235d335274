#!/usr/bin/env python3
"""gpurun_out/prof_distance/ (from scripts/profile_distance.sh) -> profiles/r01_distance_kernel_stats.{md,csv}."""
import csv, os, shutil
RND = os.environ.get("ROUND", "r05")
src = "gpurun_out/prof_distance"
shutil.copy(src + "/kernel_stats.csv", "profiles/%s_distance_kernel_stats.csv" % RND)
last = lambda f: open(src + "/" + f).read().strip().splitlines()[-1]
rows = list(csv.DictReader(open(src + "/kernel_stats.csv")))
out = ["# Round %s -- gallery distance (configs[4]) kernel profile\n" % RND[1:].lstrip("0"),
       "Produced by `bash scripts/profile_distance.sh` on an MI355X box: `python bench.py --workload distance --steps 5 --warmup 2` (unprofiled, with the CPU "
       "baseline), the same with `--precision bf16`, `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --workload distance --steps 5 "
       "--warmup 2 --no-cpu-baseline`, `python scripts/time_rank.py`; summary by `scripts/make_profile_distance_md.py`.\n",
       "Bench line, unprofiled (bf16x3 = fp32-grade split-bf16):\n\n```\n%s\n```\n" % last("bench.json"),
       "Bench line with `--precision bf16`:\n\n```\n%s\n```\n" % last("bench_bf16.json"),
       "Bench line of the profiled run:\n\n```\n%s\n```\n" % last("bench_prof.json"),
       "CMC/mAP kernel, worst case vs realistic ranking (`scripts/time_rank.py`; the kernel bins gallery entries only up to the last match, so "
       "well-separated identities stop early):\n\n```\n%s\n```\n" % "\n".join(l[:160] for l in open(src + "/rank.txt").read().strip().splitlines() if "amdgpu.ids" not in l),
       "Per-kernel summary of the profiled run (all launches of the process incl. warm-up):\n",
       "| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|"]
for r in rows[:12]:
    out.append("| `%s` | %s | %.3f | %.1f | %.1f | %.1f | %.1f |" % (r["Name"][:100].replace("|", "/"), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                                 float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["Percentage"])))
try:
    clk = open(src + "/clock.txt").read().strip()
    out.append("\nEffective shader clock during the distance kernel (`rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace` in its own run, "
               "`scripts/effective_clock.py`: GRBM_GUI_ACTIVE / 8 XCDs / kernel wall time; the 2.5 PFLOP/s dense bf16 peak is quoted at 2.4 GHz):\n\n```\n%s\n```\n" % clk)
except OSError:
    pass
if os.path.exists(src + "/pmc_mfma.md"):
    out.append("\nMFMA utilisation (`rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE` in its own run, "
               "`scripts/pmc_mfma.py`; per launch, 8 launches in the process; bf16x3 issues three MFMA products per algorithmic one):\n")
    out.append(open(src + "/pmc_mfma.md").read())
if os.path.exists(src + "/pmc_traffic.json"):
    shutil.copy(src + "/pmc_traffic.json", "profiles/%s_distance_pmc_traffic.json" % RND)
    out.append("\nHBM traffic per launch (`rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, each in its own run; `scripts/pmc_traffic_distance.py`; algorithmic: 4.0 GB of "
               "distances written + 0.9 GB of operand images read once):\n\n```\n%s\n```\n" % open(src + "/pmc_traffic.json").read().strip())
open("profiles/%s_distance_kernel_stats.md" % RND, "w").write("\n".join(out) + "\n")
print("wrote profiles/%s_distance_kernel_stats.md" % RND)
