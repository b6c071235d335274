#!/usr/bin/env python3
"""gpurun_out/prof_<wl>/ (from scripts/profile_train.sh) -> profiles/<ROUND>_<wl>_* (committed summary)."""
import csv, json, os, shutil, sys
RND = os.environ.get("ROUND", "r05")
wl = sys.argv[1] if len(sys.argv) > 1 else "train"
src, name = "gpurun_out/prof_%s" % wl, {"train": "train_step", "vit": "vit_step"}[wl]
shutil.copy(src + "/pmc_traffic.json", "profiles/%s_%s_pmc_traffic.json" % (RND, wl))
shutil.copy(src + "/kernel_stats.csv", "profiles/%s_%s_kernel_stats.csv" % (RND, name))
rows = list(csv.DictReader(open(src + "/kernel_stats.csv")))
bench = open(src + "/bench_prof.json").read().strip().splitlines()[-1]
bench_full = open(src + "/bench.json").read().strip().splitlines()[-1]
pmc = json.load(open(src + "/pmc_traffic.json"))
pmd = open(src + "/pmc_traffic.md").read()
steps = 16
out = ["# Round %s -- %s kernel profile\n" % (RND[1:].lstrip("0"), name.replace("_", " "))]
out.append("Produced by `bash scripts/profile_train.sh %s` on an MI355X box: (1) plain `python bench.py --workload %s --steps 10 --warmup 3`, "
           "(2) `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --workload %s --steps 10 --warmup 3 --no-cpu-baseline`, "
           "(3)+(4) `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, each in its own run, `-- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-distance --no-vit --no-epoch` "
           "(passes 2-5 profile the train step alone; pass 1 is the driver's default command, with the distance / vit / epoch sub-records); "
           "summary by `scripts/make_profile_md.py`.\n" % (wl, wl, wl))
out.append("Bench line of run (1) (unprofiled):\n\n```\n%s\n```\n" % bench_full)
out.append("Bench line of run (2) (under rocprofv3):\n\n```\n%s\n```\n" % bench)
out.append("## Per-kernel summary of run (2) (%d steps in the process: 3 warm-up + 10 timed + 3 event-bracketed)\n" % steps)
out.append("| kernel | calls | total ms | ms/step | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|---:|")
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
for r in rows[:40]:
    t = float(r["TotalDurationNs"]) / 1e6
    out.append("| `%s` | %s | %.3f | %.3f | %.1f | %.1f | %.1f | %.1f |" % (r["Name"][:80].replace("|", "/"), r["Calls"], t, t / steps, float(r["AverageNs"]) / 1e3,
                                                                        float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["Percentage"])))
out.append("\nSum over all kernels: %.2f ms = %.2f ms/step.\n" % (tot, tot / steps))
gem = [r for r in rows if "igemm" in r["Name"]]
gt = sum(float(r["TotalDurationNs"]) for r in gem) / 1e6
gc = sum(int(r["Calls"]) for r in gem)
flop = {"train": 6.226, "vit": 13.488}[wl]
out.append("GEMM kernels (`igemm_*`): %d launches, %.2f ms total = %.3f ms/step, average launch %.1f us -> %.0f TFLOP/s on %.3f TFLOP/step "
           "(bench.py's live figure: `roofline.kernel_ms_per_step`, `avg_launch_us`).\n" % (gc, gt, gt / steps, gt * 1e3 / gc, flop / (gt / steps * 1e-3), flop))
out.append("## HBM traffic (PMC passes 3 and 4; 7 steps in each process)\n")
out.append("Corrections: %s.\n" % pmc["corrections"])
out.append("```\n%s\n```\n" % json.dumps(pmc["per_family_bytes_per_step"], indent=1))
out.append("All kernels: %.1f GB/step; GEMM kernels %.1f GB/step.\n" % (pmc["all_kernels_hbm_bytes_per_step"] / 1e9, pmc["gemm_kernels_hbm_bytes_per_step"] / 1e9))
out.append(pmd)
if os.path.exists(src + "/pmc_mfma.md"):
    out.append("\n## MFMA utilisation (PMC pass 5: `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE`, 7 steps in the process)\n")
    out.append("mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) = share of the SIMD-cycles in which the matrix pipe executes "
               "(16 counted cycles per `v_mfma_f32_16x16x32_bf16`); `scripts/pmc_mfma.py`.\n")
    out.append(open(src + "/pmc_mfma.md").read())
open("profiles/%s_%s_kernel_stats.md" % (RND, name), "w").write("\n".join(out))
print("wrote profiles/%s_%s_kernel_stats.md" % (RND, name))
