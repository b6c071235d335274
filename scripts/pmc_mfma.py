#!/usr/bin/env python3
"""MFMA utilisation per kernel from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16
GRBM_GUI_ACTIVE; its own run, no tracing besides the kernel trace).

    python scripts/pmc_mfma.py <pmc_dir> <steps_in_process> <out.md> [out.json]

mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs): the share of the chip's SIMD-cycles in which the matrix pipe was
executing (counter_defs.yaml's MfmaUtil; rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs, MI355X_MICROARCH.md "DVFS give-back").
Calibration: SQ_VALU_MFMA_BUSY_CYCLES = 16 per v_mfma_f32_16x16x32_bf16 (layer4 conv2: 154.6 GFLOP = 9.44 M MFMAs -> 151.0 M counted).
SQ_BUSY_CU_CYCLES is collected but not turned into a ratio: its aggregation over SIMDs / shader engines is not documented for gfx950
(the naive ratio reads 3.4).  mops_tflop = SQ_INSTS_VALU_MFMA_MOPS_BF16 * 512 FLOP, the
FLOPs the MFMA instructions actually issued (padding and the bf16x3 triple products included), per step."""
import collections, csv, glob, json, sys


def main():
    d, steps, out_md = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    assert files, "no counter_collection.csv under " + d
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
    rows = []
    for k, c in agg.items():
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        if gui <= 0:
            continue
        busy, cu, mops = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("SQ_BUSY_CU_CYCLES", 0.0), c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)
        rows.append(dict(kernel=k, launches_per_step=len(calls[k]) / steps, gui_active_per_step=gui / steps, mfma_busy=busy / (gui / 8 * 1024),
                         busy_cu_quadcycles_per_step=cu / steps, mops_tflop_per_step=mops * 512 / 1e12 / steps))
    rows.sort(key=lambda r: -r["gui_active_per_step"])
    tot_gui = sum(r["gui_active_per_step"] for r in rows)
    gem = [r for r in rows if "igemm" in r["kernel"] or "pairdist" in r["kernel"] or "attention" in r["kernel"]]
    with open(out_md, "w") as f:
        f.write("| kernel | launches/step | share of GPU cycles | mfma_busy | MFMA TFLOP issued/step |\n|---|---:|---:|---:|---:|\n")
        for r in rows[:30]:
            f.write("| `%s` | %.1f | %.1f %% | %.3f | %.3f |\n" % (r["kernel"][:90].replace("|", "/"), r["launches_per_step"],
                                                                  100 * r["gui_active_per_step"] / tot_gui, r["mfma_busy"], r["mops_tflop_per_step"]))
        if gem:
            g = sum(r["gui_active_per_step"] for r in gem)
            f.write("\nMFMA kernels together: mfma_busy %.3f (cycle-weighted), %.3f TFLOP issued per step; whole step: %.3f.\n"
                    % (sum(r["mfma_busy"] * r["gui_active_per_step"] for r in gem) / g, sum(r["mops_tflop_per_step"] for r in gem),
                       sum(r["mfma_busy"] * r["gui_active_per_step"] for r in rows) / tot_gui))
    if len(sys.argv) > 4:
        json.dump(rows, open(sys.argv[4], "w"), indent=1)
    print(open(out_md).read())


if __name__ == "__main__":
    main()
