#!/bin/bash
# PMC passes for one conv shape: bash scripts/pmc_one.sh "<shape args>" ; prints per-kernel counter averages
export TMPDIR=/tmp
SHAPE="$1"
rm -rf gpurun_out/pmcA gpurun_out/pmcB
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmcA -- python scripts/one_conv.py $SHAPE > /dev/null 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d gpurun_out/pmcB -- python scripts/one_conv.py $SHAPE > /dev/null 2>&1 || exit 2
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d gpurun_out/pmcC -- python scripts/one_conv.py $SHAPE > /dev/null 2>&1 || echo "pass C failed"
python - <<'PY'
import csv, glob, collections
for d in ("pmcA", "pmcB", "pmcC"):
    fs = glob.glob("gpurun_out/%s/*/*counter_collection.csv" % d)
    if not fs: continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        agg[r["Kernel_Name"][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "igemm" in k:
            print(k)
            for c, vals in sorted(v.items()): print("    %-30s %16.0f" % (c, sum(vals) / len(vals)))
PY
