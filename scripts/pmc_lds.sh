#!/bin/bash
# LDS / issue counters of the kernels of one conv shape: bash scripts/pmc_lds.sh "<B H W cin cout k stride>"
export TMPDIR=/tmp
SHAPE="$1"
rm -rf gpurun_out/pmcL gpurun_out/pmcM
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmcL -- python scripts/one_conv.py $SHAPE > /dev/null 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS --output-format csv -d gpurun_out/pmcM -- python scripts/one_conv.py $SHAPE > /dev/null 2>&1 || echo "pass M failed"
python - <<'PY'
import csv, glob, collections
for d in ("pmcL", "pmcM"):
    fs = glob.glob("gpurun_out/%s/*/*counter_collection.csv" % d)
    if not fs: continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "igemm" in k:
            print(k)
            for c, vals in sorted(v.items()): print("    %-30s %16.0f" % (c, sum(vals) / len(vals)))
PY
