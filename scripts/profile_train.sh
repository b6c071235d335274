#!/bin/bash
# On the GPU box: bench line + rocprofv3 kernel stats + the two PMC traffic passes for the train step.
# Usage: bash scripts/profile_train.sh [train|vit]
WL=${1:-train}
export TMPDIR=/tmp
O=gpurun_out/prof_$WL
rm -rf $O; mkdir -p $O
timeout -k 10 400 python bench.py --workload $WL --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err || exit 1
tail -n 1 $O/bench.json
# (the profiled passes leave the configs[4] distance sub-record out: only the train step's kernels are in the tables)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --workload $WL --steps 10 --warmup 3 --no-cpu-baseline --no-distance --no-vit --no-epoch > $O/bench_prof.json 2> $O/stats.err || exit 2
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-distance --no-vit --no-epoch > /dev/null 2> $O/fetch.err || exit 3
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-distance --no-vit --no-epoch > /dev/null 2> $O/write.err || exit 4
# MFMA utilisation (north star: "rocprof HBM GB/s and MFMA utilisation"): SQ + GRBM counters in a pass of their own, the program directly after `--`
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -- python bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-distance --no-vit --no-epoch > /dev/null 2> $O/mfma.err || exit 5
python scripts/pmc_mfma.py $O/mfma 7 $O/pmc_mfma.md $O/pmc_mfma.json > /dev/null
python scripts/pmc_traffic.py $O/fetch $O/write 7 $O/pmc_traffic.json $O/pmc_traffic.md
cp $(ls $O/stats/*/*kernel_stats.csv | head -n 1) $O/kernel_stats.csv
