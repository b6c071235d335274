#!/bin/bash
# On the GPU box: bench lines (bf16x3, bf16), rocprofv3 kernel stats and the ranking timing for the gallery distance workload.
export TMPDIR=/tmp
O=gpurun_out/prof_distance
rm -rf $O; mkdir -p $O
timeout -k 10 300 python bench.py --workload distance --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err || exit 1
tail -n 1 $O/bench.json
timeout -k 10 300 python bench.py --workload distance --steps 5 --warmup 2 --precision bf16 --no-cpu-baseline > $O/bench_bf16.json 2>> $O/bench.err || exit 2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --workload distance --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_prof.json 2> $O/stats.err || exit 3
timeout -k 10 200 python scripts/time_rank.py > $O/rank.txt 2>> $O/bench.err || exit 4
cp $(ls $O/stats/*/*kernel_stats.csv | head -n 1) $O/kernel_stats.csv
# effective shader clock during the distance kernel (DVFS: the chip does not hold 2.4 GHz under dense MFMA load)
# MFMA utilisation of the distance kernel, counters in a pass of their own
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -- python bench.py --workload distance --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/mfma.err || exit 6
python scripts/pmc_mfma.py $O/mfma 8 $O/pmc_mfma.md > /dev/null
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/clock -- python bench.py --workload distance --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/clock.err || exit 5
python scripts/effective_clock.py $O/clock pairdist_dma > $O/clock.txt
# HBM traffic of the distance / ranking kernels: FETCH_SIZE and WRITE_SIZE, each in a pass of its own
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python bench.py --workload distance --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/fetch.err || exit 7
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python bench.py --workload distance --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/write.err || exit 8
python scripts/pmc_traffic_distance.py $O/fetch $O/write $O/pmc_traffic.json > /dev/null
