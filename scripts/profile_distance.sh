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
