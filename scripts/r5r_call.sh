#!/bin/bash
timeout -k 10 900 python -m pytest tests/test_gpu_eval.py tests/test_gpu_losses.py tests/test_gpu_trainer.py tests/test_gpu_dp.py -x -q -m gpu > gpurun_out/r5r_tests.log 2>&1; rc=$?
tail -n 12 gpurun_out/r5r_tests.log
[ $rc -ne 0 ] && exit $rc
BENCH_ARGS="--no-vit --no-epoch" bash scripts/ab_env.sh 3 "DALI_PAIRDIST_SMALL=0" "DALI_PAIRDIST_SMALL=1"
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-distance --no-vit --no-epoch > /dev/null 2>&1
python scripts/kstats.py gpurun_out/prof_r 16 80 | grep -i "dot_small\|pairdist\|rows_prep\|sum over"
