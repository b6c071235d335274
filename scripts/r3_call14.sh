#!/bin/bash
timeout -k 10 150 python -m pytest tests/test_gpu_conv.py -m gpu -q -x > gpurun_out/c14_tests.log 2>&1; rc=$?; tail -2 gpurun_out/c14_tests.log
if [ $rc -ne 0 ]; then echo "conv tests rc=$rc: stopping"; exit $rc; fi
timeout -k 10 500 python scripts/bench_wgrad.py "DALI_REDUCE_WAVES=-1" "DALI_REDUCE_WAVES=0" "DALI_REDUCE_WAVES=1" "DALI_REDUCE_WAVES=4" --reps 3 2>&1 | grep -v amdgpu.ids | tail -8
for rep in 1 2; do
for v in -1 0; do
DALI_REDUCE_WAVES=$v timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-distance 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('REDUCE_WAVES=$v: ms_per_step', d['ms_per_step'], 'gemm', d['roofline']['kernel_ms_per_step'], d['roofline']['by_class_ms_per_step'])" || exit 124
done
done
