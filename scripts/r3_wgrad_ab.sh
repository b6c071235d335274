#!/bin/bash
# weight gradient + split-K reduce variants, interleaved in one process; then the train step with the round-2 settings and with the defaults
timeout -k 10 500 python scripts/bench_wgrad.py "DALI_WGRAD_P=0 DALI_REDUCE_WAVES=-1" "DALI_REDUCE_WAVES=-1" "DALI_REDUCE_WAVES=0" "DALI_REDUCE_WAVES=1" "DALI_REDUCE_WAVES=4" --reps 5 2>&1 | grep -v amdgpu.ids
for rep in 1 2; do
DALI_WGRAD_P=0 DALI_REDUCE_WAVES=-1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-distance 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round-2 settings: ms_per_step', d['ms_per_step'], 'gemm', d['roofline']['kernel_ms_per_step'], d['roofline']['by_class_ms_per_step'])" || exit 124
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-distance 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('defaults        : ms_per_step', d['ms_per_step'], 'gemm', d['roofline']['kernel_ms_per_step'], d['roofline']['by_class_ms_per_step'])" || exit 124
done
