#!/bin/bash
# A/B of environment settings on one box, alternating processes: bash scripts/ab_env.sh <reps> "A=1 B=2" "A=0" ["A=3" ...]   (BENCH_ARGS adds bench.py arguments)
REPS=$1; shift
for rep in $(seq 1 $REPS); do
for E in "$@"; do
env $E timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-distance $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$E: ms_per_step', d['ms_per_step'], 'gemm', d['roofline']['kernel_ms_per_step'], d['roofline']['by_class_ms_per_step'])" || exit 124
done
done
