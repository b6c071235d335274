#!/bin/bash
# A/B of two environment settings on one box, alternating processes: bash scripts/ab_env.sh "A=1 B=2" "A=0" [reps] [bench args ...]
A=$1; B=$2; REPS=${3:-3}; shift 3
for rep in $(seq 1 $REPS); do
for E in "$A" "$B"; do
env $E timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-distance "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$E: ms_per_step', d['ms_per_step'], 'gemm', d['roofline']['kernel_ms_per_step'], d['roofline']['by_class_ms_per_step'])" || exit 124
done
done
