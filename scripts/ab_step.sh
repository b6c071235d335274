#!/bin/bash
# A/B of two library builds on one box: bash scripts/ab_step.sh <lib A> <lib B> [reps] [bench args ...]; A and B alternate, 30 timed steps each
A=$1; B=$2; REPS=${3:-3}; shift 3
for rep in $(seq 1 $REPS); do
for L in $A $B; do
DALIID_LIB=$PWD/$L timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-distance "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L: ms_per_step', d['ms_per_step'], 'gemm', d['roofline']['kernel_ms_per_step'], d['roofline']['by_class_ms_per_step'])" || exit 124
done
done
