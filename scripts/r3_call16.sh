#!/bin/bash
for rep in 1 2; do
for w in 128 256 512; do
DALI_BNLIN_MAXW=$w timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-distance 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('BNLIN_MAXW=$w: ms_per_step', d['ms_per_step'], 'gemm', d['roofline']['kernel_ms_per_step'], d['roofline']['by_class_ms_per_step'])" || exit 124
done
done
