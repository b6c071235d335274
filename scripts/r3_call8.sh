#!/bin/bash
for prio in 1 0; do
echo "== l4.c2 PRIO=$prio"; DALI_WGRAD3X3_PRIO=$prio timeout -k 10 100 python scripts/pp_phases.py 256 16 8 512 512 2>&1 | grep -v amdgpu.ids || exit 124
done
timeout -k 10 300 python scripts/bench_wgrad.py "DALI_WGRAD3X3_PP=0" "DALI_WGRAD3X3_PP=3 DALI_WGRAD3X3_PRIO=1" "DALI_WGRAD3X3_PP=3 DALI_WGRAD3X3_PRIO=0" --filter l4.c2 --reps 5 2>&1 | grep -v amdgpu.ids
