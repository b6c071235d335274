#!/bin/bash
timeout -k 10 150 python -m pytest tests/test_gpu_conv.py -m gpu -q -x > gpurun_out/c7_tests.log 2>&1; rc=$?; tail -3 gpurun_out/c7_tests.log
if [ $rc -ne 0 ]; then echo "conv tests rc=$rc: stopping"; exit $rc; fi
timeout -k 10 300 python scripts/bench_wgrad.py "DALI_WGRAD3X3_P=0" "DALI_WGRAD3X3_P=5" "DALI_WGRAD3X3_P=6" --filter c2 --reps 5 2>&1 | grep -v amdgpu.ids
