#!/bin/bash
timeout -k 10 300 python -m pytest tests/test_gpu_eval.py tests/test_gpu_fusion.py -m gpu -q -x > gpurun_out/c12_tests.log 2>&1; rc=$?; tail -3 gpurun_out/c12_tests.log
if [ $rc -ne 0 ]; then echo "eval tests rc=$rc: stopping"; exit $rc; fi
timeout -k 10 300 python scripts/bench_pairdist_stagger.py 2>&1 | grep -v amdgpu.ids
