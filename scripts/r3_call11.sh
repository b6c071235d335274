#!/bin/bash
timeout -k 10 200 python -m pytest tests/test_gpu_vit_ops.py -m gpu -q -x > gpurun_out/c11_tests.log 2>&1; rc=$?; tail -4 gpurun_out/c11_tests.log
if [ $rc -ne 0 ]; then echo "vit ops tests rc=$rc: stopping"; exit $rc; fi
timeout -k 10 200 python scripts/bench_attention.py 2>&1 | grep -v amdgpu.ids
T=211 timeout -k 10 200 python scripts/bench_attention.py 2>&1 | grep -v amdgpu.ids
