#!/bin/bash
# Run on the GPU box: whole GPU test suite + quick timing; logs under gpurun_out/.
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/gpu_tests.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/gpu_tests.log
timeout -k 10 200 python scripts/time_resnet.py 256 2>&1 | tail -2
