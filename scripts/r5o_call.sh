#!/bin/bash
# fused inference stem: parity, the eval twin tests, timing A/B
timeout -k 10 600 python -m pytest tests/test_gpu_stem.py tests/test_gpu_resnet_blocks.py tests/test_gpu_resnet.py tests/test_gpu_e2e_parity.py -x -q -m gpu > gpurun_out/r5o_tests.log 2>&1; rc=$?
tail -n 8 gpurun_out/r5o_tests.log
[ $rc -ge 124 ] && exit $rc
timeout -k 10 200 python scripts/time_eval_forward.py 500 20 && DALI_EVAL_STEM=0 timeout -k 10 200 python scripts/time_eval_forward.py 500 20 && timeout -k 10 200 python scripts/time_eval_forward.py 500 20
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stem -- python scripts/time_eval_forward.py 500 10 > /dev/null 2>&1
python scripts/kstats.py gpurun_out/prof_stem 13 12 | cut -c1-170
