#!/bin/bash
timeout -k 10 900 python -m pytest tests/test_gpu_bnlin.py tests/test_gpu_resnet_blocks.py tests/test_gpu_trainer.py tests/test_gpu_resnet.py -x -q -m gpu > gpurun_out/r5q_tests.log 2>&1; rc=$?
tail -n 6 gpurun_out/r5q_tests.log
[ $rc -ne 0 ] && exit $rc
bash scripts/ab_step.sh daliid_amd/libdaliid_prev.so daliid_amd/libdaliid_hip.so 3 --no-vit --no-epoch
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_q -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-distance --no-vit --no-epoch > /dev/null 2>&1
python scripts/kstats.py gpurun_out/prof_q 16 70 | grep -i "bnlin\|sum over"
