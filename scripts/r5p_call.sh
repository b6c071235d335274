#!/bin/bash
timeout -k 10 300 python -m pytest tests/test_gpu_stem.py -x -q -m gpu > gpurun_out/r5p_tests.log 2>&1; rc=$?
tail -n 5 gpurun_out/r5p_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python scripts/bench_stem.py 500 && timeout -k 10 200 python scripts/time_eval_forward.py 500 20
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stem -- python scripts/time_eval_forward.py 500 10 > /dev/null 2>&1
python scripts/kstats.py gpurun_out/prof_stem 13 40 | grep -i "stem\|sum over"
