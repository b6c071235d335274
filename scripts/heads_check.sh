#!/bin/bash
timeout -k 10 900 python -m pytest tests/test_gpu_losses.py tests/test_gpu_nnops.py tests/test_gpu_trainer.py tests/test_gpu_dp.py tests/test_gpu_vit.py -x -q -m gpu > gpurun_out/h_tests.log 2>&1; rc=$?
tail -n 6 gpurun_out/h_tests.log
[ $rc -ne 0 ] && exit $rc
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_h -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-distance --no-vit --no-epoch > gpurun_out/h_bench.json 2>/dev/null
python scripts/kstats.py gpurun_out/prof_h 16 90 | grep -i "proxy\|center\|bn1d\|l2norm\|sum over\|rowstat"
tail -n 1 gpurun_out/h_bench.json | cut -c1-120
