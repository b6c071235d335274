#!/bin/bash
# final pass 1: full GPU suite + smoke, then the train-step / gemm-table / inference profiles
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > gpurun_out/final_tests.log 2>&1; rc=$?
tail -n 4 gpurun_out/final_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" || exit 3
bash scripts/profile_train.sh train || exit 1
DALI_GEMM_PROFILE_DUMP=gpurun_out/gemm_launches.csv timeout -k 10 200 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-distance --no-vit --no-epoch > gpurun_out/gemm_dump_bench.json 2> gpurun_out/gemm_dump.err || exit 2
python scripts/gemm_launch_table.py gpurun_out/gemm_launches.csv 3 > gpurun_out/gemm_launch_table.txt
python scripts/kstats.py gpurun_out/prof_train/stats 16 60 > gpurun_out/prof_train/kstats.txt
bash scripts/profile_eval_forward.sh
