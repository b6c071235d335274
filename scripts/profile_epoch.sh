#!/bin/bash
export TMPDIR=/tmp
timeout -k 10 300 python bench.py --workload epoch --no-cpu-baseline > gpurun_out/epoch_plain.json 2>/dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_epoch -- python bench.py --workload epoch --no-cpu-baseline > gpurun_out/epoch_prof.json 2>/dev/null
python scripts/kstats.py gpurun_out/prof_epoch 1 45 | cut -c1-180
tail -n 1 gpurun_out/epoch_plain.json | cut -c1-900
