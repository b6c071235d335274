export TMPDIR=/tmp
rm -rf gpurun_out/r5i_evalprof; mkdir -p gpurun_out
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5i_evalprof -- python scripts/time_eval_forward.py 500 20 > gpurun_out/r5i_evalprof.log 2>&1 || exit 124
python scripts/kstats.py gpurun_out/r5i_evalprof 23 24 | cut -c1-150
