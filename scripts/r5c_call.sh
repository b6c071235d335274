export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_bnlin.py -q -m gpu -x > gpurun_out/r5c_tests.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 8 gpurun_out/r5c_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python scripts/bench_fused1x1.py 256 > gpurun_out/r5c_fused.log 2>&1 || exit 124
cat gpurun_out/r5c_fused.log
