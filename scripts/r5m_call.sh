export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bnlin.py tests/test_gpu_resnet_blocks.py tests/test_gpu_trainer.py tests/test_gpu_resnet.py -q -m gpu -x > gpurun_out/r5m_tests.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 4 gpurun_out/r5m_tests.log
[ $rc -ne 0 ] && exit $rc
BENCH_ARGS="--no-vit --no-epoch" bash scripts/ab_env.sh 3 "DALI_BNLIN_FUSED_FINISH=0" "DALI_BNLIN_FUSED_FINISH=1" || exit 124
