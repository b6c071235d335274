export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_e2e_parity.py tests/test_gpu_resnet.py tests/test_gpu_trainer.py tests/test_gpu_bnlin.py -q -m gpu -x > gpurun_out/r5k_tests.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 6 gpurun_out/r5k_tests.log
[ $rc -ne 0 ] && exit $rc
for f in 0 1 0 1; do DALI_EVAL_FUSED=$f timeout -k 10 120 python scripts/time_eval_forward.py 500 20 || exit 124; done
