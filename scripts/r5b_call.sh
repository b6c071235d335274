export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_conv_plan_shapes.py tests/test_gpu_e2e_parity.py tests/test_gpu_resnet.py tests/test_gpu_resnet_blocks.py tests/test_gpu_transforms.py tests/test_gpu_conv.py tests/test_gpu_trainer.py tests/test_gpu_vit.py tests/test_gpu_vit_ops.py -q -m gpu -x > gpurun_out/r5b_tests.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 5 gpurun_out/r5b_tests.log
[ $rc -ge 124 ] && exit $rc
for f in 0 1 0 1; do DALI_EVAL_FUSED=$f timeout -k 10 120 python scripts/time_eval_forward.py 500 20 || exit 124; done
for f in 0 1; do DALI_EVAL_FUSED=$f timeout -k 10 120 python scripts/time_eval_forward.py 256 20 || exit 124; done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5b_evalprof -- python scripts/time_eval_forward.py 500 20 > gpurun_out/r5b_evalprof.log 2>&1 || exit 124
python scripts/kstats.py gpurun_out/r5b_evalprof 23 40 > gpurun_out/r5b_eval_kstats.txt
head -30 gpurun_out/r5b_eval_kstats.txt
