for rep in 1 2 3; do
for L in daliid_amd/libdaliid_prev.so daliid_amd/libdaliid_hip.so; do
DALIID_LIB=$PWD/$L timeout -k 10 200 python bench.py --workload distance --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', d['value'], 'ms', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'])"
done; done
