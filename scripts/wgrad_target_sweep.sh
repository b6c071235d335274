#!/bin/bash
# rocprofv3 kernel stats of one conv shape under several split-K targets (DALI_WGRAD_TARGET)
export TMPDIR=/tmp
SHAPE="$1"; shift
for T in "$@"; do
  export DALI_WGRAD_TARGET=$T
  rm -rf gpurun_out/sw; timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sw -- python scripts/one_conv.py $SHAPE > /dev/null 2>&1 || exit 1
  echo "== shape $SHAPE target $T"
  python - <<'PY'
import csv, glob
for r in csv.DictReader(open(glob.glob("gpurun_out/sw/*/*kernel_stats.csv")[0])):
    if "splitk" in r["Name"] or "wgrad" in r["Name"]:
        print("   %-58s avg %8.1f us" % (r["Name"][:58], float(r["AverageNs"]) / 1e3))
PY
done
