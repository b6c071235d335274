#!/bin/bash
# in-kernel stamps of the weight-gradient kernels at layer4 conv3 / layer3 conv3 shapes, per variant
for shape in "256 16 8 512 2048 1" "256 16 8 256 1024 1"; do
for v in 0 1 3; do
  echo "== shape $shape DALI_WGRAD_P=$v"
  DALI_WGRAD_P=$v timeout -k 10 120 python scripts/conv_block_timeline.py $shape wgrad 2>&1 | grep -v amdgpu.ids || exit 124
done
done
