#!/bin/bash
# One conv shape under every tile configuration (separate processes): bash scripts/sweep_conv_cfg.sh H W cin cout k stride
for E in "X=0" "DALI_CONV_CFG=1" "DALI_CONV_CFG=6" "DALI_CONV_CFG=4" "DALI_CONV_CFG=7" "DALI_CONV_K64_MINK=256" "DALI_CONV_K64_MINK=256 DALI_CONV_CFG=4" "DALI_CONV_K64_MINK=256 DALI_CONV_CFG=6" "DALI_CONV_K64=3 DALI_CONV_CFG=1" "DALI_CONV_K64=6 DALI_CONV_K64_MINK=256 DALI_CONV_CFG=6"; do
  echo -n "$E :: "; env $E timeout -k 10 120 python scripts/bench_one_conv.py "$@" 2>/dev/null | tail -1
done
