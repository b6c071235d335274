#!/bin/bash
bash scripts/profile_train.sh vit || exit 1
python scripts/kstats.py gpurun_out/prof_vit/stats 16 60 > gpurun_out/prof_vit/kstats.txt
