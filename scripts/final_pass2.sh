#!/bin/bash
# final pass 2: ViT step and distance profiles
bash scripts/profile_train.sh vit || exit 1
python scripts/kstats.py gpurun_out/prof_vit/stats 16 60 > gpurun_out/prof_vit/kstats.txt
bash scripts/profile_distance.sh
