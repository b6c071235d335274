#!/bin/bash
# On the GPU box: the inference forward at batch 500 (extractFeatures' batch): kernel stats + the two PMC traffic passes -> bytes per image.
export TMPDIR=/tmp
O=gpurun_out/prof_eval
rm -rf $O; mkdir -p $O
timeout -k 10 200 python scripts/time_eval_forward.py 500 20 > $O/plain.txt 2>&1 || exit 1
DALI_EVAL_FUSED=0 timeout -k 10 200 python scripts/time_eval_forward.py 500 20 >> $O/plain.txt 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python scripts/time_eval_forward.py 500 20 > $O/prof.txt 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python scripts/time_eval_forward.py 500 4 > /dev/null 2> $O/fetch.err || exit 3
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python scripts/time_eval_forward.py 500 4 > /dev/null 2> $O/write.err || exit 4
python scripts/pmc_traffic.py $O/fetch $O/write 7 $O/pmc_traffic.json $O/pmc_traffic.md
python scripts/kstats.py $O/stats 23 30 > $O/kstats.txt
cp $(ls $O/stats/*/*kernel_stats.csv | head -n 1) $O/kernel_stats.csv
cat $O/plain.txt | grep "eval forward"
