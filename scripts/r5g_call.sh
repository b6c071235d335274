export TMPDIR=/tmp
O=gpurun_out/r5g
rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline --no-distance --no-vit --no-epoch > $O/bench_prof.json 2> $O/stats.err || exit 2
python scripts/kstats.py $O/stats 13 70 > $O/kstats.txt
cat $O/kstats.txt | cut -c1-170
