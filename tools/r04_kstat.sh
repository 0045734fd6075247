#!/bin/bash
# kernel-trace stats of the headline run only: tools/r04_kstat.sh <tag>
tag=$1
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/r04/kstat_$tag
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats -o stats --output-format csv -- python3 $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-side-legs --host-pairs 0 > $out/bench.json 2> $out/err.txt
find $out/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
rm -rf $out/stats
cut -c1-60,100- $out/kernel_stats.csv | head -14
