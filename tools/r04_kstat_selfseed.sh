#!/bin/bash
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/r04/kstat_selfseed
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats -o stats --output-format csv -- python3 $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-side-legs --host-pairs 0 --self-seed > $out/bench.json 2> $out/err.txt
find $out/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
find $out/stats -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $out/kernel_trace.csv
rm -rf $out/stats
python3 - <<P
import csv
for r in csv.DictReader(open('$out/kernel_stats.csv')):
    if 'runblk' in r['Name'] or 'noise' in r['Name']: continue
    print(r['Name'][:60].ljust(60), r['Calls'], round(float(r['AverageNs'])/1e3,1), round(float(r['MinNs'])/1e3,1), r['Percentage'])
P
