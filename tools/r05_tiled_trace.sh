#!/bin/bash
# kernel stats of 4096x2160 in 8 bands on one device (where does the protocol's cost over the untiled frame go?)
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/r05/tiled_trace
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/raw -o t --output-format csv -- python3 $root/tools/r05_tiled8.py ${1:-8} > $out/log.txt 2>&1
tail -1 $out/log.txt
f=$(find $out/raw -name "*kernel_stats.csv" | head -1)
cp $f $out/kernel_stats_tiled${1:-8}.csv
python3 - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):6d} total_ms {float(r["TotalDurationNs"])/1e6:8.2f} avg_us {float(r["AverageNs"])/1e3:7.1f} min_us {float(r["MinNs"])/1e3:6.1f} {100*float(r["TotalDurationNs"])/tot:5.1f}%')
print("total kernel ms", tot / 1e6)
P
rm -rf $out/raw
