#!/bin/bash
# plane mode: time with the window evaluation or the tile fill switched off (tuning build; results are wrong by design)
root=$GRAFT_REPO_ROOT
mkdir -p $root/gpurun_out/r04
cd /tmp && export TMPDIR=/tmp
for d in ${DBGS:-0 1 3}; do
  out=$root/gpurun_out/r04/planes_dbg_$d
  mkdir -p $out
  PM_LIB=$root/ocean-perception_amd/lib/libvehicle_pm_gpu_tuning.so PM_PLANES_DBG=$d timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/stats -o stats --output-format csv -- python3 $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-side-legs --host-pairs 0 --mode planes > $out/bench.json 2> $out/err.txt
  f=$(find $out/stats -name "*kernel_stats.csv" | head -1)
  python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'k_planes' in r['Name']: print('dbg=$d', r['Name'][9:45], 'calls', r['Calls'], 'avg_us', round(float(r['AverageNs'])/1e3,1), 'min_us', round(float(r['MinNs'])/1e3,1))
"
  python3 -c "import json,sys; j=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print('  ms_per_step', round(j['ms_per_step'],3))"
  rm -rf $out/stats
done
