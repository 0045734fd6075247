#!/bin/bash
# sweep kernels with parts of the step's memory traffic switched off (tuning build; results are WRONG by design, only the
# clock is read): PM_RUN3_DBG bit 2 = no target-record loads in row steps, bit 3 = no reference-quad loads in row steps,
# bit 4 = no target-record loads in column steps.  Upper bounds for what moving those bytes off the vector memory
# pipeline (e.g. into LDS) could buy.
root=$GRAFT_REPO_ROOT
mkdir -p $root/gpurun_out/r05
cd /tmp && export TMPDIR=/tmp
for d in ${DBGS:-0 4 12 16 28}; do
  out=$root/gpurun_out/r05/run3_dbg_$d
  mkdir -p $out
  PM_LIB=$root/ocean-perception_amd/lib/libvehicle_pm_gpu_tuning.so PM_RUN3_DBG=$d timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/stats -o stats --output-format csv -- python3 $root/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-side-legs --host-pairs 0 --pairs-per-gpu 1 > $out/bench.json 2> $out/err.txt
  f=$(find $out/stats -name "*kernel_stats.csv" | head -1)
  python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'k_runblk3' in r['Name'] or 'k_noise' in r['Name']: print('dbg=$d', r['Name'][9:40], 'calls', r['Calls'], 'avg_us', round(float(r['AverageNs'])/1e3,1), 'min_us', round(float(r['MinNs'])/1e3,1))
"
  python3 -c "import json,sys; j=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print('  dbg=$d ms_per_step', round(j['ms_per_step'],3), 'counters', j.get('run_engine_counters_per_step'))"
  rm -rf $out/stats
done
