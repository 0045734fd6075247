#!/bin/bash
root=$GRAFT_REPO_ROOT
mkdir -p $root/gpurun_out/r04
for combo in "0 0" "1 0" "1 1" "3 3"; do
  set -- $combo
  PM_LIB=$root/ocean-perception_amd/lib/libvehicle_pm_gpu_tuning.so PM_RUN3_DBG=$1 PM_NOISE_DBG=$2 timeout -k 10 200 python3 $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-side-legs --host-pairs 0 --pairs-per-gpu 1 > $root/gpurun_out/r04/floor_$1_$2.json 2> $root/gpurun_out/r04/floor_err.txt
  python3 -c "import json,sys; j=json.loads(open('$root/gpurun_out/r04/floor_$1_$2.json').read().strip().splitlines()[-1]); print('run3_dbg=$1 noise_dbg=$2 ms_per_step', round(j['ms_per_step'],3))"
done
