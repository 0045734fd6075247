#!/bin/bash
# Headline against sweep knobs of the tuning build, one "VAR=value [VAR=value ...]" configuration per argument ("" = defaults):
#   tools/ab_knobs.sh "" "PM_RUNBLK_WAVES_ROW=6" "PM_G16_ROW_AMP=2 PM_G16_COL_AMP=8"
export PM_LIB=$PWD/ocean-perception_amd/lib/libvehicle_pm_gpu_tuning.so
for cfg in "$@"; do
  for rep in 1 2; do
    env $cfg timeout -k 10 120 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --host-pairs 0 --no-side-legs 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('[$cfg]', 'pairs/s %.1f' % r['value'], {k: round(v,3) for k,v in r['kernels_ms_per_step'].items() if k.startswith('sweep')})"
  done
done
