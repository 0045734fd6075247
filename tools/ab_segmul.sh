#!/bin/bash
# Headline against segments per lane group (tuning build, PM_RUN3_SEGMUL): tools/ab_segmul.sh 1 2 3 4
export PM_LIB=ocean-perception_amd/lib/libvehicle_pm_gpu_tuning.so
for m in "$@"; do
  for rep in 1 2; do
    PM_RUN3_SEGMUL=$m timeout -k 10 120 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --host-pairs 0 --no-side-legs 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read())
c=r.get('run_engine_counters_per_step',{})
print('segmul $m', 'pairs/s %.1f' % r['value'], {k: round(v,3) for k,v in r['kernels_ms_per_step'].items() if v > 0.05}, {a:(c[a]['steps_round1'],c[a]['steps_fixup'],c[a]['fixup_rounds']) for a in c})"
  done
done
