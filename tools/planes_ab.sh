#!/bin/bash
# Plane-mode A/B on the GPU box: the shipped library and every lib/libvehicle_pm_gpu_<name>.so given as arguments, each
# through bench.py --mode planes (f32 state, one 1280x720 pair per call) -> gpurun_out/$ROUND/planes_ab.txt
round=${ROUND:-r06}
out=gpurun_out/$round
mkdir -p $out
f=$out/planes_ab.txt
: > $f
for name in "$@"; do
  if [ "$name" = shipped ]; then lib=""; else lib=$PWD/ocean-perception_amd/lib/libvehicle_pm_gpu_$name.so; fi
  for extra in "" "--state f16"; do
    PM_LIB=$lib python bench.py --mode planes $extra --steps 30 --warmup 5 --no-cpu-baseline --no-side-legs --host-pairs 0 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=j['roofline']
print('$name', '$extra', 'pairs/s', round(j['value'],1), 'ms', round(j['ms_per_step'],3), r['kernel'], round(r['avg_launch_ms']*1e3,1), 'us', {k: round(v*1e3/16.0,1) for k,v in j['kernels_ms_per_step'].items() if k.startswith('planes_s') or k.startswith('planes_v')})" >> $f
  done
done
cat $f
