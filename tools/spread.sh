#!/bin/bash
# run-to-run spread of the headline on one box: eight processes of the driver's timed region (no side legs, no CPU leg)
mkdir -p gpurun_out/${ROUND:-r06}
out=gpurun_out/${ROUND:-r06}/spread.txt
: > $out
for i in 1 2 3 4 5 6 7 8; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-side-legs --host-pairs 0 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['value'],1), round(j['ms_per_step'],3), round(j['step_ms']['median'],3))" >> $out
done
cat $out
python3 -c "
v=[float(l.split()[0]) for l in open('$out')]
v.sort(); print('pairs/s min', v[0], 'median', (v[3]+v[4])/2, 'max', v[-1])"
