#!/bin/bash
# A/B of environment knobs on the default bench: tools/r03_ab.sh "VAR=x VAR2=y" "..." ...
for cfg in "$@"; do
  echo -n "$cfg : "
  env $cfg timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --host-pairs 0 --no-side-legs --no-profile 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f pairs/s  %.3f ms (median %.3f)  det=%s'%(r['value'], r['ms_per_step'], r['step_ms']['median'], r['check']['deterministic_across_steps']))"
done
