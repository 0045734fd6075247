#!/bin/bash
# group-size / waves-per-chain sweep of PM_ENGINE_RUNBLK2 (run on the GPU box)
set -e
for sem in 0 1; do
  it=8; [ $sem = 1 ] && it=3
  for grp in 32 16; do
    for wv in 1 2 3 4 6; do
      echo "sem $sem group $grp waves $wv"
      PM_RUNBLK_GROUP=$grp PM_RUNBLK_WAVES=$wv timeout -k 10 120 python bench.py --semantics $sem --iters $it --steps 20 --warmup 3 --no-cpu-baseline --host-pairs 0 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(' ms/frame %.3f'%r['ms_per_frame'], {k: round(v,3) for k,v in r['kernels_ms_per_step'].items() if 'sweep' in k}, {a: (c['steps_round1'], c['steps_fixup']) for a, c in r['run_engine_counters_per_step'].items()})"
    done
  done
done
