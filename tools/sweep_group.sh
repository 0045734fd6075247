#!/bin/bash
# waves-per-chain / group-width sweep of PM_ENGINE_RUNBLK2 at the default bench (run on the GPU box)
for sem in 0 1; do
  it=8; [ $sem = 1 ] && it=3
  for grp in auto 32 16; do
    for wv in 2 4 8; do
      echo "sem $sem group $grp waves $wv"
      g=$grp; [ $grp = auto ] && g=0
      PM_RUNBLK_GROUP=$g PM_RUNBLK_WAVES=$wv timeout -k 10 120 python bench.py --semantics $sem --iters $it --steps 20 --warmup 3 --no-cpu-baseline --host-pairs 0 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(' ms/frame %.3f'%r['ms_per_frame'], {k: round(v,3) for k,v in r['kernels_ms_per_step'].items() if 'sweep' in k})"
    done
  done
done
