#!/bin/bash
# Headline A/B of engine builds on the GPU box: tools/ab_headline.sh <lib.so|-> [...]   ("-" = the shipped library)
# Three bare runs each (one pair per call, 40 steps), pairs/s and the per-kernel ms of the last run.
for lib in "$@"; do
  for rep in 1 2 3; do
    if [ "$lib" = "-" ]; then unset PM_LIB; else export PM_LIB=$lib; fi
    timeout -k 10 120 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --host-pairs 0 --no-side-legs 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('$lib', 'pairs/s %.1f' % r['value'], {k: round(v,3) for k,v in r['kernels_ms_per_step'].items() if v > 0.05})"
  done
done
