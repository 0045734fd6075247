#!/bin/bash
# A/B of engine builds on the GPU box: tools/ab_libs.sh <lib.so> [<lib.so> ...]
for lib in "$@"; do
  for sem in 0 1; do
    it=8; [ $sem = 1 ] && it=3
    echo "== $lib sem $sem"
    PM_LIB=$lib timeout -k 10 120 python bench.py --semantics $sem --iters $it --steps 20 --warmup 3 --no-cpu-baseline --host-pairs 0 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(' ms/frame %.3f'%r['ms_per_frame'], {k: round(v,3) for k,v in r['kernels_ms_per_step'].items()})"
  done
done
