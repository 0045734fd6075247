#!/bin/bash
# side measurements quoted in DESIGN.md 7: tools/r03_side.sh
one() {
  echo -n "$* : "
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --host-pairs 0 --no-side-legs --no-profile "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f pairs/s  %.3f ms/frame (median step %.3f ms)'%(r['value'], r['ms_per_frame'], r['step_ms']['median']))"
}
one
one --pairs-per-gpu 4
one --pairs-per-gpu 32 --steps 6
one --self-seed
one --semantics 1 --iters 3
one --mode planes
one --mode planes --pairs-per-gpu 4 --steps 8
one --mode planes --state f16 --enhance
