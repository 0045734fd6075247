#!/bin/bash
# side measurements quoted in DESIGN.md 7 that are not legs of the default line
mkdir -p gpurun_out/r04
for tag in "selfseed --self-seed" "sem1 --semantics 1 --iters 3" "sem1_selfseed --semantics 1 --iters 3 --self-seed"; do
  set -- $tag; name=$1; shift
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-legs --host-pairs 0 "$@" > gpurun_out/r04/side_$name.json 2> gpurun_out/r04/side_$name.err
  python3 -c "import json; j=json.loads(open('gpurun_out/r04/side_$name.json').read().strip().splitlines()[-1]); print('$name', round(j['ms_per_step'],3), 'ms', round(j['value'],1), 'pairs/s', j.get('sequence_device',{}).get('value'))"
done
