#!/bin/bash
# quick A/B of a sweep-kernel change: engine parity, then headline / batch / device-sequence rates
root=$GRAFT_REPO_ROOT
mkdir -p $root/gpurun_out/r05
cd $root
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_tiled.py -m gpu -x -q 2>&1 | tail -2
timeout -k 10 300 python tools/fuzz_engines.py --cases ${CASES:-150} --seed 77 2>&1 | tail -1
timeout -k 10 300 python tools/fuzz_engines.py --cases 10 --seed 78 --big 2>&1 | tail -1
for rep in 1 2; do
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-side-legs --host-pairs 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', round(j['value'],1), 'ms', round(j['ms_per_step'],3), 'row', round(j['kernels_ms_per_step'].get('sweep_row',0),3), 'col', round(j['kernels_ms_per_step'].get('sweep_col',0),3), 'noise', round(j['kernels_ms_per_step'].get('noise_cost',0),3))"
done
python bench.py --steps 6 --warmup 2 --pairs-per-gpu 32 --no-cpu-baseline --no-side-legs --host-pairs 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch32', round(j['value'],1))"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-legs --host-pairs 8 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('seq_device', round(j['sequence_device']['value'],1), 'host_seq', round(j['host_sequence_all_ranks']['value'],1))"
