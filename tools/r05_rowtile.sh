#!/bin/bash
# tiled row sweeps (k_runrow_tiled): parity on the shipped build at full size, parity of the small cases on the tuning
# build with the tiled kernel forced onto short chains and a tiny reach (many global-path fallbacks), then timing
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/r05
mkdir -p $out
cd $root
T=$root/ocean-perception_amd/lib/libvehicle_pm_gpu_tuning.so
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_size" > $out/f_tests_full.log 2>&1; tail -3 $out/f_tests_full.log
PM_LIB=$T PM_ROWTILE_MIN=16 PM_ROWTILE_REACH=8 PM_ROWTILE_PHASES=3 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q > $out/f_tests_forced.log 2>&1; tail -3 $out/f_tests_forced.log
PM_LIB=$T PM_ROWTILE_MIN=16 PM_ROWTILE_REACH=40 PM_ROWTILE_PHASES=2 timeout -k 10 300 python tools/fuzz_engines.py --cases 120 2>&1 | tail -2
PM_LIB=$T PM_ROWTILE_MIN=16 PM_ROWTILE_REACH=3 PM_ROWTILE_PHASES=5 timeout -k 10 300 python tools/fuzz_engines.py --cases 120 --seed 7 2>&1 | tail -2
timeout -k 10 300 python tools/fuzz_engines.py --cases 12 --big 2>&1 | tail -2
for ph in ${PHASES:-0 2 3 4 6 8}; do
  for reach in ${REACHES:-160}; do
    PM_LIB=$T PM_ROWTILE=$([ $ph = 0 ] && echo 0 || echo 1) PM_ROWTILE_PHASES=$ph PM_ROWTILE_REACH=$reach python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-side-legs --host-pairs 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('phases $ph reach $reach: pairs/s', round(j['value'],1), 'ms', round(j['ms_per_step'],3), 'row', round(j['kernels_ms_per_step'].get('sweep_row',0),3), 'col', round(j['kernels_ms_per_step'].get('sweep_col',0),3), j['check']['deterministic_across_steps'])"
  done
done
python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-side-legs --host-pairs 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shipped: pairs/s', round(j['value'],1), 'ms', round(j['ms_per_step'],3), j['kernels_ms_per_step'])"
