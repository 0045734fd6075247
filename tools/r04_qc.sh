#!/bin/bash
# the quad register cache of the row sweeps (PM_RUN3_QC=1, tuning build): parity, then A/B
mkdir -p gpurun_out/r04
export PM_LIB=$PWD/ocean-perception_amd/lib/libvehicle_pm_gpu_tuning.so
PM_RUN3_QC=1 timeout -k 10 600 python tools/fuzz_engines.py --cases 80 --seed 91 > gpurun_out/r04/qc_fuzz.log 2>&1 || { tail -5 gpurun_out/r04/qc_fuzz.log; exit 1; }
tail -1 gpurun_out/r04/qc_fuzz.log
PM_RUN3_QC=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_size or match_both or window_sizes or recipe" > gpurun_out/r04/qc_tests.log 2>&1 || { tail -20 gpurun_out/r04/qc_tests.log; exit 1; }
tail -1 gpurun_out/r04/qc_tests.log
out=gpurun_out/r04/qc.txt
: > $out
A="--steps 20 --warmup 5 --no-side-legs --no-cpu-baseline --host-pairs 0 --profile-every 1"
for qc in 0 1 0 1; do
  echo "## PM_RUN3_QC=$qc" >> $out
  PM_RUN3_QC=$qc python bench.py $A 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['value'],1), {k: round(v,4) for k,v in j['kernels_ms_per_step'].items()})" >> $out
  PM_RUN3_QC=$qc timeout -k 10 300 python tools/stream_matrix.py --legs pipe_dev,batch 2>&1 | grep -v amdgpu.ids >> $out
done
cat $out
