#!/bin/bash
mkdir -p gpurun_out/r04
out=gpurun_out/r04/lds_extra.txt
: > $out
export PM_LIB=$PWD/ocean-perception_amd/lib/libvehicle_pm_gpu_tuning.so
run() { echo "## $1" >> $out; timeout -k 10 300 python tools/stream_matrix.py --legs single,pipe_dev 2>&1 | grep -v amdgpu.ids >> $out; }
run default
for k in 2 4 8 16; do PM_RUN3_LDS_EXTRA_KB=$k run "PM_RUN3_LDS_EXTRA_KB=$k"; done
cat $out
