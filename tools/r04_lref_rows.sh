#!/bin/bash
# row sweeps with their reference quads staged in LDS at a 6-dword stride (three workgroups per CU), tuning build
mkdir -p gpurun_out/r04
out=gpurun_out/r04/lref_rows.txt
: > $out
export PM_LIB=$PWD/ocean-perception_amd/lib/libvehicle_pm_gpu_tuning.so
run() { echo "## $1" >> $out; timeout -k 10 300 python tools/stream_matrix.py --legs single,pipe_dev,batch 2>&1 | grep -v amdgpu.ids >> $out; }
run default
PM_RUN2_LREF=3 PM_RUN2_LREF_KB=53 run "PM_RUN2_LREF=3 PM_RUN2_LREF_KB=53"
run default
PM_RUN2_LREF=3 PM_RUN2_LREF_KB=53 run "PM_RUN2_LREF=3 PM_RUN2_LREF_KB=53"
cat $out
