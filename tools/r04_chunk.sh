#!/bin/bash
mkdir -p gpurun_out/r04
out=gpurun_out/r04/chunk.txt
: > $out
export PM_LIB=$PWD/ocean-perception_amd/lib/libvehicle_pm_gpu_tuning.so
for c in 2 3 4 2; do echo "## PM_PAIR_CHUNK=$c" >> $out; PM_PAIR_CHUNK=$c timeout -k 10 300 python tools/stream_matrix.py --legs pipe_dev,batch 2>&1 | grep -v amdgpu.ids >> $out; done
cat $out
