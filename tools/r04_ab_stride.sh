#!/bin/bash
mkdir -p gpurun_out/r04
out=gpurun_out/r04/ab_stride.txt
: > $out
for rep in 1 2; do
for lib in libab_s7.so libab_s6.so; do
  echo "## $lib" >> $out
  PM_LIB=$PWD/ocean-perception_amd/lib/$lib timeout -k 10 300 python tools/stream_matrix.py --legs single,pipe_dev,batch 2>&1 | grep -v amdgpu.ids >> $out
done
done
cat $out
