#!/bin/bash
# dedicated hardware queues (streams with a full CU mask, PM_STREAM_PRIO=2 in the tuning build) against the high class (1)
mkdir -p gpurun_out/r04
out=gpurun_out/r04/streams2.txt
: > $out
export PM_LIB=$PWD/ocean-perception_amd/lib/libvehicle_pm_gpu_tuning.so
run() { echo "## PM_STREAM_PRIO=$PM_STREAM_PRIO $*" >> $out; timeout -k 10 400 python tools/stream_matrix.py "$@" 2>&1 | grep -v amdgpu.ids >> $out; }
for prio in 2 1; do
  export PM_STREAM_PRIO=$prio
  run --alive || exit 1
  run --alive --dummies 3 --legs tiled,batch_u8_pinned,sync,pipe_dev,pipe_pinned,pipe,batch,single || exit 1
done
export PM_STREAM_PRIO=2
for leg in single batch pipe_pinned pipe_dev batch_u8_pinned tiled; do run --legs $leg || exit 1; done
cat $out
