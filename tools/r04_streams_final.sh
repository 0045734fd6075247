#!/bin/bash
# the handle-constellation matrix once more on the final build of round 4: every leg alive beside the others vs alone
mkdir -p gpurun_out/r04
out=gpurun_out/r04/streams_final.txt
: > $out
run() { echo "## $*" >> $out; timeout -k 10 400 python tools/stream_matrix.py "$@" 2>&1 | grep -v amdgpu.ids >> $out; }
run --alive || exit 1
run --alive --dummies 3 --legs tiled,batch_u8_pinned,sync,pipe_dev,pipe_pinned,pipe,batch,single || exit 1
for leg in single batch pipe pipe_pinned pipe_dev sync batch_u8 batch_u8_pinned tiled; do run --legs $leg || exit 1; done
cat $out
