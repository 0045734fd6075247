#!/bin/bash
# stream-class matrix of round 4: every leg alone (a process of its own) and with all handles alive, per PM_STREAM_PRIO
mkdir -p gpurun_out/r04
out=gpurun_out/r04/streams.txt
: > $out
run() { echo "## PM_STREAM_PRIO=${PM_STREAM_PRIO:-default} $*" >> $out; timeout -k 10 300 python tools/stream_matrix.py "$@" >> $out 2>&1; }
for prio in default 1 0; do
  if [ $prio = default ]; then unset PM_STREAM_PRIO; else export PM_STREAM_PRIO=$prio; fi
  run --alive --legs single,batch,pipe,sync,batch_u8,tiled || exit 1
  if [ $prio != 0 ]; then
    for leg in single batch pipe tiled; do run --legs $leg || exit 1; done
  fi
done
export PM_STREAM_PRIO=1
run --alive --dummies 3 --legs single,batch,pipe,sync,batch_u8,tiled
cat $out
