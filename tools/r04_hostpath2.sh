#!/bin/bash
mkdir -p gpurun_out/r04
out=gpurun_out/r04/hostpath2.txt
: > $out
timeout -k 10 900 python -m pytest --tb=short tests/test_gpu_parity.py tests/test_cpp_wrapper.py -x -q -m gpu > gpurun_out/r04/hostpath2_tests.log 2>&1 || { tail -40 gpurun_out/r04/hostpath2_tests.log; exit 1; }
tail -3 gpurun_out/r04/hostpath2_tests.log
run() { echo "## $*" >> $out; timeout -k 10 400 python tools/stream_matrix.py "$@" >> $out 2>&1; }
for leg in single batch pipe pipe_pinned pipe_dev batch_u8 batch_u8_pinned; do run --legs $leg || exit 1; done
run --alive || exit 1
run --alive --self-seed --legs single,batch,pipe,pipe_pinned,pipe_dev,batch_u8,batch_u8_pinned || exit 1
export GPU_MAX_HW_QUEUES=8
echo "## GPU_MAX_HW_QUEUES=8" >> $out
for leg in pipe_pinned pipe_dev; do run --legs $leg || exit 1; done
cat $out
