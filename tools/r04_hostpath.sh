#!/bin/bash
# round 4: parity of the rebuilt host path, then its throughput with every handle kind alive and each leg alone
mkdir -p gpurun_out/r04
out=gpurun_out/r04/hostpath.txt
: > $out
timeout -k 10 120 python tools/probe_owned.py > gpurun_out/r04/probe_owned.log 2>&1; tail -5 gpurun_out/r04/probe_owned.log
timeout -k 10 900 python -m pytest --tb=short tests/test_gpu_parity.py tests/test_seed.py tests/test_cpp_wrapper.py -x -q -m gpu > gpurun_out/r04/hostpath_tests.log 2>&1 || { tail -40 gpurun_out/r04/hostpath_tests.log; exit 1; }
tail -3 gpurun_out/r04/hostpath_tests.log
run() { echo "## $*" >> $out; timeout -k 10 400 python tools/stream_matrix.py "$@" >> $out 2>&1; }
run --alive || exit 1
run --alive --self-seed --legs single,batch,pipe,pipe_pinned,pipe_dev,batch_u8,batch_u8_pinned || exit 1
for leg in single batch pipe pipe_pinned pipe_dev batch_u8_pinned tiled; do run --legs $leg || exit 1; done
run --alive --dummies 3 || exit 1
cat $out
