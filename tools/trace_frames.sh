#!/bin/bash
# Kernel trace of a few frames of the default bench workload: tools/trace_frames.sh <outdir-under-gpurun_out>
# -> gpurun_out/<outdir>/kernel_trace.csv (start / end timestamps per launch); tools/trace_gaps.py reads it.
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/raw -o trace --output-format csv -- python3 $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-side-legs --host-pairs 0 --no-profile > $out/bench.json 2> $out/err.txt
find $out/raw -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $out/kernel_trace.csv
rm -rf $out/raw
ls -la $out
