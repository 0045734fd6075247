#!/bin/bash
# the driver's round-end sequence on one box: every -m gpu test, smoke(), then the default bench line
mkdir -p gpurun_out/r04
t0=$(date +%s)
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu --tb=short > gpurun_out/r04/full_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r04/full_tests.log; echo "pytest exit $rc after $(( $(date +%s) - t0 )) s"
[ $rc -eq 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
bash tools/r04_bench.sh
