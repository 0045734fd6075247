#!/bin/bash
mkdir -p gpurun_out/r04
timeout -k 10 900 python tools/fuzz_planes.py --cases 400 --seed 431 > gpurun_out/r04/fuzz_planes2.log 2>&1; rc=$?
tail -2 gpurun_out/r04/fuzz_planes2.log; echo "exit $rc"
