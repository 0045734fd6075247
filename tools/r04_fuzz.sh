#!/bin/bash
# the differential fuzzers on the final build of round 4 (each prints one line per case; the tails are kept)
mkdir -p gpurun_out/r04
out=gpurun_out/r04/fuzz.txt
: > $out
run() { name=$1; shift; echo "## $name $*" >> $out; timeout -k 10 900 python tools/$name "$@" > gpurun_out/r04/fuzz_$name.log 2>&1; rc=$?; tail -1 gpurun_out/r04/fuzz_$name.log >> $out; echo "exit $rc" >> $out; [ $rc -eq 0 ]; }
run fuzz_api.py --cases 400 --seed 421 || { cat $out; exit 1; }
run fuzz_engines.py --cases 300 --seed 422 || { cat $out; exit 1; }
run fuzz_planes.py --cases 300 --seed 423 || { cat $out; exit 1; }
run fuzz_seed.py --cases 200 --seed 424 || { cat $out; exit 1; }
run fuzz_selfseed.py --cases 150 --seed 425 || { cat $out; exit 1; }
run fuzz_tiled.py --cases 120 --seed 426 || { cat $out; exit 1; }
cat $out
