#!/bin/bash
# the differential fuzzers on the final build of round 4 (each prints one line per case; the tails are kept)
mkdir -p gpurun_out/r04
out=gpurun_out/r04/fuzz.txt
: > $out
run() { name=$1; shift; echo "## $name $*" >> $out; timeout -k 10 900 python tools/$name "$@" > gpurun_out/r04/fuzz_$name.log 2>&1; rc=$?; tail -1 gpurun_out/r04/fuzz_$name.log >> $out; echo "exit $rc" >> $out; [ $rc -eq 0 ]; }
run fuzz_api.py --cases 800 --seed 461 || { cat $out; exit 1; }
run fuzz_engines.py --cases 600 --seed 462 || { cat $out; exit 1; }
run fuzz_planes.py --cases 500 --seed 463 || { cat $out; exit 1; }
run fuzz_seed.py --cases 400 --seed 464 || { cat $out; exit 1; }
run fuzz_selfseed.py --cases 300 --seed 465 || { cat $out; exit 1; }
run fuzz_tiled.py --cases 200 --seed 466 || { cat $out; exit 1; }
cat $out
