#!/bin/bash
# a longer differential-fuzz campaign on the final tree (new seeds): tools/fuzz_long.sh <head-sha>
head=${1:-unknown}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/${ROUND:-r06}
mkdir -p $out
cd $root
f=$out/fuzz_long.txt
echo "Long differential-fuzz campaign on the tree of commit $head (tools/fuzz_long.sh, one MI355X; every case bit-identical or the run stops)" > $f
run() { name=$1; shift; echo "## $name $*" >> $f; timeout -k 10 1000 python tools/$name "$@" > $out/fuzzl_$name.log 2>&1; rc=$?; tail -1 $out/fuzzl_$name.log >> $f; echo "exit $rc" >> $f; echo "$name done ($rc)"; [ $rc -eq 0 ]; }
run fuzz_engines.py --cases 2500 --seed 5701 || { cat $f; exit 1; }
run fuzz_engines.py --cases 60 --seed 5707 --big || { cat $f; exit 1; }
run fuzz_planes.py --cases 2500 --seed 5702 || { cat $f; exit 1; }
run fuzz_api.py --cases 2000 --seed 5703 || { cat $f; exit 1; }
run fuzz_selfseed.py --cases 1000 --seed 5704 || { cat $f; exit 1; }
run fuzz_seed.py --cases 1200 --seed 5705 || { cat $f; exit 1; }
run fuzz_tiled.py --cases 1000 --seed 5706 || { cat $f; exit 1; }
cat $f
