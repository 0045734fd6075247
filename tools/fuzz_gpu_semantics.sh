#!/bin/bash
# differential fuzz after the round-6 rebuild of the PM_SEM_GPU run step and the seeder's launch sequence: tools/fuzz_gpu_semantics.sh <head-sha>
head=${1:-unknown}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/${ROUND:-r06}
mkdir -p $out
cd $root
f=$out/fuzz_gpu_semantics.txt
echo "Differential fuzz on the tree of commit $head (tools/fuzz_gpu_semantics.sh, one MI355X; every case bit-identical or the run stops)" > $f
run() { name=$1; shift; echo "## $name $*" >> $f; timeout -k 10 900 python tools/$name "$@" > $out/fuzzg_$name.log 2>&1; rc=$?; tail -1 $out/fuzzg_$name.log >> $f; echo "exit $rc" >> $f; echo "$name done ($rc)"; [ $rc -eq 0 ]; }
run fuzz_engines.py --cases 2000 --seed 6101 --gpu-share 1.0 || { cat $f; exit 1; }
run fuzz_engines.py --cases 40 --seed 6102 --big --gpu-share 1.0 || { cat $f; exit 1; }
run fuzz_selfseed.py --cases 800 --seed 6103 || { cat $f; exit 1; }
run fuzz_seed.py --cases 800 --seed 6104 || { cat $f; exit 1; }
run fuzz_api.py --cases 800 --seed 6105 || { cat $f; exit 1; }
cat $f
