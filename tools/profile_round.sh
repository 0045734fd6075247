#!/bin/bash
# Refresh the judged profile artefacts on the GPU box: tools/profile_round.sh <tag>
# kernel-trace stats of the default bench, then FETCH_SIZE and WRITE_SIZE passes (separate runs, --pmc only).
set -e
tag=$1
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/prof_$tag
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats -o stats --output-format csv -- python3 $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --host-pairs 0 > $out/bench_under_rocprof.json 2> $out/stats.err
echo stats done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o fetch --output-format csv -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --host-pairs 0 > $out/fetch.log 2>&1
echo fetch done
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $out/write -o write --output-format csv -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --host-pairs 0 > $out/write.log 2>&1
echo write done
python3 $root/tools/make_traffic.py $(find $out/fetch -name "*counter_collection.csv" | head -1) $(find $out/write -name "*counter_collection.csv" | head -1) $out/traffic.json
find $out/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
head -12 $out/kernel_stats.csv
