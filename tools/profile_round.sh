#!/bin/bash
# Refresh the judged profile artefacts on the GPU box: tools/profile_round.sh <tag>
# For the headline workload (scalar mode) and for the plane mode: kernel-trace stats of the bench, then FETCH_SIZE and
# WRITE_SIZE passes (separate runs, --pmc only; the profiled program is python3 itself).
set -e
tag=$1
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/prof_$tag
mkdir -p $out
common="--warmup 3 --no-cpu-baseline --no-side-legs --host-pairs 0"
for mode in scalar planes; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats_$mode -o stats --output-format csv -- python3 $root/bench.py --steps 20 $common --mode $mode > $out/bench_under_rocprof_$mode.json 2> $out/stats_$mode.err
  echo stats $mode done
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $out/fetch_$mode -o fetch --output-format csv -- python3 $root/bench.py --steps 3 $common --mode $mode > $out/fetch_$mode.log 2>&1
  echo fetch $mode done
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $out/write_$mode -o write --output-format csv -- python3 $root/bench.py --steps 3 $common --mode $mode > $out/write_$mode.log 2>&1
  echo write $mode done
  find $out/stats_$mode -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats_$mode.csv
done
python3 $root/tools/make_traffic.py $(find $out/fetch_scalar -name "*counter_collection.csv" | head -1) $(find $out/write_scalar -name "*counter_collection.csv" | head -1) $out/traffic.json
python3 $root/tools/make_traffic.py $(find $out/fetch_planes -name "*counter_collection.csv" | head -1) $(find $out/write_planes -name "*counter_collection.csv" | head -1) $out/traffic.json $out/traffic.json
head -12 $out/kernel_stats_scalar.csv
head -8 $out/kernel_stats_planes.csv
