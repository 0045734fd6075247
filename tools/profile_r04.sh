#!/bin/bash
# Round-4 profile artefacts on the GPU box: tools/profile_r04.sh <tag>   (results -> gpurun_out/prof_<tag>/)
# Kernel-trace stats of the default bench command (scalar headline and plane mode), then separate --pmc passes:
# FETCH_SIZE, WRITE_SIZE (traffic.json) and the issue side (valu.json).  rocprofv3 runs python3 directly.
tag=$1
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/prof_$tag
mkdir -p $out
common="--warmup 1 --no-cpu-baseline --no-side-legs --host-pairs 0"
pmc() {  # pmc <mode> <name> <counters...>
  mode=$1; name=$2; shift; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" -d $out/${name}_$mode -o $name --output-format csv -- python3 $root/bench.py --steps 2 $common --no-profile --mode $mode > $out/${name}_$mode.log 2>&1
  echo "$name $mode done"
}
csvof() { find $out/$1 -name "*counter_collection.csv" | head -1; }
for mode in scalar planes; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats_$mode -o stats --output-format csv -- python3 $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-side-legs --host-pairs 0 --mode $mode > $out/bench_under_rocprof_$mode.json 2> $out/stats_$mode.err
  find $out/stats_$mode -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats_$mode.csv
  echo "stats $mode done"
  pmc $mode fetch FETCH_SIZE
  pmc $mode write WRITE_SIZE
  pmc $mode insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES
  pmc $mode active SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
  pmc $mode ta TA_TA_BUSY_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE
  pmc $mode tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
  for n in insts active ta tcp; do python3 $root/tools/pmc_summary.py $(csvof ${n}_$mode) > $out/pmc_${n}_$mode.txt; done
done
python3 $root/tools/make_traffic.py $(csvof fetch_scalar) $(csvof write_scalar) $out/traffic.json > /dev/null
python3 $root/tools/make_traffic.py $(csvof fetch_planes) $(csvof write_planes) $out/traffic.json $out/traffic.json > /dev/null
python3 $root/tools/make_valu.py $(csvof insts_scalar) $(csvof active_scalar) $(csvof ta_scalar) $out/valu.json > /dev/null
python3 $root/tools/make_valu.py $(csvof insts_planes) $(csvof active_planes) $(csvof ta_planes) $out/valu.json $out/valu.json > /dev/null
# the raw counter dumps are large: keep the summaries only
rm -rf $out/fetch_* $out/write_* $out/insts_* $out/active_* $out/ta_* $out/tcp_* $out/stats_scalar $out/stats_planes
ls $out
head -14 $out/kernel_stats_scalar.csv
