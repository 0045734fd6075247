#!/bin/bash
# Profile artefacts of a round on the GPU box: ROUND=r06 tools/profile.sh <head-sha>   (results -> gpurun_out/prof_$ROUND/)
# ONE script for everything under profiles/ that describes the benchmarked tree: kernel-trace stats of the default bench
# command (scalar headline, plane mode, plane mode with two neighbours), then separate --pmc passes (never combined with a
# tracing domain): FETCH_SIZE, WRITE_SIZE (traffic.json), the issue side (valu.json, incl. the two-neighbour variant), TCP
# and TCC requests, and the FETCH_SIZE / WRITE_SIZE calibration.  The sha of the tree is written into every file.
head=${1:-unknown}
round=${ROUND:-r06}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/prof_$round
mkdir -p $out
common="--warmup 1 --no-cpu-baseline --no-side-legs --host-pairs 0"
pmc() {  # pmc <tag> "<mode args>" <name> <counters...>
  tag=$1; margs=$2; name=$3; shift; shift; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" -d $out/${name}_$tag -o $name --output-format csv -- python3 $root/bench.py --steps 2 $common --no-profile $margs > $out/${name}_$tag.log 2>&1
  echo "$name $tag done"
}
csvof() { find $out/$1 -name "*counter_collection.csv" | head -1; }
stamp() { for f in "$@"; do [ -f "$f" ] && sed -i "1i # tree of commit $head (tools/profile.sh)" "$f"; done; }
for tag in scalar planes planes2; do
  case $tag in scalar) margs="--mode scalar";; planes) margs="--mode planes";; planes2) margs="--mode planes --plane-neighbours 1";; esac
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats_$tag -o stats --output-format csv -- python3 $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-side-legs --host-pairs 0 $margs > $out/bench_under_rocprof_$tag.json 2> $out/stats_$tag.err
  find $out/stats_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats_$tag.csv
  echo "stats $tag done"
  pmc $tag "$margs" fetch FETCH_SIZE
  pmc $tag "$margs" write WRITE_SIZE
  pmc $tag "$margs" insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES
  pmc $tag "$margs" active SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
  pmc $tag "$margs" ta TA_TA_BUSY_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE
  if [ $tag != planes2 ]; then
    pmc $tag "$margs" tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
    pmc $tag "$margs" tcc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
    for n in insts active ta tcp tcc; do python3 $root/tools/pmc_summary.py $(csvof ${n}_$tag) > $out/pmc_${n}_$tag.txt; done
  fi
done
python3 $root/tools/make_traffic.py $(csvof fetch_scalar) $(csvof write_scalar) $out/traffic.json --head $head > /dev/null
python3 $root/tools/make_traffic.py $(csvof fetch_planes) $(csvof write_planes) $out/traffic.json $out/traffic.json > /dev/null
python3 $root/tools/make_traffic.py $(csvof fetch_planes2) $(csvof write_planes2) $out/traffic.json $out/traffic.json --suffix @two_neighbours > /dev/null
python3 $root/tools/make_valu.py $(csvof insts_scalar) $(csvof active_scalar) $(csvof ta_scalar) $out/valu.json --head $head > /dev/null
python3 $root/tools/make_valu.py $(csvof insts_planes) $(csvof active_planes) $(csvof ta_planes) $out/valu.json $out/valu.json > /dev/null
python3 $root/tools/make_valu.py $(csvof insts_planes2) $(csvof active_planes2) $(csvof ta_planes2) $out/valu.json $out/valu.json --suffix @two_neighbours > /dev/null
ROUND=$round bash $root/tools/fetch_calib.sh > /dev/null 2>&1
cp $root/gpurun_out/$round/fetch_calib.txt $out/calib_fetch_write.txt
stamp $out/kernel_stats_*.csv $out/pmc_*.txt $out/calib_fetch_write.txt
# the raw counter dumps are large: keep the summaries only
rm -rf $out/fetch_* $out/write_* $out/insts_* $out/active_* $out/ta_* $out/tcp_* $out/tcc_* $out/stats_scalar $out/stats_planes $out/stats_planes2
ls $out
head -14 $out/kernel_stats_scalar.csv
