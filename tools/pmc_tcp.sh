#!/bin/bash
# L1 / TA / TD detail passes of the default bench: tools/pmc_tcp.sh <tag>
tag=$1
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  out=$root/gpurun_out/pmc_${tag}_$name
  timeout -k 10 300 rocprofv3 --pmc "$@" -d $out -o $name --output-format csv -- python3 $root/bench.py --steps 4 --warmup 1 --no-cpu-baseline --host-pairs 0 --no-side-legs --no-profile > $out.log 2>&1
  f=$(find $out -name "*counter_collection.csv" | head -1)
  python3 $root/tools/pmc_summary.py $f > $root/gpurun_out/${tag}_pmc_$name.txt
  echo "pass $name done: $(wc -l < $root/gpurun_out/${tag}_pmc_$name.txt) kernels"
}
pass tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
pass tcp2 TD_TC_STALL_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum
pass tcp3 TD_LOAD_WAVEFRONT_sum TD_COALESCABLE_WAVEFRONT_sum TA_FLAT_COALESCEABLE_WAVEFRONTS_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
pass ta2 TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE TCP_TCR_TCP_STALL_CYCLES_sum
