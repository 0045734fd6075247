#!/bin/bash
# Issue-side PMC passes of the default bench (scalar headline): tools/pmc_round.sh <tag>
# Separate --pmc runs, no tracing domains, python3 directly after "--".  Summaries -> gpurun_out/<tag>_pmc_*.txt
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
pass insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS
pass active SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES
pass wait SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
pass ta TA_TA_BUSY_sum TD_TD_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
