#!/bin/bash
# LDS-side counters of the plane kernels (round 5: is the window loop LDS-bound after the packed lerp?)
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/r05/pmc_lds
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for spec in "lds1:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "lds2:SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS"; do
  name=${spec%%:*}; ctrs=${spec#*:}
  timeout -k 10 300 rocprofv3 --pmc $ctrs -d $out/$name -o $name --output-format csv -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side-legs --host-pairs 0 --no-profile --mode planes > $out/$name.log 2>&1
  f=$(find $out/$name -name "*counter_collection.csv" | head -1)
  python3 $root/tools/pmc_summary.py $f > $out/../pmc_${name}_planes.txt
  rm -rf $out/$name
done
cat $out/../pmc_lds1_planes.txt $out/../pmc_lds2_planes.txt
