#!/bin/bash
# PMC passes over bench.py with extra arguments: tools/pmc_any.sh <tag> "<extra bench args>" -- pass specs "name:C1,C2,.."
# Separate --pmc runs, no tracing domains, python3 directly after "--".  Summaries -> gpurun_out/<tag>_pmc_<name>.txt
tag=$1; extra=$2; shift; shift
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  name=${spec%%:*}; ctrs=${spec#*:}
  out=$root/gpurun_out/pmc_${tag}_$name
  timeout -k 10 300 rocprofv3 --pmc ${ctrs//,/ } -d $out -o $name --output-format csv -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --host-pairs 0 --no-side-legs --no-profile $extra > $out.log 2>&1
  f=$(find $out -name "*counter_collection.csv" | head -1)
  python3 $root/tools/pmc_summary.py $f > $root/gpurun_out/${tag}_pmc_$name.txt
  echo "pass $name done: $(wc -l < $root/gpurun_out/${tag}_pmc_$name.txt) kernels"
done
