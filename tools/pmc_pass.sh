#!/bin/bash
# One PMC pass over the default bench: tools/pmc_pass.sh <name> <counter> [<counter> ...]
# (rocprofv3 --pmc only, no tracing domains; target is python3 itself)
name=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$name
rocprofv3 --pmc "$@" -d $out -o $name --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --host-pairs 0 > $out.log 2>&1
find $out -name "*counter_collection.csv" | head -1 | xargs -I{} python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py {}
