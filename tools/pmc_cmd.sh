#!/bin/bash
# One PMC pass over an arbitrary python script: tools/pmc_cmd.sh <name> "<script and args>" <counter> [<counter> ...]
# (rocprofv3 --pmc only, no tracing domains; the profiled program is python3 itself)
name=$1; script=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$name
rocprofv3 --pmc "$@" -d $out -o $name --output-format csv -- python3 $GRAFT_REPO_ROOT/$script > $out.log 2>&1
find $out -name "*counter_collection.csv" | head -1 | xargs -I{} python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py {}
