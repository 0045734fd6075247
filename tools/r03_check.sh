#!/bin/bash
# round-3 GPU check: parity tests, then the default bench line without side legs
set -o pipefail
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a gpurun_out/r03/pytest.log
tail -5 gpurun_out/r03/pytest.log
[ $rc = 0 ] || exit $rc
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --host-pairs 0 --no-side-legs > gpurun_out/r03/bench.json 2> gpurun_out/r03/bench.err
python - <<PY
import json
r=json.loads(open("gpurun_out/r03/bench.json").read().strip().splitlines()[-1])
print(r["value"], r["ms_per_step"], {k: round(v,3) for k,v in r.get("kernels_ms_per_step",{}).items()})
PY
