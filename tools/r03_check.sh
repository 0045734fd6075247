#!/bin/bash
# round-3 GPU check: parity tests, short engine fuzz, A/B of the old and new run step
set -o pipefail
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03/pytest.log
tail -5 gpurun_out/r03/pytest.log
for r in 0 1; do
  echo "== PM_RUN3=$r"
  PM_RUN3=$r timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --host-pairs 0 --no-side-legs > gpurun_out/r03/bench_run3_$r.json 2> gpurun_out/r03/bench_run3_$r.err
  python - <<PY
import json
r=json.loads(open("gpurun_out/r03/bench_run3_$r.json").read().strip().splitlines()[-1])
print(r["value"], r["ms_per_step"], {k: round(v,3) for k,v in r.get("kernels_ms_per_step",{}).items()})
PY
done
