#!/bin/bash
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest --tb=short tests/test_gpu_parity.py tests/test_golden.py tests/test_tiled.py -x -q -m gpu > gpurun_out/r04/delta_tests.log 2>&1 || { tail -40 gpurun_out/r04/delta_tests.log; exit 1; }
tail -2 gpurun_out/r04/delta_tests.log
A="--steps 20 --warmup 5 --no-side-legs --no-cpu-baseline --host-pairs 0"
python bench.py $A > gpurun_out/r04/delta_bench.json 2> gpurun_out/r04/delta_bench.err
python bench.py $A --pairs-per-gpu 4 --steps 8 > gpurun_out/r04/delta_bench4.json 2>> gpurun_out/r04/delta_bench.err
python3 - <<'P'
import json
for f in ("delta_bench", "delta_bench4"):
    try:
        j = json.loads(open(f"gpurun_out/r04/{f}.json").read().strip().splitlines()[-1])
        print(f, round(j["value"], 1), round(j["ms_per_step"], 3), {k: round(v, 4) for k, v in j.get("kernels_ms_per_step", {}).items()})
    except Exception as e:
        print(f, "failed", e)
P
timeout -k 10 300 python tools/stream_matrix.py --legs single,replay,pipe_dev,pipe_pinned,batch 2>&1 | grep -v amdgpu.ids
