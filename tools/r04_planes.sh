#!/bin/bash
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest --tb=short tests/test_planes.py tests/test_enhance.py -x -q -m gpu > gpurun_out/r04/planes_tests.log 2>&1 || { tail -40 gpurun_out/r04/planes_tests.log; exit 1; }
tail -3 gpurun_out/r04/planes_tests.log
timeout -k 10 600 python tools/fuzz_planes.py --cases 120 --seed 41 > gpurun_out/r04/planes_fuzz.log 2>&1 || { tail -5 gpurun_out/r04/planes_fuzz.log; exit 1; }
tail -1 gpurun_out/r04/planes_fuzz.log
A="--steps 12 --warmup 4 --no-side-legs --no-cpu-baseline --host-pairs 0 --mode planes"
python bench.py $A > gpurun_out/r04/planes_f32.json 2> gpurun_out/r04/planes.err
python bench.py $A --state f16 --enhance > gpurun_out/r04/planes_f16e.json 2>> gpurun_out/r04/planes.err
python bench.py $A --pairs-per-gpu 4 --steps 6 > gpurun_out/r04/planes_f32_b4.json 2>> gpurun_out/r04/planes.err
python3 - <<'P'
import json
for f in ("planes_f32", "planes_f16e", "planes_f32_b4"):
    try:
        j = json.loads(open(f"gpurun_out/r04/{f}.json").read().strip().splitlines()[-1])
        print(f, round(j["value"], 1), round(j["ms_per_step"], 3), {k: round(v, 4) for k, v in j.get("kernels_ms_per_step", {}).items()}, j.get("check"))
    except Exception as e:
        print(f, "failed", e)
P
