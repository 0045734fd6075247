#!/bin/bash
# what loading a chain into LDS and storing it back costs (PM_RUN3_SKIP: the sweep kernels without their steps)
mkdir -p gpurun_out/r04
export PM_LIB=$PWD/ocean-perception_amd/lib/libvehicle_pm_gpu_tuning.so
A="--steps 16 --warmup 4 --profile-every 1 --no-side-legs --no-cpu-baseline --host-pairs 0"
python bench.py $A > gpurun_out/r04/skip_ref.json 2> gpurun_out/r04/skip_ref.err
PM_RUN3_SKIP=1 python bench.py $A > gpurun_out/r04/skip_on.json 2> gpurun_out/r04/skip_on.err
python3 - <<'P'
import json
for f in ("skip_ref", "skip_on"):
    try:
        j = json.loads(open(f"gpurun_out/r04/{f}.json").read().strip().splitlines()[-1])
        print(f, j["value"], j["ms_per_step"], {k: round(v, 4) for k, v in j["kernels_ms_per_step"].items()})
    except Exception as e:
        print(f, "failed", e)
P
tail -3 gpurun_out/r04/skip_on.err
