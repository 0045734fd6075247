#!/bin/bash
# the driver's round-end command on the GPU box, with its wall clock
mkdir -p gpurun_out/r03
s=$(date +%s.%N)
python bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err
echo "rc=$? wall=$(echo "$(date +%s.%N) - $s" | bc) s"
python - <<PY
import json
r=json.loads(open("gpurun_out/r03/bench_default.json").read().strip().splitlines()[-1])
print(r["value"], r["ms_per_step"], r["step_ms"])
print("roofline", {k:v for k,v in r["roofline"].items() if k not in ("note","formula")})
print("cpu", r["cpu_baseline"]["value"], r["cpu_baseline"]["all_cores"]["value"])
print("host", r["host_buffers"])
t=r["tiled_4096x2160"]; print("tiled", t.get("ms_per_frame"), t.get("eight_bands_on_this_device"), t.get("error"))
print("planes f32", r["planes"]["f32"]["value"], r["planes"]["f32"]["roofline"]["kernel"], r["planes"]["f32"]["roofline"]["frac"], "f16e", r["planes"]["f16_enhanced"]["value"], r["planes"]["f16_enhanced"]["ms_per_frame"])
PY
