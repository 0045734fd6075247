#!/bin/bash
mkdir -p gpurun_out/r04
t0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_default.json 2> gpurun_out/r04/bench_default.err
echo "exit $? after $(( $(date +%s) - t0 )) s"
tail -3 gpurun_out/r04/bench_default.err | grep -v amdgpu.ids
python3 - <<'P'
import json
j = json.loads(open("gpurun_out/r04/bench_default.json").read().strip().splitlines()[-1])
print("value", round(j["value"],1), "ms_per_step", round(j["ms_per_step"],3), "step_ms median", round(j["step_ms"]["median"],3))
print("roofline frac", round(j["roofline"]["frac"],4), "binding", j["roofline"].get("binding"))
print("host_buffers", json.dumps(j.get("host_buffers"), indent=None)[:1500])
print("sequence_device", j.get("sequence_device"))
print("batch", {k:(round(v["value"],1)) for k,v in j.get("batch",{}).items()})
print("reference_test_shape", j.get("reference_test_shape"))
print("planes", {k:((round(v["value"],1), v.get("check",{}).get("foreground_within_1px_of_truth")) if isinstance(v,dict) and "value" in v else None) for k,v in j.get("planes",{}).items()})
t = j.get("tiled_4096x2160", {})
print("tiled", t.get("ms_per_frame"), (t.get("eight_bands_on_this_device") or {}).get("ms_per_frame"), t.get("error"))
print("cpu_baseline", j.get("cpu_baseline",{}).get("value"), j.get("cpu_baseline",{}).get("all_cores",{}).get("value"))
P
