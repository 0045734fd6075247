#!/usr/bin/env python3
"""Quick timing of PM_MODE_PLANES at 1280x720 (per-stage HIP event times). Usage: bench_planes.py [f16] [pairs]"""
import json, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ocean-perception_amd", "python"))
import numpy as np, torch
import pm_ctypes as pm, synth
f16 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows, cols = 720, 1280
dev = torch.device("cuda:0")
pairs = [synth.make_pair(i, rows, cols) for i in range(min(nb, 4))]
st = lambda k: torch.from_numpy(np.stack([pairs[i % len(pairs)][k] for i in range(nb)])).to(dev).contiguous()
L, R = st("left"), st("right")
DL = torch.empty((nb, rows, cols), dtype=torch.float32, device=dev); DR = torch.empty_like(DL)
prm = pm.default_params(0, patch=11, patchmatch_iters=8, mode=pm.PM_MODE_PLANES, state_dtype=f16)
e = pm.Engine(prm, max_rows=rows, max_cols=cols, max_batch=nb)
step = lambda: e.match_device(nb, L.data_ptr(), R.data_ptr(), rows, cols, None, None, DL.data_ptr(), DR.data_ptr())
for _ in range(3): step()
e.synchronize()
t0 = time.perf_counter(); K = 10
for _ in range(K): step()
e.synchronize(); dt = (time.perf_counter() - t0) / K
e.profile_enable(True); e.profile_read(); step(); prof = e.profile_read(); e.profile_enable(False)
gt = torch.from_numpy(pairs[0]["gt"]).to(dev); d = DL[0]; ok = d > 0
print(json.dumps({"f16": f16, "pairs": nb, "ms_per_step": dt * 1e3, "pairs_per_s": nb / dt,
                  "kernels_ms": {k: v[1] for k, v in prof.items() if v[0]},
                  "launches": {k: v[0] for k, v in prof.items() if v[0]},
                  "valid": float(ok.float().mean()), "within1": float(((d - gt).abs()[ok] < 1).float().mean())}))
