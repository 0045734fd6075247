#!/usr/bin/env python3
"""Batch throughput with the batch split over H handles fed from one thread (all launches are asynchronous).
usage: tools/bench_two_handles.py [pairs_total] [handles]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ocean-perception_amd", "python"))
import numpy as np, torch
import pm_ctypes as pm, synth

total = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows, cols = 720, 1280
nb = total // H
uniq = [synth.make_pair(i, rows, cols) for i in range(4)]
pairs = [uniq[i % 4] for i in range(nb)]
dev = torch.device("cuda:0")
st = lambda k: torch.from_numpy(np.stack([p[k] for p in pairs])).to(dev).contiguous()
L, R, SL, SR = st("left"), st("right"), st("seed_l"), st("seed_r")
params = pm.default_params(0, patch=11, patchmatch_iters=8)
engs = [pm.Engine(params, max_rows=rows, max_cols=cols, max_batch=nb) for _ in range(H)]
outs = [(torch.empty((nb, rows, cols), device=dev), torch.empty((nb, rows, cols), device=dev)) for _ in range(H)]
def step():
    for e, (dl, dr) in zip(engs, outs):
        e.match_device(nb, L.data_ptr(), R.data_ptr(), rows, cols, SL.data_ptr(), SR.data_ptr(), dl.data_ptr(), dr.data_ptr())
def sync():
    for e in engs: e.synchronize()
step(); sync()
t0 = time.perf_counter(); n = 3
for _ in range(n): step()
sync()
dt = (time.perf_counter() - t0) / n
print(json.dumps({"pairs_per_step": nb * H, "handles": H, "pairs_per_s": nb * H / dt, "ms_per_pair": 1e3 * dt / (nb * H),
                  "same_result": bool(torch.equal(outs[0][0], outs[-1][0]))}))
for e in engs: e.close()
