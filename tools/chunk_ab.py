"""Pairs per chunk of a batch (PM_PAIR_CHUNK, tuning build): 24 resident pairs per pm_match_device call, best of three passes.

    make tuning; for c in 1 2 3 4 6; do PM_LIB=$PWD/ocean-perception_amd/lib/libvehicle_pm_gpu_tuning.so PM_PAIR_CHUNK=$c python tools/chunk_ab.py; done

Round 6, one MI355X: 452 / 501 / 456 / 436 / 427 pairs/s for 1 / 2 / 3 / 4 / 6 -- two pairs per launch stays the optimum."""
import os, sys, time
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "ocean-perception_amd", "python"))
import numpy as np, torch
import pm_ctypes as pm, synth
pm.load()
rows, cols, nb = 720, 1280, 24
dev = torch.device("cuda:0")
prm = pm.default_params(0, patch=11, patchmatch_iters=8)
uniq = [synth.make_pair(i, rows, cols) for i in range(4)]
stack = lambda k: torch.from_numpy(np.stack([uniq[i % 4][k] for i in range(nb)])).to(dev).contiguous()
L, R, SL, SR = stack("left"), stack("right"), stack("seed_l"), stack("seed_r")
DL = torch.empty((nb, rows, cols), dtype=torch.float32, device=dev); DR = torch.empty_like(DL)
with pm.Engine(prm, max_rows=rows, max_cols=cols, max_batch=nb) as e:
    run = lambda: e.match_device(nb, L.data_ptr(), R.data_ptr(), rows, cols, SL.data_ptr(), SR.data_ptr(), DL.data_ptr(), DR.data_ptr())
    run(); e.synchronize()
    best = 0
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(3): run()
        e.synchronize()
        best = max(best, 3 * nb / (time.perf_counter() - t0))
print("PM_PAIR_CHUNK", os.environ.get("PM_PAIR_CHUNK", "default"), round(best, 1), "pairs/s")
