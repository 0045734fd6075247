import sys, time, numpy as np
sys.path.insert(0, "ocean-perception_amd/python")
import pm_ctypes as pm, synth
rows, cols = 2160, 4096
params = pm.default_params(pm.PM_SEM_CPU, patch=11, patchmatch_iters=8)
pair = synth.make_pair(0, rows, cols, n_points=200 * (rows * cols) // (720 * 1280))
for bands in (8, 4, 2):
    with pm.TiledEngine(params, rows, cols, bands) as te:
        te.upload(pair["left"], pair["right"], pair["seed_l"], pair["seed_r"])
        te.run(2)
        for rounds in (1, 2, 3, 4, bands - 1):
            t0 = time.perf_counter(); info = te.run(rounds); t = time.perf_counter() - t0
            print(bands, "bands, rounds", rounds, "-> %.1f ms" % (1e3 * t), info, flush=True)
