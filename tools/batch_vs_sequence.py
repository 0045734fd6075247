"""Does pm_match_device(n = 32) lose against 32 pm_submit_device frames, or do bench.py's two legs match different pairs?
Both entry points on the SAME pairs (content A: synthetic pairs 0..3 repeated, the headline's rotation; content B: pairs
0..15 as bench.py's batch leg drew them until round 5), same handle parameters, one process."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ocean-perception_amd", "python"))
import numpy as np
import torch
import pm_ctypes as pm
import synth

rows, cols, nb = 720, 1280, 32
dev = torch.device("cuda:0")
prm = pm.default_params(0, patch=11, patchmatch_iters=8)
uniq = [synth.make_pair(i, rows, cols) for i in range(16)]
stack = lambda idx, k: torch.from_numpy(np.stack([uniq[i][k] for i in idx])).to(dev).contiguous()
for name, idx in (("pairs 0..3 x 8", [i % 4 for i in range(nb)]), ("pairs 0..15 x 2", [i % 16 for i in range(nb)]),
                  ("pairs 4..7 x 8", [4 + i % 4 for i in range(nb)]), ("pairs 12..15 x 8", [12 + i % 4 for i in range(nb)])):
    L, R, SL, SR = stack(idx, "left"), stack(idx, "right"), stack(idx, "seed_l"), stack(idx, "seed_r")
    DL = torch.empty((nb, rows, cols), dtype=torch.float32, device=dev)
    DR = torch.empty_like(DL)
    res = {}
    with pm.Engine(prm, max_rows=rows, max_cols=cols, max_batch=nb) as e:
        run = lambda: e.match_device(nb, L.data_ptr(), R.data_ptr(), rows, cols, SL.data_ptr(), SR.data_ptr(), DL.data_ptr(), DR.data_ptr())
        run(); e.synchronize()
        best = 0
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(3):
                run()
            e.synchronize()
            best = max(best, 3 * nb / (time.perf_counter() - t0))
        res["batch32"] = best
        ref = DL.clone()
    for depth in (4,):
        with pm.Engine(prm, max_rows=rows, max_cols=cols, max_batch=depth) as e:
            def seq(k):
                for i in range(k):
                    if e.in_flight() == depth:
                        e.collect_device()
                    q = i % nb
                    e.submit_device(L[q].data_ptr(), R[q].data_ptr(), rows, cols, SL[q].data_ptr(), SR[q].data_ptr(),
                                    DL[q].data_ptr(), DR[q].data_ptr(), tag=i)
                while e.in_flight():
                    e.collect_device()
            seq(8)
            best = 0
            for _ in range(3):
                t0 = time.perf_counter()
                seq(96)
                best = max(best, 96 / (time.perf_counter() - t0))
            res[f"sequence_depth{depth}"] = best
            assert torch.equal(DL, ref)
    print(name, {k: round(v, 1) for k, v in res.items()}, flush=True)
