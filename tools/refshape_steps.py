"""Run-step counters of the sweeps of the reference's own call (376x240, PM_SEM_GPU, 3 iterations, self-seeded): what a
chain's launch consists of -- speculative steps, fix-up steps, rounds -- per wavefront and per chain, averaged over the
launches of one Match (pm_debug_counters)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ocean-perception_amd", "python"))
import numpy as np
import pm_ctypes as pm

pm.load()
iters = 3
if os.environ.get("SHAPE") == "720p":  # PM_SEM_GPU on the headline's synthetic pair, given seeds
    import synth
    p = synth.make_pair(0, 720, 1280)
    l, r = p["left"], p["right"]
    rows, cols = l.shape
    prm = pm.default_params(pm.PM_SEM_GPU, patchmatch_iters=iters)
    args = (l, r, p["seed_l"], p["seed_r"])
else:
    g = np.load(os.path.join(ROOT, "tests", "golden", "farmsim_fs1_376x240.npz"))
    l, r = np.ascontiguousarray(g["left"]), np.ascontiguousarray(g["right"])
    rows, cols = l.shape
    prm = pm.default_params(pm.PM_SEM_GPU, cost_alpha=0.9, patchmatch_iters=iters, sparse_init=1)
    args = (l, r)
with pm.Engine(prm, max_rows=rows, max_cols=cols) as e:
    e.debug_counters_enable(True)
    e.match(*args)
    d = e.debug_counters()
waves = int(os.environ.get("PM_RUNBLK_WAVES", "4"))
if os.environ.get("PHASES"):  # a -DPM_RUN2_PHASES=1 (sums) / =2 (maxima over all launches) build: device wall clock per phase, 10 ns ticks
    for ax, chains in (("row", rows), ("col", cols)):
        c = list(d[ax].values())
        div = 2 * iters * 2 * (chains - 2) if os.environ["PHASES"] == "1" else 1
        print(ax, "sweeps, us per chain:", " ".join(f"{n} {v / div / 100:.2f}" for n, v in zip(("load", "round 1", "fix-up", "write-back"), c)),
              "(mean over chains and launches)" if div > 1 else "(maximum over chains and launches)")
    sys.exit(0)
for ax, chains, n in (("row", rows, cols), ("col", cols, rows)):
    c = d[ax]
    launches = 2 * iters * 2  # two directions, two views
    wl = launches * chains * waves  # wavefront-launches (interior chains are a few fewer)
    print(f"{ax} sweeps: chains of {n} positions; per wavefront and launch {c['steps_round1'] / wl:.2f} speculative + "
          f"{c['steps_fixup'] / wl:.2f} fix-up steps; fix-up rounds per chain {c['fixup_rounds'] / (launches * chains):.2f}; "
          f"positions per launch {c['positions'] / launches:.0f}")
