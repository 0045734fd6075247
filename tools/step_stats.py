"""Run-step statistics of the sweep engine (the device counters of the run engine, opt-in through pm_debug_counters_enable; a build with -DPM_RUN3_STATS,
loaded through PM_LIB, additionally prints the per-workgroup round / tick statistics of pm_run3.hpp on stderr).

    PM_LIB=ocean-perception_amd/lib/ab_stats.so python tools/step_stats.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ocean-perception_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pm_ctypes as pm
import synth

pm.load()
rows, cols = 720, 1280
p = synth.make_pair(0, rows, cols)
for iters in (1, 2, 4, 8):
    prm = pm.default_params(0, patch=11, patchmatch_iters=iters)  # the benchmark window (the default is the 3x3 of the reference test)
    with pm.Engine(prm, max_rows=rows, max_cols=cols) as e:
        e.debug_counters_enable(True)
        e.match(p["left"], p["right"], p["seed_l"], p["seed_r"])
        d = e.debug_counters()
        c = list(d["row"].values()) + list(d["col"].values())
    ws, fix, wev, gs, gev, adv, pred = c[:7]
    print(f"iters {iters}: wave-steps {ws} (+{fix} fix-up), evaluating {wev} ({100*wev/max(ws,1):.1f} %); "
          f"group-steps {gs}, evaluating {gev} ({100*gev/max(gs,1):.1f} %); positions/group-step {adv/max(gs,1):.2f}; "
          f"ending in a reject at the first evaluated position {100*pred/max(gs,1):.1f} %")
