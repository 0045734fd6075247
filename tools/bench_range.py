#!/usr/bin/env python3
"""HBM roofline of the fused disparity -> range -> RemoveBackscatter -> CorrectAttenuation pass
(pm_range_enhance, include/pm/imaging.h) on one GPU.  Prints one JSON line.

Algorithmic bytes per pixel: pass 1 reads the disparity (4 B); pass 2 reads disparity (4) + BGR float (12) and
writes the corrected BGR (12) and the range map (4): 36 B/px."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ocean-perception_amd", "python"))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2160)
    ap.add_argument("--cols", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=50)
    args = ap.parse_args()
    import torch
    import pm_ctypes as pm
    rows, cols = args.rows, args.cols
    g = torch.Generator(device="cuda").manual_seed(1)
    disp = torch.rand((rows, cols), device="cuda", generator=g) * 94 + 2
    disp[torch.rand((rows, cols), device="cuda", generator=g) < 0.2] = 0
    bgr = torch.rand((rows, cols, 3), device="cuda", generator=g)
    out, rng = torch.empty_like(bgr), torch.empty_like(disp)
    B, bB = (0.132, 0.115, 0.0559), (0.358, 0.695, 1.11)
    X = (0.30, 0.25, 0.40, -0.20, -0.15, -0.30, 0.10, 0.12, 0.08, -0.05, -0.04, -0.06)
    with pm.Engine(pm.default_params(0, patch=5), max_rows=64, max_cols=64) as e:
        run = lambda: e.range_enhance(bgr.data_ptr(), disp.data_ptr(), rows, cols, 400.0, 0.1, B, bB, X, rng.data_ptr(),
                                      out.data_ptr())
        for _ in range(5):
            run()
        e.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run()
        e.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
    nbytes = rows * cols * 36
    print(json.dumps({"kernel": "pm_range_enhance (k_disp_min_positive + k_range_enhance<7>)", "rows": rows, "cols": cols,
                      "ms": dt * 1e3, "algorithmic_bytes": nbytes, "achieved_GBps": nbytes / dt / 1e9,
                      "peak_GBps": 8000.0, "frac": nbytes / dt / 1e9 / 8000.0,
                      "mpix_per_s": rows * cols / dt / 1e6}))


if __name__ == "__main__":
    main()
