#!/usr/bin/env python3
"""Time of pm_stereo_ready (8-bit BGR -> stereo-ready 8-bit gray, include/pm/imaging.h) per image. One JSON line.

Algorithmic bytes per pixel: 3 (bgr8 in) + 1 (gray8 out); the intermediate float planes (row pass 12 B, quotient 12 B,
each written once and read once) add 48 B/px of scratch traffic."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ocean-perception_amd", "python"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=720)
    ap.add_argument("--cols", type=int, default=1280)
    ap.add_argument("--steps", type=int, default=30)
    args = ap.parse_args()
    import torch
    import pm_ctypes as pm
    rows, cols = args.rows, args.cols
    g = torch.Generator(device="cuda").manual_seed(1)
    bgr = (torch.rand((rows, cols, 3), device="cuda", generator=g) * 255).to(torch.uint8)
    gray = torch.empty((rows, cols), dtype=torch.uint8, device="cuda")
    with pm.Engine(pm.default_params(0, patch=5), max_rows=64, max_cols=64) as e:
        e.profile_enable(False)
        run = lambda: e.stereo_ready(bgr.data_ptr(), rows, cols, None, gray.data_ptr())
        for _ in range(3):
            run()
        e.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run()
        e.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
    third = cols // 3
    print(json.dumps({"op": "pm_stereo_ready", "rows": rows, "cols": cols, "gaussian_taps": third + (1 - third % 2),
                      "ms": dt * 1e3, "mpix_per_s": rows * cols / dt / 1e6}))


if __name__ == "__main__":
    main()
