#!/usr/bin/env python3
"""Does the sweep engine's tuning hold on imagery it was not tuned on?

The group-width rule of pm_sweeps.hip (16- or 32-lane chain segments by noise amplitude and sweep direction) and the
wavefronts-per-chain choice were measured on bench.py's synthetic pairs.  This tool times Match() (scalar mode,
BASELINE configs[1] parameters, inputs resident, HIP events on the engine's stream) on a set of different scenes; the
knobs are read once per process from the environment, so tools/scene_ab.sh runs it once per setting:

    python tools/scene_ab.py [--steps 12]      -> one line per scene: ms per frame

Scenes (1280x720):
  bench0      the benchmark's own pair 0
  seed7       another synthetic scene
  shallow     d_max 32      deep      d_max 192
  sparse      20 seed points        dense     2000 seed points
  smooth      the synthetic texture low-passed (little fine detail: flat cost landscapes)
  caddy       the CADDY underwater pair of tests/golden (640x480, real imagery) tiled 2x2 and cropped, self-seeded
  farmsim     the reference's own test pair (376x240) tiled 4x3 and cropped, self-seeded
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ocean-perception_amd", "python"))
import numpy as np

ROWS, COLS = 720, 1280


def tile_to(img, rows, cols):
    ry, rx = -(-rows // img.shape[0]), -(-cols // img.shape[1])
    return np.ascontiguousarray(np.tile(img, (ry, rx))[:rows, :cols])


def scenes(synth):
    from scipy.ndimage import gaussian_filter
    out = []
    out.append(("bench0", synth.make_pair(0, ROWS, COLS), False))
    out.append(("seed7", synth.make_pair(7, ROWS, COLS), False))
    out.append(("shallow", synth.make_pair(11, ROWS, COLS, d_max=32.0), False))
    out.append(("deep", synth.make_pair(12, ROWS, COLS, d_max=192.0), False))
    out.append(("sparse", synth.make_pair(13, ROWS, COLS, n_points=20), False))
    out.append(("dense", synth.make_pair(14, ROWS, COLS, n_points=2000), False))
    p = synth.make_pair(15, ROWS, COLS)
    for k in ("left", "right"):
        p[k] = np.clip(np.rint(gaussian_filter(p[k].astype(np.float32), 2.5)), 0, 255).astype(np.uint8)
    out.append(("smooth", p, False))
    g = os.path.join(ROOT, "tests", "golden")
    z = np.load(os.path.join(g, "caddy_32_gray_640x480.npz"))
    out.append(("caddy", {"left": tile_to(z["left"], ROWS, COLS), "right": tile_to(z["right"], ROWS, COLS)}, True))
    z = np.load(os.path.join(g, "farmsim_fs1_376x240.npz"))
    out.append(("farmsim", {"left": tile_to(z["left"], ROWS, COLS), "right": tile_to(z["right"], ROWS, COLS)}, True))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--only", default="")
    ap.add_argument("--kernels", action="store_true", help="also print per-kernel-class ms of 4 profiled frames")
    args = ap.parse_args()
    import torch
    import pm_ctypes as pm
    import synth
    pm.load()
    dev = torch.device("cuda:0")
    total = 0.0
    for name, p, self_seed in scenes(synth):
        if args.only and name not in args.only.split(","):
            continue
        prm = pm.default_params(0, patch=11, patchmatch_iters=8, sparse_init=1 if self_seed else 0)
        with pm.Engine(prm, max_rows=ROWS, max_cols=COLS) as e:
            L = torch.from_numpy(p["left"]).to(dev)
            R = torch.from_numpy(p["right"]).to(dev)
            SL = None if self_seed else torch.from_numpy(p["seed_l"]).to(dev)
            SR = None if self_seed else torch.from_numpy(p["seed_r"]).to(dev)
            DL = torch.empty((ROWS, COLS), dtype=torch.float32, device=dev)
            DR = torch.empty_like(DL)
            stream = torch.cuda.ExternalStream(e.stream())

            def step():
                e.match_device(1, L.data_ptr(), R.data_ptr(), ROWS, COLS, SL.data_ptr() if SL is not None else None,
                               SR.data_ptr() if SR is not None else None, DL.data_ptr(), DR.data_ptr())

            for _ in range(3):
                step()
            e.synchronize()
            first = DL.clone()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
            ev[0].record(stream)
            for i in range(args.steps):
                step()
                ev[i + 1].record(stream)
            e.synchronize()
            ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps))
            med = ms[len(ms) // 2]
            det = bool(torch.equal(first, DL))
            valid = float((DL > 0).float().mean())
            total += med
            if args.kernels:
                e.profile_enable(True)
                for _ in range(4):
                    step()
                prof = e.profile_read()
                e.profile_enable(False)
                print("         " + "  ".join(f"{k} {v[1] / 4:.3f}" for k, v in prof.items() if v[0]))
            print(f"{name:8s} {med:7.3f} ms/frame (min {ms[0]:.3f})  valid {100 * valid:5.1f} %  deterministic {det}")
    print(f"sum of medians {total:.3f} ms")


if __name__ == "__main__":
    main()
