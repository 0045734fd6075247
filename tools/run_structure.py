"""How many run steps does a chain need -- under the step rule as shipped (one candidate per step: the step ends at the
first position that does not end up holding it) and under a TWO-candidate rule (every position also tests its
predecessor's OLD value, so a step that has met its first stop continues through positions that keep their own value and
ends at the first one that adopts its predecessor's)?  Taken from the disparity planes before and after every sweep
of a Match driven stage by stage (pm_tile_*, the whole image as one band): position p passes iff new[p] == new[p - 1].

    python tools/run_structure.py [--shape ref|720p] [--semantics 1] [--iters 3] [--patch 3] [--out file]
"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ocean-perception_amd", "python"))
import numpy as np
import torch
import pm_ctypes as pm
import synth


def steps_of(pass_, keep, lo, hi, cap, two):
    """steps a group needs for positions lo..hi-1 of one chain (sweep order).  pass_[p]: p ends up holding its
    predecessor's new value; keep[p]: p keeps its old value."""
    i, n = lo, 0
    while i < hi:
        end = min(hi, i + cap)
        j = i
        while j < end and pass_[j]:
            j += 1
        if j < end:
            stopped_keeping = keep[j]
            j += 1  # the stop position itself is decided in this step
            if two and stopped_keeping:
                while j < end and keep[j] and not pass_[j]:
                    j += 1
                if j < end and not keep[j]:
                    j += 1  # the first position that adopts its predecessor's old value ends the step
        i = j
        n += 1
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="ref")
    ap.add_argument("--semantics", type=int, default=1)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--patch", type=int, default=3)
    ap.add_argument("--group", type=int, default=16)
    ap.add_argument("--segments", type=int, default=16)
    ap.add_argument("--chains", type=int, default=48, help="chains sampled per sweep")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    pm.load()
    dev = torch.device("cuda:0")
    if a.shape == "ref":
        g = np.load(os.path.join(ROOT, "tests", "golden", "farmsim_fs1_376x240.npz"))
        l, r = np.ascontiguousarray(g["left"]), np.ascontiguousarray(g["right"])
        rows, cols = l.shape
        prm = pm.default_params(a.semantics, cost_alpha=0.9, patchmatch_iters=a.iters)
        with pm.Engine(prm, max_rows=rows, max_cols=cols) as e:  # the seeds Match() would make for itself (view 1: on the mirrored pair)
            sl = e.sparse_init(l, r, prm.init_dilate_factor)
            sr = np.ascontiguousarray(e.sparse_init(np.ascontiguousarray(r[:, ::-1]), np.ascontiguousarray(l[:, ::-1]), prm.init_dilate_factor)[:, ::-1])
    else:
        rows, cols = 720, 1280
        p = synth.make_pair(0, rows, cols)
        l, r, sl, sr = p["left"], p["right"], p["seed_l"], p["seed_r"]
        prm = pm.default_params(a.semantics, patch=a.patch, patchmatch_iters=a.iters)
    cap = a.group - 2 if a.semantics == 1 else a.group - (a.patch - 1) - 1
    t = lambda x, dt: torch.from_numpy(np.ascontiguousarray(x)).to(dev, dt).contiguous()
    L, R, SL, SR = t(l, torch.uint8), t(r, torch.uint8), t(sl, torch.float32), t(sr, torch.float32)
    lines = [f"# {cols}x{rows}, semantics {a.semantics}, {a.iters} iterations; {a.segments} segments per chain, {cap} positions per step; "
             f"{a.chains} chains sampled per sweep and view",
             "it sweep | steps per segment, one candidate: mean / slowest segment of a chain (mean over chains) / slowest chain | "
             "two candidates: the same | ratio of the chain means"]
    with pm.Engine(prm, max_rows=rows, max_cols=cols) as e:
        e.tile_begin(pm.PmTile(rows, 0, 0, rows), L.data_ptr(), R.data_ptr(), rows, cols, SL.data_ptr(), SR.data_ptr())

        def planes():
            out = torch.empty((rows, 2, cols), dtype=torch.float32, device=dev)
            for y in range(rows):
                e.tile_get_row(y, out[y].data_ptr())
            e.synchronize()
            return out.cpu().numpy()

        tot = [0.0, 0.0]
        for it in range(a.iters):
            e.tile_noise(it)
            for k in range(4):
                before = planes()
                e.tile_sweep(it, k)
                after = planes()
                axis, d = k % 2, 1 if k < 2 else -1
                res = [[], []]
                worst = [[], []]
                for view in range(2):
                    b, n = before[:, view, :], after[:, view, :]
                    if axis == 1:
                        b, n = b.T, n.T
                    if d < 0:
                        b, n = b[:, ::-1], n[:, ::-1]
                    nch, ln = b.shape
                    pick = np.linspace(2, nch - 3, a.chains).astype(int)
                    seg = -(-ln // a.segments)
                    for c in pick:
                        pass_ = np.zeros(ln, bool)
                        pass_[1:] = n[c, 1:] == n[c, :-1]
                        keep = n[c] == b[c]
                        for two in (0, 1):
                            st = [steps_of(pass_, keep, s0, min(ln, s0 + seg), cap, two) for s0 in range(1, ln, seg)]
                            res[two].append(np.mean(st))
                            worst[two].append(np.max(st))
                m = [np.mean(worst[0]), np.mean(worst[1])]
                tot[0] += m[0]
                tot[1] += m[1]
                lines.append(f"{it} {['row+', 'col+', 'row-', 'col-'][k]} | {np.mean(res[0]):5.2f} / {m[0]:5.2f} / {np.max(worst[0]):3d} | "
                             f"{np.mean(res[1]):5.2f} / {m[1]:5.2f} / {np.max(worst[1]):3d} | {m[1] / m[0]:.2f}")
        lines.append(f"# sum over the sweeps of 'slowest segment of a chain': one candidate {tot[0]:.1f}, two candidates {tot[1]:.1f} "
                     f"({tot[1] / tot[0]:.2f})")
    text = "\n".join(lines)
    print(text)
    if a.out:
        open(a.out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
