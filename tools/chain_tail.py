"""Where does a sweep launch spend its time?  Per iteration and sweep direction of the headline workload (1280x720, 8
iterations, 11x11, PM_SEM_CPU, both views side by side as in bench.py): launch time, and per CHAIN (= workgroup) its
duration and the run steps of its slowest wavefront -- min / median / p99 / max -- so that "the launch waits for a few
slow chains" can be read off (or refuted).

Needs the stats build of the library (a per-workgroup log in k_runblk3, device wall clock at 100 MHz):

    make tuning TUNE_DEFS=-DPM_RUN3_STATS && cp ocean-perception_amd/lib/libvehicle_pm_gpu_tuning.so \
        ocean-perception_amd/lib/libvehicle_pm_gpu_stats.so
    PM_LIB=ocean-perception_amd/lib/libvehicle_pm_gpu_stats.so python tools/chain_tail.py [--matches 4] [--out file]
"""
import argparse
import ctypes as C
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ocean-perception_amd", "python"))
import numpy as np
import torch

import pm_ctypes as pm
import synth

NAMES = {(0, 1): "row+", (1, 1): "col+", (0, -1): "row-", (1, -1): "col-"}


def read_log(path):
    with open(path, "rb") as f:
        assert f.read(8) == b"RUN3LOG1"
        (nl,) = struct.unpack("<q", f.read(8))
        out = []
        for _ in range(nl):
            stream, axis, dr, gs, n, chains, waves, _ = struct.unpack("<8q", f.read(64))
            w = np.frombuffer(f.read(32 * chains), np.uint32).reshape(chains, 8)
            out.append(dict(stream=stream, axis=axis, dir=dr, gs=gs, n=n, chains=chains, waves=waves, w=w))
    return out


def q(a, p):
    return float(np.percentile(a, p))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--matches", type=int, default=4)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--patch", type=int, default=11)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    lib = pm.load()
    if not hasattr(lib, "pm_run3_stats_enable"):
        sys.exit("this library is not the stats build (PM_LIB=.../libvehicle_pm_gpu_stats.so)")
    rows, cols = 720, 1280
    dev = torch.device("cuda:0")
    pairs = [synth.make_pair(i, rows, cols) for i in range(2)]
    t = lambda k, dt: [torch.from_numpy(p[k]).to(dev, dt).contiguous() for p in pairs]
    L, R, SL, SR = t("left", torch.uint8), t("right", torch.uint8), t("seed_l", torch.float32), t("seed_r", torch.float32)
    DL = torch.empty((rows, cols), dtype=torch.float32, device=dev)
    DR = torch.empty_like(DL)
    prm = pm.default_params(0, patch=args.patch, patchmatch_iters=args.iters)
    lines = []
    with pm.Engine(prm, max_rows=rows, max_cols=cols) as e:
        run = lambda i: e.match_device(1, L[i % 2].data_ptr(), R[i % 2].data_ptr(), rows, cols, SL[i % 2].data_ptr(),
                                       SR[i % 2].data_ptr(), DL.data_ptr(), DR.data_ptr())
        for i in range(3):
            run(i)
        e.synchronize()
        lib.pm_run3_stats_enable(1)
        for i in range(args.matches):
            run(i)
        e.synchronize()
        path = "/tmp/run3log.bin"
        n = lib.pm_run3_stats_dump(path.encode())
        lib.pm_run3_stats_enable(0)
    log = read_log(path)
    per_stream = {}
    for r in log:
        per_stream.setdefault(r["stream"], []).append(r)
    per_match = 4 * args.iters
    cells = {}  # (iteration, k) -> list of launches (both views, all matches)
    for s, rs in per_stream.items():
        assert len(rs) == per_match * args.matches, (len(rs), per_match, args.matches)
        for j, r in enumerate(rs):
            it, k = (j % per_match) // 4, j % 4
            assert NAMES[(r["axis"], r["dir"])] == ["row+", "col+", "row-", "col-"][k]
            cells.setdefault((it, k), []).append(r)
    lines.append(f"# k_runblk3 launches of {args.matches} headline Matches ({n} launches logged; both views run side by side; "
                 f"times in us from the device wall clock, 10 ns resolution)")
    lines.append("# launch = last chain's end - first chain's start; chain = one workgroup; steps = run steps (speculative "
                 "round + fix-up rounds) of the chain's slowest wavefront; tail = chains still running when 80 % of the "
                 "launch time has passed; idle = 1 - sum(chain time) / (chains x launch time)")
    lines.append("it sweep gs waves | launch us (mean of %d) | chain us min / med / p99 / max | steps min / med / p99 / max | fix-up steps med / max | rounds max | tail chains | idle"
                 % len(next(iter(cells.values()))))
    tot_launch = 0.0
    for (it, k), rs in sorted(cells.items()):
        launch, cdur, steps, fix, rounds, tail, idle = [], [], [], [], [], [], []
        for r in rs:
            w = r["w"].astype(np.int64)
            t0, t1 = w[:, 4], w[:, 5]
            base = t0.min()
            d0 = (t0 - base) & 0xffffffff
            d1 = (t1 - base) & 0xffffffff
            lt = d1.max() / 100.0
            launch.append(lt)
            cd = (d1 - d0) / 100.0
            cdur.append(cd)
            steps.append(w[:, 1] + w[:, 2])
            fix.append(w[:, 2])
            rounds.append((w[:, 3] & 0xff).max())
            tail.append(int((d1 / 100.0 > 0.8 * lt).sum()))
            idle.append(1.0 - cd.sum() / (len(cd) * lt))
        cd, st, fx = np.concatenate(cdur), np.concatenate(steps), np.concatenate(fix)
        tot_launch += float(np.mean(launch))
        lines.append(f"{it} {['row+', 'col+', 'row-', 'col-'][k]} {rs[0]['gs']:2d} {rs[0]['waves']} | {np.mean(launch):6.1f} | "
                     f"{cd.min():5.1f} / {q(cd, 50):5.1f} / {q(cd, 99):5.1f} / {cd.max():5.1f} | "
                     f"{st.min():3d} / {q(st, 50):5.1f} / {q(st, 99):5.1f} / {st.max():3d} | {q(fx, 50):4.1f} / {fx.max():3d} | "
                     f"{max(rounds)} | {np.mean(tail):6.1f} of {rs[0]['chains']} | {np.mean(idle):.2f}")
    lines.append(f"# sum of the mean launch times over one view's 32 sweeps: {tot_launch / 1e3:.3f} ms")
    # ---- balance INSIDE the chains: steps per wavefront (words 6 / 7 of the log: 8 bits per wavefront, chains of <= 4) ---------
    def per_wave(w, word, nwv):
        return np.stack([(w[:, word] >> (8 * i)) & 255 for i in range(nwv)], 1).astype(np.float64)
    lines.append("# inside the chains (wavefront i holds the i-th quarter of the chain in SWEEP order; steps = round 1 + fix-up rounds):")
    lines.append("it sweep | the 8 chains that end a launch: slowest wavefront, round-1 / fix-up steps | mean over their wavefronts | "
                 "steps by wavefront 0..3, these chains | steps by wavefront 0..3, all chains")
    crit_now = crit_bal = 0.0
    for (it, k), rs in sorted(cells.items()):
        acc, by_top, by_all = [], [], []
        for r in rs:
            w = r["w"].astype(np.int64)
            nwv = min(4, int(r["waves"]))
            s1, sf = per_wave(w, 6, nwv), per_wave(w, 7, nwv)
            tot = (w[:, 1] + w[:, 2]).astype(np.float64)
            top = np.argsort(tot)[-8:]
            acc.append([w[top, 1].mean(), w[top, 2].mean(), s1[top].mean(), sf[top].mean()])
            by_top.append((s1 + sf)[top].mean(0))
            by_all.append((s1 + sf).mean(0))
            crit_now += tot.max() / len(rs)
            crit_bal += (s1 + sf).mean(1).max() / len(rs)
        a = np.mean(acc, 0)
        fmt = lambda v: " ".join("%5.1f" % x for x in np.mean(v, 0))
        lines.append(f"{it} {['row+', 'col+', 'row-', 'col-'][k]} | {a[0]:5.1f} / {a[1]:5.1f} | {a[2]:5.1f} / {a[3]:5.1f} | "
                     f"{fmt(by_top)} | {fmt(by_all)}")
    lines.append(f"# sum over one view's 32 sweeps of the slowest chain's steps: {crit_now:.0f}; if every chain took the MEAN of its "
                 f"wavefronts (perfect balance inside a chain, nothing else changed): {crit_bal:.0f}")
    # ---- is a chain's step count predictable from the SAME sweep of the iteration before? ------------------------------
    lines.append("# prediction: chains whose steps in sweep (it - 1, k) exceeded F x that launch's mean are 'predicted slow' for (it, k)")
    lines.append("it sweep | corr(steps it-1, steps it) | F=1.25: predicted / chains, actual max steps, max steps of the NOT predicted chains | F=1.5: the same")
    bystream = {}
    for s_, rs in per_stream.items():
        for j, r in enumerate(rs):
            bystream[(s_, j // per_match, (j % per_match) // 4, j % 4)] = r
    keys = sorted(set((it, k) for (_, _, it, k) in bystream))
    what_if = {1.25: 0.0, 1.5: 0.0}
    base_sum = 0.0
    for (it, k) in keys:
        if it == 0:
            continue
        cor, info = [], {1.25: [], 1.5: []}
        for (s_, m, it2, k2), r in bystream.items():
            if (it2, k2) != (it, k):
                continue
            prev = bystream[(s_, m, it - 1, k)]
            a = (prev["w"][:, 1] + prev["w"][:, 2]).astype(np.float64)
            b = (r["w"][:, 1] + r["w"][:, 2]).astype(np.float64)
            # same chain order? the log is indexed by workgroup; map through the chain ids
            oa = np.argsort(prev["w"][:, 0]); ob = np.argsort(r["w"][:, 0])
            a, b = a[oa], b[ob]
            cor.append(np.corrcoef(a, b)[0, 1])
            for F in (1.25, 1.5):
                pred = a > F * a.mean()
                info[F].append((int(pred.sum()), b.max(), b[~pred].max() if (~pred).any() else 0.0,
                                max(b[~pred].max() if (~pred).any() else 0.0, 0.6 * b[pred].max() if pred.any() else 0.0)))
        row = f"{it} {['row+', 'col+', 'row-', 'col-'][k]} | {np.mean(cor):.2f}"
        for F in (1.25, 1.5):
            v = np.array(info[F], dtype=np.float64)
            row += f" | {v[:, 0].mean():6.1f} / {rs[0]['chains'] if False else ''}{v[:, 1].mean():5.1f}, {v[:, 2].mean():5.1f}"
            what_if[F] += v[:, 3].mean()
        base_sum += np.array(info[1.25], dtype=np.float64)[:, 1].mean()
        lines.append(row)
    lines.append(f"# sum over iterations 1.. of the launches' max steps: {base_sum:.0f}; if predicted chains took 0.6 x their steps: "
                 f"F=1.25 {what_if[1.25]:.0f}, F=1.5 {what_if[1.5]:.0f}")
    text = "\n".join(lines)
    print(text)
    if args.out:
        open(args.out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
