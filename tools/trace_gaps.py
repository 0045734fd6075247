#!/usr/bin/env python3
"""Frame-boundary analysis of a rocprofv3 kernel trace of bench.py (tools/trace_frames.sh).

    python tools/trace_gaps.py gpurun_out/<dir>/kernel_trace.csv

Per frame (one k_finalize each): the time line of its launches per stream, the intervals in which fewer than two sweep /
noise kernels are running, and the head (previous k_finalize end -> first noise + cost start) and tail (last sweep
end -> k_finalize end)."""
import csv
import sys
from collections import defaultdict


def short(name):
    for k in ("k_runblk3", "k_noise_cost_tiled", "k_background_tiled", "k_finalize", "k_setup", "k_prep", "k_seed"):
        if k in name:
            if k == "k_runblk3":
                args = name.split("<")[1].split(">")[0].replace(" ", "").split(",")
                return ("row" if args[1] == "0" else "col") + ("+" if args[3] == "1" else "-") + args[0]
            return k[2:]
    return name[:24]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"], r["Stream_Id"])
          for r in rows if "fillBuffer" not in r["Kernel_Name"] and "copyBuffer" not in r["Kernel_Name"]]
    ev.sort()
    fins = [e for e in ev if e[2] == "finalize"]
    print(f"{len(ev)} launches, {len(fins)} frames")
    for fi in range(max(1, len(fins) - 3), len(fins)):
        t0, t1 = fins[fi - 1][1], fins[fi][1]
        fr = [e for e in ev if e[0] >= t0 - 1000 and e[1] <= t1]
        heavy = [e for e in fr if e[2][:3] in ("row", "col", "noi")]
        first_noise = min(e[0] for e in fr if e[2].startswith("noise"))
        last_sweep = max(e[1] for e in heavy)
        # busy-count timeline of heavy kernels
        pts = sorted([(e[0], 1) for e in heavy] + [(e[1], -1) for e in heavy])
        cur, last, under = 0, t0, defaultdict(int)
        for t, d in pts:
            under[cur] += t - last
            cur += d
            last = t
        under[0] += t1 - last
        print(f"frame {fi}: {1e-3 * (t1 - t0):.1f} us; head (finalize -> first noise_cost) {1e-3 * (first_noise - t0):.1f} us; "
              f"tail (last sweep -> finalize end) {1e-3 * (t1 - last_sweep):.1f} us; heavy kernels running: "
              + ", ".join(f"{k}: {1e-3 * v:.0f} us" for k, v in sorted(under.items())))
        by_stream = defaultdict(list)
        for e in fr:
            by_stream[e[4]].append(e)
        for sid, es in sorted(by_stream.items()):
            gaps = [es[i + 1][0] - es[i][1] for i in range(len(es) - 1)]
            busy = sum(e[1] - e[0] for e in es)
            print(f"   stream {sid} (queue {es[0][3]}): {len(es)} launches, busy {1e-3 * busy:.0f} us, gaps between consecutive "
                  f"launches: sum {1e-3 * sum(gaps):.0f} us, median {1e-3 * sorted(gaps)[len(gaps) // 2]:.1f} us, max {1e-3 * max(gaps):.1f} us")
        if fi == len(fins) - 1:
            for sid, es in sorted(by_stream.items()):
                print(f"   stream {sid}: " + " ".join(f"{e[2]}[{1e-3 * (e[0] - t0):.0f}-{1e-3 * (e[1] - t0):.0f}]" for e in es[:9]) + " ... "
                      + " ".join(f"{e[2]}[{1e-3 * (e[0] - t0):.0f}-{1e-3 * (e[1] - t0):.0f}]" for e in es[-4:]))


if __name__ == "__main__":
    main()
