#!/usr/bin/env python3
"""profiles/traffic.json from two rocprofv3 counter_collection.csv files (FETCH_SIZE pass, WRITE_SIZE pass).

usage: tools/make_traffic.py <fetch.csv> <write.csv> <out.json> [<merge-into.json>] [--suffix @variant] [--head SHA]
Per-launch averages in KiB for the dominant kernels (the counters report KiB; summed over XCD instances), and the
CORRECTED byte counts: /opt/skills/guides/MI355X_MICROARCH.md (HBM section) -- on gfx950 FETCH_SIZE reports exactly 1/2 of
the bytes read, confirmed on this engine's own access widths by tools/probe/fetch_calib.hip (profiles/r06_calib_fetch_write.txt:
4, 8 and 16 bytes per lane streaming and 16-byte records in 16-lane runs all read 0.500); WRITE_SIZE is exact for
whole-line stores and counts whole 32-byte sectors for sparse ones (every fourth dword of a stretch: 4.0 x the bytes
stored), so it is an upper bound of the bytes written where a kernel stores sparsely (the sweeps' write-back)."""
import csv, collections, json, re, sys


CLASSES = (("sweep_row", r"k_runblk3<\d+, 0,"), ("sweep_col", r"k_runblk3<\d+, 1,"),
           ("noise_cost", r"k_noise_cost_tiled"), ("planes_init", r"k_planes<\d+, 0,"),
           ("planes_spatial", r"k_planes<\d+, 1,"), ("planes_view", r"k_planes<\d+, 2,"),
           ("planes_refine", r"k_planes<\d+, 3,"), ("planes_view_refine", r"k_planes<\d+, 4,"))


def per_launch(path, counter):
    """class -> (mean counter value per launch over all kernels of the class, kernel names, launches)"""
    tot, disp, names = collections.defaultdict(float), collections.defaultdict(set), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for cls, pat in CLASSES:
            if re.search(pat, r["Kernel_Name"]):
                tot[cls] += float(r["Counter_Value"])
                disp[cls].add(r["Dispatch_Id"])
                names[cls].add(r["Kernel_Name"].split("(")[0].replace("void ", ""))
    return {c: (tot[c] / len(disp[c]), sorted(names[c]), len(disp[c])) for c in tot}


FETCH_CORRECTION = 2.0  # gfx950: FETCH_SIZE = TCC_EA0_RDREQ x 64 B with 128-byte requests tallied at 64 (guide + own calibration)


def main():
    argv = list(sys.argv[1:])
    suffix = head = ""
    for flag in ("--suffix", "--head"):
        if flag in argv:
            i = argv.index(flag)
            val = argv[i + 1]
            del argv[i:i + 2]
            if flag == "--suffix":
                suffix = val
            else:
                head = val
    fetch, write = per_launch(argv[0], "FETCH_SIZE"), per_launch(argv[1], "WRITE_SIZE")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no tracing domains) -- python3 "
                     "bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side-legs --host-pairs 0 --no-profile [--mode "
                     "planes [--plane-neighbours 1]]; per-launch averages",
           "correction": "bytes = fetch_kib * 1024 * 2 + write_kib * 1024: gfx950's FETCH_SIZE reports exactly half of the "
                         "bytes read (MI355X_MICROARCH.md, HBM section; confirmed for 4 / 8 / 16 B per lane and for 16-byte "
                         "record gathers by tools/probe/fetch_calib.hip, profiles/r06_calib_fetch_write.txt); WRITE_SIZE is exact "
                         "for whole-line stores and counts whole 32-byte sectors for sparse ones (an upper bound there). "
                         "Both count the L2's memory-side requests: Infinity-Cache hits included.",
           "kernels": {}}
    if len(argv) > 3:  # merge into an existing file (second mode)
        try:
            old = json.load(open(argv[3]))
            out["kernels"] = old["kernels"]
            if "head" in old and not head:
                head = old["head"]
        except Exception:
            pass
    if head:
        out["head"] = head
    for cls, _ in CLASSES:
        if cls in fetch:
            f_kib, w_kib = fetch[cls][0], write.get(cls, (0.0,))[0] or 0.0
            out["kernels"][cls + suffix] = {"kernels": fetch[cls][1], "launches_sampled": fetch[cls][2],
                                            "fetch_kib": f_kib, "write_kib": w_kib,
                                            "bytes_read": f_kib * 1024.0 * FETCH_CORRECTION, "bytes_written_upper": w_kib * 1024.0,
                                            "bytes": f_kib * 1024.0 * FETCH_CORRECTION + w_kib * 1024.0}
    json.dump(out, open(argv[2], "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main()
