#!/usr/bin/env python3
"""profiles/traffic.json from two rocprofv3 counter_collection.csv files (FETCH_SIZE pass, WRITE_SIZE pass).

usage: tools/make_traffic.py <fetch.csv> <write.csv> <out.json>
Per-launch averages in KiB for the dominant kernels (the counters report KiB; summed over XCD instances)."""
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


def main():
    fetch, write = per_launch(sys.argv[1], "FETCH_SIZE"), per_launch(sys.argv[2], "WRITE_SIZE")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 "
                     "--warmup 1 --no-cpu-baseline --no-side-legs --host-pairs 0 [--mode planes]; per-launch averages, "
                     "KiB; FETCH_SIZE is uncalibrated for 1-4 B/lane loads (MI355X_MICROARCH.md HBM section)",
           "kernels": {}}
    if len(sys.argv) > 4:  # merge into an existing file (second mode)
        try:
            out["kernels"] = json.load(open(sys.argv[4]))["kernels"]
        except Exception:
            pass
    for cls, _ in CLASSES:
        if cls in fetch:
            out["kernels"][cls] = {"kernels": fetch[cls][1], "launches_sampled": fetch[cls][2],
                                   "fetch_kib": fetch[cls][0], "write_kib": write.get(cls, (None,))[0]}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main()
