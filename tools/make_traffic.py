#!/usr/bin/env python3
"""profiles/traffic.json from two rocprofv3 counter_collection.csv files (FETCH_SIZE pass, WRITE_SIZE pass).

usage: tools/make_traffic.py <fetch.csv> <write.csv> <out.json>
Per-launch averages in KiB for the dominant kernels (the counters report KiB; summed over XCD instances)."""
import csv, collections, json, sys


def per_launch(path, counter):
    tot, disp = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        tot[k] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    return {k: tot[k] / len(disp[k]) for k in tot}


def pick(d, *subs):
    for k, v in d.items():
        if all(s in k for s in subs):
            return k, v
    return None, None


def main():
    fetch, write = per_launch(sys.argv[1], "FETCH_SIZE"), per_launch(sys.argv[2], "WRITE_SIZE")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 "
                     "--warmup 1 --no-cpu-baseline --host-pairs 0; per-launch averages, KiB; FETCH_SIZE is uncalibrated "
                     "for 1-4 B/lane loads (MI355X_MICROARCH.md HBM section)", "kernels": {}}
    for name, subs in (("sweep_row", ("k_runblk2<0, 32, 0,",)), ("sweep_col", ("k_runblk2<0, 32, 1,",)),
                       ("noise_cost", ("k_noise_cost_tiled",))):
        kf, f = pick(fetch, *subs)
        kw, w = pick(write, *subs)
        if kf:
            out["kernels"][name] = {"kernel": kf.split("(")[0].replace("void ", ""), "fetch_kib": f, "write_kib": w}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main()
