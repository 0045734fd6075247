#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on this box: builds tools/probe/fetch_calib, two separate --pmc passes (no tracing
# domains), prints counter KiB per kernel against the known byte count.  -> gpurun_out/$ROUND/fetch_calib.txt
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/${ROUND:-r06}
mkdir -p $out
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/fetch_calib $root/tools/probe/fetch_calib.hip || exit 1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c -d /tmp/calib_$c -o calib --output-format csv -- /tmp/fetch_calib > /tmp/calib_$c.log 2>&1
done
python3 - > $out/fetch_calib.txt <<'P'
import csv, glob, collections
MiB = 512.0
known = {"rd<2>": MiB, "rd<4>": MiB, "rd<8>": MiB, "rd<16>": MiB, "rd_rec16": MiB, "wr<4>": MiB, "wr<16>": MiB, "wr_strided4": MiB / 4}
print("FETCH_SIZE / WRITE_SIZE (KiB counters, summed over XCDs) against known traffic of tools/probe/fetch_calib.hip (512 MiB buffer, every byte once)")
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/calib_{c}/**/*counter_collection.csv", recursive=True)[0]
    per = collections.OrderedDict()  # (kernel, dispatch) -> value summed over XCD rows
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        per[(k, int(r["Dispatch_Id"]))] = per.get((k, int(r["Dispatch_Id"])), 0.0) + float(r["Counter_Value"])
    tot, cnt = collections.OrderedDict(), collections.Counter()
    for (k, d), v in per.items():  # per-dispatch average (rd<4> runs twice: the process's first dispatch is a warm-up)
        tot[k] = tot.get(k, 0.0) + v
        cnt[k] += 1
    for k in tot:
        tot[k] /= cnt[k]
    for k, v in tot.items():
        kb = known.get(k)
        if kb is None:
            continue
        relevant = (c == "FETCH_SIZE") == k.startswith("rd")
        print(f"  {c:10s} {k:12s} counter {v / 1024.0:9.1f} MiB   known {kb:7.1f} MiB   counter / known = {v / 1024.0 / kb:6.3f}" + ("" if relevant else "   (the other direction)"))
P
cat $out/fetch_calib.txt
