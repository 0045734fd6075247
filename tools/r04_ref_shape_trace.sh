#!/bin/bash
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/r04/ref_shape
mkdir -p $out
timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace -d $out/raw -o t --output-format csv -- python3 $root/tools/ref_shape_loop.py > $out/log.txt 2>&1
tail -1 $out/log.txt
kt=$(find $out/raw -name "*kernel_trace.csv" | head -1); mc=$(find $out/raw -name "*memory_copy_trace.csv" | head -1)
python3 - "$kt" "$mc" <<'P'
import csv, sys
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:], "q" + r["Queue_Id"]))
try:
    for r in csv.DictReader(open(sys.argv[2])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", r.get("Name", "")), ""))
except Exception as e:
    print("no copy trace", e)
ev.sort()
fins = [i for i, e in enumerate(ev) if "k_finalize" in e[2]]
a, b = fins[-3], fins[-2]
t0 = ev[a][1]
print("one call:", round(1e-3 * (ev[b][1] - t0), 1), "us between finalize ends;", b - a, "events")
for e in ev[a + 1:b + 3]:
    print(f"{1e-3*(e[0]-t0):8.1f} {1e-3*(e[1]-t0):8.1f} {1e-3*(e[1]-e[0]):6.1f}  {e[3]:4s} {e[2]}")
P
rm -rf $out/raw
