#!/bin/bash
# Kernel trace of the reference's own call pattern (376x240, tools/ref_shape_loop.py): the time line of ONE call.
# -> gpurun_out/$ROUND/ref_shape_timeline.txt
root=$GRAFT_REPO_ROOT
round=${ROUND:-r06}
out=$root/gpurun_out/$round
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace -d /tmp/refshape -o rs --output-format csv -- python3 $root/tools/ref_shape_loop.py > $out/ref_shape_trace.log 2>&1
f=$(find /tmp/refshape -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $out/ref_shape_timeline.txt <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pm::", "")[:60], r["Queue_Id"]) for r in rows)
fin = [i for i, e in enumerate(ev) if "k_finalize" in e[2]]
# the call of median length among the steady-state ones (launches from behind one k_finalize up to the next)
calls = sorted(((ev[fin[j + 1]][1] - ev[fin[j] + 1][0]), j) for j in range(5, len(fin) - 1))
j = calls[len(calls) // 2][1]
a, b = fin[j] + 1, fin[j + 1] + 1
t0 = ev[a][0]
print("launches of the median call of %d: %d; first start -> last end %.1f us; previous call's finalize end -> first start %.1f us" % (len(calls), b - a, (ev[b - 1][1] - t0) / 1e3, (t0 - ev[a - 1][1]) / 1e3))
qs = sorted(set(e[3] for e in ev[a:b]))
for e in ev[a:b]:
    print("%8.1f %8.1f  %6.1f us  q%-2d %s" % ((e[0] - t0) / 1e3, (e[1] - t0) / 1e3, (e[1] - e[0]) / 1e3, qs.index(e[3]), e[2]))
busy = sum(e[1] - e[0] for e in ev[a:b]) / 1e3
print("sum of kernel times %.1f us" % busy)
P
tail -3 $out/ref_shape_trace.log
cat $out/ref_shape_timeline.txt
