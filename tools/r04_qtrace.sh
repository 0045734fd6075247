#!/bin/bash
# which hardware queue does each stream's work land on?  kernel trace of the alive matrix (queue id per dispatch)
mkdir -p gpurun_out/r04/qtrace
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04/qtrace -o alive -- python3 $GRAFT_REPO_ROOT/tools/stream_matrix.py --alive --legs single,batch,pipe_dev,pipe_pinned > $GRAFT_REPO_ROOT/gpurun_out/r04/qtrace/run.log 2>&1
cd $GRAFT_REPO_ROOT
tail -2 gpurun_out/r04/qtrace/run.log
f=$(ls gpurun_out/r04/qtrace/*kernel_trace.csv gpurun_out/r04/qtrace/*/*kernel_trace.csv 2>/dev/null | head -1)
echo $f; head -2 $f
python3 - "$f" <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), "dispatches; columns:", list(rows[0].keys()))
qk = "Queue_Id"
sk = "Stream_Id" if "Stream_Id" in rows[0] else None
c = collections.Counter((r.get(sk, "?") if sk else "?", r[qk]) for r in rows)
for (s, q), n in sorted(c.items()):
    print("stream", s, "queue", q, "dispatches", n)
P
# keep the merged-back files small
find gpurun_out/r04/qtrace -name "*.csv" -size +20M -delete
