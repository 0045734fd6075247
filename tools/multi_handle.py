"""Aggregate Match() throughput of NH handles of one process, one host thread each, inputs resident (pm_match_device):
    python tools/multi_handle.py NH [start offset between the threads in ms]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ocean-perception_amd", "python"))
import numpy as np, torch
import pm_ctypes as pm, synth
pm.load()
ROWS, COLS = 720, 1280
dev = torch.device("cuda:0")
NH = int(sys.argv[1]) if len(sys.argv) > 1 else 2
OFFSET_MS = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
N = 40
# DUMMY=n: n streams created first (shifts the streams of the handles onto other hardware queues)
dummies = [torch.cuda.Stream() for _ in range(int(os.environ.get("DUMMY", "0")))]
for sd in dummies:  # a stream takes its hardware queue with its first submission
    with torch.cuda.stream(sd):
        torch.zeros(1024, device=dev).add_(1)
torch.cuda.synchronize()
# MODE=planes_bgr: plane mode with f16 state on BGR inputs (pm_match_bgr_device, BASELINE configs[4]) instead of the scalar headline
BGR = os.environ.get("MODE") == "planes_bgr"
PLANES = os.environ.get("MODE") == "planes"  # plane mode, f32 state, gray inputs
NB = int(os.environ.get("BATCH", "1"))       # pairs per call (the same pair NB times)
prm = pm.default_params(0, patch=11, patchmatch_iters=8)
if BGR:
    prm = pm.default_params(0, patch=11, patchmatch_iters=8, mode=pm.PM_MODE_PLANES, state_dtype=pm.PM_STATE_F16)
if PLANES:
    prm = pm.default_params(0, patch=11, patchmatch_iters=8, mode=pm.PM_MODE_PLANES)
engines = [pm.Engine(prm, max_rows=ROWS, max_cols=COLS, max_batch=NB) for _ in range(NH)]
# DUMMY_AFTER=n: n more streams that take their queues AFTER the handles exist but before their first Match
after = [torch.cuda.Stream() for _ in range(int(os.environ.get("DUMMY_AFTER", "0")))]
for sd in after:
    with torch.cuda.stream(sd):
        torch.zeros(1024, device=dev).add_(1)
torch.cuda.synchronize()
bufs = []
for i in range(NH):
    p = synth.make_pair(i, ROWS, COLS)
    rep = lambda a: np.ascontiguousarray(np.stack([a] * NB))
    t = [torch.from_numpy(rep(p[k])).to(dev) for k in ("left", "right", "seed_l", "seed_r")]
    t += [torch.empty((NB, ROWS, COLS), dtype=torch.float32, device=dev) for _ in range(2)]
    if BGR:
        t += [torch.from_numpy(rep(synth.to_bgr(p["left"], 1))).to(dev).contiguous(), torch.from_numpy(rep(synth.to_bgr(p["right"], 2))).to(dev).contiguous()]
    bufs.append(t)
def run(i, n):
    e, t = engines[i], bufs[i]
    if OFFSET_MS and n > 3:
        time.sleep(i * OFFSET_MS / 1000.0)
    for _ in range(n):
        if BGR:
            e.match_bgr_device(NB, t[6].data_ptr(), t[7].data_ptr(), ROWS, COLS, None, None, t[4].data_ptr(), t[5].data_ptr())
            continue
        e.match_device(NB, t[0].data_ptr(), t[1].data_ptr(), ROWS, COLS, t[2].data_ptr(), t[3].data_ptr(), t[4].data_ptr(), t[5].data_ptr())
    e.synchronize()
for i in range(NH): run(i, 3)
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(i, N)) for i in range(NH)]
[x.start() for x in th]; [x.join() for x in th]
dt = time.perf_counter() - t0
print(f"{NH} handles x {NB} pairs per call (start offset {OFFSET_MS} ms): {NH * N * NB / dt:.1f} pairs/s ({1000 * dt / (NH * N * NB):.3f} ms/frame)")
