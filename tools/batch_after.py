"""Why does a batch run slower in a process that has used another handle before?  usage: batch_after.py [none|alive|closed]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ocean-perception_amd", "python"))
import numpy as np, torch
import pm_ctypes as pm, synth
pm.load()
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
ROWS, COLS, NB = 720, 1280, 4
dev = torch.device("cuda:0")
prs = [synth.make_pair(i, ROWS, COLS) for i in range(NB)]
st = lambda k: torch.from_numpy(np.stack([p[k] for p in prs])).to(dev).contiguous()
L, R, SL, SR = st("left"), st("right"), st("seed_l"), st("seed_r")
DL = torch.empty((NB, ROWS, COLS), dtype=torch.float32, device=dev); DR = torch.empty_like(DL)
prm = pm.default_params(0, patch=11, patchmatch_iters=8)
a = None
pad = torch.empty(int(os.environ.get("PAD_MB", "0")) << 20, dtype=torch.uint8, device=dev) if os.environ.get("PAD_MB") else None
if mode == "work":  # GPU work without a handle: a few large torch kernels
    x = torch.randn(4096, 4096, device=dev)
    for _ in range(20):
        x = x @ x * 1e-4
    torch.cuda.synchronize()
elif mode != "none":
    a = pm.Engine(prm, max_rows=ROWS, max_cols=COLS)
    for _ in range(10):
        a.match_device(1, L.data_ptr(), R.data_ptr(), ROWS, COLS, SL.data_ptr(), SR.data_ptr(), DL.data_ptr(), DR.data_ptr())
    a.synchronize()
    if mode == "closed":
        a.close(); a = None
dummies = [torch.cuda.Stream() for _ in range(int(os.environ.get("DUMMY", "0")))]
for sd in dummies:  # a stream gets its hardware queue with its first submission
    with torch.cuda.stream(sd):
        torch.zeros(1024, device=dev).add_(1)
torch.cuda.synchronize()
with pm.Engine(prm, max_rows=ROWS, max_cols=COLS, max_batch=NB) as e:
    run = lambda: e.match_device(NB, L.data_ptr(), R.data_ptr(), ROWS, COLS, SL.data_ptr(), SR.data_ptr(), DL.data_ptr(), DR.data_ptr())
    for _ in range(2): run()
    e.synchronize()
    t0 = time.perf_counter()
    for _ in range(8): run()
    e.synchronize()
    dt = time.perf_counter() - t0
print(f"dummy streams {len(dummies)}, first handle {mode}: batch of {NB}: {NB * 8 / dt:.1f} pairs/s")
