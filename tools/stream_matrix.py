"""Does a leg's throughput depend on which other handles the process holds?  

    python tools/stream_matrix.py --legs single,batch,pipe,sync,batch_u8,tiled [--alive] [--dummies N]

Every leg is one way of calling Match() on 1280x720 pairs (PM_SEM_CPU, 8 iterations, 11x11, both views + cross-check);
`tiled` is 4096x2160 in 8 bands on this one device.  Without --alive every leg creates its handle, is measured and
destroys it; with --alive ALL handles of the listed legs are created (and warmed) first and stay alive while each leg is
measured in turn -- the constellation of a vehicle process that keeps a single-pair handle, a submit / collect handle
and a tiled plan.  One JSON line: {leg: pairs/s}.  "Alone" = the same leg run as the only leg of a process.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ocean-perception_amd", "python"))
import numpy as np
import torch
import pm_ctypes as pm
import synth

ap = argparse.ArgumentParser()
ap.add_argument("--legs", default="single,batch,pipe,pipe_pinned,pipe_dev,sync,batch_u8,batch_u8_pinned,tiled")
ap.add_argument("--depth", type=int, default=4)
ap.add_argument("--self-seed", action="store_true")
ap.add_argument("--alive", action="store_true")
ap.add_argument("--dummies", type=int, default=0, help="foreign queue-owning streams created before the handles")
ap.add_argument("--reps", type=int, default=1)
args = ap.parse_args()
legs = args.legs.split(",")
pm.load()
dev = torch.device("cuda:0")
ROWS, COLS = 720, 1280
dummies = [torch.cuda.Stream() for _ in range(args.dummies)]
for sd in dummies:
    with torch.cuda.stream(sd):
        torch.zeros(1024, device=dev).add_(1)
torch.cuda.synchronize()
prm = pm.default_params(0, patch=11, patchmatch_iters=8, sparse_init=1 if args.self_seed else 0)
NB = 4
D = args.depth
prs = [synth.make_pair(i, ROWS, COLS) for i in range(NB)]
st = lambda k: torch.from_numpy(np.stack([p[k] for p in prs])).to(dev).contiguous()
L, R, SL, SR = st("left"), st("right"), st("seed_l"), st("seed_r")
DL = torch.empty((NB, ROWS, COLS), dtype=torch.float32, device=dev)
DR = torch.empty_like(DL)
outs = [(np.zeros((ROWS, COLS), np.float32), np.zeros((ROWS, COLS), np.float32)) for _ in range(8)]


class Leg:
    def __init__(self, name):
        self.name = name
        self.e = None

    def create(self):
        n = self.name
        if n == "single":
            self.e = pm.Engine(prm, max_rows=ROWS, max_cols=COLS)
        elif n == "replay":
            self.e = pm.Engine(prm, max_rows=ROWS, max_cols=COLS)
            a = (1, L.data_ptr(), R.data_ptr(), ROWS, COLS, None if args.self_seed else SL.data_ptr(),
                 None if args.self_seed else SR.data_ptr(), DL.data_ptr(), DR.data_ptr())
            self.e.match_device(*a)
            self.e.synchronize()
            self.e.capture_begin()
            self.e.match_device(*a)
            self.e.capture_end()
        elif n == "batch":
            self.e = pm.Engine(prm, max_rows=ROWS, max_cols=COLS, max_batch=NB)
        elif n in ("pipe", "pipe_pinned", "pipe_dev"):
            self.e = pm.Engine(prm, max_rows=ROWS, max_cols=COLS, max_batch=D)
            if n == "pipe_pinned":
                pin = lambda a: (lambda b: (np.copyto(b, a), b)[1])(self.e.host_alloc(a.shape, a.dtype))
                self.pin_in = [{k: pin(p[k]) for k in ("left", "right", "seed_l", "seed_r")} for p in prs]
                self.pin_out = [(self.e.host_alloc((ROWS, COLS), np.float32), self.e.host_alloc((ROWS, COLS), np.float32))
                                for _ in range(D)]
        elif n == "sync":
            self.e = pm.Engine(prm, max_rows=ROWS, max_cols=COLS)
        elif n in ("batch_u8", "batch_u8_pinned"):
            self.e = pm.Engine(prm, max_rows=ROWS, max_cols=COLS, max_batch=8)
            if n == "batch_u8_pinned":
                pin = lambda a: (lambda b: (np.copyto(b, a), b)[1])(self.e.host_alloc(a.shape, a.dtype))
                self.pin_in = [{k: pin(p[k]) for k in ("left", "right", "seed_l", "seed_r")} for p in prs]
                self.bout = ([self.e.host_alloc((ROWS, COLS), np.float32) for _ in range(8)],
                             [self.e.host_alloc((ROWS, COLS), np.float32) for _ in range(8)])
            else:
                self.bout = ([np.zeros((ROWS, COLS), np.float32) for _ in range(8)],
                             [np.zeros((ROWS, COLS), np.float32) for _ in range(8)])
        elif n == "tiled":
            self.big = synth.make_pair(3, 2160, 4096, n_points=1500)
            self.e = pm.TiledEngine(prm, 2160, 4096, 8)
            self.e.upload(self.big["left"], self.big["right"], self.big["seed_l"], self.big["seed_r"])
        else:
            raise SystemExit("unknown leg " + n)
        self.run(warm=True)

    def close(self):
        if self.e is not None:
            self.e.close()
            self.e = None

    def run(self, warm=False):
        """-> (pairs, seconds, extra)"""
        n, e = self.name, self.e
        extra = {}
        if n == "single":
            k = 3 if warm else 40
            a = (1, L.data_ptr(), R.data_ptr(), ROWS, COLS, None if args.self_seed else SL.data_ptr(),
                 None if args.self_seed else SR.data_ptr(), DL.data_ptr(), DR.data_ptr())
            e.synchronize()
            t0 = time.perf_counter()
            for _ in range(k):
                e.match_device(*a)
            t1 = time.perf_counter()
            e.synchronize()
            extra["enqueue_ms_per_pair"] = round(1e3 * (t1 - t0) / k, 3)
            return k, time.perf_counter() - t0, extra
        if n == "replay":
            k = 3 if warm else 40
            e.synchronize()
            t0 = time.perf_counter()
            for _ in range(k):
                e.replay()
            t1 = time.perf_counter()
            e.synchronize()
            extra["enqueue_ms_per_pair"] = round(1e3 * (t1 - t0) / k, 3)
            return k, time.perf_counter() - t0, extra
        if n == "batch":
            k = 2 if warm else 8
            a = (NB, L.data_ptr(), R.data_ptr(), ROWS, COLS, None if args.self_seed else SL.data_ptr(),
                 None if args.self_seed else SR.data_ptr(), DL.data_ptr(), DR.data_ptr())
            e.synchronize()
            t0 = time.perf_counter()
            for _ in range(k):
                e.match_device(*a)
            e.synchronize()
            return k * NB, time.perf_counter() - t0, extra
        if n in ("pipe", "pipe_pinned"):
            k = 2 * D if warm else 80
            ts = tc = 0.0
            j = 0
            src = self.pin_in if n == "pipe_pinned" else prs
            sd = (lambda p: (None, None)) if args.self_seed else (lambda p: (p["seed_l"], p["seed_r"]))
            t0 = time.perf_counter()
            for i in range(k):
                p = src[i % NB]
                if e.in_flight() == D:
                    a = time.perf_counter()
                    e.collect() if n == "pipe_pinned" else e.collect(out=outs[j & 7])
                    tc += time.perf_counter() - a
                    j += 1
                a = time.perf_counter()
                if n == "pipe_pinned":
                    e.submit(p["left"], p["right"], *sd(p), tag=i, out=self.pin_out[i % D])
                else:
                    e.submit(p["left"], p["right"], *sd(p), tag=i)
                ts += time.perf_counter() - a
            while e.in_flight():
                a = time.perf_counter()
                e.collect() if n == "pipe_pinned" else e.collect(out=outs[j & 7])
                tc += time.perf_counter() - a
                j += 1
            dt = time.perf_counter() - t0
            extra["submit_ms"] = round(1e3 * ts / k, 3)
            extra["collect_ms"] = round(1e3 * tc / k, 3)
            return k, dt, extra
        if n == "pipe_dev":
            k = 2 * D if warm else 80
            ssl = (lambda i: None) if args.self_seed else (lambda i: SL[i].data_ptr())
            ssr = (lambda i: None) if args.self_seed else (lambda i: SR[i].data_ptr())
            t0 = time.perf_counter()
            for i in range(k):
                if e.in_flight() == D:
                    e.collect_device()
                q = i % NB
                e.submit_device(L[q].data_ptr(), R[q].data_ptr(), ROWS, COLS, ssl(q), ssr(q), DL[q].data_ptr(),
                                DR[q].data_ptr(), tag=i)
            while e.in_flight():
                e.collect_device()
            return k, time.perf_counter() - t0, extra
        if n == "sync":
            k = 2 if warm else 20
            t0 = time.perf_counter()
            for i in range(k):
                p = prs[i % NB]
                e.match(p["left"], p["right"], None if args.self_seed else p["seed_l"],
                        None if args.self_seed else p["seed_r"], out=outs[i & 7])
            return k, time.perf_counter() - t0, extra
        if n in ("batch_u8", "batch_u8_pinned"):
            k = 1 if warm else 4
            src = self.pin_in if n == "batch_u8_pinned" else prs
            ls = [src[i % NB]["left"] for i in range(8)]
            rs = [src[i % NB]["right"] for i in range(8)]
            sl = None if args.self_seed else [src[i % NB]["seed_l"] for i in range(8)]
            sr = None if args.self_seed else [src[i % NB]["seed_r"] for i in range(8)]
            t0 = time.perf_counter()
            for _ in range(k):
                e.match_batch(ls, rs, sl, sr, out=self.bout)
            return k * 8, time.perf_counter() - t0, extra
        if n == "tiled":
            k = 1 if warm else 4
            for b in e.bands:
                b.synchronize()
            t0 = time.perf_counter()
            for _ in range(k):
                info = e.run(-1)
                for b in e.bands:
                    b.synchronize()
            dt = time.perf_counter() - t0
            extra["ms_per_frame"] = round(1e3 * dt / k, 2)
            extra["repeated"] = info["repeated"]
            return k, dt, extra


res = {"alive": bool(args.alive), "dummies": args.dummies, "depth": D, "self_seed": bool(args.self_seed)}
objs = [Leg(n) for n in legs]
if args.alive:
    for o in objs:
        o.create()
for o in objs:
    if not args.alive:
        o.create()
    best = None
    for _ in range(args.reps):
        k, dt, extra = o.run()
        v = k / dt
        if best is None or v > best[0]:
            best = (v, extra)
    res[o.name] = round(best[0], 1)
    for kk, vv in best[1].items():
        res[o.name + "." + kk] = vv
    if not args.alive:
        o.close()
for o in objs:
    o.close()
print(json.dumps(res))
