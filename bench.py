#!/usr/bin/env python3
"""Benchmark of the hot path: PatchMatch stereo Match() on MI355X.

Workload (BASELINE.json configs[1]): one 1280x720 synthetic stereo pair per GPU, 8 iterations,
11x11 window, fp32 cost, reference-CPU semantics (PM_SEM_CPU), left + right view + cross-check.
A "step" is one Match() through the C ABI entry point pm_match_device with the u8 pair and the seed
maps already resident in HBM and the disparity maps written to HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank r matches pair r (stereo pairs are independent: no data-path collective, weak scaling).  The K
timed steps are bracketed by barrier + synchronize on both sides; rank 0 prints ONE JSON line with
the whole-job pairs/s (max elapsed over ranks), the roofline of the dominant kernel (per-launch
duration from HIP events recorded by the engine on its own stream during the timed steps) and the
CPU baseline (the oracle -- a port of the reference CPU path -- timed on this host, rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ocean-perception_amd", "python"))

ROWS, COLS, ITERS, PATCH = 720, 1280, 8, 11
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured streaming
# Algorithmic bytes of one directional sweep: per pixel and view the sweep reads the four
# f32-equivalent image planes once (16 B), reads the disparity (4 B) and writes it (4 B).
SWEEP_BYTES_PER_PX = 24


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL) on GPUs; gloo with --dry-run for CPU tests")
    ap.add_argument("--dry-run", action="store_true",
                    help="exercise sharding/barrier/reduction without a GPU (no compute, no engine)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rows", type=int, default=ROWS)
    ap.add_argument("--cols", type=int, default=COLS)
    ap.add_argument("--iters", type=int, default=ITERS)
    ap.add_argument("--patch", type=int, default=PATCH)
    ap.add_argument("--engine", type=int, default=0)
    ap.add_argument("--pairs-per-gpu", type=int, default=1,
                    help="pairs matched per step and GPU (1 = BASELINE configs[1]; 32 = configs[2]'s per-GPU share)")
    ap.add_argument("--self-seed", action="store_true",
                    help="let Match() compute its seeds with the device SparseInit (side measurement)")
    ap.add_argument("--profile-every", type=int, default=4,
                    help="per-kernel HIP events are recorded on every n-th timed step (the ~180 event records of a "
                         "fully timed step cost 6 %% of the step; 1 = every step)")
    ap.add_argument("--no-profile", action="store_true",
                    help="experiment: no per-kernel HIP events in the timed region (the roofline object is then empty)")
    ap.add_argument("--host-pairs", type=int, default=24,
                    help="pairs of the untimed host-buffer leg (PCIe-inclusive rates, reported beside `value`); 0 = skip")
    ap.add_argument("--semantics", type=int, default=0,
                    help="0 = PM_SEM_CPU (the benchmark configuration), 1 = PM_SEM_GPU (side measurement)")
    return ap.parse_args()


class Dist:
    """One process per GPU; torch.distributed only for the barrier and the max over ranks."""

    def __init__(self, args):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            if args.backend == "nccl":
                import torch
                torch.cuda.set_device(self.local_rank)
            dist.init_process_group(backend=args.backend, rank=self.rank, world_size=self.world)
            self.dist = dist
        self.device = f"cuda:{self.local_rank}" if args.backend == "nccl" and not args.dry_run else "cpu"

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def max_over_ranks(self, value):
        if not self.dist:
            return value
        import torch
        t = torch.tensor([value], dtype=torch.float64, device=self.device if self.device != "cpu" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.dist:
            self.dist.destroy_process_group()


def pmc_traffic(kernel_class):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/traffic.json, written by tools in DESIGN.md 7: separate --pmc FETCH_SIZE / WRITE_SIZE runs of
    this same command; FETCH_SIZE/WRITE_SIZE are KiB counters of the L2's memory-side requests).  None if the
    file is missing."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        k = t["kernels"][kernel_class]
        return float(k["fetch_kib"] + k["write_kib"]) * 1024.0
    except Exception:
        return None


def shard(rank, world, steps):
    """Pair index matched by `rank` at each step: independent pairs, contiguous by rank."""
    return [rank for _ in range(steps)]


def cpu_baseline(args):
    """The oracle (port of src/vehicle/stereo_matching/patchmatch.cpp + the test recipe, literal call
    structure, single thread like the reference) on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    import synth
    p = synth.make_pair(0, args.rows, args.cols)
    band_rows = min(160, args.rows)  # ~10-15 s of single-thread CPU work on the GPU node's host
    y0 = (args.rows - band_rows) // 2
    band = slice(y0, y0 + band_rows)
    prm = O.default_params(O.SEM_CPU, patch=args.patch, n_iters=args.iters, nthreads=1, literal=1, left_right_check=1)
    t0 = time.perf_counter()
    O.match(prm, p["left"][band], p["right"][band], p["seed_l"][band], p["seed_r"][band])
    t = time.perf_counter() - t0
    h = args.patch // 2
    scale = (args.rows - 2 * h) / float(band_rows - 2 * h)  # swept rows of the full image / of the band
    return {
        "value": 1.0 / (t * scale), "unit": "pairs/s", "cores": 1, "kind": "port",
        "sample": f"oracle (literal getRectSubPix+functor port, 1 thread) on a {band_rows}-row full-width band of "
                  f"pair 0, both views, {args.iters} iterations, {args.patch}x{args.patch}: {t:.2f} s; scaled by "
                  f"swept rows {args.rows - 2 * h}/{band_rows - 2 * h}",
        "host_cores": os.cpu_count(),
    }


def host_buffer_leg(pm, params, args, pair, device):
    """PCIe-inclusive rates (never `value`): pageable host images in, host maps out.  `synchronous` is
    pm_match_u8 call by call (what the reference's Match() does); `pipelined` keeps 3 pairs in flight with
    pm_submit_u8 / pm_collect, so packing, upload and download overlap the matching of the neighbours."""
    n = args.host_pairs
    seeds = (None, None) if args.self_seed else (pair["seed_l"], pair["seed_r"])
    out = {"pairs": n, "depth": 3, "unit": "pairs/s"}
    with pm.Engine(params, device=device, max_rows=args.rows, max_cols=args.cols, max_batch=3) as e:
        e.match(pair["left"], pair["right"], *seeds)
        t0 = time.perf_counter()
        for _ in range(n):
            e.match(pair["left"], pair["right"], *seeds)
        out["synchronous"] = n / (time.perf_counter() - t0)
        t0 = time.perf_counter()
        for i in range(n):
            if e.in_flight() == 3:
                e.collect()
            e.submit(pair["left"], pair["right"], *seeds, tag=i)
        while e.in_flight():
            e.collect()
        out["pipelined"] = n / (time.perf_counter() - t0)
    # two handles, pairs alternating between them: two matches (four view streams) share the chip
    with pm.Engine(params, device=device, max_rows=args.rows, max_cols=args.cols, max_batch=2) as e0, \
            pm.Engine(params, device=device, max_rows=args.rows, max_cols=args.cols, max_batch=2) as e1:
        engs = (e0, e1)
        for e in engs:
            e.match(pair["left"], pair["right"], *seeds)
        t0 = time.perf_counter()
        for i in range(n):
            e = engs[i & 1]
            if e.in_flight() == 2:
                e.collect()
            e.submit(pair["left"], pair["right"], *seeds, tag=i)
        for e in engs:
            while e.in_flight():
                e.collect()
        out["pipelined_two_handles"] = n / (time.perf_counter() - t0)
    return out


def main():
    args = parse()
    d = Dist(args)
    steps, warmup = args.steps, args.warmup
    result = {
        "metric": "stereo_pairs_per_sec_1280x720_patchmatch", "unit": "pairs/s", "n_gpus": d.world, "steps": steps,
        "warmup": warmup, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{max(1, args.pairs_per_gpu)} synthetic {args.cols}x{args.rows} stereo pair(s) per GPU, "
                               f"{args.iters} iterations, "
                               f"{args.patch}x{args.patch} window, fp32 cost, {'PM_SEM_CPU' if args.semantics == 0 else 'PM_SEM_GPU (5-tap)'}, left+right view + cross-check "
                               "(BASELINE.json configs[1])",
                   "pairs_per_gpu_per_step": max(1, args.pairs_per_gpu),
                   "sharding": "pair index = rank * pairs_per_gpu + i, no collective"},
    }
    if args.dry_run:
        # plumbing only: same barrier / reduction path, no engine, no numbers worth reading
        d.barrier()
        t0 = time.perf_counter()
        for _ in shard(d.rank, d.world, steps):
            time.sleep(0.001 * (1 + d.rank))
        elapsed = d.max_over_ranks(time.perf_counter() - t0)
        d.barrier()
        if d.rank == 0:
            result.update(value=d.world * steps / elapsed, ms_per_step=1e3 * elapsed / steps, dry_run=True)
            print(json.dumps(result), flush=True)
        d.close()
        return

    import numpy as np
    import torch
    import pm_ctypes as pm
    import synth

    torch.cuda.set_device(d.local_rank)
    dev = torch.device(f"cuda:{d.local_rank}")
    nb = max(1, args.pairs_per_gpu)
    # rank r owns pairs r*nb .. r*nb+nb-1 (a few distinct pairs are generated and repeated to fill the batch)
    uniq = [synth.make_pair(d.rank * nb + i, args.rows, args.cols) for i in range(min(nb, 4))]
    pairs = [uniq[i % len(uniq)] for i in range(nb)]
    pair = pairs[0]
    stack = lambda k: torch.from_numpy(np.stack([p[k] for p in pairs])).to(dev).contiguous()
    L, R, SL, SR = stack("left"), stack("right"), stack("seed_l"), stack("seed_r")
    DLb = torch.empty((nb, args.rows, args.cols), dtype=torch.float32, device=dev)
    DRb = torch.empty_like(DLb)
    DL = DLb[0]
    params = pm.default_params(args.semantics, patch=args.patch, patchmatch_iters=args.iters, engine=args.engine,
                               sparse_init=1 if args.self_seed else 0)
    eng = pm.Engine(params, device=d.local_rank, max_rows=args.rows, max_cols=args.cols, max_batch=nb)

    def step():
        eng.match_device(nb, L.data_ptr(), R.data_ptr(), args.rows, args.cols,
                         None if args.self_seed else SL.data_ptr(), None if args.self_seed else SR.data_ptr(),
                         DLb.data_ptr(), DRb.data_ptr())

    for _ in range(warmup):
        step()
    eng.synchronize()
    ref = DL.clone()
    eng.profile_read()
    every = max(1, args.profile_every)
    n_prof = 0

    torch.cuda.synchronize()
    d.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i, _ in enumerate(shard(d.rank, d.world, steps)):
        timed = (not args.no_profile) and i % every == 0
        eng.profile_enable(timed)
        n_prof += 1 if timed else 0
        step()
    eng.synchronize()
    torch.cuda.synchronize()
    elapsed_local = time.perf_counter() - t0
    d.barrier()
    torch.cuda.synchronize()
    elapsed = d.max_over_ranks(elapsed_local)

    prof = eng.profile_read()
    eng.profile_enable(False)
    # work counters: one extra untimed step (the counting atomics would distort the timed ones)
    eng.debug_counters_enable(True)
    eng.debug_counters()
    step()
    counters = eng.debug_counters()
    eng.debug_counters_enable(False)
    deterministic = bool(torch.equal(ref, DL))
    fg = float((DL > 0).float().mean().item())
    err = (DL - torch.from_numpy(pair["gt"]).to(dev)).abs()
    within1 = float((err[DL > 0] < 1.0).float().mean().item()) if fg > 0 else 0.0

    if d.rank == 0:
        px_views = args.rows * args.cols * 2 * nb
        dom = max(("sweep_row", "sweep_col"), key=lambda k: prof[k][1])
        n_launch, total_ms = prof[dom]
        avg_ms = total_ms / max(n_launch, 1)
        # a class runs 2 sweeps per iteration over every pixel of both views; with the views on their own streams
        # (default) a launch covers one view, and two launches run concurrently on the chip
        launches_per_step = n_launch / max(n_prof, 1)
        concurrency = max(1, round(launches_per_step / (2 * args.iters)))
        bytes_per_launch = SWEEP_BYTES_PER_PX * px_views / concurrency
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if n_launch else 0.0
        gpu_ms = sum(v[1] for v in prof.values())
        result.update(
            value=d.world * nb * steps / elapsed, ms_per_step=1e3 * elapsed / steps,
            ms_per_frame=1e3 * elapsed / steps / nb,
            roofline={"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(dom) if nb == 1 else None,
                      "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": avg_ms,
                      "launches": n_launch, "profiled_steps": n_prof, "concurrent_launches": concurrency,
                      "achieved_all_concurrent": achieved * concurrency,
                      "note": "per-launch figure as specified, HIP events on every --profile-every-th timed step "
                              "(`profiled_steps` of them); `concurrent_launches` launches of this class (one per view, "
                              "own streams) share the chip, so the chip-level rate is achieved_all_concurrent"},
            kernels_ms_per_step={k: v[1] / max(n_prof, 1) for k, v in prof.items()},
            gpu_busy_ms_per_step=gpu_ms / max(n_prof, 1),
            run_engine_counters_per_step=counters,
            check={"deterministic_across_steps": deterministic, "foreground_fraction": fg,
                   "foreground_within_1px_of_truth": within1},
        )
        if d.world == 1 and args.host_pairs > 0:
            result["host_buffers"] = host_buffer_leg(pm, params, args, pair, d.local_rank)
        if d.world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(result), flush=True)
    eng.close()
    d.close()


if __name__ == "__main__":
    main()
