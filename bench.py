#!/usr/bin/env python3
"""Benchmark of the hot path: PatchMatch stereo Match() on MI355X.

Headline workload (BASELINE.json configs[1]): one 1280x720 synthetic stereo pair per GPU, 8 iterations,
11x11 window, fp32 cost, reference-CPU semantics (PM_SEM_CPU, the parity-bearing scalar mode), left + right
view + cross-check.  A "step" is one Match() through the C ABI entry point pm_match_device with the u8 pair and
the seed maps already resident in HBM and the disparity maps written to HBM; the timed steps rotate over four
distinct resident pairs, so the inputs are not the same cache-hot pair every step.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank r matches its own pairs (stereo pairs are independent: no data-path collective, weak scaling).  The K
timed steps are bracketed by barrier + synchronize on both sides; rank 0 prints ONE JSON line with the whole-job
pairs/s (max elapsed over ranks), the roofline of the dominant kernel (per-launch duration from HIP events
recorded by the engine on its own stream during the timed steps) and the CPU baseline (the oracle -- a port of
the reference CPU path -- timed on this host, rank 0, N=1).

Side legs in the same line (rank 0, N=1, never `value`; all measured in THIS process, beside the headline's handle):
    planes            the slanted-plane mode (north_star kernels) at the same shape: fp32 state (configs[1] shape) and
                      fp16 state behind the on-device stereo-ready enhancement (configs[4] shape)
    host_buffers      PCIe-inclusive rates of the host-buffer entry points (pageable and page-locked caller memory,
                      seeded and self-seeded): synchronous, the frame sequence, batches of 32
    sequence_device   the frame sequence on device-resident pairs (pm_submit_device)
    host_sequence_all_ranks  the PCIe-inclusive frame sequence on page-locked buffers on EVERY rank of the run, reduced like
                      `value` (configs[2] incl. H2D / D2H; also in multi-rank runs)
    batch             4 and 32 pairs per pm_match_device call (configs[2]'s per-GPU share)
    reference_test_shape  the reference's own timed call pattern (patchmatch_gpu_test.cpp:68-88) with its CPU side
    tiled_4096x2160   configs[3]: untiled, and through the C-ABI driver in 8 bands on this device
Other workloads: --mode planes [--state f16] [--enhance] makes the plane mode the timed one; --pairs-per-gpu 32 is
configs[2]'s per-GPU share; --tiled runs configs[3] (one 4096x2160 pair row-tiled over the ranks).
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
# Algorithmic bytes.  Scalar mode: a directional sweep reads the four f32-equivalent image planes once
# (16 B), reads the disparity (4 B) and writes it (4 B) per pixel and view; a whole pair is N * (78 + 216 * I).
SWEEP_BYTES_PER_PX = 24
# Plane mode: state = 12 B plane + 4 B cost per pixel and view (8 B with fp16 state); per iteration and view the
# spatial stage moves 48 B/px (two colour launches of 24), view propagation 64, refinement 48: N * (74 + 320 * I).
# bytes per px and view of ONE launch of a class: a red or black launch of the spatial stage is half of its 48; the fused
# view propagation + refinement launch of the default schedule is 64 + 48
PLANE_LAUNCH_BYTES = {"planes_spatial": 24, "planes_view": 64, "planes_refine": 48, "planes_view_refine": 112}
# views one launch of the class covers (the spatial stage sweeps both views of a pair in one launch)
PLANE_LAUNCH_VIEWS = {"planes_spatial": 2, "planes_view": 1, "planes_refine": 2, "planes_view_refine": 1}
N_ROTATE = 4  # distinct device-resident pairs the timed steps rotate over


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL) on GPUs; gloo with --dry-run for CPU tests")
    ap.add_argument("--dry-run", action="store_true",
                    help="exercise sharding/barrier/reduction without a GPU (no compute, no engine)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true",
                    help="time the single-threaded oracle on the WHOLE frame whatever the band sample predicts")
    ap.add_argument("--cpu-budget-s", type=float, default=80.0,
                    help="the single-threaded whole-frame CPU baseline is timed (unscaled) when the band sample predicts "
                         "at most this many seconds (1280x720: ~67 s); otherwise the band-scaled figure is reported")
    ap.add_argument("--rows", type=int, default=ROWS)
    ap.add_argument("--cols", type=int, default=COLS)
    ap.add_argument("--iters", type=int, default=ITERS)
    ap.add_argument("--patch", type=int, default=PATCH)
    ap.add_argument("--engine", type=int, default=0)
    ap.add_argument("--pairs-per-gpu", type=int, default=1,
                    help="pairs matched per step and GPU (1 = BASELINE configs[1]; 32 = configs[2]'s per-GPU share)")
    ap.add_argument("--self-seed", action="store_true",
                    help="let Match() compute its seeds with the device SparseInit (side measurement)")
    ap.add_argument("--profile-every", type=int, default=8,
                    help="per-kernel HIP events are recorded on every n-th timed step (the ~180 event records of a "
                         "fully timed step cost 6 %% of the step; 1 = every step)")
    ap.add_argument("--no-profile", action="store_true",
                    help="experiment: no per-kernel HIP events in the timed region (the roofline object is then empty)")
    ap.add_argument("--host-pairs", type=int, default=64,
                    help="pairs of the untimed host-buffer leg (PCIe-inclusive rates, reported beside `value`); 0 = skip")
    ap.add_argument("--semantics", type=int, default=0,
                    help="0 = PM_SEM_CPU (the benchmark configuration), 1 = PM_SEM_GPU (side measurement)")
    ap.add_argument("--mode", choices=("scalar", "planes"), default="scalar",
                    help="scalar = the reference's algorithm (headline); planes = slanted-plane mode as the timed workload")
    ap.add_argument("--state", choices=("f32", "f16"), default="f32", help="plane / cost storage type (--mode planes)")
    ap.add_argument("--enhance", action="store_true",
                    help="--mode planes: inputs are BGR images, the stereo-ready enhancement (pm_stereo_ready) of both "
                         "runs on the device in front of every Match (BASELINE configs[4])")
    ap.add_argument("--plane-neighbours", type=int, choices=(0, 1), default=0,
                    help="--mode planes: 1 = the spatial stage's two-neighbour option (PM_PL_NEIGH_TWO)")
    ap.add_argument("--no-side-legs", action="store_true", help="skip the plane-mode side legs of the default run")
    ap.add_argument("--tiled", action="store_true",
                    help="BASELINE configs[3]: one 4096x2160 pair row-tiled over the ranks (see python/tiled.py)")
    ap.add_argument("--tiled-rccl", action="store_true",
                    help="multi-rank configs[3] leg over RCCL (python/tiled.py, one rank per GPU) instead of the "
                         "single-process C-ABI driver (pm_tiled_*)")
    ap.add_argument("--tiled-timeout", type=int, default=150,
                    help="deadline (s) of the crash-isolated configs[3] leg of a multi-rank run")
    ap.add_argument("--rehearse-tiled-leg", action="store_true",
                    help="with --dry-run: start the tiled children anyway (without a GPU they fail) -- tests that a "
                         "failing leg ends up as an error entry and the line is still printed")
    return ap.parse_args()


class Dist:
    """One process per GPU; torch.distributed only for the barrier and the max over ranks."""

    def __init__(self, args):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        # PM_BENCH_FORCE_DIST=1: a process group of ONE rank (rehearsal on a one-GPU box: the barrier and the all-reduce
        # then run real RCCL kernels on RCCL's own stream beside the engine's streams)
        if self.world > 1 or os.environ.get("PM_BENCH_FORCE_DIST") == "1":
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

    def ranks_seen(self):
        return self.dist.get_world_size() if self.dist else 1

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
    """Memory-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/traffic.json,
    tools/make_traffic.py: separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same command), CORRECTED as
    MI355X_MICROARCH.md's HBM section prescribes: on gfx950 FETCH_SIZE reports exactly half of the bytes read (confirmed on
    this engine's access widths: profiles/r06_calib_fetch_write.txt), so bytes = 2 x fetch + write; WRITE_SIZE counts whole
    32-byte sectors of sparse stores (an upper bound for the sweeps' write-back).  The counters sit on the L2's memory
    side: Infinity-Cache hits are included.  None if the file is missing."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        k = t["kernels"][kernel_class]
        return float(k["fetch_kib"]) * 1024.0 * 2.0 + float(k["write_kib"] or 0.0) * 1024.0
    except Exception:
        return None


def pmc_valu(kernel_class, prof=None, n_prof=0, variant=""):
    """The issue-side roofs of the window kernels beside the HBM figure BASELINE.json asks for, from the committed
    rocprofv3 --pmc passes of this same command (profiles/valu.json, written by tools/make_valu.py from separate SQ /
    TA / TD runs; rocprofv3 serialises the launches, so the fractions are those of a kernel ALONE on the chip):
    issue_frac = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x kernel cycles); ta / td_busy_frac = texture address / data
    unit busy cycles over 256 CUs x kernel cycles.  insts_per_step = vector instructions of one timed step: per-launch
    counts of the profiled classes x this run's launches per step.  `variant`: suffix of a kernel variant that has a PMC
    pass of its OWN in the file (e.g. "@two_neighbours": the same kernel name executes about half the instructions);
    classes the variant's pass did not cover, or a variant without a pass, give None -- counts measured on another
    variant are never applied.  None if the file is missing."""
    try:
        with open(os.path.join(ROOT, "profiles", "valu.json")) as f:
            t = json.load(f)
        if variant:
            have = {c[:-len(variant)]: v for c, v in t["kernels"].items() if c.endswith(variant)}
            if prof and any(n and c.startswith("planes_") and c not in have for c, (n, _) in prof.items()):
                return None
            t = {"kernels": have}
        k = t["kernels"][kernel_class]
        out = {"issue_frac": k["valu_issue_frac"], "ta_busy_frac": k.get("ta_busy_frac"),
               "td_busy_frac": k.get("td_busy_frac"), "insts_valu_per_launch": k.get("insts_valu_per_launch"),
               "source": "profiles/valu.json (committed rocprofv3 --pmc passes of this command; kernels alone on the chip)"}
        if prof and n_prof:
            out["insts_per_step"] = sum(v["insts_valu_per_launch"] * prof[c][0] / n_prof
                                        for c, v in t["kernels"].items() if c in prof and v.get("insts_valu_per_launch"))
        return out
    except Exception:
        return None


ENGINE_CLOCK_GHZ = 2.4  # MI355X peak engine clock (MI355X_MICROARCH.md); 256 CUs x 4 SIMDs, a wave64 op holds its SIMD 4 cycles


def binding_roof(kernel_class, prof, n_prof, ms_per_step, variant=""):
    """What actually bounds the step, computed from THIS run's ms_per_step: the chip-level vector-issue fraction -- vector
    instructions of one step (per-launch counts of the committed PMC passes x this run's launches per step) x 4 cycles
    over (1024 SIMDs x clock x the step's wall time) -- beside the texture address / data unit occupancy of the dominant
    kernel (committed PMC passes: kernels alone on the chip; with both views' kernels running the units are shared)."""
    v = pmc_valu(kernel_class, prof, n_prof, variant)
    if not v or not v.get("insts_per_step") or not ms_per_step:
        return None
    issue = v["insts_per_step"] * 4.0 / (1024.0 * ENGINE_CLOCK_GHZ * 1e9 * ms_per_step * 1e-3)
    return {"issue_frac_chip": issue, "ta_busy": v.get("ta_busy_frac"), "td_busy": v.get("td_busy_frac"),
            "insts_valu_per_step": v["insts_per_step"], "clock_ghz": ENGINE_CLOCK_GHZ,
            "formula": "insts_valu_per_step * 4 / (1024 SIMDs * clock * ms_per_step); ta / td: the dominant kernel "
                       "alone on the chip (profiles/valu.json)"}


def no_cpu_leg(result, world):
    """The contract asks for the CPU leg on rank 0 at N = 1 only: the N = 1 point of a scaling run is the BENCH line, which
    carries it (and check.equals_oracle_full_frame) -- an N > 1 line says so instead of leaving the key out."""
    result["cpu_baseline"] = None
    result["cpu_baseline_reason"] = ("timed on rank 0 at N = 1 only (bench.py --gpus 1 carries it together with "
                                     "check.equals_oracle_full_frame); this is an N = %d line" % world)


def shard(rank, world, steps, nb):
    """Pair indices matched by `rank` at each step (what Workload builds): rank r owns the N_ROTATE distinct pairs
    r * N_ROTATE ...; step s matches them in rotation from pair s on -- one pair at nb = 1, the same pairs repeated in a
    batch.  No pair is shared between ranks."""
    return [[rank * N_ROTATE + (s + i) % N_ROTATE for i in range(nb)] for s in range(steps)]


def cpu_baseline(args):
    """The oracle (port of src/vehicle/stereo_matching/patchmatch.cpp + the test recipe, literal call structure) timed on
    this host: single-threaded like the reference -- first a 160-row full-width band (10-15 s), which also predicts the
    whole frame; the WHOLE frame, unscaled, is then timed too when that prediction fits the budget (--cpu-budget-s,
    default 80 s; 1280x720 takes ~67 s) and becomes `value` -- and on all host cores (the whole frame; rows / columns of a
    sweep are independent)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    import synth
    p = synth.make_pair(0, args.rows, args.cols)
    band_rows = min(160, args.rows)
    y0 = (args.rows - band_rows) // 2
    band = slice(y0, y0 + band_rows)
    prm = O.default_params(O.SEM_CPU, patch=args.patch, n_iters=args.iters, nthreads=1, literal=1, left_right_check=1)
    t0 = time.perf_counter()
    O.match(prm, p["left"][band], p["right"][band], p["seed_l"][band], p["seed_r"][band])
    t_band = time.perf_counter() - t0
    h = args.patch // 2
    scale = (args.rows - 2 * h) / float(band_rows - 2 * h)  # swept rows of the full image / of the band
    predicted = t_band * scale
    t_full = None
    if band_rows < args.rows and (args.cpu_baseline_full or predicted <= args.cpu_budget_s):
        t0 = time.perf_counter()
        O.match(prm, p["left"], p["right"], p["seed_l"], p["seed_r"])
        t_full = time.perf_counter() - t0
    elif band_rows == args.rows:
        t_full = t_band
    ncpu = os.cpu_count() or 1
    nthr = min(ncpu, 16)  # the GPU box's CPU share for one GPU is 16 cores; more threads only oversubscribe it
    prm_all = O.default_params(O.SEM_CPU, patch=args.patch, n_iters=args.iters, nthreads=nthr, literal=1,
                               left_right_check=1)
    t0 = time.perf_counter()
    maps_all = O.match(prm_all, p["left"], p["right"], p["seed_l"], p["seed_r"])
    t_all = time.perf_counter() - t0
    what = (f"oracle (literal getRectSubPix+functor port, 1 thread: the reference CPU path has no threading), pair 0, both "
            f"views, {args.iters} iterations, {args.patch}x{args.patch}")
    band_note = (f"{band_rows}-row full-width band: {t_band:.2f} s, scaled by swept rows {args.rows - 2 * h}/"
                 f"{band_rows - 2 * h} -> {predicted:.1f} s per frame")
    return {
        "value": 1.0 / (t_full if t_full is not None else predicted), "unit": "pairs/s", "cores": 1, "kind": "port",
        "sampled": "whole frame, unscaled" if t_full is not None else "band sample scaled by swept rows (x%.3f)" % scale,
        "sample": (f"{what}: the WHOLE {args.cols}x{args.rows} frame in {t_full:.2f} s, unscaled (cross-check: {band_note})"
                   if t_full is not None else f"{what}: {band_note} (the whole frame was not timed: over --cpu-budget-s)"),
        "band_scaled": {"value": 1.0 / predicted, "unit": "pairs/s", "seconds_band": t_band, "scale": scale},
        "all_cores": {"value": 1.0 / t_all, "unit": "pairs/s", "cores": nthr, "kind": "port",
                      "sample": f"the same oracle on the WHOLE {args.cols}x{args.rows} frame with {nthr} OpenMP threads "
                                f"(rows / columns of a sweep in parallel; the box's CPU share for one GPU is 16 cores "
                                f"of {ncpu}): {t_all:.2f} s, unscaled"},
        "host_cores": ncpu,
        # pair 0's maps of the all-cores run (whole frame): the checker of `check.equals_oracle_full_frame`; popped by
        # main() before the line is printed
        "_maps": maps_all,
    }


def cpu_baseline_planes(args, f16, neighbours=0):
    """The plane mode's CPU definition (oracle/pm_planes_oracle.c) on all host cores, whole frame.  `_maps` = pair 0's
    maps (the checker of `check.equals_oracle_full_frame`; popped before the line is printed)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    import synth
    p = synth.make_pair(0, args.rows, args.cols)
    import pm_ctypes as pm
    nthr = min(os.cpu_count() or 1, 16)
    # the parameter struct the engine's plane legs run with (Workload), translated field by field
    prm = O.planes_params(**O.planes_kwargs_of(pm.default_params(
        0, patch=args.patch, patchmatch_iters=args.iters, mode=pm.PM_MODE_PLANES,
        state_dtype=pm.PM_STATE_F16 if f16 else pm.PM_STATE_F32, plane_neighbours=neighbours), nthreads=nthr))
    t0 = time.perf_counter()
    maps = O.planes_match(prm, p["left"], p["right"])
    t = time.perf_counter() - t0
    return {"_maps": maps, "value": 1.0 / t, "unit": "pairs/s", "cores": nthr, "kind": "port",
            "sample": f"oracle/pm_planes_oracle.c (this mode's own CPU definition; the reference has no slanted-plane "
                      f"code) on the whole {args.cols}x{args.rows} frame, {nthr} OpenMP threads: {t:.2f} s"}


def equals_oracle(mine, theirs, what):
    """Whole-frame comparison of the engine's maps of pair 0 with the oracle's (value equality of the float32 maps,
    tolerance 0, as tests/conftest.py::assert_same); the reference pattern it stands in for is the whole-image recipe of
    test/stereo_matching/patchmatch_test.cpp:149-183 (which asserts nothing)."""
    import numpy as np
    nl, nr = int((mine[0] != theirs[0]).sum()), int((mine[1] != theirs[1]).sum())
    return {"left": nl == 0, "right": nr == 0, "differing_pixels": nl + nr, "pixels_compared": int(mine[0].size + mine[1].size),
            "tolerance": 0, "checker": what}


def host_buffer_leg(pm, args, pairs, device):
    """PCIe-inclusive rates (never `value`): host images in, host maps out, seeded (seed maps uploaded) and self-seeded
    (SparseInit on the device, as the reference's Match() does).  Every way in exists for PAGEABLE caller memory (the
    engine packs into its pinned slab and unpacks) and for memory made known through pm_host_alloc / pm_host_register
    (DMA in place, nothing staged):
      synchronous   pm_match_u8 call by call -- what the reference's host Match() does (patchmatch_gpu.cu:331-376)
      pipelined     the frame sequence pm_submit_u8 / pm_submit_bound_u8 + pm_collect, `depth` frames in flight
      batch         pm_match_batch_u8: `batch_pairs` pairs per call (BASELINE configs[2]'s per-GPU share incl. H2D / D2H)
      sequence_device  pm_submit_device: the same frame sequence on device-resident pairs (no copies at all)"""
    import numpy as np
    n, depth, nbatch = args.host_pairs, 4, 32
    rows, cols = args.rows, args.cols
    out = {"pairs": n, "depth": depth, "batch_pairs": nbatch, "unit": "pairs/s",
           "note": "the caller re-uses its buffers from frame to frame; `pinned` = buffers of pm_host_alloc / "
                   "pm_host_register memory (DMA in place), `pageable` = plain numpy arrays (staged through the "
                   "engine's pinned slab by four host threads)"}
    fmaps = lambda k: [(np.zeros((rows, cols), np.float32), np.zeros((rows, cols), np.float32)) for _ in range(k)]
    for self_seed in (False, True):
        params = pm.default_params(args.semantics, patch=args.patch, patchmatch_iters=args.iters, engine=args.engine,
                                   sparse_init=1 if self_seed else 0)
        seeds = (lambda p: (None, None)) if self_seed else (lambda p: (p["seed_l"], p["seed_r"]))
        res = {}
        with pm.Engine(params, device=device, max_rows=rows, max_cols=cols, max_batch=depth) as e:
            bufs = fmaps(depth)
            pin = lambda a_: (lambda b_: (np.copyto(b_, a_), b_)[1])(e.host_alloc(a_.shape, a_.dtype))
            pinned = [{k: pin(p[k]) for k in ("left", "right", "seed_l", "seed_r")} for p in pairs]
            pbufs = [(e.host_alloc((rows, cols), np.float32), e.host_alloc((rows, cols), np.float32)) for _ in range(depth)]
            for tag, src, dst in (("pageable", pairs, bufs), ("pinned", pinned, pbufs)):
                e.match(src[0]["left"], src[0]["right"], *seeds(src[0]), out=dst[0])
                t0 = time.perf_counter()
                for i in range(n):
                    p = src[i % len(src)]
                    e.match(p["left"], p["right"], *seeds(p), out=dst[i % depth])
                res["synchronous_" + tag] = n / (time.perf_counter() - t0)
                for rep in range(2):  # the first pass warms the ring (first-use allocations of the view streams' scratch)
                    k = 2 * depth if rep == 0 else n
                    t0 = time.perf_counter()
                    for i in range(k):
                        if e.in_flight() == depth:
                            e.collect() if tag == "pinned" else e.collect(out=dst[(i - depth) % depth])
                        p = src[i % len(src)]
                        e.submit(p["left"], p["right"], *seeds(p), tag=i, out=dst[i % depth] if tag == "pinned" else None)
                    j = k - e.in_flight()
                    while e.in_flight():
                        e.collect() if tag == "pinned" else e.collect(out=dst[j % depth])
                        j += 1
                    res["pipelined_" + tag] = k / (time.perf_counter() - t0)
        with pm.Engine(params, device=device, max_rows=rows, max_cols=cols, max_batch=nbatch) as e:
            pin = lambda a_: (lambda b_: (np.copyto(b_, a_), b_)[1])(e.host_alloc(a_.shape, a_.dtype))
            pinned = [{k: pin(p[k]) for k in ("left", "right", "seed_l", "seed_r")} for p in pairs]
            for tag, src in (("pageable", pairs), ("pinned", pinned)):
                ls = [src[i % len(src)]["left"] for i in range(nbatch)]
                rs = [src[i % len(src)]["right"] for i in range(nbatch)]
                sl = None if self_seed else [src[i % len(src)]["seed_l"] for i in range(nbatch)]
                sr = None if self_seed else [src[i % len(src)]["seed_r"] for i in range(nbatch)]
                if tag == "pinned":
                    outs = ([e.host_alloc((rows, cols), np.float32) for _ in range(nbatch)],
                            [e.host_alloc((rows, cols), np.float32) for _ in range(nbatch)])
                else:
                    outs = ([np.zeros((rows, cols), np.float32) for _ in range(nbatch)],
                            [np.zeros((rows, cols), np.float32) for _ in range(nbatch)])
                e.match_batch(ls[:4], rs[:4], sl[:4] if sl else None, sr[:4] if sr else None, out=(outs[0][:4], outs[1][:4]))
                calls = max(1, n // nbatch)
                t0 = time.perf_counter()
                for _ in range(calls):
                    e.match_batch(ls, rs, sl, sr, out=outs)
                res["batch_" + tag] = calls * nbatch / (time.perf_counter() - t0)
        out["self_seeded" if self_seed else "seeded"] = res
    return out


def host_sequence_all_ranks(pm, args, pairs, d):
    """BASELINE configs[2] INCLUDING the PCIe transfers ("aggregate pairs/s incl. H2D/D2H (pinned,
    overlapped)"), on EVERY rank of the run: host images in, host maps out through the frame sequence pm_submit_bound_u8 /
    pm_collect, all caller buffers page-locked (pm_host_alloc: DMA in place, nothing staged), `depth` frames in flight.
    Bracketed like the headline -- barrier on both sides, max elapsed over ranks -- and reduced like `value`: frames of
    all ranks / that time.  Collective: every rank must call it.  What it mirrors: upload ... download inside the
    reference's host Match() (patchmatch_gpu.cu:343-375); eight ranks with their own pinned buffers and DMA queues are
    where a host path could stop scaling, which a device-resident headline cannot see.  Never `value`."""
    import numpy as np
    n, depth = args.host_pairs, 4
    rows, cols = args.rows, args.cols
    if args.dry_run:  # plumbing only (tests/test_dist.py): same barriers and reduction, no engine
        d.barrier()
        t0 = time.perf_counter()
        time.sleep(0.001 * n * (1 + d.rank))
        local = time.perf_counter() - t0
    else:
        params = pm.default_params(args.semantics, patch=args.patch, patchmatch_iters=args.iters, engine=args.engine,
                                   sparse_init=1 if args.self_seed else 0)
        seeds = (lambda p: (None, None)) if args.self_seed else (lambda p: (p["seed_l"], p["seed_r"]))
        with pm.Engine(params, device=d.local_rank, max_rows=rows, max_cols=cols, max_batch=depth) as e:
            pin = lambda a_: (lambda b_: (np.copyto(b_, a_), b_)[1])(e.host_alloc(a_.shape, a_.dtype))
            src = [{k: pin(p[k]) for k in ("left", "right", "seed_l", "seed_r")} for p in pairs]
            dst = [(e.host_alloc((rows, cols), np.float32), e.host_alloc((rows, cols), np.float32)) for _ in range(depth)]

            def run(k):
                for i in range(k):
                    if e.in_flight() == depth:
                        e.collect()
                    p = src[i % len(src)]
                    e.submit(p["left"], p["right"], *seeds(p), tag=i, out=dst[i % depth])
                while e.in_flight():
                    e.collect()
            run(2 * depth)
            d.barrier()
            t0 = time.perf_counter()
            run(n)
            local = time.perf_counter() - t0
    d.barrier()
    elapsed = d.max_over_ranks(local)
    return {"value": d.world * n / elapsed, "unit": "pairs/s", "frames_per_rank": n, "depth": depth, "ranks": d.world,
            "ms_per_frame_per_rank": 1e3 * elapsed / n,
            "note": "PCIe-inclusive: every rank's frame sequence on page-locked caller buffers (2 x 0.92 MB in + 2 x 3.69 MB "
                    "seeds in + 2 x 3.69 MB maps out per pair), frames of all ranks / max elapsed over ranks"}


def sequence_device_leg(pm, torch, w, args, device, depth=4, frames=96):
    """pm_submit_device / pm_collect on the headline's resident pairs: one pair per call as in the headline, but the engine
    may overlap consecutive frames (two frames advanced through every launch together while the device is busy)."""
    a = args
    with pm.Engine(w.params, device=device, max_rows=a.rows, max_cols=a.cols, max_batch=depth) as e:
        DL = torch.empty((depth, a.rows, a.cols), dtype=torch.float32, device=w.DL.device)
        DR = torch.empty_like(DL)
        seeded = w.mode == "scalar" and not a.self_seed  # as Workload.step passes them
        L = torch.cat([w.L[g][:1] for g in range(N_ROTATE)]).contiguous()
        R = torch.cat([w.R[g][:1] for g in range(N_ROTATE)]).contiguous()
        SL = torch.cat([w.SL[g][:1] for g in range(N_ROTATE)]).contiguous()
        SR = torch.cat([w.SR[g][:1] for g in range(N_ROTATE)]).contiguous()
        torch.cuda.synchronize()

        def run(k):
            for i in range(k):
                if e.in_flight() == depth:
                    e.collect_device()
                q = i % N_ROTATE
                e.submit_device(L[q].data_ptr(), R[q].data_ptr(), a.rows, a.cols, SL[q].data_ptr() if seeded else None,
                                SR[q].data_ptr() if seeded else None, DL[i % depth].data_ptr(), DR[i % depth].data_ptr(), tag=i)
            while e.in_flight():
                e.collect_device()
        run(2 * depth)
        t0 = time.perf_counter()
        run(frames)
        dt = time.perf_counter() - t0
        # the map of the last frame of pair 0 equals the headline's (same pair, same parameters)
        last0 = ((frames - 1) // N_ROTATE) * N_ROTATE
        w.step(0)
        w.eng.synchronize()
        same = bool(torch.equal(DL[last0 % depth], w.DL[0]))
    return {"value": frames / dt, "unit": "pairs/s", "ms_per_frame": 1e3 * dt / frames, "frames": frames, "depth": depth,
            "equals_the_headline_maps": same,
            "note": ("one device-resident pair per pm_submit_device call, up to `depth` in flight: frames run back to "
                     "back on the handle's two view streams (no fork / join per frame), two of them through every "
                     "launch together while the device is busy") if w.mode == "scalar" else
                    ("one device-resident pair per pm_submit_device call, up to `depth` in flight: while the device is busy "
                     "a frame waits for the next one and the two run as a batch of two lanes on two streams")}


def batch_leg(pm, torch, np, synth, args, dev, device, nb, steps):
    """The headline workload as a batch -- `nb` pairs per pm_match_device call, BASELINE configs[2]'s per-GPU shape --
    in THIS process, beside the other handles (round 3 needed a process of its own for it: tools/stream_matrix.py)."""
    class A:
        pass
    a2 = A()
    a2.__dict__.update(vars(args))
    a2.pairs_per_gpu = nb
    w = Workload(a2, pm, torch, np, synth, dev, device, 0, "scalar", "f32", False)
    for s_ in range(2):
        w.step(s_)
    w.eng.synchronize()
    t0 = time.perf_counter()
    for s_ in range(steps):
        w.step(s_)
    w.eng.synchronize()
    dt = time.perf_counter() - t0
    q = w.quality()
    # every slot holds ITS pair's map: slots repeat the rank's distinct pairs with period N_ROTATE
    per = min(nb, N_ROTATE)
    q["slots_equal_their_pairs_first_slot"] = bool(all(torch.equal(w.DL[i], w.DL[i % per]) for i in range(nb)))
    q["passes"] = bool(q["foreground_within_1px_of_truth"] >= 0.95 and q["slots_equal_their_pairs_first_slot"])
    w.eng.close()
    return {"pairs_per_call": nb, "calls": steps, "value": nb * steps / dt, "unit": "pairs/s", "ms_per_pair": 1e3 * dt / steps / nb,
            "check": q, "note": "one pm_match_device call per batch: chunks of two pairs one after the other on the "
                                 "handle's two view streams, every chunk's cross-check on a third"}


def reference_test_shape_leg(pm, args, device):
    """The only timing the reference itself does (test/stereo_matching/patchmatch_gpu_test.cpp:68-88): its farmsim test pair
    halved to 376x240, PatchmatchGpu with cost_alpha 0.9 and 3 iterations, `Match(iml, imr, disp, dispr)` five times under
    a Timer -- host images in, host maps out, self-seeded by SparseInit(4) inside the call.  The pair comes from
    tests/golden/farmsim_fs1_376x240.npz (decoded and halved by tests/golden/make_golden.py; the reference tree itself is
    not read here).  CPU side: the oracle's restatement of the same call on one thread, the whole frame, unscaled."""
    import numpy as np
    g = np.load(os.path.join(ROOT, "tests", "golden", "farmsim_fs1_376x240.npz"))
    l, r = np.ascontiguousarray(g["left"]), np.ascontiguousarray(g["right"])
    rows, cols = l.shape
    rowsum = lambda d: d.view(np.uint32).astype(np.uint64).sum(axis=1)  # as tests/test_golden.py pins them
    prm = pm.default_params(pm.PM_SEM_GPU, cost_alpha=0.9, patchmatch_iters=3, sparse_init=1)
    calls = []
    with pm.Engine(prm, device=device, max_rows=rows, max_cols=cols) as e:
        out = (np.zeros((rows, cols), np.float32), np.zeros((rows, cols), np.float32))
        for i in range(5):
            t0 = time.perf_counter()
            e.match(l, r, out=out)
            calls.append(1e3 * (time.perf_counter() - t0))
        steady = []
        for i in range(60):
            t0 = time.perf_counter()
            e.match(l, r, out=out)
            steady.append(1e3 * (time.perf_counter() - t0))
        same = bool(np.array_equal(rowsum(out[0]), g["gpu_test_rows_l"]) and np.array_equal(rowsum(out[1]), g["gpu_test_rows_r"]))
    med = float(np.median(steady))
    res = {"workload": f"{cols}x{rows} farmsim test pair, PM_SEM_GPU (the CUDA module's own 5-tap semantics), cost_alpha "
                       "0.9, 3 iterations, self-seeded, host images in / host maps out, Match x 5 "
                       "(patchmatch_gpu_test.cpp:68-88)",
           "ms_per_call_first_five": [round(c, 3) for c in calls], "ms_per_call_steady_median": med,
           "pairs_per_s_steady": 1e3 / med, "equals_the_golden_row_checksums": same,
           "note": "round 6: 0.64 -> 0.45 ms.  The sweeps of this 3x3-window semantics were bound by run steps that ended at "
                   "every position rejecting its predecessor's value (13 steps per 23-position segment); the step now "
                   "also follows the predecessor's OLD value through the positions that decline it, from a per-chain "
                   "pre-pass (4.9 steps), clamped positions no longer take a step each, the seeder lost two memset "
                   "launches per view, the view that ends last stays on the handle's stream (no cross-queue wake-up in "
                   "front of the cross-check) and the pair goes up as one kernel copy.  What is left: 12 sweeps of "
                   "13-27 us per view back to back, a 107 us seeder head (its selection kernel 46 us), ~70 us on the host "
                   "between two calls: profiles/r06_reference_call_timeline.txt"}
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        t0 = time.perf_counter()
        sp = O.seed_params()
        sl = O.sparse_init(l, r, 4, sp)
        sr = np.ascontiguousarray(O.sparse_init(np.ascontiguousarray(r[:, ::-1]), np.ascontiguousarray(l[:, ::-1]), 4, sp)[:, ::-1])
        op = O.default_params(O.SEM_GPU, n_iters=3, nthreads=1, cost_alpha=0.9)
        O.match(op, l, r, sl, sr)
        t = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": 1.0 / t, "unit": "pairs/s", "cores": 1, "kind": "port", "ms_per_call": 1e3 * t,
                               "sample": "oracle (SparseInit on both views + the 5-tap Match), one thread, the whole "
                                         "376x240 frame, unscaled"}
    return res


class Workload:
    """Device-resident inputs/outputs of one rank and the step function of the selected mode."""

    def __init__(self, args, pm, torch, np, synth, dev, local_rank, rank, mode, state, enhance, plane_neighbours=0):
        self.args, self.pm, self.torch, self.mode, self.enhance = args, pm, torch, mode, enhance
        nb = max(1, args.pairs_per_gpu)
        self.nb = nb
        # rank r owns the N_ROTATE distinct pairs r * N_ROTATE ...; group g (what step g matches) holds them in rotation
        # from pair g on: one pair per group at nb = 1 -- the headline -- and the SAME four pairs repeated in every batch.
        # (Until round 5 a batch drew 4 x N_ROTATE other pairs; scenes differ by +-4 % in their sweep work, and the
        # batch leg read 3.6 % below the frame sequence for that reason alone: profiles/r06_batch_vs_sequence.txt.)
        uniq = [synth.make_pair(rank * N_ROTATE + i, args.rows, args.cols) for i in range(N_ROTATE)]
        self.pairs = uniq
        groups = [[uniq[(g + i) % N_ROTATE] for i in range(nb)] for g in range(N_ROTATE)]
        stack = lambda grp, k: torch.from_numpy(np.stack([p[k] for p in grp])).to(dev).contiguous()
        self.L = [stack(g, "left") for g in groups]
        self.R = [stack(g, "right") for g in groups]
        self.SL = [stack(g, "seed_l") for g in groups]
        self.SR = [stack(g, "seed_r") for g in groups]
        self.gt = [torch.from_numpy(g[0]["gt"]).to(dev) for g in groups]  # truth of slot 0 of every group
        self.last_g = 0
        self.DL = torch.empty((nb, args.rows, args.cols), dtype=torch.float32, device=dev)
        self.DR = torch.empty_like(self.DL)
        if mode == "planes":
            self.params = pm.default_params(0, patch=args.patch, patchmatch_iters=args.iters, mode=pm.PM_MODE_PLANES,
                                            state_dtype=pm.PM_STATE_F16 if state == "f16" else pm.PM_STATE_F32,
                                            sparse_init=1 if args.self_seed else 0, plane_neighbours=plane_neighbours)
        else:
            self.params = pm.default_params(args.semantics, patch=args.patch, patchmatch_iters=args.iters,
                                            engine=args.engine, sparse_init=1 if args.self_seed else 0)
        self.eng = pm.Engine(self.params, device=local_rank, max_rows=args.rows, max_cols=args.cols, max_batch=nb)
        if enhance:
            # BGR inputs; the stereo-ready enhancement (imaging::Normalize(NormalizeColorIlluminant(.)) -> gray) runs on the
            # device, on the engine's stream, folded into every Match's load path (pm_match_bgr_device)
            self.BL = [torch.from_numpy(np.stack([synth.to_bgr(p["left"], 1) for p in g])).to(dev).contiguous() for g in groups]
            self.BR = [torch.from_numpy(np.stack([synth.to_bgr(p["right"], 2) for p in g])).to(dev).contiguous() for g in groups]

    def step(self, s):
        a, e, g = self.args, self.eng, s % N_ROTATE
        self.last_g = g
        seeded = self.mode == "scalar" and not a.self_seed
        if self.enhance:
            # BGR inputs: pm_match_bgr_device -- per image the two Gaussian passes and two small min / max passes, the
            # per-pixel tail of the enhancement inside the prep kernel (no gray image in memory), then the Match
            e.match_bgr_device(self.nb, self.BL[g].data_ptr(), self.BR[g].data_ptr(), a.rows, a.cols, None, None,
                               self.DL.data_ptr(), self.DR.data_ptr())
            return
        left, right = self.L[g].data_ptr(), self.R[g].data_ptr()
        e.match_device(self.nb, left, right, a.rows, a.cols, self.SL[g].data_ptr() if seeded else None,
                       self.SR[g].data_ptr() if seeded else None, self.DL.data_ptr(), self.DR.data_ptr())

    def quality(self):
        """Slot 0 of the group the LAST step matched, against that pair's own synthetic truth."""
        self.eng.synchronize()
        d = self.DL[0]
        ok = d > 0
        fg = float(ok.float().mean().item())
        err = (d - self.gt[self.last_g]).abs()
        return {"foreground_fraction": fg, "pair": "slot 0 of group %d (the last step's)" % self.last_g,
                "foreground_within_1px_of_truth": float((err[ok] < 1.0).float().mean().item()) if fg > 0 else 0.0}

    def maps0(self):
        """Host copies of slot 0's maps (call after step(0): pair 0 of this rank)."""
        self.eng.synchronize()
        return self.DL[0].cpu().numpy(), self.DR[0].cpu().numpy()


def roofline_of(args, prof, n_prof, nb, mode, state, ms_per_step=None, variant=""):
    px_views = args.rows * args.cols * 2 * nb
    if mode == "planes":
        dom = max(PLANE_LAUNCH_BYTES, key=lambda k: prof.get(k, (0, 0.0))[1])
        n_launch, total_ms = prof[dom]
        avg_ms = total_ms / max(n_launch, 1)
        scale = 0.5 if state == "f16" else 1.0  # "with fp16 state replace 16 -> 8 B" 
        bytes_per_launch = PLANE_LAUNCH_BYTES[dom] * scale * args.rows * args.cols * nb * PLANE_LAUNCH_VIEWS[dom]
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if n_launch else 0.0
        return {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic(dom + variant) if (nb == 1 and state == "f32") else None,
                "traffic_source": "profiles/traffic.json: 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction, profiles/r06_calib_fetch_write.txt); a LOWER bound for these kernels: their reference fill loads 2 bytes per lane, which FETCH_SIZE does not count at all"
                                  if (nb == 1 and state == "f32") else None,
                "valu": pmc_valu(dom, prof, n_prof, variant) if (nb == 1 and state == "f32") else None,
                "binding": binding_roof(dom, prof, n_prof, ms_per_step, variant) if (nb == 1 and state == "f32") else None,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "avg_launch_ms": avg_ms, "launches": n_launch, "profiled_steps": n_prof,
                "formula": "N * (74 + 320 * I) B per pair, plane state; stage bytes per px and view: spatial 48 (two "
                           "launches of 24: red, black), view 64, refine 48, fused view + refine launch 112 (halved "
                           "for fp16 state)",
                "note": "the window cost makes these kernels VALU-bound (about 100 vector instructions per window row "
                        "and candidate, DESIGN.md): the HBM fraction is the figure BASELINE.json asks for, not the "
                        "binding roof"}
    dom = max(("sweep_row", "sweep_col"), key=lambda k: prof[k][1])
    n_launch, total_ms = prof[dom]
    avg_ms = total_ms / max(n_launch, 1)
    # a class runs 2 sweeps per iteration over every pixel of both views; with the views on their own streams
    # (default) a launch covers one view, and two launches run concurrently on the chip
    launches_per_step = n_launch / max(n_prof, 1)
    concurrency = max(1, round(launches_per_step / (2 * args.iters)))
    bytes_per_launch = SWEEP_BYTES_PER_PX * px_views / concurrency
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if n_launch else 0.0
    return {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(dom) if nb == 1 else None,
            "traffic_source": "profiles/traffic.json (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                              "command, not measured in this run): 2 x FETCH_SIZE + WRITE_SIZE -- gfx950's FETCH_SIZE reports "
                              "half of the bytes read (MI355X_MICROARCH.md; profiles/r06_calib_fetch_write.txt), WRITE_SIZE whole "
                              "32-byte sectors of the sparse write-back (an upper bound); the L2's memory side, Infinity-Cache "
                              "hits included" if nb == 1 else None,
            "valu": pmc_valu(dom, prof, n_prof) if nb == 1 else None,
            "binding": binding_roof(dom, prof, n_prof, ms_per_step) if nb == 1 else None,
            "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": avg_ms,
            "launches": n_launch, "profiled_steps": n_prof, "concurrent_launches": concurrency,
            "achieved_all_concurrent": achieved * concurrency,
            "formula": "N * (78 + 216 * I) B per pair, scalar state; 24 B per px and view and sweep",
            "note": "per-launch figure as specified, HIP events on every --profile-every-th timed step "
                    "(`profiled_steps` of them); `concurrent_launches` launches of this class (one per view, "
                    "own streams) share the chip, so the chip-level rate is achieved_all_concurrent"}


def timed_loop(w, d, steps, warmup, every, no_profile):
    """W untimed steps, then exactly `steps` timed ones between barrier + synchronize on both sides (wall clock: the
    contract's figure).  Beside it every step boundary is a HIP event recorded on the ENGINE's stream (torch's own
    events only see torch's current stream): per-step durations without any host synchronisation, for the median."""
    torch = w.torch
    eng = w.eng
    d.barrier()  # (the first collective builds the communicator -- hundreds of ms with the GPU idle: before the warm-up,
                 # not between it and the timed steps)
    for s in range(warmup):
        w.step(s)
    eng.synchronize()
    eng.profile_read()
    n_prof = 0
    es = torch.cuda.ExternalStream(eng.stream())
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    torch.cuda.synchronize()
    d.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record(es)
    for s in range(steps):
        timed = (not no_profile) and s % every == 0
        eng.profile_enable(timed)
        n_prof += 1 if timed else 0
        w.step(s)
        marks[s + 1].record(es)
    eng.synchronize()
    torch.cuda.synchronize()
    elapsed_local = time.perf_counter() - t0
    d.barrier()
    torch.cuda.synchronize()
    elapsed = d.max_over_ranks(elapsed_local)
    prof = eng.profile_read()
    eng.profile_enable(False)
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    stats = {"median": step_ms[steps // 2] if steps % 2 else 0.5 * (step_ms[steps // 2 - 1] + step_ms[steps // 2]),
             "min": step_ms[0], "max": step_ms[-1], "mean": sum(step_ms) / steps,
             "clock": "HIP events on the engine's stream at every step boundary (this rank)"} if steps else None
    return elapsed, prof, n_prof, stats


def side_leg(args, pm, torch, np, synth, dev, d, state, enhance, plane_neighbours=0, nb=1):
    """One plane-mode measurement for the default run's JSON line (rank 0, N=1).  plane_neighbours = 1: the spatial stage's
    two-neighbour option (pm_params.plane_neighbours = PM_PL_NEIGH_TWO), an option, not the default.  nb > 1: batches of nb
    pairs per call (BASELINE configs[2] / [4] are batches: the engine advances the two halves of a batch on two streams)."""
    import argparse
    args = argparse.Namespace(**vars(args))
    args.pairs_per_gpu = nb
    class NoDist:
        def barrier(self):
            pass

        def max_over_ranks(self, v):
            return v
    w = Workload(args, pm, torch, np, synth, dev, d.local_rank, d.rank, "planes", state, enhance, plane_neighbours)
    steps = 16 if nb == 1 else max(4, 32 // nb)
    # (six warm-up steps: the engine's creation leaves the GPU idle for a few hundred ms and its clocks take ~20 ms of
    # work to come back; with two warm-up steps the leg read 3 % low.  Every 8th step carries per-kernel events, as in
    # the headline loop: a profiled step is ~3 % slower -- 33 launches bracketed by events -- and with every 4th of 12
    # steps profiled the same build read 460 pairs/s where it now reads 470)
    elapsed, prof, n_prof, step_stats = timed_loop(w, NoDist(), steps, 6, 8, False)
    w.step(0)
    w.eng.synchronize()
    variant = "@two_neighbours" if plane_neighbours else ""
    out = {"_maps": None if enhance else w.maps0(), "workload": f"PM_MODE_PLANES, {args.cols}x{args.rows}, {args.iters} iterations, {args.patch}x{args.patch}, "
                       f"{state} plane/cost state" + (", stereo-ready enhancement of both BGR images fused into the "
                                                      "Match's load path (pm_match_bgr_device; BASELINE configs[4] per-GPU shape)" if enhance
                                                      else " (BASELINE configs[1] shape)") +
                       ("; spatial stage with TWO neighbours per pass (PM_PL_NEIGH_TWO: an option, not the default)" if plane_neighbours else "") +
                       (f"; batches of {nb} pairs per call, the halves of a batch on two streams" if nb > 1 else ""),
           "value": w.nb * steps / elapsed, "unit": "pairs/s", "ms_per_frame": 1e3 * elapsed / steps / w.nb, "steps": steps,
           "step_ms": step_stats,
           "dtype": "u8 window cost, " + state + " state",
           "roofline": roofline_of(args, prof, n_prof, w.nb, "planes", state, 1e3 * elapsed / steps, variant),
           "kernels_ms_per_step": {k: v[1] / max(n_prof, 1) for k, v in prof.items() if v[0]},
           "check": w.quality()}
    if nb == 1 and not enhance and not plane_neighbours:
        # the frame sequence in plane mode (pm_submit_device / pm_collect, the reference's Sequence caller): one pair per
        # call like the leg above, but consecutive frames may share the chip
        seq = sequence_device_leg(pm, torch, w, args, d.local_rank, depth=4, frames=64)
        seq["equals_this_leg's_maps"] = seq.pop("equals_the_headline_maps")
        out["sequence_device"] = seq
    w.eng.close()
    return out


def run_tiled(args, d):
    """BASELINE configs[3]: one 4096x2160 pair row-tiled over the ranks; delegates to python/tiled.py."""
    import tiled
    tiled.bench(args, d)


def tiled_children(args):
    """The configs[3] leg of a multi-rank run, crash-isolated: a CHILD process started before this process touches the
    GPU or joins its process group, waited for with a deadline; whatever it does -- exception, hang, abort -- ends in
    an {"error": ...} entry and never costs the headline.  Default: rank 0 alone starts ONE child that drives all
    WORLD_SIZE GPUs through the C-ABI driver (pm_tiled_*: one process, one band per GPU; the default configuration --
    pipelined schedule, boundary rows by hipMemcpyPeerAsync + events -- first, then the speculative schedule and the
    direct exchange as variants of the same child, a JSON line each) while the other ranks wait at the rendezvous.  --tiled-rccl: every rank starts a
    `python/tiled.py` rank of its own (RCCL neighbour exchange on the engine's stream, own rendezvous port)."""
    import subprocess
    env = dict(os.environ)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    script = os.path.join(ROOT, "ocean-perception_amd", "python", "tiled.py")
    base = [sys.executable, script, "--rows", "2160", "--cols", "4096", "--iters", str(args.iters), "--patch",
            str(args.patch), "--steps", "2"]
    if args.tiled_rccl or args.dry_run:
        env["MASTER_ADDR"] = env.get("MASTER_ADDR", "127.0.0.1")
        env["MASTER_PORT"] = str((int(env.get("MASTER_PORT", "29500")) - 1024 + 4099) % 60000 + 1024)
        for k in ("TORCHELASTIC_RUN_ID", "TORCHELASTIC_USE_AGENT_STORE", "TORCH_NCCL_ASYNC_ERROR_HANDLING"):
            env.pop(k, None)  # the child rendezvous is a plain env:// TCP store of its own
        cmd = base + ["--backend", args.backend]
        single = False
    else:
        single = True
        if rank != 0:
            return None
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID",
                  "TORCHELASTIC_USE_AGENT_STORE"):
            env.pop(k, None)
        cmd = base + ["--single-process", str(world)]
    try:
        p = subprocess.Popen(cmd + (["--variants", "speculative,direct"] if single else []), env=env,
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    except OSError as e:
        return {"error": "could not start the tiled leg: %r" % (e,)}
    timed_out = False
    try:
        out, err = p.communicate(timeout=args.tiled_timeout)
    except subprocess.TimeoutExpired:
        p.kill()
        out, err = p.communicate()
        timed_out = True
    if rank != 0:
        return None
    # One JSON line per variant, each printed as soon as it was measured: the default configuration first (pipelined
    # schedule, peer copies across devices), then -- single-process driver only -- `speculative` (all bands sweep at once and
    # re-sweep what changed) and `direct` (the receiving band's kernel reads the boundary row across the link; only where
    # peer access is enabled on every boundary; last, because it is the one thing that has never run on real devices).  A variant that hangs or kills the child costs itself, not the lines
    # printed before it.
    lines = []
    for line in (out or "").strip().splitlines():
        if line.startswith("{"):
            try:
                lines.append(json.loads(line))
            except ValueError:
                pass
    if not lines:
        if timed_out:
            return {"error": "tiled leg did not finish within %d s (killed)" % args.tiled_timeout}
        return {"error": "tiled leg exited with code %d" % p.returncode, "stderr_tail": (err or "")[-400:]}
    res = lines[0]
    for extra in lines[1:]:
        res[{"direct": "direct_exchange", "speculative": "speculative_schedule"}.get(extra.get("variant"), str(extra.get("variant")))] = extra
    if single and (timed_out or p.returncode != 0):
        res["variants_note"] = ("the child %s after %d of its variants" %
                                ("was killed at the deadline" if timed_out else "exited with code %d" % p.returncode, len(lines)))
    return res


def device_count():
    """GPUs visible to this job, read in a CHILD process: the parent must not touch the GPU before it starts its
    ranks (a forked / spawned rank of a process that has initialised HIP is not safe)."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                           capture_output=True, text=True, timeout=300)
        return int(r.stdout.strip().splitlines()[-1])
    except Exception:  # noqa: BLE001
        return 0


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one per GPU, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment, rendezvous on 127.0.0.1), wait for all of them, relay rank 0's JSON
    line and fail if any rank failed.  The parent never imports torch or touches a GPU."""
    import subprocess
    want = args.gpus
    n = want
    note = None
    if not args.dry_run:
        have = device_count()
        if have < 1:
            print(json.dumps({"error": "no GPU visible", "n_gpus_requested": want}), flush=True)
            return 3
        if have < want:
            n, note = have, f"--gpus {want} requested, {have} GPU(s) visible: ran {have}; the {want}-GPU point is UNRUN"
    env = dict(os.environ)
    env.update(WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
               PM_BENCH_GPUS_REQUESTED=str(want))
    if note:
        env["PM_BENCH_LAUNCH_NOTE"] = note
    import tempfile
    procs = []
    with tempfile.TemporaryFile("w+") as out0:
        for r in range(n):
            e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                          stdout=out0 if r == 0 else subprocess.DEVNULL))
        # a rank that dies leaves the others waiting at a barrier for ever: end them
        codes = [None] * n
        while any(c is None for c in codes):
            for r, p in enumerate(procs):
                if codes[r] is None:
                    codes[r] = p.poll()
            if any(c not in (None, 0) for c in codes):
                for r, p in enumerate(procs):
                    if codes[r] is None:
                        p.kill()
                        codes[r] = p.wait()
            time.sleep(0.2)
        out0.seek(0)
        for line in out0.read().splitlines():  # the JSON line to stdout, library chatter (e.g. "[Gloo] Rank 0 ...") to stderr
            print(line, file=sys.stdout if line.startswith("{") else sys.stderr)
        sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print("bench.py: rank(s) failed (rank, exit code): %s" % bad, file=sys.stderr)
        return 1
    return 0


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    planes = args.mode == "planes"
    tiled_result = None
    if (int(os.environ.get("WORLD_SIZE", "1")) > 1 and not planes and not args.no_side_legs and not args.tiled
            and max(1, args.pairs_per_gpu) == 1 and (not args.dry_run or args.rehearse_tiled_leg)):
        import torch  # noqa: F401 -- pages the libraries in before the children import them
        tiled_result = tiled_children(args)
    if os.environ.get("PM_BENCH_FAIL_RANK") == os.environ.get("RANK", "0"):  # fault injection of tests/test_dist.py
        sys.exit(7)
    d = Dist(args)
    steps, warmup = args.steps, args.warmup
    nb = max(1, args.pairs_per_gpu)
    sem_name = "PM_SEM_CPU" if args.semantics == 0 else "PM_SEM_GPU (5-tap)"
    if planes:
        what = (f"PM_MODE_PLANES (random plane init, red-black / view propagation, refinement), {args.state} state"
                + (", BGR inputs enhanced on the device (BASELINE.json configs[4] per-GPU shape)" if args.enhance else ""))
    else:
        what = f"fp32 cost, {sem_name}"
    result = {
        "metric": "stereo_pairs_per_sec_1280x720_patchmatch", "unit": "pairs/s", "n_gpus": d.world, "steps": steps,
        "warmup": warmup, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("u8 window cost, " + args.state + " state") if planes else "f32",
        "data": "synthetic",
        # the world size the timed barrier saw (torch.distributed; backend nccl = RCCL); 1 = no process group
        "rccl_ranks": d.ranks_seen(), "backend": args.backend if d.world > 1 else None,
        "config": {"workload": f"{nb} synthetic {args.cols}x{args.rows} stereo pair(s) per GPU and step, "
                               f"{args.iters} iterations, {args.patch}x{args.patch} window, {what}, left+right view + "
                               "cross-check" + (" (BASELINE.json configs[1])" if not planes and nb == 1 else ""),
                   "pairs_per_gpu_per_step": nb, "distinct_resident_pairs_rotated": N_ROTATE,
                   "sharding": "rank r owns the distinct pairs 4 r .. 4 r + 3 (a batch repeats them), no collective"},
    }
    if os.environ.get("PM_BENCH_LAUNCH_NOTE"):
        result["n_gpus_requested"] = int(os.environ.get("PM_BENCH_GPUS_REQUESTED", d.world))
        result["launch_note"] = os.environ["PM_BENCH_LAUNCH_NOTE"]
    if args.dry_run:
        # plumbing only: same barrier / reduction path, no engine, no numbers worth reading
        d.barrier()
        t0 = time.perf_counter()
        for _ in shard(d.rank, d.world, steps, nb):
            time.sleep(0.001 * (1 + d.rank))
        elapsed = d.max_over_ranks(time.perf_counter() - t0)
        d.barrier()
        host_seq = host_sequence_all_ranks(None, args, None, d) if (not planes and args.host_pairs > 0) else None
        if d.rank == 0:
            result.update(value=d.world * steps / elapsed, ms_per_step=1e3 * elapsed / steps, dry_run=True)
            if host_seq is not None:
                result["host_sequence_all_ranks"] = host_seq
            if tiled_result is not None:
                result["tiled_4096x2160"] = tiled_result
            if d.world > 1:
                no_cpu_leg(result, d.world)
            print(json.dumps(result), flush=True)
        d.close()
        return
    if args.tiled:
        run_tiled(args, d)
        d.close()
        return

    import numpy as np
    import torch
    import pm_ctypes as pm
    import synth

    torch.cuda.set_device(d.local_rank)
    dev = torch.device(f"cuda:{d.local_rank}")
    w = Workload(args, pm, torch, np, synth, dev, d.local_rank, d.rank, args.mode, args.state, args.enhance,
                 args.plane_neighbours if planes else 0)
    eng = w.eng
    elapsed, prof, n_prof, step_stats = timed_loop(w, d, steps, warmup, max(1, args.profile_every), args.no_profile)

    # determinism: the same resident pair twice
    w.step(0)
    eng.synchronize()
    ref = w.DL[0].clone()
    counters = None
    if not planes:
        # work counters: one extra untimed step (the counting atomics would distort the timed ones)
        eng.debug_counters_enable(True)
        eng.debug_counters()
        w.step(0)
        counters = eng.debug_counters()
        eng.debug_counters_enable(False)
    else:
        w.step(0)
    eng.synchronize()
    deterministic = bool(torch.equal(ref, w.DL[0]))
    check = w.quality()
    check["deterministic_across_steps"] = deterministic
    # pair 0's maps (rank 0: synth pair 0, the pair the CPU baselines run) for the whole-frame comparison below
    maps0 = w.maps0() if (d.rank == 0 and not args.enhance and not args.self_seed) else None

    # the PCIe-inclusive frame sequence on every rank (collective: barriers + max over ranks inside)
    host_seq = None
    if not planes and args.host_pairs > 0 and nb == 1:
        host_seq = host_sequence_all_ranks(pm, args, w.pairs[:N_ROTATE], d)
    if d.rank == 0:
        gpu_ms = sum(v[1] for v in prof.values())
        result.update(
            value=d.world * nb * steps / elapsed, ms_per_step=1e3 * elapsed / steps,
            ms_per_frame=1e3 * elapsed / steps / nb,
            step_ms=step_stats,
            roofline=roofline_of(args, prof, n_prof, nb, args.mode, args.state, 1e3 * elapsed / steps,
                                 "@two_neighbours" if planes and args.plane_neighbours else "") if n_prof else None,
            kernels_ms_per_step={k: v[1] / max(n_prof, 1) for k, v in prof.items() if v[0]},
            gpu_busy_ms_per_step=gpu_ms / max(n_prof, 1),
            check=check,
        )
        if counters is not None:
            result["run_engine_counters_per_step"] = counters
        if host_seq is not None:
            result["host_sequence_all_ranks"] = host_seq
        side = d.world == 1 and not planes and nb == 1 and not args.no_side_legs
        if d.world == 1 and not planes and args.host_pairs > 0:
            # every leg below runs in THIS process, beside the headline's handle (one stream policy: DESIGN.md 6)
            result["host_buffers"] = host_buffer_leg(pm, args, w.pairs[:N_ROTATE], d.local_rank)
            result["sequence_device"] = sequence_device_leg(pm, torch, w, args, d.local_rank)
        if side and not args.self_seed:
            result["batch"] = {"4": batch_leg(pm, torch, np, synth, args, dev, d.local_rank, 4, 8),
                               "32": batch_leg(pm, torch, np, synth, args, dev, d.local_rank, 32, 2)}
        if side and args.semantics == 0:
            try:
                result["reference_test_shape"] = reference_test_shape_leg(pm, args, d.local_rank)
            except Exception as e:  # noqa: BLE001 -- report, never lose the headline
                result["reference_test_shape"] = {"error": repr(e)}
    eng.close()
    del w
    if not planes and not args.no_side_legs and nb == 1:
        # BASELINE configs[3] beside the headline, never `value`: one 4096x2160 pair row-tiled over the ranks of this
        # run.  One rank: the untiled large image, in this process.  Several ranks: measured by tiled_children()
        # before this process touched the GPU.
        if d.world == 1:
            import tiled
            try:
                tiled_result = tiled.bench(args, d, steps=2, quiet=True)
                # the C-ABI driver with 8 bands on THIS device: the protocol (boundary rows, masked re-sweeps, flag) at
                # work, its cost beside the untiled frame; a real caller has one band per GPU
                tiled_result["eight_bands_on_this_device"] = tiled.bench_single_process(args, [d.local_rank] * 8, steps=2)
                tiled_result["eight_bands_on_this_device_speculative"] = tiled.bench_single_process(
                    args, [d.local_rank] * 8, steps=2, schedule=0)
                # four bands: the fastest way through one device for an image of this size (bands in different phases
                # fill each other's launch tails; more bands add hops: profiles/r06_tiled_bands_one_device.txt)
                tiled_result["four_bands_on_this_device"] = tiled.bench_single_process(args, [d.local_rank] * 4, steps=2)
            except Exception as e:  # noqa: BLE001 -- report, never lose the headline
                tiled_result = {"error": repr(e)}
        if d.rank == 0 and tiled_result is not None:
            result["tiled_4096x2160"] = tiled_result
    if d.rank == 0:
        if d.world == 1 and not planes and not args.no_side_legs and nb == 1:
            result["planes"] = {"f32": side_leg(args, pm, torch, np, synth, dev, d, "f32", False),
                                "f16_enhanced": side_leg(args, pm, torch, np, synth, dev, d, "f16", True),
                                "f32_two_neighbours": side_leg(args, pm, torch, np, synth, dev, d, "f32", False, 1),
                                "f32_batch4": side_leg(args, pm, torch, np, synth, dev, d, "f32", False, nb=4),
                                "f16_enhanced_batch4": side_leg(args, pm, torch, np, synth, dev, d, "f16", True, nb=4)}
            pl = result["planes"]
            if pl["f32"].get("_maps") is not None and pl["f32_batch4"].get("_maps") is not None:
                pl["f32_batch4"]["check"]["slot_0_equals_the_single_pair_leg"] = bool(
                    np.array_equal(pl["f32"]["_maps"][0], pl["f32_batch4"]["_maps"][0]) and
                    np.array_equal(pl["f32"]["_maps"][1], pl["f32_batch4"]["_maps"][1]))
        plane_maps = {k: v.pop("_maps", None) for k, v in result.get("planes", {}).items()}
        if d.world == 1 and not args.no_cpu_baseline:
            # the CPU baselines run the oracle on the WHOLE frame of pair 0 anyway: their maps are the checker of the
            # engine's maps of the same pair (tolerance 0) -- the whole-image recipe of patchmatch_test.cpp:149-183
            if planes:
                cb = cpu_baseline_planes(args, args.state == "f16", args.plane_neighbours)
                if maps0 is not None:
                    result["check"]["equals_oracle_full_frame"] = equals_oracle(
                        maps0, cb["_maps"], "oracle/pm_planes_oracle.c (this mode's CPU definition), whole frame")
            else:
                cb = cpu_baseline(args)
                if maps0 is not None and args.semantics == 0:
                    result["check"]["equals_oracle_full_frame"] = equals_oracle(
                        maps0, cb["_maps"], "oracle/pm_oracle.c (PM_SEM_CPU, literal getRectSubPix + functor form, "
                                            "%d threads), whole frame" % cb["all_cores"]["cores"])
            cb.pop("_maps", None)
            result["cpu_baseline"] = cb
            if not planes and "planes" in result:
                for leg, neigh in (("f32", 0), ("f32_two_neighbours", 1)):
                    pb = cpu_baseline_planes(args, False, neigh)
                    if plane_maps.get(leg) is not None:
                        result["planes"][leg]["check"]["equals_oracle_full_frame"] = equals_oracle(
                            plane_maps[leg], pb["_maps"], "oracle/pm_planes_oracle.c (this mode's CPU definition), whole frame")
                    pb.pop("_maps", None)
                    result["planes"]["cpu_baseline" if neigh == 0 else "cpu_baseline_two_neighbours"] = pb
        elif d.world > 1:
            no_cpu_leg(result, d.world)
        print(json.dumps(result), flush=True)
    d.close()


if __name__ == "__main__":
    main()
