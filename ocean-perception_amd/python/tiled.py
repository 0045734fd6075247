"""One large stereo pair row-tiled over several GPUs (BASELINE.json configs[3]: 4096x2160 over 8 MI355X).

Every rank owns a horizontal band of the image.  Rectified stereo only looks along rows, so the
noise / cost stage, the two horizontal sweeps, the background mask and the cross-check are local to
a band; the band carries patch_h/2 + 1 halo rows of IMAGE data so windows and Sobel gradients of its
own rows see their true neighbours.  Only the two vertical sweeps of an iteration cross bands: along a
column, the first row of a band continues from the value the last row of the band above ended on
(src/vehicle/stereo_matching/patchmatch.cpp:276-285, :301-310 read in pass order).

Exact semantics without serialising the GPUs: speculate and fix up at band granularity (the scheme
the kernels use inside a chain, pm_run.hpp).  Each rank sweeps with the neighbour's OLD boundary row,
the new boundary rows travel one hop (one row of disparities per view: 16 KB at 4096 columns), and a rank
re-sweeps exactly the COLUMNS whose incoming value changed (columns of a vertical sweep are independent
chains) from the snapshot it took before the sweep.  Band k is final after round k+1; the fixpoint is the
untiled sweep.

Everything stays on the device timeline: the exchange is RCCL send/recv (torch.distributed P2P over xGMI)
enqueued on the engine's own stream, "which columns changed" is a device mask handed to
pm_tile_restore_cols / pm_tile_sweep_masked, and the number of rounds per sweep is FIXED (2 by default; a round
in which nothing changed costs three empty launches), so no rank ever waits on the host inside a Match().
Whether the fixed rounds sufficed is one device flag per rank -- "my boundary row still changed after the
last round" -- reduced over the ranks ONCE, after the Match; only if it is set (a value crossed more than
`rounds` band boundaries in one sweep) is the Match repeated with world - 1 rounds, which always suffices.
`LocalComm` runs the same protocol with the ranks as threads of one process (tests on a single GPU).

Two schedules, same maps (pm_tiled_schedule of include/pm/patchmatch.h; `pipelined=` of match_band): the one above is the
SPECULATIVE one.  The default since round 6 is PIPELINED: in a vertical sweep the ranks take turns along the sweep
direction -- a rank receives its predecessor's FINAL boundary row, stores it in front of its chains, sweeps once and sends
its own last row on.  Nothing is guessed, so there is no snapshot, no mask, no re-sweep, no flag and no repeat; everything
that does not cross a boundary (noise / cost, horizontal sweeps) still overlaps between the ranks.  Measured with eight
bands on one device through the C driver: 28.5 ms per 4096x2160 frame against 35.7 speculative and 29.5 untiled.
"""
import threading

import numpy as np
import torch

import pm_ctypes as pm

DEFAULT_ROUNDS = 1 << 20  # clamped to world - 1: always exact without a repeat (measured: tools/tiled_rounds.py)


def band_of(rank, world, global_rows, halo):
    """(own_row0, own_rows, band_row0, band_rows): rows split as evenly as possible, halo clipped to the image."""
    base, rem = divmod(global_rows, world)
    own_row0 = rank * base + min(rank, rem)
    own_rows = base + (1 if rank < rem else 0)
    band_row0 = max(0, own_row0 - halo)
    band_end = min(global_rows, own_row0 + own_rows + halo)
    return own_row0, own_rows, band_row0, band_end - band_row0


def halo_rows(params):
    if params.semantics == pm.PM_SEM_CPU:
        ph = max([params.patch_h[i] for i in range(params.patchmatch_iters)] + [params.bg_patch_h])
        return ph // 2 + 1
    return 2


class DistComm:
    """Neighbour exchange over torch.distributed (backend "nccl" = RCCL on ROCm), ordered on the CURRENT torch
    stream -- the caller makes the engine's stream current -- so no host synchronisation is involved."""

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.exchanges = 0
        self.timed = False   # bench: bracket every exchange with events on the current (= the engine's) stream
        self._events = []

    def shift(self, row, down, send=True, recv=True):
        """Send `row` to the next rank in sweep direction, return the row of the previous one (or None).  send / recv =
        False: this rank's side of the exchange is known to be redundant in this round (match_band) -- both ends of a
        transfer derive that from their positions, so every send still meets its receive."""
        dist = self.dist
        dst = self.rank + 1 if down else self.rank - 1
        src = self.rank - 1 if down else self.rank + 1
        ops, recv_buf = [], None
        want_recv = recv
        recv = None
        if send and 0 <= dst < self.world:
            ops.append(dist.P2POp(dist.isend, row, dst))
        if want_recv and 0 <= src < self.world:
            recv = torch.empty_like(row)
            ops.append(dist.P2POp(dist.irecv, recv, src))
        if ops:
            ev = None
            if self.timed and row.is_cuda:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record(torch.cuda.current_stream())
            for req in dist.batch_isend_irecv(ops):
                req.wait()  # NCCL backend: makes the current stream wait for the transfer, the host does not block
            if ev is not None:
                ev[1].record(torch.cuda.current_stream())
                self._events.append(ev)
            self.exchanges += 1
        return recv

    def send(self, row, dst, down=True):
        """One boundary row to rank `dst`, enqueued on the current stream (the pipelined schedule's hand-over)."""
        self._p2p(self.dist.P2POp(self.dist.isend, row, dst), row)

    def recv(self, like_shape, device, src, down=True):
        got = torch.empty(like_shape, dtype=torch.float32, device=device)
        self._p2p(self.dist.P2POp(self.dist.irecv, got, src), got)
        return got

    def _p2p(self, op, t):
        ev = None
        if self.timed and t.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(torch.cuda.current_stream())
        for req in self.dist.batch_isend_irecv([op]):
            req.wait()  # NCCL backend: a stream dependency, the host does not block
        if ev is not None:
            ev[1].record(torch.cuda.current_stream())
            self._events.append(ev)
        self.exchanges += 1

    def exchange_ms(self):
        """Stream time between the start and the end of every timed exchange since the last call (the transfer
        itself plus waiting for the neighbour to reach its side of it).  Call after synchronising the stream."""
        ms = sum(a.elapsed_time(b) for a, b in self._events)
        self._events = []
        return ms

    def any(self, flag_tensor):
        """Logical OR of a one-element device flag over the ranks -> python bool (the one host read per Match)."""
        t = flag_tensor.clone()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return bool(t.item())


class LocalComm:
    """The same protocol between threads of one process (one thread, one engine handle, one stream per band).
    The hand-over between two bands' streams is an event: the receiver's stream waits for the sender's row copy."""

    class Shared:
        def __init__(self, world):
            self.world = world
            self.slots = [[None, None] for _ in range(world)]  # [rank][0 = row for rank+1, 1 = row for rank-1]
            self.done = [[None, None] for _ in range(world)]   # [rank][dir]: the reader has copied that row (event)
            self.flags = [False] * world
            self.barrier = threading.Barrier(world)
            import queue
            self.q = [[queue.Queue(), queue.Queue()] for _ in range(world)]  # [sender][0 down, 1 up]: (row, event)
            self.kept = [[None, None] for _ in range(world)]

    def __init__(self, shared, rank):
        self.s, self.rank, self.world = shared, rank, shared.world
        self.exchanges = 0

    def shift(self, row, down, send=True, recv=True):
        s = self.s
        d = 0 if down else 1
        ev = None
        if row.is_cuda and send:
            # the row handed over last time stays referenced by the slot until now; its reader recorded an event after
            # copying it, and this stream waits for that event before the slot lets go of the tensor -- so whatever
            # this stream's allocator does with the block next is ordered behind the reader's copy.  (record_stream()
            # would say the same, but it crashes on a torch.cuda.ExternalStream in this torch build.)
            if s.done[self.rank][d] is not None:
                torch.cuda.current_stream().wait_event(s.done[self.rank][d])
                s.done[self.rank][d] = None
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
        if send:
            s.slots[self.rank][d] = (row, ev)
        s.barrier.wait()
        src = self.rank - 1 if down else self.rank + 1
        got = None
        if recv and 0 <= src < self.world:
            src_row, src_ev = s.slots[src][d]
            if src_ev is not None:
                torch.cuda.current_stream().wait_event(src_ev)
            got = src_row.clone()
            if src_ev is not None:
                done = torch.cuda.Event()
                done.record(torch.cuda.current_stream())
                s.done[src][d] = done
            self.exchanges += 1
        s.barrier.wait()
        return got

    def send(self, row, dst, down=True):
        """Pipelined hand-over: the row and an event behind its producer go into the sender's queue; the tensor stays
        referenced here until the next send in this direction, which first waits for the reader's "copied" event."""
        s, d = self.s, 0 if down else 1
        ev = None
        if row.is_cuda:
            if s.done[self.rank][d] is not None:
                torch.cuda.current_stream().wait_event(s.done[self.rank][d])
                s.done[self.rank][d] = None
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
        s.kept[self.rank][d] = row
        s.q[self.rank][d].put((row, ev))

    def recv(self, like_shape, device, src, down=True):
        s, d = self.s, 0 if down else 1
        row, ev = s.q[src][d].get(timeout=120)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
        got = row.clone()
        if ev is not None:
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream())
            s.done[src][d] = done
        self.exchanges += 1
        return got

    def any(self, flag_tensor):
        s = self.s
        s.flags[self.rank] = bool(flag_tensor.item())
        s.barrier.wait()
        res = any(s.flags)
        s.barrier.wait()
        return res


def match_band(engine, comm, params, left_band, right_band, seed_l_band, seed_r_band, global_rows, own_row0, own_rows,
               band_row0, rounds=DEFAULT_ROUNDS, pipelined=True):
    """Enqueues this rank's part of Match() for device tensors of its band on the engine's stream and returns
    (disp_l, disp_r, flag): the maps of the owned rows and a one-element device tensor that is non-zero if this rank's
    boundary row still changed after the last exchange round of some sweep (then `rounds` was too small).
    Nothing in here waits on the host."""
    band_rows, cols = left_band.shape
    n_views = 2 if params.left_right_check else 1
    dev = left_band.device
    tile = pm.PmTile(global_rows, band_row0, own_row0, own_rows)
    ptr = lambda t: t.data_ptr() if t is not None else None
    ext = torch.cuda.ExternalStream(engine.stream(), device=dev)
    ext.wait_stream(torch.cuda.current_stream(dev))  # the inputs were produced on the caller's stream
    with torch.cuda.stream(ext):
        engine.tile_begin(tile, ptr(left_band), ptr(right_band), band_rows, cols, ptr(seed_l_band), ptr(seed_r_band))
        own_end = own_row0 + own_rows
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        mask = torch.zeros((n_views, cols), dtype=torch.int32, device=dev)

        def get_row(r):
            t = torch.empty((n_views, cols), dtype=torch.float32, device=dev)
            engine.tile_get_row(r, t.data_ptr())
            return t

        for it in range(params.patchmatch_iters):
            engine.tile_noise(it)
            for k in range(4):
                if k in (0, 2):  # horizontal sweeps never leave the band
                    engine.tile_sweep(it, k)
                    continue
                down = k == 1
                out_row = own_end - 1 if down else own_row0
                pred_row = own_row0 - 1 if down else own_end
                if pipelined:
                    # the ranks take turns along the sweep direction: my predecessor's final row in front of my chains,
                    # one sweep, my last row on to my successor
                    pos = comm.rank if down else comm.world - 1 - comm.rank
                    pred = comm.rank - 1 if down else comm.rank + 1
                    succ = comm.rank + 1 if down else comm.rank - 1
                    if pos > 0:
                        row = comm.recv((n_views, cols), dev, pred, down)
                        engine.tile_set_row(pred_row, row.data_ptr())
                    engine.tile_sweep(it, k)
                    if pos < comm.world - 1:
                        comm.send(get_row(out_row), succ, down)
                    continue
                sent = get_row(out_row)
                used = comm.shift(sent, down)  # the neighbour's value before the sweep: the guess
                # the guess into the planes and the snapshot in one launch
                engine.tile_presweep(pred_row, used.data_ptr() if used is not None else None)
                engine.tile_sweep(it, k)
                # The band at position pos of the sweep direction (0 = the band without a predecessor) is final after
                # round pos - 1: from round pos on it would receive the row it already has.  It skips those rounds and
                # its predecessor does not send: world (world - 1) / 2 band-rounds instead of (world - 1)^2.
                pos = comm.rank if down else comm.world - 1 - comm.rank
                for r in range(rounds):
                    recv = pos > r            # my predecessor's row can still have changed
                    send = pos + 1 > r and 0 <= (comm.rank + 1 if down else comm.rank - 1) < comm.world
                    if send:
                        sent = get_row(out_row)
                    new_in = comm.shift(sent, down, send=send, recv=recv)
                    if new_in is not None:
                        # columns whose incoming value changed: flagged, put back to the snapshot, the incoming row
                        # stored -- one launch -- then the masked sweep (pm_tile_exchange_round)
                        engine.tile_exchange_round(it, k, pred_row, new_in.data_ptr(), used.data_ptr(), new_in.data_ptr(), mask.data_ptr())
                        used = new_in
                # did my boundary row move after the last row I sent?  then my successor is stale.  Only a rank
                # that HAS a successor in this sweep's direction asks: the last band's row is an image border
                # nobody consumes, and a change there must not make every rank repeat the Match.
                succ = comm.rank + 1 if down else comm.rank - 1
                if 0 <= succ < comm.world:
                    engine.tile_row_moved(out_row, sent.data_ptr(), flag.data_ptr())
        engine.tile_background()
        out_l = torch.empty((own_rows, cols), dtype=torch.float32, device=dev)
        out_r = torch.empty_like(out_l) if n_views > 1 else None
        engine.tile_finish(out_l.data_ptr(), ptr(out_r))
    torch.cuda.current_stream(dev).wait_stream(ext)
    return out_l, out_r, flag


def match_band_exact(engine, comm, params, *band_args, rounds=DEFAULT_ROUNDS, pipelined=True):
    """match_band + (speculative schedule only) the one convergence check per Match: (disp_l, disp_r, rounds_used, repeated)."""
    if pipelined:
        out_l, out_r, _ = match_band(engine, comm, params, *band_args, pipelined=True)
        return out_l, out_r, 0, False
    rounds = min(rounds, max(comm.world - 1, 0))
    out_l, out_r, flag = match_band(engine, comm, params, *band_args, rounds=rounds, pipelined=False)
    if comm.world > 1 and comm.any(flag):
        rounds = comm.world - 1  # band k is final after round k + 1: always enough
        out_l, out_r, flag = match_band(engine, comm, params, *band_args, rounds=rounds, pipelined=False)
        return out_l, out_r, rounds, True
    return out_l, out_r, rounds, False


def match_tiled_local(params, left, right, seed_l, seed_r, world, device=0, rounds=DEFAULT_ROUNDS, pipelined=False):
    """Single-process emulation: `world` bands on one GPU, one thread and one engine handle per band.
    numpy in, numpy out (whole image)."""
    rows, cols = left.shape
    halo = halo_rows(params)
    shared = LocalComm.Shared(world)
    dev = torch.device(f"cuda:{device}")
    torch.zeros(1, device=dev)  # initialise the device context in the main thread before the band threads start
    results = [None] * world
    errors = []

    def work(rank):
        try:
            own_row0, own_rows, band_row0, band_rows = band_of(rank, world, rows, halo)
            sl = slice(band_row0, band_row0 + band_rows)
            t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a[sl])).to(dev, dt).contiguous() if a is not None else None
            with pm.Engine(params, device=device, max_rows=band_rows, max_cols=cols) as eng:
                comm = LocalComm(shared, rank)
                ol, orr, used, repeated = match_band_exact(
                    eng, comm, params, t(left, torch.uint8), t(right, torch.uint8), t(seed_l, torch.float32),
                    t(seed_r, torch.float32), rows, own_row0, own_rows, band_row0, rounds=rounds, pipelined=pipelined)
                eng.synchronize()
                res = (ol.cpu(), orr.cpu() if orr is not None else None, (used, repeated, comm.exchanges))
            results[rank] = (own_row0, res)
        except Exception as e:  # keep the other threads from waiting forever
            errors.append(e)
            shared.barrier.abort()

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    if errors:
        raise errors[0]
    disp_l = np.zeros((rows, cols), np.float32)
    disp_r = np.zeros((rows, cols), np.float32) if params.left_right_check else None
    info = None
    for own_row0, (ol, orr, rr) in results:
        disp_l[own_row0:own_row0 + ol.shape[0]] = ol.numpy()
        if disp_r is not None:
            disp_r[own_row0:own_row0 + orr.shape[0]] = orr.numpy()
        info = rr
    return disp_l, disp_r, {"rounds": info[0], "repeated": info[1], "exchanges_per_rank": info[2]}


def bench(args, d, steps=None, rows=2160, cols=4096, quiet=False):
    """BASELINE configs[3]: one `cols` x `rows` pair row-tiled over the ranks of `d` (bench.py's Dist: one process per
    GPU).  Every rank builds the same seeded synthetic pair, matches its band `steps` times; rank 0 returns / prints
    one JSON object: ms/frame (max over ranks), exchange rounds per vertical sweep, exchanges, repeats."""
    import json
    import time

    import synth
    world, rank, local = d.world, d.rank, d.local_rank
    steps = steps if steps is not None else max(1, args.steps)
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    comm = DistComm() if world > 1 else LocalComm(LocalComm.Shared(1), 0)
    params = pm.default_params(pm.PM_SEM_CPU, patch=args.patch, patchmatch_iters=args.iters)
    pair = synth.make_pair(0, rows, cols, n_points=200 * (rows * cols) // (720 * 1280))
    own_row0, own_rows, band_row0, band_rows = band_of(rank, world, rows, halo_rows(params))
    sl = slice(band_row0, band_row0 + band_rows)
    t = lambda k, dt: torch.from_numpy(np.ascontiguousarray(pair[k][sl])).to(dev, dt).contiguous()
    L, R, SL, SR = t("left", torch.uint8), t("right", torch.uint8), t("seed_l", torch.float32), t("seed_r", torch.float32)
    times, ex_ms, used, repeats = [], [], 0, 0
    with pm.Engine(params, device=local, max_rows=band_rows, max_cols=cols) as eng:
        if isinstance(comm, DistComm):
            comm.timed = True
        for step in range(steps + 1):
            d.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out_l, out_r, used, repeated = match_band_exact(eng, comm, params, L, R, SL, SR, rows, own_row0, own_rows,
                                                            band_row0)
            eng.synchronize()
            torch.cuda.synchronize()
            ex = comm.exchange_ms() if isinstance(comm, DistComm) else 0.0
            if step > 0:
                times.append(time.perf_counter() - t0)
                ex_ms.append(ex)
                repeats += 1 if repeated else 0
        exchanges = comm.exchanges / float(steps + 1 + repeats)
    ms = d.max_over_ranks(1e3 * float(np.median(times)))
    ex_max = d.max_over_ranks(float(np.median(ex_ms)))
    res = None
    if rank == 0:
        gt = torch.from_numpy(pair["gt"][own_row0:own_row0 + own_rows]).to(dev)
        fg = out_l > 0
        res = {"workload": f"one {cols}x{rows} pair row-tiled over {world} GPU(s) (BASELINE.json configs[3]), "
                           f"{args.iters} iterations, {args.patch}x{args.patch}, PM_SEM_CPU", "n_gpus": world,
               "ms_per_frame": ms, "pairs_per_s": 1e3 / ms, "steps": steps,
               "schedule": "pipelined (the ranks sweep in order along the sweep direction)",
               "exchange_rounds_per_vertical_sweep": 1 + used, "boundary_exchanges_per_match_and_rank": exchanges,
               "matches_repeated_with_more_rounds": repeats, "host_syncs_inside_a_match": 0,
               # engine-stream time inside the neighbour exchanges of one Match (transfer + waiting for the neighbour),
               # max over ranks; 0 with one rank
               "exchange_ms_per_match": ex_max,
               "rank0_foreground_within_1px": float(((out_l - gt).abs()[fg] < 1).float().mean().item())}
        if not quiet:
            print(json.dumps(res), flush=True)
    return res


def bench_single_process(args, devices, steps=2, rows=2160, cols=4096, rounds=-1, exchange=0, schedule=1):
    """BASELINE configs[3] through the C-ABI driver (pm_tiled_* of include/pm/patchmatch.h): ONE process, band k on
    devices[k], boundary rows by hipMemcpyPeerAsync + events (no RCCL, no torch.distributed).  The pair is uploaded
    once and stays resident in the bands' HBM; a timed step is one pm_tiled_run (it returns after the one flag read
    per Match).  Returns the JSON object of the leg."""
    import time

    import synth
    params = pm.default_params(pm.PM_SEM_CPU, patch=args.patch, patchmatch_iters=args.iters)
    pair = synth.make_pair(0, rows, cols, n_points=200 * (rows * cols) // (720 * 1280))
    n = len(devices)
    with pm.TiledEngine(params, rows, cols, n, devices, exchange=exchange, schedule=schedule) as te:
        topology = te.topology()
        te.upload(pair["left"], pair["right"], pair["seed_l"], pair["seed_r"])
        te.run(rounds)  # untimed: first-use allocations of the band handles
        times, infos = [], []
        for _ in range(steps):
            t0 = time.perf_counter()
            infos.append(te.run(rounds))
            times.append(time.perf_counter() - t0)
        dl, _ = te.download()
    ms = 1e3 * float(np.median(times))
    fg = dl > 0
    return {"workload": f"one {cols}x{rows} pair row-tiled into {n} band(s) on device(s) {sorted(set(devices))} "
                        f"(BASELINE.json configs[3]), {args.iters} iterations, {args.patch}x{args.patch}, PM_SEM_CPU",
            "driver": "pm_tiled_* (C ABI, one process, hipMemcpyPeerAsync + events)", "n_gpus": len(set(devices)),
            "exchange": {0: "auto (in place on one device, peer copy across devices)", 1: "copy", 2: "direct (kernel reads across the link)"}[exchange],
            "schedule": {0: "speculative (all bands sweep at once, boundary rows travel one hop per round, changed columns are re-swept)",
                         1: "pipelined (the bands sweep in order along the sweep direction; nothing is re-swept)"}[schedule],
            "device_boundaries": topology[0], "peer_links": topology[1],
            "bands": n, "ms_per_frame": ms, "pairs_per_s": 1e3 / ms, "steps": steps,
            "exchange_rounds_per_vertical_sweep": 1 + infos[-1]["rounds"],
            "boundary_rows_moved_per_match": infos[-1]["exchanges"],
            "matches_repeated_with_more_rounds": sum(1 for i in infos if i["repeated"]),
            "host_syncs_inside_a_match": "one flag read per attempt",
            "foreground_within_1px": float((np.abs(dl - pair["gt"])[fg] < 1).mean()) if fg.any() else 0.0}


def main():
    """torchrun entry (same as `bench.py --tiled`): one rank per GPU.
        python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
            ocean-perception_amd/python/tiled.py --rows 2160 --cols 4096"""
    import argparse
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import bench as B
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2160)
    ap.add_argument("--cols", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--patch", type=int, default=11)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--dry-run", action="store_true")
    ap.add_argument("--single-process", type=int, default=0, metavar="N",
                    help="run the C-ABI driver (pm_tiled_*) in THIS process over devices 0..N-1 instead of one rank per GPU")
    ap.add_argument("--bands", type=int, default=0, help="--single-process: bands (default: one per device)")
    ap.add_argument("--exchange", type=int, default=0, help="--single-process: pm_tiled_exchange (0 auto, 1 copy, 2 direct)")
    ap.add_argument("--schedule", type=int, default=1, help="--single-process: pm_tiled_schedule (1 pipelined: the default; 0 speculative)")
    ap.add_argument("--variants", default="", help="--single-process: after the configured run, also (comma separated) "
                    "`direct` = PM_TILED_EXCHANGE_DIRECT where every boundary has peer access, `speculative` = "
                    "PM_TILED_SCHEDULE_SPECULATIVE; one JSON line each, printed as soon as it exists (a variant that "
                    "kills the process cannot take the lines before it along)")
    args = ap.parse_args()
    if args.single_process > 0:
        import json
        nb = args.bands if args.bands > 0 else args.single_process
        devices = [k * args.single_process // nb for k in range(nb)]
        first = bench_single_process(args, devices, steps=args.steps, rows=args.rows, cols=args.cols,
                                     exchange=args.exchange, schedule=args.schedule)
        first["variant"] = "default"
        print(json.dumps(first), flush=True)
        for v in [v for v in args.variants.split(",") if v]:
            if v == "direct" and not (first["peer_links"] == first["device_boundaries"] > 0):
                continue  # kernel reads of peer memory only where peer access is enabled on every boundary
            kw = {"direct": dict(exchange=2, schedule=args.schedule), "speculative": dict(exchange=args.exchange, schedule=0)}.get(v)
            if kw is None:
                continue
            res = bench_single_process(args, devices, steps=args.steps, rows=args.rows, cols=args.cols, **kw)
            res["variant"] = v
            print(json.dumps(res), flush=True)
        return
    d = B.Dist(args)
    bench(args, d, rows=args.rows, cols=args.cols)
    d.close()


if __name__ == "__main__":
    main()
