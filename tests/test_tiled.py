"""Row-tiled large-image mode (BASELINE configs[3]): the banded result must equal the untiled one bit for
bit.  Runs the real protocol (boundary-row exchange, snapshot / re-sweep until no incoming row changes) with
the ranks as threads of one process on one GPU; the band arithmetic is also checked on CPU."""
import os

import numpy as np
import pytest

from conftest import assert_same, small_pair


def test_band_partition_cpu():
    import importlib
    import sys
    # tiled.py imports torch lazily at module import; fine on CPU
    tiled = importlib.import_module("tiled")
    for rows, world, halo in ((2160, 8, 6), (100, 3, 2), (17, 4, 6)):
        covered = []
        for r in range(world):
            own0, own, band0, band = tiled.band_of(r, world, rows, halo)
            covered += list(range(own0, own0 + own))
            assert band0 == max(0, own0 - halo) and band0 + band == min(rows, own0 + own + halo)
        assert covered == list(range(rows))


@pytest.mark.gpu
@pytest.mark.parametrize("band_rows", [37, 40, 41, 43])  # the state planes keep four rows interleaved: every remainder
def test_boundary_row_get_and_set(pm, synth, band_rows):
    """pm_tile_get_row / pm_tile_set_row: image row r of every view as a tight [n_views][cols] buffer -- after the
    band's seeds went in, a row reads back as the seed maps' row (view 1 mirrored); a row written is the row read."""
    import torch
    cols = 150
    l, r, sl, sr, _ = small_pair(synth, 90, band_rows, cols, n_points=40, dilate_factor=3)
    dev = torch.device("cuda:0")
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    L, R, SL, SR = up(l), up(r), up(sl), up(sr)
    params = pm.default_params(0, patch=5, patchmatch_iters=1)
    row0 = 0  # the top band of a taller image: its owned rows and the halo below them
    tile = pm.PmTile(band_rows + 100, 0, 0, band_rows - 5 // 2 - 1)
    with pm.Engine(params, max_rows=band_rows, max_cols=cols) as e:
        e.tile_begin(tile, L.data_ptr(), R.data_ptr(), band_rows, cols, SL.data_ptr(), SR.data_ptr())
        buf = torch.empty((2, cols), dtype=torch.float32, device=dev)
        for rr in (0, 1, 2, 3, band_rows // 2, band_rows - 2, band_rows - 1):
            e.tile_get_row(row0 + rr, buf.data_ptr())
            e.synchronize()
            got = buf.cpu().numpy()
            assert_same(got[0], sl[rr], f"view 0 row {rr}")
            assert_same(got[1], sr[rr, ::-1], f"view 1 row {rr} (mirrored)")
        rng = np.random.default_rng(band_rows)
        for rr in (1, band_rows - 1):
            new = rng.uniform(0, 40, (2, cols)).astype(np.float32)
            e.tile_set_row(row0 + rr, up(new).data_ptr())
            e.tile_get_row(row0 + rr, buf.data_ptr())
            e.synchronize()
            assert_same(buf.cpu().numpy(), new, f"row {rr} written and read")
            # the rows beside it are untouched
            e.tile_get_row(row0 + rr - 1, buf.data_ptr())
            e.synchronize()
            assert_same(buf.cpu().numpy()[0], sl[rr - 1], f"row {rr - 1} beside a written row")


@pytest.mark.gpu
@pytest.mark.parametrize("sem,patch,world,rounds", [(0, 5, 4, 2), (0, 11, 3, 2), (1, 3, 4, 2), (0, 3, 7, 2), (0, 3, 7, 0),
                                                    (0, 5, 4, 1), (0, 5, 2, 2), (1, 3, 2, 2)])
def test_tiled_equals_untiled(pm, oracle, synth, sem, patch, world, rounds):
    """rounds = fixed exchange rounds per vertical sweep; 0 forces the "a boundary row still moved" flag and with it
    the repeat with world - 1 rounds (the exactness guarantee), 2 is the default."""
    import tiled
    rows, cols = 150, 200
    l, r, sl, sr, _ = small_pair(synth, 80 + world, rows, cols, n_points=60, dilate_factor=3)
    params = pm.default_params(sem, patch=patch, patchmatch_iters=3)
    # the pipelined schedule (the ranks sweep in order: the default of match_band) first, then the speculative one
    pl, pr, pinfo = tiled.match_tiled_local(params, l, r, sl, sr, world, pipelined=True)
    assert pinfo["rounds"] == 0 and not pinfo["repeated"] and pinfo["exchanges_per_rank"] > 0
    dl, dr, info = tiled.match_tiled_local(params, l, r, sl, sr, world, rounds=rounds)
    assert_same(pl, dl, "pipelined vs speculative schedule (left)")
    assert_same(pr, dr, "pipelined vs speculative schedule (right)")
    if rounds == 0:
        assert info["repeated"] and info["rounds"] == world - 1
    if world == 2 and rounds >= 1:
        # two bands are exact after one round; the last band's re-sweep moves its own border row, which nobody consumes
        # and which therefore must not raise the "repeat the Match" flag (it did before round 3)
        assert not info["repeated"]
    with pm.Engine(params, max_rows=rows, max_cols=cols) as e:
        ul, ur = e.match(l, r, sl, sr)
    assert_same(dl, ul, "tiled vs untiled (left)")
    assert_same(dr, ur, "tiled vs untiled (right)")
    el, er = oracle.match(oracle.default_params(sem, patch=patch, n_iters=3, nthreads=8), l, r, sl, sr)
    assert_same(dl, el, "tiled vs oracle (left)")
    assert_same(dr, er, "tiled vs oracle (right)")
    assert info["exchanges_per_rank"] > 0  # boundary rows did travel


@pytest.mark.gpu
def test_tiled_full_size_4096x2160_eight_bands(pm, oracle, synth):
    """BASELINE configs[3] at full size on one GPU (8 bands as threads): tiled == untiled bit for bit, and a 64-row
    band ACROSS a band boundary (rows 238..301 around the boundary at 270) against the oracle -- as its own
    independent problem, matched by the tiled driver with two bands whose boundary cuts it in the middle."""
    import tiled
    rows, cols, world = 2160, 4096, 8
    p = synth.make_pair(0, rows, cols, n_points=200 * (rows * cols) // (720 * 1280))
    l, r, sl, sr = p["left"], p["right"], p["seed_l"], p["seed_r"]
    params = pm.default_params(0, patch=11, patchmatch_iters=8)
    dl, dr, info = tiled.match_tiled_local(params, l, r, sl, sr, world)
    with pm.Engine(params, max_rows=rows, max_cols=cols) as e:
        ul, ur = e.match(l, r, sl, sr)
    assert_same(dl, ul, "4096x2160, 8 bands vs untiled (left)")
    assert_same(dr, ur, "4096x2160, 8 bands vs untiled (right)")
    pl, pr, _ = tiled.match_tiled_local(params, l, r, sl, sr, world, pipelined=True)
    assert_same(pl, ul, "4096x2160, 8 bands in order vs untiled (left)")
    assert_same(pr, ur, "4096x2160, 8 bands in order vs untiled (right)")
    fg = dl > 0
    assert fg.mean() > 0.15 and (np.abs(dl - p["gt"])[fg] < 1.0).mean() > 0.95
    band = np.s_[238:302, :]
    bl, br, _ = tiled.match_tiled_local(params, l[band], r[band], sl[band], sr[band], 2)
    el, er = oracle.match(oracle.default_params(0, patch=11, n_iters=8, nthreads=16), l[band], r[band], sl[band], sr[band])
    assert_same(bl, el, "64-row band across a tile boundary vs oracle (left)")
    assert_same(br, er, "64-row band across a tile boundary vs oracle (right)")


def _comm_worker(rank, world, port, q):
    import os
    import sys
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ocean-perception_amd",
                                    "python"))
    os.environ["PM_NO_TORCH_PRELOAD"] = "1"
    import tiled
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    comm = tiled.DistComm()
    row = torch.full((2, 5), float(rank))
    down = comm.shift(row, True)    # from rank-1
    up = comm.shift(row, False)     # from rank+1
    # a later round of a downward sweep (r = 1): bands already final neither send nor receive -- only rank 1 -> rank 2
    # is left, and both ends derive that from their positions (match_band)
    r, pos = 1, rank
    late = comm.shift(row + 10.0, True, send=pos + 1 > r and rank + 1 < world, recv=pos > r)
    # the pipelined hand-over: a value travels down the ranks and back up, every rank adding its own
    chain = torch.zeros((2, 5))
    if rank > 0:
        chain = comm.recv((2, 5), torch.device("cpu"), rank - 1, True)
    chain = chain + 1.0
    if rank + 1 < world:
        comm.send(chain, rank + 1, True)
        back = comm.recv((2, 5), torch.device("cpu"), rank + 1, False)
    else:
        back = chain
    if rank > 0:
        comm.send(back + 10.0, rank - 1, False)
    res = (None if down is None else float(down[0, 0]), None if up is None else float(up[0, 0]),
           comm.any(torch.tensor([1 if rank == 1 else 0], dtype=torch.int32)), comm.any(torch.zeros(1, dtype=torch.int32)),
           None if late is None else float(late[0, 0]), float(chain[0, 0]), float(back[1, 4]))
    q.put((rank, res))
    dist.destroy_process_group()


def test_dist_comm_protocol_gloo():
    """DistComm (the RCCL path) on CPU with gloo, 3 ranks: neighbour rows arrive from the right side and the
    1-word all-reduce is an OR."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 3
    procs = [ctx.Process(target=_comm_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # (the last two: the pipelined hand-over -- 1, 2, 3 on the way down, + 10 per hop on the way back up)
    assert out[0] == (None, 1.0, True, False, None, 1.0, 23.0)
    assert out[1] == (0.0, 2.0, True, False, None, 2.0, 13.0)
    assert out[2] == (1.0, None, True, False, 11.0, 3.0, 3.0)


@pytest.mark.gpu
def test_differential_fuzz_tiled_vs_untiled():
    """tools/fuzz_tiled.py: random sizes, band counts 2..6, windows, iteration counts, exchange rounds 0..3 (0 and 1
    mostly end in the repeat path), both semantics: tiled == untiled bit for bit (2500 cases were run; 40 here)."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_tiled.py"), "--cases", "40", "--seed", "8"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "bit-identical" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", [0, 2])
def test_single_process_bench_child_command_line(exchange):
    """What bench.py's multi-rank configs[3] leg starts as a child (tiled_children): python/tiled.py --single-process N
    [--exchange 2] prints one JSON object.  Here N = 1 with three bands on the one device and a small image."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    cmd = [sys.executable, os.path.join(ROOT, "ocean-perception_amd", "python", "tiled.py"), "--rows", "300", "--cols",
           "400", "--iters", "2", "--patch", "5", "--steps", "1", "--single-process", "1", "--bands", "3", "--exchange",
           str(exchange), "--variants", "speculative,direct"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    # the configured run, then the speculative schedule; `direct` is skipped: no boundary between devices here
    assert [l["variant"] for l in lines] == ["default", "speculative"]
    assert "pipelined" in lines[0]["schedule"] and "speculative" in lines[1]["schedule"]
    out = lines[0]
    assert out["bands"] == 3 and out["n_gpus"] == 1 and out["ms_per_frame"] > 0
    assert out["device_boundaries"] == 0 and out["peer_links"] == 0
    assert ("direct" in out["exchange"]) == (exchange == 2)
