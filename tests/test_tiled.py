"""Row-tiled large-image mode (BASELINE configs[3]): the banded result must equal the untiled one bit for
bit.  Runs the real protocol (boundary-row exchange, snapshot / re-sweep until no incoming row changes) with
the ranks as threads of one process on one GPU; the band arithmetic is also checked on CPU."""
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
@pytest.mark.parametrize("sem,patch,world", [(0, 5, 4), (0, 11, 3), (1, 3, 4), (0, 3, 7)])
def test_tiled_equals_untiled(pm, oracle, synth, sem, patch, world):
    import tiled
    rows, cols = 150, 200
    l, r, sl, sr, _ = small_pair(synth, 80 + world, rows, cols, n_points=60, dilate_factor=3)
    params = pm.default_params(sem, patch=patch, patchmatch_iters=3)
    dl, dr, rounds = tiled.match_tiled_local(params, l, r, sl, sr, world)
    with pm.Engine(params, max_rows=rows, max_cols=cols) as e:
        ul, ur = e.match(l, r, sl, sr)
    assert_same(dl, ul, "tiled vs untiled (left)")
    assert_same(dr, ur, "tiled vs untiled (right)")
    el, er = oracle.match(oracle.default_params(sem, patch=patch, n_iters=3, nthreads=8), l, r, sl, sr)
    assert_same(dl, el, "tiled vs oracle (left)")
    assert_same(dr, er, "tiled vs oracle (right)")
    assert rounds >= 1  # values did cross band boundaries, i.e. the fix-up path was exercised


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
    res = (None if down is None else float(down[0, 0]), None if up is None else float(up[0, 0]),
           comm.any(rank == 1), comm.any(False))
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
    assert out[0] == (None, 1.0, True, False)
    assert out[1] == (0.0, 2.0, True, False)
    assert out[2] == (1.0, None, True, False)
