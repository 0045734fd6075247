"""The stream policy of a handle (pm_engine.hip::create_handle_streams) under guard.

The engine runs the two views of a pair on two streams and relies on them landing on DIFFERENT hardware queues; how
streams are bound to queues is runtime behaviour read off rocprofv3 traces (profiles/r04_queue_assignment.txt), not a
documented contract.  A runtime update that changes it would show up as a silent 25-30 % loss (measured 430 -> 329 and
384 -> 275 pairs/s for two views on one queue), so this test measures the three device-resident legs alone and with every
other handle kind (and foreign queue-owning streams) alive in the process and fails below 85 % of the alone figure --
what tools/stream_matrix.py does as a tool.  The reference object has no effect on the rest of the process
(patchmatch_gpu.h:118-123: scratch GpuMats only); pm_params.stream_priority is the knob a host application has.
"""
import time

import numpy as np
import pytest

from conftest import assert_same

pytestmark = pytest.mark.gpu

ROWS, COLS, NB, DEPTH = 720, 1280, 4, 4


def _resident(synth):
    import torch
    dev = torch.device("cuda:0")
    prs = [synth.make_pair(i, ROWS, COLS) for i in range(NB)]
    st = lambda k: torch.from_numpy(np.stack([p[k] for p in prs])).to(dev).contiguous()
    t = {k: st(k) for k in ("left", "right", "seed_l", "seed_r")}
    t["dl"] = torch.empty((NB, ROWS, COLS), dtype=torch.float32, device=dev)
    t["dr"] = torch.empty_like(t["dl"])
    torch.cuda.synchronize()
    return t


def _run_single(e, t, k):
    a = (1, t["left"].data_ptr(), t["right"].data_ptr(), ROWS, COLS, t["seed_l"].data_ptr(), t["seed_r"].data_ptr(),
         t["dl"].data_ptr(), t["dr"].data_ptr())
    e.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        e.match_device(*a)
    e.synchronize()
    return k / (time.perf_counter() - t0)


def _run_batch(e, t, k):
    a = (NB, t["left"].data_ptr(), t["right"].data_ptr(), ROWS, COLS, t["seed_l"].data_ptr(), t["seed_r"].data_ptr(),
         t["dl"].data_ptr(), t["dr"].data_ptr())
    e.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        e.match_device(*a)
    e.synchronize()
    return k * NB / (time.perf_counter() - t0)


def _run_seq(e, t, k):
    t0 = time.perf_counter()
    for i in range(k):
        if e.in_flight() == DEPTH:
            e.collect_device()
        q = i % NB
        e.submit_device(t["left"][q].data_ptr(), t["right"][q].data_ptr(), ROWS, COLS, t["seed_l"][q].data_ptr(),
                        t["seed_r"][q].data_ptr(), t["dl"][i % DEPTH].data_ptr(), t["dr"][i % DEPTH].data_ptr(), tag=i)
    while e.in_flight():
        e.collect_device()
    return k / (time.perf_counter() - t0)


LEGS = {"single": (1, _run_single, 30), "batch": (NB, _run_batch, 6), "sequence": (DEPTH, _run_seq, 48)}


def _measure(run, e, t, k):
    run(e, t, max(2, k // 6))  # warm
    return max(run(e, t, k) for _ in range(3))  # best of three: a rate, not a latency -- robust against a noisy moment


@pytest.mark.rate
def test_legs_keep_their_rate_beside_other_handles(pm, synth):
    import torch
    t = _resident(synth)
    prm = pm.default_params(0, patch=11, patchmatch_iters=8)
    alone = {}
    for name, (mb, run, k) in LEGS.items():
        with pm.Engine(prm, max_rows=ROWS, max_cols=COLS, max_batch=mb) as e:
            alone[name] = _measure(run, e, t, k)
    # every handle kind alive at once, a host-buffer sequence handle that has worked, and three foreign streams that own
    # hardware queues (a framework's side streams)
    foreign = [torch.cuda.Stream() for _ in range(3)]
    for s in foreign:
        with torch.cuda.stream(s):
            torch.zeros(1024, device=t["dl"].device).add_(1)
    torch.cuda.synchronize()
    engines = {name: pm.Engine(prm, max_rows=ROWS, max_cols=COLS, max_batch=mb) for name, (mb, _, _) in LEGS.items()}
    host = pm.Engine(prm, max_rows=ROWS, max_cols=COLS, max_batch=2)
    try:
        p = synth.make_pair(0, ROWS, COLS)
        host.submit(p["left"], p["right"], p["seed_l"], p["seed_r"], tag=0)
        host.collect()
        for name, (_, run, k) in LEGS.items():
            run(engines[name], t, max(2, k // 6))
        beside = {name: _measure(run, engines[name], t, k) for name, (_, run, k) in LEGS.items()}
    finally:
        for e in list(engines.values()) + [host]:
            e.close()
    report = {n: (round(alone[n], 1), round(beside[n], 1)) for n in LEGS}
    # What the guard is for: both views of a pair landing on ONE hardware queue costs 28 % (384 -> 275 pairs/s),
    # mixed priority classes cost 30-50 %.  Measured with the policy in place: every leg within 3 % (0.97-1.02).  The floor
    # sits between the two, far enough from the healthy value that a busy box does not trip it; this test is collected
    # last (marker `rate`) and states a rate, never a result.
    for n in LEGS:
        assert beside[n] >= 0.80 * alone[n], f"{n}: {report} (pairs/s alone, beside the other handles)"
    # and the two views of a pair do run side by side: a single pair is no slower than 75 % of the batch rate
    assert alone["single"] >= 0.75 * alone["batch"], report


@pytest.mark.parametrize("prio", [-1, 0])
def test_stream_priority_classes_give_the_same_maps(pm, synth, prio):
    """pm_params.stream_priority (ABI 6): the class of the handle's streams is a scheduling choice, never a result."""
    rows, cols = 96, 160
    p = synth.make_pair(31, rows=rows, cols=cols, n_points=30, dilate_factor=2)
    with pm.Engine(pm.default_params(0, patch=5, patchmatch_iters=2), max_rows=rows, max_cols=cols) as e:
        want = e.match(p["left"], p["right"], p["seed_l"], p["seed_r"])
    prm = pm.default_params(0, patch=5, patchmatch_iters=2, stream_priority=prio)
    with pm.Engine(prm, max_rows=rows, max_cols=cols, max_batch=2) as e:
        got = e.match(p["left"], p["right"], p["seed_l"], p["seed_r"])
        e.submit(p["left"], p["right"], p["seed_l"], p["seed_r"], tag=1)
        seq = e.collect()
    for g in (got, seq):
        assert_same(g[0], want[0], "left")
        assert_same(g[1], want[1], "right")


def test_stream_priority_out_of_range_is_refused(pm):
    for field, bad in (("stream_priority", 2), ("stream_priority", -2)):
        with pytest.raises(pm.PmError) as ex:
            pm.Engine(pm.default_params(0, **{field: bad}), max_rows=64, max_cols=64)
        assert ex.value.status == pm.PM_ERR_INVALID_ARG
