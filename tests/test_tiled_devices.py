"""The row-tiled driver's multi-device discipline (csrc/pm_tiled.hip), proven on ONE GPU.

With one band per GPU every event, stream and allocation of pm_tiled.hip belongs to one device: an event may only be
recorded on a stream of its own device, a stream only be used with its device current, and a band's kernels may read
another device's memory only where the exchange mode says so.  No box of this build has two GPUs, so the plan can be
created with the bands accounted to LOGICAL devices (include/pm/testing.h: pm_tiled_create_logical): every runtime call
is logged with the logical devices involved and the log must be clean -- while the maps still equal the untiled Match().
The reference has one GPU and nothing to compare with (src/vehicle/patchmatch_gpu/patchmatch_gpu.cu:331-376)."""
import numpy as np
import pytest

from conftest import assert_same, small_pair

ROWS, COLS = 150, 200


def _pair(synth, seed=91):
    return small_pair(synth, seed, ROWS, COLS, n_points=60, dilate_factor=3)


def _untiled(pm, params, l, r, sl, sr):
    with pm.Engine(params, max_rows=ROWS, max_cols=COLS) as e:
        return e.match(l, r, sl, sr)


def _check_log(recs, bad, n_bands, logical):
    assert bad == 0, [r for r in recs if r["violation"]][:5]
    by = lambda name: [r for r in recs if r["call_name"] == name]
    # an event is recorded on a stream of ITS device, with that device current
    assert by("event_record")
    for r in by("event_record"):
        assert r["stream_device"] == r["object_device"] == r["current_device"] == logical[r["band"]], r
    # every use of a stream happens with the stream's device current
    for name in ("stream_wait_event", "stream_sync", "memset", "copy_h2d", "copy_d2h", "copy_peer", "stage"):
        for r in by(name):
            assert r["stream_device"] == r["current_device"] == logical[r["band"]], r
    # allocations and events are made on the device of the band they are made for
    assert len(by("malloc")) == 12 * n_bands and len(by("event_create")) == 6 * n_bands
    for r in by("malloc") + by("event_create"):
        assert r["object_device"] == r["current_device"] == logical[r["band"]], r
    # a peer copy lands in the receiving band's memory
    for r in by("copy_peer"):
        assert r["object_device"] == r["current_device"], r
        assert r["source_device"] == logical[r["detail"]], r


@pytest.mark.gpu
@pytest.mark.parametrize("logical,peer,exchange", [
    ([0, 1, 2, 3], 0, 0), ([0, 1, 2, 3], 1, 0), ([0, 1, 2, 3], 1, 2), ([0, 1, 2, 3], 0, 2), ([0, 1, 2, 3], 1, 1),
    ([0, 0, 1, 1], 1, 0), ([0, 0, 1, 1], 1, 2), ([3, 1, 0], 1, 2)])
def test_bands_on_logical_devices_keep_the_device_discipline(pm, synth, logical, peer, exchange):
    """The speculative schedule (rounds, snapshots, masked re-sweeps): the richer protocol, every object of it audited."""
    n = len(logical)
    l, r, sl, sr, _ = _pair(synth)
    params = pm.default_params(0, patch=5, patchmatch_iters=3)
    ul, ur = _untiled(pm, params, l, r, sl, sr)
    with pm.TiledEngine(params, ROWS, COLS, n, logical_devices=logical, simulate_peer_access=peer,
                        exchange=exchange, schedule=pm.PM_TILED_SCHEDULE_SPECULATIVE) as t:
        boundaries = sum(1 for a, b in zip(logical, logical[1:]) if a != b)
        assert t.topology() == (boundaries, boundaries if peer else 0)
        dl, dr, info = t.match(l, r, sl, sr)
        recs, bad = t.audit()
        assert_same(dl, ul, "bands on logical devices vs untiled (left)")
        assert_same(dr, ur, "bands on logical devices vs untiled (right)")
        assert info["exchanges"] > 0
        _check_log(recs, bad, n, logical)
        foreign = [r for r in recs if r["call_name"] == "stage_arg" and r["source_device"] != r["current_device"]]
        crossing = [r for r in recs if r["call_name"] == "copy_peer" and r["source_device"] != r["current_device"]]
        waits_abroad = [r for r in recs if r["call_name"] == "stream_wait_event" and r["object_device"] != r["current_device"]]
        assert waits_abroad  # bands do wait for events of other devices: legal, and the only cross-device signal
        if exchange == pm.PM_TILED_EXCHANGE_DIRECT and peer:
            # kernels read the neighbour's row across the (simulated) link, and only exchange rounds do
            assert foreign and all(r["stage"] == "exchange_round" and r["arg"] == 1 and r["foreign_allowed"] for r in foreign)
        else:
            assert not foreign  # AUTO / COPY, or no peer access: no kernel ever sees another device's memory
            assert crossing
        # a second Match on the same plan (buffers and events reused) stays clean
        t.audit_reset()
        dl2, dr2, _ = t.match(l, r, sl, sr)
        recs2, bad2 = t.audit()
        assert bad2 == 0 and not [r for r in recs2 if r["call_name"] in ("malloc", "event_create")]
        assert_same(dl2, ul, "second Match (left)")
        assert_same(dr2, ur, "second Match (right)")


@pytest.mark.gpu
@pytest.mark.parametrize("logical,peer,exchange", [([0, 1, 2, 3], 1, 0), ([0, 1, 2, 3], 1, 2), ([0, 0, 1, 1], 0, 0), ([2, 1, 0], 1, 1)])
def test_pipelined_schedule_keeps_the_discipline_and_the_maps(pm, synth, logical, peer, exchange):
    """PM_TILED_SCHEDULE_PIPELINED: the bands sweep in order along the sweep direction -- no snapshot, no rounds, no
    re-sweep.  Same maps, same rules for events, streams and memory; kernels see foreign memory only as the source of
    set_row in DIRECT mode."""
    n = len(logical)
    l, r, sl, sr, _ = _pair(synth, 95)
    params = pm.default_params(0, patch=7, patchmatch_iters=3)
    ul, ur = _untiled(pm, params, l, r, sl, sr)
    with pm.TiledEngine(params, ROWS, COLS, n, logical_devices=logical, simulate_peer_access=peer, exchange=exchange,
                        schedule=pm.PM_TILED_SCHEDULE_PIPELINED) as t:
        for _ in range(2):  # the second Match reuses buffers and events
            t.audit_reset() if _ else None
            dl, dr, info = t.match(l, r, sl, sr)
            recs, bad = t.audit()
            assert_same(dl, ul, "pipelined schedule vs untiled (left)")
            assert_same(dr, ur, "pipelined schedule vs untiled (right)")
            assert bad == 0, [x for x in recs if x["violation"]][:5]
            assert info["rounds"] == 0 and not info["repeated"] and info["exchanges"] == 2 * 3 * (n - 1)
            stages = {x["stage"] for x in recs if x["call_name"] == "stage"}
            assert "set_row" in stages and not stages & {"presweep", "exchange_round", "row_moved"}
            for x in [x for x in recs if x["call_name"] == "event_record"]:
                assert x["stream_device"] == x["object_device"] == x["current_device"] == logical[x["band"]], x
            foreign = [x for x in recs if x["call_name"] == "stage_arg" and x["source_device"] != x["current_device"]]
            if exchange == pm.PM_TILED_EXCHANGE_DIRECT and peer:
                assert foreign and all(x["stage"] == "set_row" and x["foreign_allowed"] for x in foreign)
            else:
                assert not foreign


@pytest.mark.gpu
@pytest.mark.parametrize("schedule", [0, 1])
@pytest.mark.parametrize("what,bit,name", [(1, 2, "event_record"), (2, 1, "stream_wait_event")])
def test_the_log_catches_a_deliberate_breach(pm, synth, what, bit, name, schedule):
    """The auditor is not vacuous: with the round-5 arrangement injected (an event of the publisher's device recorded on the
    reader's stream) or a stream used under another band's current device, the log marks exactly those calls -- and only
    at boundaries between DIFFERENT logical devices; the maps are unaffected (on one physical device the breach is harmless)."""
    l, r, sl, sr, _ = _pair(synth, 94)
    params = pm.default_params(0, patch=5, patchmatch_iters=2)
    ul, ur = _untiled(pm, params, l, r, sl, sr)
    logical = [0, 0, 1, 2]
    with pm.TiledEngine(params, ROWS, COLS, 4, logical_devices=logical, simulate_peer_access=1, schedule=schedule) as t:
        t.debug_inject(what)
        dl, dr, _ = t.match(l, r, sl, sr)
        recs, bad = t.audit()
        assert_same(dl, ul, "left")
        assert_same(dr, ur, "right")
        marked = [x for x in recs if x["violation"]]
        assert bad == len(marked) > 0
        assert all(x["call_name"] == name and x["violation"] & bit for x in marked), marked[:3]
        # bands 0 and 1 share logical device 0: the hand-overs between them breach nothing even with the injection on, so
        # band 0 (whose only neighbour is band 1) is never the reader of a marked call; bands 2 and 3 always are
        readers = {x["band"] for x in marked}
        assert 0 not in readers and {2, 3} <= readers, readers
        t.debug_inject(0)
        t.audit_reset()
        t.match(l, r, sl, sr)
        assert t.audit()[1] == 0


@pytest.mark.gpu
def test_exchange_modes_agree_on_one_device(pm, synth):
    """PM_TILED_EXCHANGE_COPY forces the hipMemcpyPeerAsync path also between bands of one device (the path real devices
    take by default); all three modes give the untiled maps."""
    l, r, sl, sr, _ = _pair(synth, 92)
    params = pm.default_params(0, patch=11, patchmatch_iters=3)
    ul, ur = _untiled(pm, params, l, r, sl, sr)
    for mode in (pm.PM_TILED_EXCHANGE_AUTO, pm.PM_TILED_EXCHANGE_COPY, pm.PM_TILED_EXCHANGE_DIRECT):
        for schedule in (pm.PM_TILED_SCHEDULE_SPECULATIVE, pm.PM_TILED_SCHEDULE_PIPELINED):
            with pm.TiledEngine(params, ROWS, COLS, 5, exchange=mode, schedule=schedule) as t:
                dl, dr, _ = t.match(l, r, sl, sr)
            assert_same(dl, ul, f"exchange mode {mode}, schedule {schedule} (left)")
            assert_same(dr, ur, f"exchange mode {mode}, schedule {schedule} (right)")
    with pm.TiledEngine(params, ROWS, COLS, 2) as t:
        with pytest.raises(pm.PmError):
            t.set_exchange(7)
        with pytest.raises(pm.PmError):
            t.set_schedule(2)
        with pytest.raises(pm.PmError):
            t.audit()  # an ordinary plan keeps no log


def _gpus():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


@pytest.mark.gpu
@pytest.mark.parametrize("schedule", [0, 1])
@pytest.mark.parametrize("exchange", [0, 1, 2])
def test_two_bands_on_two_real_devices(pm, synth, exchange, schedule):
    """The same on hardware, where a box has it: bands on devices 0 and 1, boundary rows over the link."""
    if _gpus() < 2:
        pytest.skip("needs two GPUs")
    l, r, sl, sr, _ = _pair(synth, 93)
    params = pm.default_params(0, patch=5, patchmatch_iters=3)
    ul, ur = _untiled(pm, params, l, r, sl, sr)
    with pm.TiledEngine(params, ROWS, COLS, 2, devices=[0, 1], exchange=exchange, schedule=schedule) as t:
        assert t.topology()[0] == 1
        dl, dr, _ = t.match(l, r, sl, sr)
    assert_same(dl, ul, "two devices vs untiled (left)")
    assert_same(dr, ur, "two devices vs untiled (right)")
