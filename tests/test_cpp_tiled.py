"""The row-tiled large image from C / C++ in ONE process (pm_tiled_* of include/pm/patchmatch.h, bm::pm::TiledPatchmatchGpu):
tests/cpp/tiled_main.cpp, compiled with plain g++, runs the untiled Match() and the tiled one (n bands, here all on
device 0, boundary rows by hipMemcpyPeerAsync + events) and the maps must agree bit for bit -- at small sizes also with
the oracle, and at BASELINE configs[3]'s 4096x2160 with 8 bands."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, assert_same, small_pair

PKG = os.path.join(ROOT, "ocean-perception_amd")
LIBDIR = os.path.join(PKG, "lib")


@pytest.fixture(scope="module")
def tiled_exe(tmp_path_factory):
    out = tmp_path_factory.mktemp("cpp") / "tiled_main"
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(PKG, "host"), os.path.join(ROOT, "tests", "cpp", "tiled_main.cpp"), "-L" + LIBDIR,
           "-lvehicle_pm_gpu", "-Wl,-rpath," + LIBDIR, "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return str(out)


def run(exe, tmp_path, l, r, sl, sr, sem, patch, iters, bands, rounds):
    rows, cols = l.shape
    for name, arr in (("left.u8", l), ("right.u8", r), ("seed_l.f32", sl), ("seed_r.f32", sr)):
        np.ascontiguousarray(arr).tofile(os.path.join(tmp_path, name))
    res = subprocess.run([exe, str(tmp_path), str(rows), str(cols), str(sem), str(patch), str(iters), str(bands),
                          str(rounds)], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout + res.stderr
    rd = lambda n: np.fromfile(os.path.join(tmp_path, n), np.float32).reshape(rows, cols)
    info = dict(zip(res.stdout.split()[0::2], map(int, res.stdout.split()[1::2])))
    return rd("disp_l.f32"), rd("disp_r.f32"), rd("tiled_l.f32"), rd("tiled_r.f32"), info


def test_tiled_driver_builds_with_gxx(tiled_exe):
    assert os.path.exists(tiled_exe)


def _gpus():
    try:
        import torch
        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:
        return 0


def test_bands_on_devices_that_do_not_exist_fail_loudly(tiled_exe):
    """The multi-device constructor path (pm_tiled_create over band handles of DISTINCT devices) on a box that does not
    have those devices: no band is silently moved to another device, nothing falls back to the CPU -- the constructor
    throws what pm_create said.  Without a GPU device 0 fails the same way; with one GPU device 1 does."""
    n = _gpus()
    res = subprocess.run([tiled_exe, "devices", f"{n},{n + 1}"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 10 and "exception" in res.stdout, res.stdout + res.stderr
    assert ("no HIP device" in res.stdout) or ("out of range" in res.stdout), res.stdout
    if n >= 1:  # the first band can be created, the second cannot: the constructor cleans up and still throws
        res = subprocess.run([tiled_exe, "devices", f"0,{n}"], capture_output=True, text=True, timeout=300)
        assert res.returncode == 10 and "out of range" in res.stdout, res.stdout + res.stderr


@pytest.mark.gpu
def test_bands_sharing_a_device_request_no_peer_access(tiled_exe):
    """All bands on device 0: no device boundary, hipDeviceEnablePeerAccess is never called (pm_tiled_topology says so);
    on a box with two or more GPUs the same call over distinct devices reports one boundary per neighbouring pair."""
    res = subprocess.run([tiled_exe, "devices", "0,0,0"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "bands 3 device_boundaries 0 peer_links 0" in res.stdout, res.stdout + res.stderr
    if _gpus() >= 2:
        res = subprocess.run([tiled_exe, "devices", "0,1"], capture_output=True, text=True, timeout=300)
        assert res.returncode == 0 and "bands 2 device_boundaries 1" in res.stdout, res.stdout + res.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("sem,patch,bands,rounds", [(0, 5, 4, 2), (0, 11, 3, 2), (1, 3, 4, 2), (0, 3, 7, 0), (0, 5, 2, 2),
                                                    (0, 5, 4, -2), (1, 3, 6, -2)])  # -2: the pipelined schedule
def test_cpp_tiled_equals_untiled_and_oracle(tiled_exe, tmp_path, oracle, synth, sem, patch, bands, rounds):
    rows, cols = 150, 200
    l, r, sl, sr, _ = small_pair(synth, 90 + bands, rows, cols, n_points=60, dilate_factor=3)
    ul, ur, tl, tr, info = run(tiled_exe, tmp_path, l, r, sl, sr, sem, patch, 3, bands, rounds)
    assert_same(tl, ul, "tiled vs untiled (left)")
    assert_same(tr, ur, "tiled vs untiled (right)")
    el, er = oracle.match(oracle.default_params(sem, patch=patch, n_iters=3, nthreads=8), l, r, sl, sr)
    assert_same(tl, el, "tiled vs oracle (left)")
    assert_same(tr, er, "tiled vs oracle (right)")
    assert info["exchanges"] > 0
    if rounds == -2:
        assert info["repeated"] == 0 and info["rounds_used"] == 0 and info["exchanges"] == 2 * 3 * (bands - 1)
    if rounds == 0:
        assert info["repeated"] == 1 and info["rounds_used"] == bands - 1
    if bands == 2 and rounds >= 1:
        assert info["repeated"] == 0


@pytest.mark.gpu
def test_cpp_tiled_configs3_full_size(tiled_exe, tmp_path, synth):
    """BASELINE configs[3]: one 4096x2160 pair, 8 bands, 8 iterations, 11x11 -- from a C++ caller."""
    rows, cols = 2160, 4096
    p = synth.make_pair(0, rows, cols, n_points=200 * (rows * cols) // (720 * 1280))
    ul, ur, tl, tr, info = run(tiled_exe, tmp_path, p["left"], p["right"], p["seed_l"], p["seed_r"], 0, 11, 8, 8, 2)
    assert_same(tl, ul, "4096x2160, 8 bands vs untiled (left)")
    assert_same(tr, ur, "4096x2160, 8 bands vs untiled (right)")
    fg = tl > 0
    assert fg.mean() > 0.15 and (np.abs(tl - p["gt"])[fg] < 1.0).mean() > 0.95
    # the same frame with the bands sweeping in order (PM_TILED_SCHEDULE_PIPELINED)
    _, _, pl, pr, info = run(tiled_exe, tmp_path, p["left"], p["right"], p["seed_l"], p["seed_r"], 0, 11, 8, 8, -2)
    assert_same(pl, ul, "4096x2160, 8 bands in order vs untiled (left)")
    assert_same(pr, ur, "4096x2160, 8 bands in order vs untiled (right)")
    assert info["exchanges"] == 2 * 8 * 7


@pytest.mark.gpu
def test_python_tiled_engine_three_step_api(pm, oracle, synth):
    """pm_tiled_upload_u8 / pm_tiled_run / pm_tiled_download through the ctypes TiledEngine (what bench.py's configs[3]
    leg drives): the resident pair can be run repeatedly, every run gives the untiled result."""
    rows, cols, bands = 150, 200, 4
    l, r, sl, sr, _ = small_pair(synth, 97, rows, cols, n_points=60, dilate_factor=3)
    prm = pm.default_params(0, patch=7, patchmatch_iters=3)
    el, er = oracle.match(oracle.default_params(0, patch=7, n_iters=3, nthreads=8), l, r, sl, sr)
    with pm.TiledEngine(prm, rows, cols, bands, schedule=pm.PM_TILED_SCHEDULE_SPECULATIVE) as te:
        te.upload(l, r, sl, sr)
        for rounds in (2, 0, 1):
            info = te.run(rounds)
            dl, dr = te.download()
            assert_same(dl, el, f"rounds {rounds} left")
            assert_same(dr, er, f"rounds {rounds} right")
            assert info["exchanges"] > 0 and (rounds != 0 or info["repeated"])
        dl, dr, info = te.match(l, r, sl, sr)
        assert_same(dl, el, "match() left")
