"""The C-ABI shared library loads and exports every symbol include/pm/patchmatch.h declares, and
fails loudly without a GPU (no compute is attempted here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    """Every function declared by the public headers include/pm/*.h."""
    names = set()
    inc = os.path.join(ROOT, "include", "pm")
    for header in sorted(os.listdir(inc)):
        text = open(os.path.join(inc, header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(pm_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_header_declares_what_the_binding_lists(pm):
    assert declared_functions() == sorted(pm.EXPORTS)


def test_library_exports_every_declared_symbol(pm):
    lib = pm.load()
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in include/pm/*.h but not exported"


def test_params_default_matches_reference_defaults(pm):
    p = pm.default_params(pm.PM_SEM_GPU)
    assert p.struct_size == C.sizeof(pm.PmParams) and p.abi_version == pm.PM_ABI_VERSION
    # patchmatch_gpu.h:85-88
    assert abs(p.cost_alpha - 0.9) < 1e-7 and p.patchmatch_iters == 3
    assert p.init_dilate_factor == 4 and abs(p.cost_improve_factor - 0.8) < 1e-7
    # noise schedule 32 / 2^i (patchmatch_gpu.cu:395), seed 123 (patchmatch_gpu.cu:341)
    assert [p.noise_amp[i] for i in range(4)] == [32.0, 16.0, 8.0, 4.0] and p.noise_seed == 123
    assert p.left_right_check == 1
    q = pm.default_params(pm.PM_SEM_CPU)
    assert abs(q.functor_alpha - 0.7) < 1e-7 and q.functor_tau_color == 50.0 and q.functor_tau_grad == 20.0
    assert abs(q.win_by_factor - 1.5) < 1e-7


def test_status_strings(pm):
    lib = pm.load()
    assert lib.pm_status_string(0) == b"ok"
    assert b"device" in lib.pm_status_string(pm.PM_ERR_NO_DEVICE)
    assert lib.pm_kernel_name(3) == b"sweep_row"


def test_create_rejects_bad_arguments_without_touching_a_device(pm):
    lib = pm.load()
    h = C.c_void_p()
    assert lib.pm_create(None, 0, 64, 64, 1, C.byref(h)) == pm.PM_ERR_INVALID_ARG
    p = pm.default_params(pm.PM_SEM_CPU, patch=4)  # even window: patchmatch.cpp:257-258 CHECKs oddness
    rc = lib.pm_create(C.byref(p), 0, 64, 64, 1, C.byref(h))
    assert rc == pm.PM_ERR_INVALID_ARG and b"odd" in lib.pm_last_error(h)
    lib.pm_destroy(h)
    p = pm.default_params(pm.PM_SEM_CPU)
    p.struct_size = 8
    rc = lib.pm_create(C.byref(p), 0, 64, 64, 1, C.byref(h))
    assert rc == pm.PM_ERR_INVALID_ARG
    lib.pm_destroy(h)


def test_create_checks_the_seeder_options(pm):
    """feature_detector.hpp:34-44 / stereo_matcher.hpp:25-26: the Harris response and cv::cornerSubPix on corners and
    matches are honoured; values outside their ranges are refused by pm_create (before it looks for a device), never
    silently replaced."""
    lib = pm.load()
    h = C.c_void_p()
    bad = (dict(subpixel_corners=2), dict(subpixel_refinement=-1), dict(subpixel_corners=1, subpix_winsize=0),
           dict(subpixel_corners=1, subpix_winsize=16), dict(subpix_maxiters=0), dict(subpix_epsilon=-1.0))
    for kw in bad:
        p = pm.default_params(pm.PM_SEM_GPU, **kw)
        rc = lib.pm_create(C.byref(p), 0, 64, 64, 1, C.byref(h))
        assert rc == pm.PM_ERR_INVALID_ARG and b"cornerSubPix" in lib.pm_last_error(h), kw
        lib.pm_destroy(h)
    for kw in (dict(gftt_use_harris=2), dict(gftt_use_harris=1, gftt_k=-0.1), dict(gftt_k=float("nan"))):
        p = pm.default_params(pm.PM_SEM_GPU, **kw)
        rc = lib.pm_create(C.byref(p), 0, 64, 64, 1, C.byref(h))
        assert rc == pm.PM_ERR_INVALID_ARG and b"gftt" in lib.pm_last_error(h), kw
        lib.pm_destroy(h)
    d = pm.default_params(pm.PM_SEM_GPU)
    assert d.gftt_use_harris == 0 and abs(d.gftt_k - 0.04) < 1e-12 and d.subpixel_corners == 0
    assert (d.subpix_winsize, d.subpix_zerozone, d.subpix_maxiters) == (10, -1, 10) and abs(d.subpix_epsilon - 0.01) < 1e-9


def test_the_shipped_library_carries_no_tuning_knob():
    """The A/B knobs of the tuning build (pm_tune.hpp: PM_STREAM_PRIO, PM_PAIR_CHUNK, PM_RUNBLK_*, PM_G16_*, ...) are
    compiled out of the product: none of their names is left in the binary, so no schedule or LDS budget can depend
    on the host's environment."""
    import subprocess
    so = os.path.join(ROOT, "ocean-perception_amd", "lib", "libvehicle_pm_gpu.so")
    names = re.findall(r"^PM_[A-Z0-9_]+$", subprocess.run(["strings", so], capture_output=True, text=True).stdout, re.M)
    assert names == [], names
    src = os.path.join(ROOT, "ocean-perception_amd", "csrc")
    for f in sorted(os.listdir(src)):
        if f != "pm_tune.hpp" and os.path.isfile(os.path.join(src, f)):  # (csrc/experimental/ holds records, not sources)
            assert "getenv" not in open(os.path.join(src, f)).read().replace("tune_env", ""), f


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU failure mode")
def test_no_gpu_means_loud_failure_not_fallback(pm):
    with pytest.raises(pm.PmError) as e:
        pm.Engine(pm.default_params(pm.PM_SEM_CPU), max_rows=64, max_cols=64)
    assert e.value.status == pm.PM_ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)
