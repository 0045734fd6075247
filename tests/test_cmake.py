"""The CMake target that replaces `vehicle_pm_gpu` (src/vehicle/patchmatch_gpu/CMakeLists.txt:1-18) is configured and
built, and a consumer project links the C++ mirror against it by target name, as the reference's test tree does
(test/CMakeLists.txt:49-64).  north_star: "host side stays C++/CMake".  No GPU needed: hipcc cross-compiles gfx950."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(shutil.which("cmake") is None, reason="cmake not installed")
def test_cmake_target_builds_and_a_consumer_links_it(tmp_path):
    src = os.path.join(ROOT, "tests", "cmake_consumer")
    bld = str(tmp_path / "build")
    gen = ["-G", "Ninja"] if shutil.which("ninja") else []
    r = subprocess.run(["cmake", "-S", src, "-B", bld, "-DPM_REPO=" + ROOT, "-DCMAKE_BUILD_TYPE=Release"] + gen,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run(["cmake", "--build", bld, "--parallel", "8"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-4000:]
    so = os.path.join(bld, "patchmatch_gpu", "libvehicle_pm_gpu.so")
    exe = os.path.join(bld, "wrapper_main")
    assert os.path.isfile(so) and os.path.isfile(exe)
    # the library the target built exports the C ABI, and the consumer is bound to THAT file
    syms = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True).stdout
    for s in ("pm_create", "pm_match_u8", "pm_submit_u8", "pm_tiled_create", "pm_host_register"):
        assert (" T " + s) in syms, s
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert os.path.realpath(so) in os.path.realpath(ldd.split("libvehicle_pm_gpu.so =>")[1].split()[0])
    if not _has_gpu():  # the consumer runs and fails the way a box without a GPU must
        r = subprocess.run([exe, str(tmp_path), "32", "48", "1", "3", "2"], capture_output=True, text=True)
        assert r.returncode == 10 and "no HIP device" in r.stdout
