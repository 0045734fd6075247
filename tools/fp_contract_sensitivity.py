"""How much would the reference's maps move if ITS compiler contracted the cost functor's `alpha * ec + (1 - alpha) * eg`
(test/stereo_matching/patchmatch_test.cpp:44) into a fused multiply-add?  The reference builds with g++'s default
-ffp-contract=fast (CMakeLists.txt:17-25); this build defines the expression unfused.  Nothing in the reference
pins either choice -- this counts what is at stake: the oracle is built three times (uncontracted; the first product
fused; the second product fused: oracle/pm_oracle.c functor_mix, -DPMO_FUNCTOR_FMA) and run on the reference's own test pair
with the recipe of patchmatch_test.cpp:149-183 and on a band of the benchmark pair with the benchmark's settings.  CPU only.

    python tools/fp_contract_sensitivity.py > profiles/r06_fp_contract_sensitivity.txt
"""
import ctypes as C
import os
import subprocess
import sys
import importlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "ocean-perception_amd", "python"))
import numpy as np

SRCS = ["pm_oracle.c", "pm_seed_oracle.c", "pm_imaging_oracle.c", "pm_enhance_oracle.c", "pm_planes_oracle.c"]


def build(tag, define):
    out = f"/tmp/libpm_oracle_{tag}.so"
    cmd = ["gcc", "-O3", "-march=native", "-ffp-contract=off", "-fopenmp", "-fPIC", "-std=c11", "-shared", "-o", out] + \
          ([f"-DPMO_FUNCTOR_FMA={define}"] if define else []) + [os.path.join(ROOT, "oracle", s) for s in SRCS] + ["-lm"]
    subprocess.run(cmd, check=True)
    return out


def oracle_with(lib_path):
    import oracle_lib
    importlib.reload(oracle_lib)
    oracle_lib.LIB_PATH = lib_path
    oracle_lib._lib = None
    oracle_lib.load()
    return oracle_lib


def run_all(O):
    import synth
    res = {}
    z = np.load(os.path.join(ROOT, "tests", "golden", "farmsim_fs1_376x240.npz"))
    l, r = z["left"], z["right"]
    sp = O.seed_params(templ_cols=31, templ_rows=11, max_disp=128, max_matching_cost=0.15)
    seeds = O.cpu_initialize(l, r, 1, sp)
    prm = O.default_params(0, n_iters=4, bg_patch_w=3, bg_patch_h=3, bg_factor=1.5, left_right_check=0, nthreads=8,
                           noise_amp=[32.0, 8.0, 2.0, 0.5], patch_w=[5, 5, 3, 3], patch_h=[5, 5, 3, 3])
    res["farmsim 376x240, recipe of patchmatch_test.cpp:149-183 (left map)"] = O.match(prm, l, r, seeds, None)[0]
    p = synth.make_pair(0, 720, 1280)
    band = slice(280, 440)
    prm = O.default_params(0, patch=11, n_iters=8, nthreads=8)
    dl, dr = O.match(prm, p["left"][band], p["right"][band], p["seed_l"][band], p["seed_r"][band])
    res["benchmark pair 0, rows 280-439 as its own problem, 8 iterations, 11x11 (left map)"] = dl
    res["the same, right map"] = dr
    return res


def main():
    base = run_all(oracle_with(build("nofma", 0)))
    print("# Sensitivity of the maps to a contraction of alpha * ec + (1 - alpha) * eg in the cost functor (tools/fp_contract_sensitivity.py;")
    print("# CPU oracle only).  `defined` = two products and a sum (this build's definition, oracle and engine); fma1 = fmaf(alpha, ec,")
    print("# (1 - alpha) * eg); fma2 = fmaf(1 - alpha, eg, alpha * ec).  A differing pixel is one whose disparity is not bit-equal.")
    for tag, d in (("fma1", 1), ("fma2", 2)):
        other = run_all(oracle_with(build(tag, d)))
        for k in base:
            a, b = base[k], other[k]
            diff = a != b
            fg = (a > 0) | (b > 0)
            big = np.abs(a - b) > 1.0
            print(f"{tag}  {k}: {int(diff.sum())} of {a.size} pixels differ ({100.0 * diff.mean():.3f} %; of the foreground "
                  f"{100.0 * diff[fg].mean() if fg.any() else 0.0:.3f} %), {int(big.sum())} by more than one pixel of disparity, "
                  f"{int(((a > 0) != (b > 0)).sum())} change between foreground and background")


if __name__ == "__main__":
    main()
