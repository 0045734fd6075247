"""ctypes binding of the C ABI in include/pm/patchmatch.h (libvehicle_pm_gpu.so).

This is the Python-side stub a maintainer would add to call the engine; tests/ and bench.py go
through it, so every GPU test exercises the C ABI itself.  There is no fallback of any kind: if the
library is missing the import fails, and without a HIP device pm_create raises.
"""
import ctypes as C
import os

import numpy as np

PM_ABI_VERSION = 6
PM_MAX_ITERS = 16
PM_MAX_PATCH = 15
PM_SEM_CPU, PM_SEM_GPU = 0, 1
PM_ENGINE_AUTO, PM_ENGINE_SERIAL, PM_ENGINE_WAVE, PM_ENGINE_RUNBLK2 = 0, 1, 2, 5
PM_OK = 0
PM_ERR_INVALID_ARG, PM_ERR_SIZE, PM_ERR_HIP, PM_ERR_NO_DEVICE, PM_ERR_NOMEM, PM_ERR_BUSY, PM_ERR_STATE = -1, -2, -3, -4, -5, -6, -7
PM_K_COUNT = 12
PM_MODE_SCALAR, PM_MODE_PLANES = 0, 1
PM_STATE_F32, PM_STATE_F16 = 0, 1
PM_PL_SPATIAL, PM_PL_VIEW, PM_PL_REFINE, PM_PL_VIEW_REFINE = 1, 2, 3, 4
PM_PL_WINDOW_FULL, PM_PL_WINDOW_CHECKER = 0, 1
PM_PL_NEIGH_FOUR, PM_PL_NEIGH_TWO = 0, 1

_HERE = os.path.dirname(os.path.abspath(__file__))
# PM_LIB: experiment knob to load another build of the same library (e.g. a different unroll factor)
LIB_PATH = os.environ.get("PM_LIB") or os.path.normpath(os.path.join(_HERE, "..", "lib", "libvehicle_pm_gpu.so"))

# every symbol include/pm/patchmatch.h declares
EXPORTS = [
    "pm_params_default", "pm_create", "pm_destroy", "pm_last_error", "pm_status_string",
    "pm_match_u8", "pm_match_batch_u8", "pm_match_device", "pm_synchronize", "pm_stream",
    "pm_submit_u8", "pm_submit_bound_u8", "pm_submit_device", "pm_submit_device_after", "pm_collect", "pm_flush", "pm_in_flight",
    "pm_host_alloc", "pm_host_free", "pm_host_register", "pm_host_unregister",
    "pm_capture_begin", "pm_capture_end", "pm_replay", "pm_debug_capture_fork",
    "pm_disp_to_range", "pm_remove_backscatter", "pm_correct_attenuation", "pm_range_enhance",
    "pm_compute_intensity", "pm_find_dark", "pm_stereo_ready", "pm_gaussian_blur", "pm_normalize",
    "pm_normalize_color_illuminant", "pm_match_bgr_device", "pm_device_malloc", "pm_device_free", "pm_upload", "pm_download",
    "pm_gradient_magnitude", "pm_unit_noise", "pm_add_noise", "pm_propagate",
    "pm_remove_background", "pm_mask_occlusions", "pm_foreground_texture_mask", "pm_sparse_init", "pm_corner_subpix", "pm_profile_enable", "pm_profile_read",
    "pm_kernel_name", "pm_debug_counters", "pm_debug_counters_enable",
    "pm_tile_begin", "pm_tile_noise", "pm_tile_sweep", "pm_tile_snapshot", "pm_tile_restore", "pm_tile_get_row",
    "pm_tile_set_row", "pm_tile_background", "pm_tile_finish", "pm_tile_restore_cols", "pm_tile_sweep_masked",
    "pm_tile_exchange_round", "pm_tile_row_moved", "pm_tile_presweep",
    "pm_match_view_device", "pm_set_unit_noise", "pm_initialize",
    "pm_planes_begin", "pm_planes_step", "pm_planes_read", "pm_planes_write", "pm_planes_finish",
    "pm_tiled_band_rows", "pm_tiled_create", "pm_tiled_destroy", "pm_tiled_match_u8", "pm_tiled_last_error",
    "pm_tiled_upload_u8", "pm_tiled_run", "pm_tiled_download", "pm_tiled_topology", "pm_tiled_set_exchange",
    "pm_tiled_set_schedule",
    "pm_tiled_create_logical", "pm_tiled_audit", "pm_tiled_audit_reset", "pm_tiled_debug_inject",
]


class PmParams(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("abi_version", C.c_uint32),
        ("cost_alpha", C.c_float),
        ("patchmatch_iters", C.c_int),
        ("init_dilate_factor", C.c_int),
        ("cost_improve_factor", C.c_float),
        ("semantics", C.c_int),
        ("engine", C.c_int),
        ("noise_amp", C.c_float * PM_MAX_ITERS),
        ("patch_w", C.c_int * PM_MAX_ITERS),
        ("patch_h", C.c_int * PM_MAX_ITERS),
        ("bg_patch_w", C.c_int),
        ("bg_patch_h", C.c_int),
        ("win_by_factor", C.c_float),
        ("functor_alpha", C.c_float),
        ("functor_tau_color", C.c_float),
        ("functor_tau_grad", C.c_float),
        ("noise_seed", C.c_uint64),
        ("left_right_check", C.c_int),
        ("sparse_init", C.c_int),
        ("max_features_per_frame", C.c_int),
        ("min_distance_btw_features", C.c_int),
        ("gftt_block_size", C.c_int),
        ("gftt_quality_level", C.c_double),
        ("templ_cols", C.c_int),
        ("templ_rows", C.c_int),
        ("max_disp", C.c_int),
        ("max_matching_cost", C.c_double),
        ("gftt_use_harris", C.c_int),
        ("gftt_k", C.c_double),
        ("subpixel_corners", C.c_int),
        ("subpix_winsize", C.c_int),
        ("subpix_zerozone", C.c_int),
        ("subpix_maxiters", C.c_int),
        ("subpix_epsilon", C.c_float),
        ("subpixel_refinement", C.c_int),
        ("cpu_initialize_factor", C.c_int),
        ("mode", C.c_int),
        ("state_dtype", C.c_int),
        ("plane_refine_steps", C.c_int),
        ("plane_slope_max", C.c_float),
        ("plane_slope_init", C.c_float),
        ("plane_slope_per_disp", C.c_float),
        ("plane_lr_tol", C.c_float),
        ("plane_window", C.c_int),
        ("plane_neighbours", C.c_int),
        ("stream_priority", C.c_int),
    ]


class PmTile(C.Structure):
    _fields_ = [("global_rows", C.c_int), ("band_row0", C.c_int), ("own_row0", C.c_int), ("own_rows", C.c_int)]


class PmTiledInfo(C.Structure):
    _fields_ = [("rounds_used", C.c_int), ("repeated", C.c_int), ("exchanges", C.c_int)]


class PmTiledAuditRecord(C.Structure):  # include/pm/testing.h
    _fields_ = [("call", C.c_int), ("band", C.c_int), ("detail", C.c_int), ("current_device", C.c_int),
                ("stream_device", C.c_int), ("object_device", C.c_int), ("source_device", C.c_int),
                ("foreign_allowed", C.c_int), ("violation", C.c_int)]


class PmProfile(C.Structure):
    _fields_ = [("launches", C.c_uint64 * PM_K_COUNT), ("total_ms", C.c_double * PM_K_COUNT)]


class PmError(RuntimeError):
    def __init__(self, status, what, detail=""):
        self.status = status
        super().__init__(f"{what}: status {status} ({detail})")


_lib = None


def load():
    """dlopen the engine (cached).  Raises OSError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # A process must hold ONE HIP runtime.  PyTorch wheels bundle their own libamdhip64; if this library
    # is loaded first it pulls in the system runtime, torch later loads its bundled copy, and whichever of
    # the two initialises second finds no device.  Loading torch first makes this library bind to the
    # runtime torch brought (same SONAME).  Pure C/C++ callers are unaffected.
    if os.environ.get("PM_NO_TORCH_PRELOAD") is None:
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    lib = C.CDLL(LIB_PATH)
    u8p, f32p, vp = C.c_void_p, C.c_void_p, C.c_void_p
    lib.pm_params_default.argtypes = [C.POINTER(PmParams), C.c_int]
    lib.pm_params_default.restype = None
    lib.pm_create.argtypes = [C.POINTER(PmParams), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    lib.pm_create.restype = C.c_int
    lib.pm_destroy.argtypes = [vp]
    lib.pm_destroy.restype = None
    lib.pm_last_error.argtypes = [vp]
    lib.pm_last_error.restype = C.c_char_p
    lib.pm_status_string.argtypes = [C.c_int]
    lib.pm_status_string.restype = C.c_char_p
    lib.pm_match_u8.argtypes = [vp, u8p, u8p, C.c_int, C.c_int, C.c_size_t, f32p, f32p, C.c_size_t, f32p, f32p,
                                C.c_size_t]
    lib.pm_match_u8.restype = C.c_int
    lib.pm_match_batch_u8.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_int, C.POINTER(vp),
                                      C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    lib.pm_match_batch_u8.restype = C.c_int
    lib.pm_match_device.argtypes = [vp, C.c_int, u8p, u8p, C.c_int, C.c_int, f32p, f32p, f32p, f32p]
    lib.pm_match_device.restype = C.c_int
    lib.pm_submit_u8.argtypes = [vp, u8p, u8p, C.c_int, C.c_int, C.c_size_t, f32p, f32p, C.c_size_t, C.c_uint64]
    lib.pm_submit_u8.restype = C.c_int
    lib.pm_submit_bound_u8.argtypes = [vp, u8p, u8p, C.c_int, C.c_int, C.c_size_t, f32p, f32p, C.c_size_t, f32p, f32p,
                                       C.c_size_t, C.c_uint64]
    lib.pm_submit_bound_u8.restype = C.c_int
    lib.pm_submit_device.argtypes = [vp, u8p, u8p, C.c_int, C.c_int, f32p, f32p, f32p, f32p, C.c_uint64]
    lib.pm_submit_device.restype = C.c_int
    lib.pm_submit_device_after.argtypes = [vp, u8p, u8p, C.c_int, C.c_int, f32p, f32p, f32p, f32p, C.c_uint64, vp]
    lib.pm_submit_device_after.restype = C.c_int
    lib.pm_collect.argtypes = [vp, f32p, f32p, C.c_size_t, C.POINTER(C.c_uint64)]
    lib.pm_collect.restype = C.c_int
    lib.pm_flush.argtypes = [vp]
    lib.pm_flush.restype = C.c_int
    lib.pm_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.pm_host_alloc.restype = C.c_int
    lib.pm_host_free.argtypes = [vp, vp]
    lib.pm_host_free.restype = C.c_int
    lib.pm_host_register.argtypes = [vp, vp, C.c_size_t]
    lib.pm_host_register.restype = C.c_int
    lib.pm_host_unregister.argtypes = [vp, vp]
    lib.pm_host_unregister.restype = C.c_int
    lib.pm_debug_capture_fork.argtypes = [vp]
    lib.pm_debug_capture_fork.restype = C.c_int
    lib.pm_in_flight.argtypes = [vp]
    lib.pm_in_flight.restype = C.c_int
    # pm/imaging.h: raw device addresses
    f3 = C.POINTER(C.c_float)
    lib.pm_disp_to_range.argtypes = [vp, vp, C.c_int, C.c_int, C.c_double, C.c_double, vp]
    lib.pm_remove_backscatter.argtypes = [vp, vp, vp, C.c_int, C.c_int, f3, f3, vp]
    lib.pm_correct_attenuation.argtypes = [vp, vp, vp, C.c_int, C.c_int, f3, vp]
    lib.pm_range_enhance.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_double, C.c_double, f3, f3, f3, vp, vp]
    lib.pm_compute_intensity.argtypes = [vp, vp, C.c_int, C.c_int, vp]
    lib.pm_find_dark.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_float, vp, f3]
    lib.pm_stereo_ready.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp]
    lib.pm_match_bgr_device.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp]
    lib.pm_match_bgr_device.restype = C.c_int
    lib.pm_gaussian_blur.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, vp]
    lib.pm_normalize.argtypes = [vp, vp, C.c_int, C.c_int, vp]
    lib.pm_normalize_color_illuminant.argtypes = [vp, vp, C.c_int, C.c_int, vp]
    lib.pm_normalize_color_illuminant.restype = C.c_int
    lib.pm_device_malloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.pm_device_free.argtypes = [vp, vp]
    lib.pm_upload.argtypes = [vp, vp, vp, C.c_size_t]
    lib.pm_download.argtypes = [vp, vp, vp, C.c_size_t]
    for name in ("pm_device_malloc", "pm_device_free", "pm_upload", "pm_download"):
        getattr(lib, name).restype = C.c_int
    for name in ("pm_disp_to_range", "pm_remove_backscatter", "pm_correct_attenuation", "pm_range_enhance",
                 "pm_compute_intensity", "pm_find_dark", "pm_stereo_ready", "pm_gaussian_blur", "pm_normalize"):
        getattr(lib, name).restype = C.c_int
    for name in ("pm_capture_begin", "pm_capture_end", "pm_replay"):
        getattr(lib, name).argtypes = [vp]
        getattr(lib, name).restype = C.c_int
    lib.pm_synchronize.argtypes = [vp]
    lib.pm_synchronize.restype = C.c_int
    lib.pm_stream.argtypes = [vp]
    lib.pm_stream.restype = C.c_void_p
    lib.pm_gradient_magnitude.argtypes = [vp, u8p, C.c_int, C.c_int, f32p]
    lib.pm_gradient_magnitude.restype = C.c_int
    lib.pm_unit_noise.argtypes = [vp, C.c_int, C.c_int, f32p]
    lib.pm_unit_noise.restype = C.c_int
    lib.pm_add_noise.argtypes = [vp, f32p, C.c_int, C.c_int, C.c_float]
    lib.pm_add_noise.restype = C.c_int
    lib.pm_propagate.argtypes = [vp, u8p, u8p, C.c_int, C.c_int, f32p, C.c_int, C.c_int, C.c_int]
    lib.pm_propagate.restype = C.c_int
    lib.pm_remove_background.argtypes = [vp, u8p, u8p, C.c_int, C.c_int, f32p, C.c_int, C.c_int, C.c_float]
    lib.pm_remove_background.restype = C.c_int
    lib.pm_mask_occlusions.argtypes = [vp, f32p, f32p, C.c_int, C.c_int]
    lib.pm_mask_occlusions.restype = C.c_int
    lib.pm_foreground_texture_mask.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, vp]
    lib.pm_foreground_texture_mask.restype = C.c_int
    lib.pm_corner_subpix.argtypes = [vp, u8p, C.c_int, C.c_int, f32p, f32p, C.c_int]
    lib.pm_corner_subpix.restype = C.c_int
    lib.pm_sparse_init.argtypes = [vp, u8p, u8p, C.c_int, C.c_int, C.c_int, f32p]
    lib.pm_sparse_init.restype = C.c_int
    lib.pm_initialize.argtypes = [vp, u8p, u8p, C.c_int, C.c_int, C.c_int, f32p]
    lib.pm_initialize.restype = C.c_int
    lib.pm_tile_begin.argtypes = [vp, C.POINTER(PmTile), u8p, u8p, C.c_int, C.c_int, f32p, f32p]
    lib.pm_tile_noise.argtypes = [vp, C.c_int]
    lib.pm_tile_sweep.argtypes = [vp, C.c_int, C.c_int]
    lib.pm_tile_snapshot.argtypes = [vp]
    lib.pm_tile_restore.argtypes = [vp]
    lib.pm_tile_get_row.argtypes = [vp, C.c_int, f32p]
    lib.pm_tile_set_row.argtypes = [vp, C.c_int, f32p]
    lib.pm_tile_background.argtypes = [vp]
    lib.pm_tile_finish.argtypes = [vp, f32p, f32p]
    lib.pm_tiled_band_rows.argtypes = [C.POINTER(PmParams), C.c_int, C.c_int]
    lib.pm_tiled_band_rows.restype = C.c_int
    lib.pm_tiled_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    lib.pm_tiled_create.restype = C.c_int
    lib.pm_tiled_destroy.argtypes = [vp]
    lib.pm_tiled_destroy.restype = None
    lib.pm_tiled_match_u8.argtypes = [vp, vp, vp, C.c_size_t, vp, vp, C.c_size_t, vp, vp, C.c_size_t, C.c_int,
                                      C.POINTER(PmTiledInfo)]
    lib.pm_tiled_match_u8.restype = C.c_int
    lib.pm_tiled_upload_u8.argtypes = [vp, vp, vp, C.c_size_t, vp, vp, C.c_size_t]
    lib.pm_tiled_upload_u8.restype = C.c_int
    lib.pm_tiled_run.argtypes = [vp, C.c_int, C.POINTER(PmTiledInfo)]
    lib.pm_tiled_run.restype = C.c_int
    lib.pm_tiled_download.argtypes = [vp, vp, vp, C.c_size_t]
    lib.pm_tiled_download.restype = C.c_int
    lib.pm_tiled_last_error.argtypes = [vp]
    lib.pm_tiled_last_error.restype = C.c_char_p
    lib.pm_tiled_topology.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.pm_tiled_topology.restype = C.c_int
    lib.pm_tiled_set_exchange.argtypes = [vp, C.c_int]
    lib.pm_tiled_set_exchange.restype = C.c_int
    lib.pm_tiled_set_schedule.argtypes = [vp, C.c_int]
    lib.pm_tiled_set_schedule.restype = C.c_int
    lib.pm_tiled_create_logical.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int,
                                            C.POINTER(vp)]
    lib.pm_tiled_create_logical.restype = C.c_int
    lib.pm_tiled_audit.argtypes = [vp, C.POINTER(PmTiledAuditRecord), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.pm_tiled_audit.restype = C.c_int
    lib.pm_tiled_audit_reset.argtypes = [vp]
    lib.pm_tiled_audit_reset.restype = C.c_int
    lib.pm_tiled_debug_inject.argtypes = [vp, C.c_int]
    lib.pm_tiled_debug_inject.restype = C.c_int
    lib.pm_tile_restore_cols.argtypes = [vp, vp]
    lib.pm_tile_sweep_masked.argtypes = [vp, C.c_int, C.c_int, vp]
    lib.pm_tile_exchange_round.argtypes = [vp, C.c_int, C.c_int, C.c_int, f32p, f32p, f32p, vp]
    lib.pm_tile_exchange_round.restype = C.c_int
    lib.pm_tile_row_moved.argtypes = [vp, C.c_int, f32p, vp]
    lib.pm_tile_row_moved.restype = C.c_int
    lib.pm_tile_presweep.argtypes = [vp, C.c_int, f32p]
    lib.pm_tile_presweep.restype = C.c_int
    for name in ("pm_tile_begin", "pm_tile_noise", "pm_tile_sweep", "pm_tile_snapshot", "pm_tile_restore",
                 "pm_tile_get_row", "pm_tile_set_row", "pm_tile_background", "pm_tile_finish", "pm_tile_restore_cols",
                 "pm_tile_sweep_masked"):
        getattr(lib, name).restype = C.c_int
    lib.pm_match_view_device.argtypes = [vp, f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_size_t, f32p, C.c_size_t, vp]
    lib.pm_match_view_device.restype = C.c_int
    lib.pm_set_unit_noise.argtypes = [vp, f32p, C.c_int, C.c_int]
    lib.pm_set_unit_noise.restype = C.c_int
    lib.pm_planes_begin.argtypes = [vp, C.c_int, u8p, u8p, C.c_int, C.c_int, f32p, f32p]
    lib.pm_planes_step.argtypes = [vp, C.c_int, C.c_int]
    lib.pm_planes_read.argtypes = [vp, C.c_int, C.c_int, f32p]
    lib.pm_planes_write.argtypes = [vp, C.c_int, C.c_int, f32p]
    lib.pm_planes_finish.argtypes = [vp, f32p, f32p]
    for name in ("pm_planes_begin", "pm_planes_step", "pm_planes_read", "pm_planes_write", "pm_planes_finish"):
        getattr(lib, name).restype = C.c_int
    lib.pm_profile_enable.argtypes = [vp, C.c_int]
    lib.pm_profile_enable.restype = C.c_int
    lib.pm_profile_read.argtypes = [vp, C.POINTER(PmProfile)]
    lib.pm_profile_read.restype = C.c_int
    lib.pm_kernel_name.argtypes = [C.c_int]
    lib.pm_kernel_name.restype = C.c_char_p
    lib.pm_debug_counters.argtypes = [vp, C.POINTER(C.c_uint64 * 8)]
    lib.pm_debug_counters.restype = C.c_int
    lib.pm_debug_counters_enable.argtypes = [vp, C.c_int]
    lib.pm_debug_counters_enable.restype = C.c_int
    _lib = lib
    return lib


def default_params(semantics=PM_SEM_CPU, **kw):
    """pm_params_default + keyword overrides.  `patch` sets every per-iteration and background
    window; `noise_amp` / `patch_w` / `patch_h` accept sequences."""
    p = PmParams()
    load().pm_params_default(C.byref(p), semantics)
    patch = kw.pop("patch", None)
    if patch is not None:
        for i in range(PM_MAX_ITERS):
            p.patch_w[i] = patch
            p.patch_h[i] = patch
        p.bg_patch_w = patch
        p.bg_patch_h = patch
    for k, v in kw.items():
        if k in ("noise_amp", "patch_w", "patch_h"):
            arr = getattr(p, k)
            for i, x in enumerate(v):
                arr[i] = x
        else:
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, v)
    return p


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(C.c_void_p)


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.c_void_p)


class Engine:
    """One pm_handle.  Methods mirror the C entry points; numpy arrays in, numpy arrays out."""

    def __init__(self, params=None, device=0, max_rows=720, max_cols=1280, max_batch=1):
        self.lib = load()
        self.params = params if params is not None else default_params()
        self.h = C.c_void_p()
        rc = self.lib.pm_create(C.byref(self.params), device, max_rows, max_cols, max_batch, C.byref(self.h))
        if rc != PM_OK:
            detail = self.lib.pm_last_error(self.h).decode() if self.h else ""
            status = self.lib.pm_status_string(rc).decode()
            if self.h:
                self.lib.pm_destroy(self.h)
                self.h = C.c_void_p()
            raise PmError(rc, "pm_create", f"{status}: {detail}")

    def close(self):
        if self.h:
            self.lib.pm_destroy(self.h)  # (also gives back / un-registers every pm_host_alloc / pm_host_register range)
            self.h = C.c_void_p()
        self._registered = {}

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != PM_OK:
            raise PmError(rc, what, self.lib.pm_last_error(self.h).decode())

    # --- whole path -----------------------------------------------------------------------------
    def match(self, left, right, seed_l=None, seed_r=None, out=None):
        """out = (disp_l, disp_r): contiguous float32 arrays to write into (a caller that re-uses its buffers does not pay
        the page faults of two fresh 3.7 MB arrays per call)."""
        left, pl = _u8(left)
        right, pr = _u8(right)
        rows, cols = left.shape
        sl = sr = None
        psl = psr = None
        if seed_l is not None:
            sl, psl = _f32(seed_l)
        if seed_r is not None:
            sr, psr = _f32(seed_r)
        dl, dr = out if out is not None else (np.empty((rows, cols), np.float32), np.empty((rows, cols), np.float32))
        lr = bool(self.params.left_right_check)
        self._pl_shape = (rows, cols)
        self._check(self.lib.pm_match_u8(self.h, pl, pr, rows, cols, 0, psl, psr, 0, dl.ctypes.data_as(C.c_void_p),
                                         dr.ctypes.data_as(C.c_void_p) if lr else None, 0), "pm_match_u8")
        return (dl, dr) if lr else (dl, None)

    def match_batch(self, lefts, rights, seeds_l=None, seeds_r=None, out=None):
        """out = (list of left maps, list of right maps) to write into (e.g. views of pm_host_alloc memory)."""
        n = len(lefts)
        keep = []

        def arr(items, conv):
            out_ = (C.c_void_p * n)()
            for i, it in enumerate(items):
                if it is None:
                    out_[i] = None
                else:
                    a, p = conv(it)
                    keep.append(a)
                    out_[i] = p
            return out_

        rows, cols = np.asarray(lefts[0]).shape
        pl, pr = arr(lefts, _u8), arr(rights, _u8)
        psl = arr(seeds_l, _f32) if seeds_l is not None else None
        psr = arr(seeds_r, _f32) if seeds_r is not None else None
        if out is not None:
            dls, drs = out
        else:
            dls = [np.empty((rows, cols), np.float32) for _ in range(n)]
            drs = [np.empty((rows, cols), np.float32) for _ in range(n)]
        raw = lambda maps: (C.c_void_p * n)(*[m.ctypes.data for m in maps])
        pdl, pdr = raw(dls), raw(drs)
        lr = bool(self.params.left_right_check)
        self._check(self.lib.pm_match_batch_u8(self.h, n, pl, pr, rows, cols, psl, psr, pdl, pdr if lr else None),
                    "pm_match_batch_u8")
        return dls, (drs if lr else None)

    # --- pipelined sequence: submit without waiting, collect the oldest --------------------------------
    def submit(self, left, right, seed_l=None, seed_r=None, tag=0, out=None):
        """out = (disp_l, disp_r): maps bound at submission (pm_submit_bound_u8); collect() may then be called without
        buffers.  Arrays are passed as they are (no copy): a caller that hands in pm_host_alloc views gets the DMA path."""
        left, pl = _u8(left)
        right, pr = _u8(right)
        rows, cols = left.shape
        sl = sr = None
        psl = psr = None
        if seed_l is not None:
            sl, psl = _f32(seed_l)
        if seed_r is not None:
            sr, psr = _f32(seed_r)
        self._shape_q = getattr(self, "_shape_q", [])
        if out is None:
            self._check(self.lib.pm_submit_u8(self.h, pl, pr, rows, cols, 0, psl, psr, 0, tag), "pm_submit_u8")
        else:
            lr = bool(self.params.left_right_check)
            self._check(self.lib.pm_submit_bound_u8(self.h, pl, pr, rows, cols, 0, psl, psr, 0, out[0].ctypes.data,
                                                    out[1].ctypes.data if lr else None, 0, tag), "pm_submit_bound_u8")
        self._shape_q.append((rows, cols, out))

    def submit_device(self, d_left, d_right, rows, cols, d_seed_l, d_seed_r, d_disp_l, d_disp_r, tag=0, ready_event=None):
        """Raw device addresses; nothing is copied.  collect_device() waits for the frame.  ready_event: a hipEvent_t
        (integer handle, e.g. torch.cuda.Event.cuda_event) recorded behind the producer of the inputs
        (pm_submit_device_after); without it the inputs must be complete when this is called.  The event is consumed
        inside the call (an internal stream waits for it): the caller may drop or re-record it as soon as this returns."""
        self._shape_q = getattr(self, "_shape_q", [])
        if ready_event:
            self._check(self.lib.pm_submit_device_after(self.h, d_left, d_right, rows, cols, d_seed_l, d_seed_r, d_disp_l,
                                                        d_disp_r, tag, C.c_void_p(int(ready_event))),
                        "pm_submit_device_after")
        else:
            self._check(self.lib.pm_submit_device(self.h, d_left, d_right, rows, cols, d_seed_l, d_seed_r, d_disp_l,
                                                  d_disp_r, tag), "pm_submit_device")
        self._shape_q.append((rows, cols, "device"))

    def collect_device(self):
        tag = C.c_uint64(0)
        self._check(self.lib.pm_collect(self.h, None, None, 0, C.byref(tag)), "pm_collect")
        self._shape_q.pop(0)
        return int(tag.value)

    def collect(self, out=None):
        rows, cols, bound = self._shape_q[0] if getattr(self, "_shape_q", None) else (1, 1, None)
        tag = C.c_uint64(0)
        lr = bool(self.params.left_right_check)
        if bound is not None and out is None:
            dl, dr = bound
            self._check(self.lib.pm_collect(self.h, None, None, 0, C.byref(tag)), "pm_collect")
        else:
            dl, dr = out if out is not None else (np.empty((rows, cols), np.float32), np.empty((rows, cols), np.float32))
            self._check(self.lib.pm_collect(self.h, dl.ctypes.data_as(C.POINTER(C.c_float)),
                                            dr.ctypes.data_as(C.POINTER(C.c_float)) if lr else None, 0, C.byref(tag)),
                        "pm_collect")
        self._shape_q.pop(0)
        return dl, (dr if lr else None), int(tag.value)

    def flush(self):
        self._check(self.lib.pm_flush(self.h), "pm_flush")

    # --- page-locked caller memory (pm_host_alloc / pm_host_register) ----------------------------------
    def host_alloc(self, shape, dtype, owned=False):
        """A numpy array over page-locked memory: images, seed maps and output maps kept in such arrays are transferred
        by DMA without a staging copy.  Default: numpy's own memory, page-locked in place (pm_host_register) -- the
        array stays valid after close().  owned=True: memory handed out by pm_host_alloc, which pm_host_free /
        pm_destroy give back: the array must not be touched after host_free(array) / close()."""
        dtype = np.dtype(dtype)
        if not owned:
            a = np.empty(shape, dtype)
            self.host_register(a)
            return a
        nbytes = int(np.prod(shape)) * dtype.itemsize
        ptr = C.c_void_p()
        self._check(self.lib.pm_host_alloc(self.h, nbytes, C.byref(ptr)), "pm_host_alloc")
        buf = (C.c_char * nbytes).from_address(ptr.value)
        a = np.frombuffer(buf, dtype=dtype).reshape(shape)
        self._host = getattr(self, "_host", {})
        self._host[a.ctypes.data] = ptr.value
        return a

    def host_free(self, a):
        if a.ctypes.data in getattr(self, "_host", {}):
            ptr = self._host.pop(a.ctypes.data)
            self._check(self.lib.pm_host_free(self.h, ptr), "pm_host_free")
        else:
            self.host_unregister(a)

    def host_register(self, a):
        """Page-locks the array's memory in place.  The engine keeps a reference to the array until host_unregister /
        close(): memory the library treats as page-locked (DMA in place, k_download's direct stores) must not go back to
        the allocator while the range is registered."""
        self._check(self.lib.pm_host_register(self.h, a.ctypes.data, a.nbytes), "pm_host_register")
        self._registered = getattr(self, "_registered", {})
        self._registered[a.ctypes.data] = a

    def host_unregister(self, a):
        self._check(self.lib.pm_host_unregister(self.h, a.ctypes.data), "pm_host_unregister")
        getattr(self, "_registered", {}).pop(a.ctypes.data, None)

    def debug_capture_fork(self):
        self._check(self.lib.pm_debug_capture_fork(self.h), "pm_debug_capture_fork")

    def in_flight(self):
        return int(self.lib.pm_in_flight(self.h))

    # --- pm/imaging.h (device addresses as ints; float parameter vectors as sequences) ---------------------
    @staticmethod
    def _fv(values, n):
        a = (C.c_float * n)(*[float(v) for v in values])
        return a

    def disp_to_range(self, d_disp, rows, cols, fx, baseline, d_range):
        self._check(self.lib.pm_disp_to_range(self.h, d_disp, rows, cols, fx, baseline, d_range), "pm_disp_to_range")

    def remove_backscatter(self, d_bgr, d_range, rows, cols, B, beta_B, d_out):
        self._check(self.lib.pm_remove_backscatter(self.h, d_bgr, d_range, rows, cols, self._fv(B, 3),
                                                   self._fv(beta_B, 3), d_out), "pm_remove_backscatter")

    def correct_attenuation(self, d_bgr, d_range, rows, cols, X, d_out):
        self._check(self.lib.pm_correct_attenuation(self.h, d_bgr, d_range, rows, cols, self._fv(X, 12), d_out),
                    "pm_correct_attenuation")

    def range_enhance(self, d_bgr, d_disp, rows, cols, fx, baseline, B, beta_B, X, d_range_out, d_out):
        self._check(self.lib.pm_range_enhance(self.h, d_bgr, d_disp, rows, cols, fx, baseline, self._fv(B, 3),
                                              self._fv(beta_B, 3), self._fv(X, 12), d_range_out, d_out),
                    "pm_range_enhance")

    def compute_intensity(self, d_bgr, rows, cols, d_gray):
        self._check(self.lib.pm_compute_intensity(self.h, d_bgr, rows, cols, d_gray), "pm_compute_intensity")

    def find_dark(self, d_intensity, d_range, rows, cols, percentile, d_mask):
        thr = C.c_float(0)
        self._check(self.lib.pm_find_dark(self.h, d_intensity, d_range, rows, cols, percentile, d_mask, C.byref(thr)),
                    "pm_find_dark")
        return float(thr.value)

    def stereo_ready(self, d_bgr8, rows, cols, d_J, d_gray8):
        self._check(self.lib.pm_stereo_ready(self.h, d_bgr8, rows, cols, d_J, d_gray8), "pm_stereo_ready")

    def gaussian_blur(self, d_src, rows, cols, channels, ksize, sigma, d_dst):
        self._check(self.lib.pm_gaussian_blur(self.h, d_src, rows, cols, channels, ksize, sigma, d_dst),
                    "pm_gaussian_blur")

    def normalize(self, d_bgr, rows, cols, d_out):
        self._check(self.lib.pm_normalize(self.h, d_bgr, rows, cols, d_out), "pm_normalize")

    def match_device(self, n, d_left, d_right, rows, cols, d_seed_l, d_seed_r, d_disp_l, d_disp_r):
        """All arguments are raw device addresses (ints)."""
        self._pl_shape = (rows, cols)
        self._check(self.lib.pm_match_device(self.h, n, d_left, d_right, rows, cols, d_seed_l, d_seed_r, d_disp_l,
                                             d_disp_r), "pm_match_device")

    def match_bgr_device(self, n, d_left_bgr8, d_right_bgr8, rows, cols, d_seed_l, d_seed_r, d_disp_l, d_disp_r):
        """Match() on 8-bit BGR pairs, the stereo-ready enhancement folded into the load path (raw device addresses)."""
        self._pl_shape = (rows, cols)
        self._check(self.lib.pm_match_bgr_device(self.h, n, d_left_bgr8, d_right_bgr8, rows, cols, d_seed_l, d_seed_r,
                                                 d_disp_l, d_disp_r), "pm_match_bgr_device")

    def match_view_device(self, d_iml, d_imr, d_gl, d_gr, rows, cols, step, d_disp, disp_step=0, stream=None):
        """One view with caller-supplied float images and gradients (device addresses as ints)."""
        self._check(self.lib.pm_match_view_device(self.h, d_iml, d_imr, d_gl, d_gr, rows, cols, step, d_disp,
                                                  disp_step, stream), "pm_match_view_device")

    def set_unit_noise(self, noise):
        a, p = _f32(noise)
        self._check(self.lib.pm_set_unit_noise(self.h, p, a.shape[0], a.shape[1]), "pm_set_unit_noise")

    def capture_begin(self):
        self._check(self.lib.pm_capture_begin(self.h), "pm_capture_begin")

    def capture_end(self):
        self._check(self.lib.pm_capture_end(self.h), "pm_capture_end")

    def replay(self):
        self._check(self.lib.pm_replay(self.h), "pm_replay")

    def synchronize(self):
        self._check(self.lib.pm_synchronize(self.h), "pm_synchronize")

    def stream(self):
        return self.lib.pm_stream(self.h)

    # --- single stages --------------------------------------------------------------------------
    def gradient_magnitude(self, image):
        image, p = _u8(image)
        g = np.empty(image.shape, np.float32)
        self._check(self.lib.pm_gradient_magnitude(self.h, p, image.shape[0], image.shape[1],
                                                   g.ctypes.data_as(C.c_void_p)), "pm_gradient_magnitude")
        return g

    def unit_noise(self, rows, cols):
        n = np.empty((rows, cols), np.float32)
        self._check(self.lib.pm_unit_noise(self.h, rows, cols, n.ctypes.data_as(C.c_void_p)), "pm_unit_noise")
        return n

    def add_noise(self, disp, amount):
        d = np.array(disp, dtype=np.float32, order="C", copy=True)
        self._check(self.lib.pm_add_noise(self.h, d.ctypes.data_as(C.c_void_p), d.shape[0], d.shape[1], amount),
                    "pm_add_noise")
        return d

    def propagate(self, left, right, disp, patch_h, patch_w, pass_mask=15):
        left, pl = _u8(left)
        right, pr = _u8(right)
        d = np.array(disp, dtype=np.float32, order="C", copy=True)
        self._check(self.lib.pm_propagate(self.h, pl, pr, d.shape[0], d.shape[1], d.ctypes.data_as(C.c_void_p),
                                          patch_h, patch_w, pass_mask), "pm_propagate")
        return d

    def remove_background(self, left, right, disp, patch_h, patch_w, factor):
        left, pl = _u8(left)
        right, pr = _u8(right)
        d = np.array(disp, dtype=np.float32, order="C", copy=True)
        self._check(self.lib.pm_remove_background(self.h, pl, pr, d.shape[0], d.shape[1],
                                                  d.ctypes.data_as(C.c_void_p), patch_h, patch_w, factor),
                    "pm_remove_background")
        return d

    def sparse_init(self, left, right, dilate_factor=4):
        left, pl = _u8(left)
        right, pr = _u8(right)
        seed = np.empty(left.shape, np.float32)
        self._check(self.lib.pm_sparse_init(self.h, pl, pr, left.shape[0], left.shape[1], dilate_factor,
                                            seed.ctypes.data_as(C.c_void_p)), "pm_sparse_init")
        return seed

    def corner_subpix(self, image, xs, ys):
        image, p = _u8(image)
        xs = np.array(xs, dtype=np.float32, copy=True)
        ys = np.array(ys, dtype=np.float32, copy=True)
        self._check(self.lib.pm_corner_subpix(self.h, p, image.shape[0], image.shape[1], xs.ctypes.data_as(C.c_void_p),
                                              ys.ctypes.data_as(C.c_void_p), len(xs)), "pm_corner_subpix")
        return xs, ys

    def initialize(self, left, right, downsample_factor=1):
        left, pl = _u8(left)
        right, pr = _u8(right)
        f = downsample_factor
        seed = np.empty((left.shape[0] // f, left.shape[1] // f), np.float32)
        self._check(self.lib.pm_initialize(self.h, pl, pr, left.shape[0], left.shape[1], f,
                                           seed.ctypes.data_as(C.c_void_p)), "pm_initialize")
        return seed

    def foreground_texture_mask(self, d_gray, rows, cols, ksize, min_grad, downsize, d_mask):
        """Raw device addresses (u8 planes)."""
        self._check(self.lib.pm_foreground_texture_mask(self.h, d_gray, rows, cols, ksize, min_grad, downsize, d_mask),
                    "pm_foreground_texture_mask")

    def mask_occlusions(self, disp_l, disp_r):
        dl = np.array(disp_l, dtype=np.float32, order="C", copy=True)
        dr, pdr = _f32(disp_r)
        self._check(self.lib.pm_mask_occlusions(self.h, dl.ctypes.data_as(C.c_void_p), pdr, dl.shape[0],
                                                dl.shape[1]), "pm_mask_occlusions")
        return dl

    # --- row-tiled mode (device addresses as ints) -----------------------------------------------
    def tile_begin(self, tile, d_left, d_right, band_rows, cols, d_seed_l, d_seed_r):
        self._check(self.lib.pm_tile_begin(self.h, C.byref(tile), d_left, d_right, band_rows, cols, d_seed_l,
                                           d_seed_r), "pm_tile_begin")

    def tile_noise(self, it):
        self._check(self.lib.pm_tile_noise(self.h, it), "pm_tile_noise")

    def tile_sweep(self, it, k):
        self._check(self.lib.pm_tile_sweep(self.h, it, k), "pm_tile_sweep")

    def tile_snapshot(self):
        self._check(self.lib.pm_tile_snapshot(self.h), "pm_tile_snapshot")

    def tile_restore(self):
        self._check(self.lib.pm_tile_restore(self.h), "pm_tile_restore")

    def tile_restore_cols(self, d_mask):
        self._check(self.lib.pm_tile_restore_cols(self.h, d_mask), "pm_tile_restore_cols")

    def tile_sweep_masked(self, it, k, d_mask):
        self._check(self.lib.pm_tile_sweep_masked(self.h, it, k, d_mask), "pm_tile_sweep_masked")

    def tile_exchange_round(self, it, k, pred_image_row, d_incoming, d_used, d_used_next, d_mask):
        self._check(self.lib.pm_tile_exchange_round(self.h, it, k, pred_image_row, d_incoming, d_used, d_used_next, d_mask),
                    "pm_tile_exchange_round")

    def tile_presweep(self, pred_image_row, d_row):
        self._check(self.lib.pm_tile_presweep(self.h, pred_image_row, d_row), "pm_tile_presweep")

    def tile_row_moved(self, image_row, d_ref_row, d_flag):
        self._check(self.lib.pm_tile_row_moved(self.h, image_row, d_ref_row, d_flag), "pm_tile_row_moved")

    def tile_get_row(self, image_row, d_dst):
        self._check(self.lib.pm_tile_get_row(self.h, image_row, d_dst), "pm_tile_get_row")

    def tile_set_row(self, image_row, d_src):
        self._check(self.lib.pm_tile_set_row(self.h, image_row, d_src), "pm_tile_set_row")

    def tile_background(self):
        self._check(self.lib.pm_tile_background(self.h), "pm_tile_background")

    def tile_finish(self, d_out_l, d_out_r):
        self._check(self.lib.pm_tile_finish(self.h, d_out_l, d_out_r), "pm_tile_finish")

    # --- PM_MODE_PLANES stage by stage (device addresses as ints) ---------------------------------------
    def planes_begin(self, n, d_left, d_right, rows, cols, d_seed_l=None, d_seed_r=None):
        self._pl_shape = (rows, cols)
        self._check(self.lib.pm_planes_begin(self.h, n, d_left, d_right, rows, cols, d_seed_l, d_seed_r),
                    "pm_planes_begin")

    def planes_step(self, stage, arg):
        self._check(self.lib.pm_planes_step(self.h, stage, arg), "pm_planes_step")

    def planes_read(self, pair, view):
        """(4, rows, cols) float32: a, b, z, cost of (pair, view); view 1 in mirrored coordinates."""
        rows, cols = self._pl_shape
        out = np.empty((4, rows, cols), np.float32)
        self._check(self.lib.pm_planes_read(self.h, pair, view, out.ctypes.data_as(C.c_void_p)), "pm_planes_read")
        return out

    def planes_write(self, pair, view, planes):
        a, p = _f32(planes)
        self._check(self.lib.pm_planes_write(self.h, pair, view, p), "pm_planes_write")

    def planes_finish(self, d_disp_l, d_disp_r):
        self._check(self.lib.pm_planes_finish(self.h, d_disp_l, d_disp_r), "pm_planes_finish")

    def debug_counters_enable(self, on=True):
        self._check(self.lib.pm_debug_counters_enable(self.h, 1 if on else 0), "pm_debug_counters_enable")

    def debug_counters(self):
        out = (C.c_uint64 * 8)()
        self._check(self.lib.pm_debug_counters(self.h, C.byref(out)), "pm_debug_counters")
        names = ("steps_round1", "steps_fixup", "fixup_rounds", "positions")
        return {ax: {n: int(out[k * 4 + j]) for j, n in enumerate(names)} for k, ax in enumerate(("row", "col"))}

    # --- profiling ------------------------------------------------------------------------------
    def profile_enable(self, on=True):
        self._check(self.lib.pm_profile_enable(self.h, 1 if on else 0), "pm_profile_enable")

    def profile_read(self):
        prof = PmProfile()
        self._check(self.lib.pm_profile_read(self.h, C.byref(prof)), "pm_profile_read")
        return {self.lib.pm_kernel_name(k).decode(): (int(prof.launches[k]), float(prof.total_ms[k]))
                for k in range(PM_K_COUNT)}


PM_TILED_EXCHANGE_AUTO, PM_TILED_EXCHANGE_COPY, PM_TILED_EXCHANGE_DIRECT = 0, 1, 2
PM_TILED_SCHEDULE_SPECULATIVE, PM_TILED_SCHEDULE_PIPELINED = 0, 1
# include/pm/testing.h: pm_tiled_audit_call
TILED_CALLS = {1: "set_device", 2: "malloc", 3: "event_create", 4: "event_record", 5: "stream_wait_event",
               6: "stream_sync", 7: "memset", 8: "copy_h2d", 9: "copy_d2h", 10: "copy_peer", 11: "stage", 12: "stage_arg"}
TILED_STAGES = {1: "begin", 2: "noise", 3: "sweep", 4: "get_row", 5: "presweep", 6: "exchange_round", 7: "row_moved",
                8: "background", 9: "finish", 10: "set_row"}


class TiledEngine:
    """pm_tiled_*: one large pair row-tiled over `n_bands` handles of this process (devices[k] = the device of band k).
    logical_devices (include/pm/testing.h): the bands are ACCOUNTED to these device ids while they run on devices[k], and
    the plan logs every runtime call with the logical devices involved (audit())."""

    def __init__(self, params, rows, cols, n_bands, devices=None, logical_devices=None, simulate_peer_access=0,
                 exchange=None, schedule=None):
        self.lib = load()
        self.rows, self.cols, self.n = rows, cols, n_bands
        self.params = params
        band_rows = self.lib.pm_tiled_band_rows(C.byref(params), rows, n_bands)
        if band_rows < 0:
            raise PmError(band_rows, "pm_tiled_band_rows")
        devices = devices or [0] * n_bands
        self.bands = [Engine(params, device=devices[k], max_rows=band_rows, max_cols=cols) for k in range(n_bands)]
        arr = (C.c_void_p * n_bands)(*[b.h for b in self.bands])
        self.plan = C.c_void_p()
        if logical_devices is None:
            rc = self.lib.pm_tiled_create(arr, n_bands, rows, cols, C.byref(self.plan))
        else:
            ld = (C.c_int * n_bands)(*logical_devices)
            rc = self.lib.pm_tiled_create_logical(arr, n_bands, rows, cols, ld, int(simulate_peer_access),
                                                  C.byref(self.plan))
        if rc != PM_OK:
            msg = self.lib.pm_tiled_last_error(self.plan).decode() if self.plan else ""
            self.close()
            raise PmError(rc, "pm_tiled_create", msg)
        if exchange is not None:
            self.set_exchange(exchange)
        if schedule is not None:
            self.set_schedule(schedule)

    def set_exchange(self, mode):
        self._tcheck(self.lib.pm_tiled_set_exchange(self.plan, int(mode)), "pm_tiled_set_exchange")

    def set_schedule(self, schedule):
        self._tcheck(self.lib.pm_tiled_set_schedule(self.plan, int(schedule)), "pm_tiled_set_schedule")

    def audit(self):
        """(records as dicts, number of violations) of a plan made with logical_devices"""
        total, bad = C.c_int(0), C.c_int(0)
        self._tcheck(self.lib.pm_tiled_audit(self.plan, None, 0, C.byref(total), C.byref(bad)), "pm_tiled_audit")
        recs = (PmTiledAuditRecord * max(total.value, 1))()
        self._tcheck(self.lib.pm_tiled_audit(self.plan, recs, total.value, C.byref(total), C.byref(bad)), "pm_tiled_audit")
        out = []
        for r in recs[:total.value]:
            d = {f: getattr(r, f) for f, _ in PmTiledAuditRecord._fields_}
            d["call_name"] = TILED_CALLS.get(r.call, str(r.call))
            if r.call in (11, 12):
                d["stage"] = TILED_STAGES.get(r.detail // 16, str(r.detail // 16))
                d["arg"] = r.detail % 16
            out.append(d)
        return out, bad.value

    def audit_reset(self):
        self._tcheck(self.lib.pm_tiled_audit_reset(self.plan), "pm_tiled_audit_reset")

    def debug_inject(self, what):
        self._tcheck(self.lib.pm_tiled_debug_inject(self.plan, int(what)), "pm_tiled_debug_inject")

    def topology(self):
        """(neighbouring bands on different devices, of which with direct peer access)"""
        a, b = C.c_int(-1), C.c_int(-1)
        self._tcheck(self.lib.pm_tiled_topology(self.plan, C.byref(a), C.byref(b)), "pm_tiled_topology")
        return a.value, b.value

    def match(self, left, right, seed_l=None, seed_r=None, rounds=-1):
        self.upload(left, right, seed_l, seed_r)
        info = self.run(rounds)
        dl, dr = self.download()
        return dl, dr, info

    def _tcheck(self, rc, what):
        if rc != PM_OK:
            raise PmError(rc, what, self.lib.pm_tiled_last_error(self.plan).decode())

    def upload(self, left, right, seed_l=None, seed_r=None):
        (left, pl), (right, pr) = _u8(left), _u8(right)
        sl, psl = _f32(seed_l) if seed_l is not None else (None, None)
        sr, psr = _f32(seed_r) if seed_r is not None else (None, None)
        if left.shape != (self.rows, self.cols) or right.shape != left.shape:
            raise ValueError("image size differs from the plan")
        self._tcheck(self.lib.pm_tiled_upload_u8(self.plan, pl, pr, 0, psl, psr, 0), "pm_tiled_upload_u8")
        for b in self.bands:
            b.synchronize()  # the host arrays may go away after this call

    def run(self, rounds=-1):
        info = PmTiledInfo()
        self._tcheck(self.lib.pm_tiled_run(self.plan, rounds, C.byref(info)), "pm_tiled_run")
        return {"rounds": info.rounds_used, "repeated": bool(info.repeated), "exchanges": info.exchanges}

    def download(self):
        dl = np.empty((self.rows, self.cols), np.float32)
        dr = np.empty((self.rows, self.cols), np.float32)
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)
        self._tcheck(self.lib.pm_tiled_download(self.plan, ptr(dl), ptr(dr), 0), "pm_tiled_download")
        return dl, (dr if self.params.left_right_check else None)

    def close(self):
        if getattr(self, "plan", None):
            self.lib.pm_tiled_destroy(self.plan)
            self.plan = None
        for b in getattr(self, "bands", []):
            b.close()
        self.bands = []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

