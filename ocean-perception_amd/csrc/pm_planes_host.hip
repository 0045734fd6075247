// pm_planes_host.hip -- PM_MODE_PLANES on the host side: the translation unit of the plane-mode kernels
// (pm_planes.hpp), the schedule of oracle/pm_planes_oracle.c::pmo_planes_match, and the pm_planes_* entry points
// (include/pm/patchmatch.h).
#include <cmath>

#include "pm_handle.hpp"
#include "pm_planes.hpp"
#include "pm_tune.hpp"

using namespace pm;
using namespace pm::eng;

namespace {

dim3 pixel_grid(int cols, int rows, int z) { return dim3((unsigned)((cols + 255) / 256), (unsigned)rows, (unsigned)z); }

PlanesParams planes_params(const pm_params& p) {
  PlanesParams pp;
  pp.patch = p.patch_w[0];
  pp.max_disp = p.max_disp;
  pp.refine_steps = p.plane_refine_steps;
  // the bound itself must be a fixed point of the state's rounding (oracle: slope_bound)
  pp.slope_max = p.state_dtype == PM_STATE_F16 ? (float)(_Float16)p.plane_slope_max : p.plane_slope_max;
  // columns a window can reach beyond [x - h - max_disp, x + h]: h * (|a| + |b|) on either side, + rounding slack
  pp.margin = (int)std::ceil(2.0 * (pp.patch / 2) * (double)pp.slope_max) + 2;
  pp.slope_init = p.plane_slope_init;
  pp.slope_per_disp = p.plane_slope_per_disp;
  pp.alpha = p.functor_alpha;
  pp.one_minus_alpha = 1.f - p.functor_alpha;
  pp.tau_color = p.functor_tau_color;
  pp.tau_grad = p.functor_tau_grad;
  pp.window = p.plane_window;
  pp.neighbours = p.plane_neighbours;
  // taps that count: all of them, or those with i + j even (the centre and every other tap: (P * P + 1) / 2)
  pp.inv_n = 1.0f / (float)(pp.window == PM_PL_WINDOW_CHECKER ? (pp.patch * pp.patch + 1) / 2 : pp.patch * pp.patch);
  pp.lr_tol = p.plane_lr_tol;
  pp.seed = p.noise_seed;
  pp.n_views = p.left_right_check ? 2 : 1;
  return pp;
}

}  // namespace

namespace pm {
// tuning build: PM_PLANES_DBG (timing switches of k_planes); 0 in the shipped library
static int planes_dbg() {
  static const int v = [] {
    const char* e = pm::tune_env("PM_PLANES_DBG");
    return e ? atoi(e) : 0;
  }();
  return v;
}
namespace eng {

int planes_alloc(pm_handle* h) {
  if (h->planes_state) return PM_OK;
  {  // the tile of the widest stage must fit the CU's LDS: (128 + P-1 + max_disp + slope margin) x (8 + P-1) entries
    const PlanesParams pp = planes_params(h->params);
    const size_t need = pl_lds_bytes<PL_SPATIAL>(pp.patch, pp);
    if (need > kChainLdsMax) {
      set_err(h, "PM_MODE_PLANES: window %d, max_disp %d and slope_max %.2f need %zu KB of LDS per tile (limit %zu KB): "
                 "lower max_disp or the window", pp.patch, pp.max_disp, (double)pp.slope_max, need / 1024,
              kChainLdsMax / 1024);
      return PM_ERR_INVALID_ARG;
    }
  }
  const size_t plane = (size_t)h->max_rows * h->max_pitch;
  const size_t bytes = sizeof(float) * ((size_t)h->max_batch * 2 * 4 * plane + 64);
  PM_HIP(h, hipMalloc(&h->planes_state, bytes));
  PM_HIP(h, hipMemsetAsync(h->planes_state, 0, bytes, h->stream));
  return PM_OK;
}

}  // namespace eng
}  // namespace pm

namespace {

// PM_PLANES_LANES=1 (tuning builds): the whole batch through every launch on one stream, as until round 6
bool planes_two_lanes() {
  static const bool on = [] {
    const char* e = pm::tune_env("PM_PLANES_LANES");
    return !(e && atoi(e) == 1);
  }();
  return on;
}

// pair0: the launch covers the pairs from pair0 on (`ps` is the plan's plane set; slots counts from that pair)
template <int STAGE>
int planes_stage(pm_handle* h, const PlaneSet& ps0, const PlArgs& ar, int slots, int klass, const char* what, int pair0 = 0) {
  Launch l(h, klass);
  const bool f16 = h->params.state_dtype == PM_STATE_F16;
  const PlaneSet ps = pair0 ? pair_plane_set(ps0, pair0) : ps0;
  // PlaneState::arr: a pair holds 2 views x 4 arrays of `plane` elements
  char* state = (char*)h->planes_state + (size_t)pair0 * 8 * ps0.plane * (f16 ? sizeof(_Float16) : sizeof(float));
  const hipError_t e = pl_launch<STAGE>(ps, state, f16, planes_params(h->params), ar, slots, h->stream);
  if (e != hipSuccess) {
    set_err(h, "launch of planes %s failed: %s", what, hipGetErrorString(e));
    return PM_ERR_HIP;
  }
  return PM_OK;
}

int planes_step(pm_handle* h, const PlaneSet& ps, int n, int stage, int arg, int pair0 = 0) {
  const int nv = ps.n_views;
  PlArgs ar{};
  ar.dbg = planes_dbg();
  ar.stage = stage;
  ar.arg = arg;
  ar.view_fixed = -1;
  switch (stage) {
    case PM_PL_SPATIAL:
      return planes_stage<PL_SPATIAL>(h, ps, ar, n * nv, PM_K_PL_SPATIAL, "spatial propagation", pair0);
    case PM_PL_VIEW:
      if (nv < 2) return PM_OK;
      ar.view_fixed = arg;
      return planes_stage<PL_VIEW>(h, ps, ar, n, PM_K_PL_VIEW, "view propagation", pair0);
    case PM_PL_REFINE:
      ar.refine_amp = h->params.noise_amp[arg];
      return planes_stage<PL_REFINE>(h, ps, ar, n * nv, PM_K_PL_REFINE, "refinement", pair0);
    case PM_PL_VIEW_REFINE: {  // arg = iteration * 2 + view: view propagation into `view`, then its refinement
      const int view = arg & 1, it = arg >> 1;
      if (view >= nv) return PM_OK;
      ar.arg = it;
      ar.view_fixed = view;
      ar.refine_amp = h->params.noise_amp[it];
      if (nv < 2) return planes_stage<PL_REFINE>(h, ps, ar, n, PM_K_PL_REFINE, "refinement", pair0);
      return planes_stage<PL_VIEW_REFINE>(h, ps, ar, n, PM_K_PL_VIEW_REFINE, "view propagation + refinement", pair0);
    }
    default:
      set_err(h, "unknown planes stage %d", stage);
      return PM_ERR_INVALID_ARG;
  }
}

// prep (images, gradients, packed planes) + seeds + random initialisation of n pairs
int planes_begin(pm_handle* h, int n, const uint8_t* d_left, const uint8_t* d_right, int rows, int cols,
                 const float* d_seed_l, const float* d_seed_r) {
  if (int rc = planes_alloc(h)) return rc;
  const int nv = h->params.left_right_check ? 2 : 1;
  PlaneSet ps = plane_set(h, rows, cols, nv);
  {
    Launch l(h, PM_K_PREP);
    launch_prep(h, ps, d_left, d_right, n, (size_t)cols);
  }
  if (int rc = launch_check(h, "prep")) return rc;
  const float* sl = d_seed_l;
  const float* sr = d_seed_r;
  PlArgs ar{};
  ar.dbg = planes_dbg();
  if (h->params.sparse_init) {
    // SparseInit on the device (patchmatch_gpu.cu:414-442) into the scalar engine's disparity planes, from
    // which the initialisation kernel takes the seeds (view 1's plane is already in mirrored coordinates)
    for (int v = 0; v < nv; ++v) {
      if (v == 0 ? sl != nullptr : sr != nullptr) continue;
      Launch l(h, PM_K_SEED);
      for (int b = 0; b < n; ++b)
        if (int rc = run_sparse_init(h, ps, b, v, 0)) return rc;
      ar.seed_in_disp |= 1 << v;
    }
  }
  ar.stage = PL_INIT;
  ar.view_fixed = -1;
  ar.seed_l = sl;
  ar.seed_r = sr;
  if (int rc = planes_stage<PL_INIT>(h, ps, ar, n * nv, PM_K_PL_INIT, "initialisation")) return rc;
  h->pl_rows = rows;
  h->pl_cols = cols;
  h->pl_n = n;
  h->pl_on = true;
  return PM_OK;
}

int planes_finish(pm_handle* h, float* d_disp_l, float* d_disp_r) {
  const int nv = h->params.left_right_check ? 2 : 1;
  const PlaneSet ps = plane_set(h, h->pl_rows, h->pl_cols, nv);
  const PlanesParams pp = planes_params(h->params);
  Launch l(h, PM_K_FINALIZE);
  if (h->params.state_dtype == PM_STATE_F16) {
    PlaneState<_Float16> st{(_Float16*)h->planes_state, ps.plane, ps.pitch / 2};
    hipLaunchKernelGGL(k_planes_finish<_Float16>, pixel_grid(ps.cols, ps.rows, h->pl_n), dim3(256), 0, h->stream, ps,
                       st, pp, d_disp_l, d_disp_r, (size_t)ps.cols);
  } else {
    PlaneState<float> st{(float*)h->planes_state, ps.plane, ps.pitch / 2};
    hipLaunchKernelGGL(k_planes_finish<float>, pixel_grid(ps.cols, ps.rows, h->pl_n), dim3(256), 0, h->stream, ps, st,
                       pp, d_disp_l, d_disp_r, (size_t)ps.cols);
  }
  return launch_check(h, "planes finish");
}

int planes_check(pm_handle* h, const char* what, bool need_begin) {
  if (!h) return PM_ERR_INVALID_ARG;
  if (h->params.mode != PM_MODE_PLANES) {
    set_err(h, "%s: the handle was created with mode != PM_MODE_PLANES", what);
    return PM_ERR_INVALID_ARG;
  }
  if (need_begin && !h->pl_on) {
    set_err(h, "%s: call pm_planes_begin (or a Match) first", what);
    return PM_ERR_INVALID_ARG;
  }
  PM_HIP(h, hipSetDevice(h->device));
  prof_break_all(h);  // (profiling brackets do not chain across API calls)
  return PM_OK;
}

}  // namespace

namespace pm {
namespace eng {

// The whole schedule of oracle/pm_planes_oracle.c::pmo_planes_match.
int planes_match(pm_handle* h, int n, const uint8_t* d_left, const uint8_t* d_right, int rows, int cols,
                 const float* d_seed_l, const float* d_seed_r, float* d_disp_l, float* d_disp_r) {
  if (int rc = planes_begin(h, n, d_left, d_right, rows, cols, d_seed_l, d_seed_r)) return rc;
  const int nv = h->params.left_right_check ? 2 : 1;
  const PlaneSet ps = plane_set(h, rows, cols, nv);
  // A batch runs as TWO LANES: the first half of its pairs through the iterations on the handle's stream, the second
  // half on view1_stream, launch k of both before launch k + 1 of either.  The 33 launches of a pair's iterations each end
  // with a tail in which a few tiles hold the chip; the other lane's launch fills it (two handles side by side matched
  // 560 pairs/s where one matched 469 / 521 / 552 with 1 / 2 / 4 pairs per launch: tools/multi_handle.py, round 6).
  const int n_lane[2] = {planes_two_lanes() && n >= 2 ? (n + 1) / 2 : n, planes_two_lanes() && n >= 2 ? n / 2 : 0};
  hipStream_t lane_stream[2] = {h->stream, h->stream};
  if (n_lane[1]) {
    if (int rc = lane_fork(h)) return rc;
    lane_stream[1] = h->view1_stream;
  }
  struct Restore {
    pm_handle* h;
    hipStream_t s;
    ~Restore() { h->stream = s; }
  } restore{h, h->stream};
  auto both = [&](int stage, int arg) -> int {
    for (int lane = 0; lane < 2; ++lane) {
      if (!n_lane[lane]) continue;
      h->stream = lane_stream[lane];  // every launch helper enqueues on h->stream
      if (int rc = planes_step(h, ps, n_lane[lane], stage, arg, lane ? n_lane[0] : 0)) return rc;
    }
    h->stream = restore.s;
    return PM_OK;
  };
  for (int it = 0; it < h->params.patchmatch_iters; ++it) {
    if (int rc = both(PM_PL_SPATIAL, 2 * it)) return rc;
    if (int rc = both(PM_PL_SPATIAL, 2 * it + 1)) return rc;
    // per view: view propagation then refinement, fused in one launch (one tile fill for 1 + R candidates)
    for (int v = 0; v < nv; ++v)
      if (int rc = both(PM_PL_VIEW_REFINE, it * 2 + v)) return rc;
  }
  h->stream = restore.s;
  if (n_lane[1])
    if (int rc = lane_join(h)) return rc;
  return planes_finish(h, d_disp_l, d_disp_r);
}

}  // namespace eng
}  // namespace pm

extern "C" {

int pm_planes_begin(pm_handle* h, int n, const uint8_t* d_left, const uint8_t* d_right, int rows, int cols,
                    const float* d_seed_l, const float* d_seed_r) {
  if (int rc = planes_check(h, "pm_planes_begin", false)) return rc;
  if (!d_left || !d_right) {
    set_err(h, "pm_planes_begin: null image pointer");
    return PM_ERR_INVALID_ARG;
  }
  if (int rc = check_size(h, rows, cols, n)) return rc;
  return planes_begin(h, n, d_left, d_right, rows, cols, d_seed_l, d_seed_r);
}

int pm_planes_step(pm_handle* h, int stage, int arg) {
  if (int rc = planes_check(h, "pm_planes_step", true)) return rc;
  const int nv = h->params.left_right_check ? 2 : 1;
  const bool ok = (stage == PM_PL_SPATIAL && arg >= 0 && arg < 2 * PM_MAX_ITERS) ||
                  (stage == PM_PL_VIEW && (arg == 0 || arg == 1)) ||
                  (stage == PM_PL_REFINE && arg >= 0 && arg < PM_MAX_ITERS) ||
                  (stage == PM_PL_VIEW_REFINE && arg >= 0 && arg < 2 * PM_MAX_ITERS);
  if (!ok) {
    set_err(h, "pm_planes_step: stage %d / argument %d out of range", stage, arg);
    return PM_ERR_INVALID_ARG;
  }
  return planes_step(h, plane_set(h, h->pl_rows, h->pl_cols, nv), h->pl_n, stage, arg);
}

static int planes_rw(pm_handle* h, int pair, int view, float* planes, int to_state, const char* what) {
  if (int rc = planes_check(h, what, true)) return rc;
  if (!planes || pair < 0 || pair >= h->pl_n || view < 0 || view > 1) {
    set_err(h, "%s: null buffer or pair / view out of range", what);
    return PM_ERR_INVALID_ARG;
  }
  const int nv = h->params.left_right_check ? 2 : 1;
  const PlaneSet ps = plane_set(h, h->pl_rows, h->pl_cols, nv);
  const size_t count = 4 * (size_t)ps.rows * ps.cols;
  // staged through the disparity staging buffers (4 * rows * cols floats fit st_disp_l .. only when max_batch
  // allows; a scratch allocation keeps this tool path independent of the plan)
  float* d_buf = nullptr;
  PM_HIP(h, hipMalloc((void**)&d_buf, sizeof(float) * count));
  int rc = PM_OK;
  if (to_state && hipMemcpyAsync(d_buf, planes, sizeof(float) * count, hipMemcpyHostToDevice, h->stream) != hipSuccess)
    rc = PM_ERR_HIP;
  if (rc == PM_OK) {
    const dim3 grid((unsigned)((ps.cols + 255) / 256), (unsigned)ps.rows, 4);
    if (h->params.state_dtype == PM_STATE_F16) {
      PlaneState<_Float16> st{(_Float16*)h->planes_state, ps.plane, ps.pitch / 2};
      hipLaunchKernelGGL(k_planes_copy<_Float16>, grid, dim3(256), 0, h->stream, ps, st, pair, view, d_buf, to_state);
    } else {
      PlaneState<float> st{(float*)h->planes_state, ps.plane, ps.pitch / 2};
      hipLaunchKernelGGL(k_planes_copy<float>, grid, dim3(256), 0, h->stream, ps, st, pair, view, d_buf, to_state);
    }
    rc = launch_check(h, what);
  }
  if (rc == PM_OK && !to_state &&
      hipMemcpyAsync(planes, d_buf, sizeof(float) * count, hipMemcpyDeviceToHost, h->stream) != hipSuccess)
    rc = PM_ERR_HIP;
  if (hipStreamSynchronize(h->stream) != hipSuccess && rc == PM_OK) rc = PM_ERR_HIP;
  (void)hipFree(d_buf);
  if (rc == PM_ERR_HIP && !h->err[0]) set_err(h, "%s: copy failed", what);
  return rc;
}

int pm_planes_read(pm_handle* h, int pair, int view, float* planes) {
  return planes_rw(h, pair, view, planes, 0, "pm_planes_read");
}
int pm_planes_write(pm_handle* h, int pair, int view, const float* planes) {
  return planes_rw(h, pair, view, const_cast<float*>(planes), 1, "pm_planes_write");
}

int pm_planes_finish(pm_handle* h, float* d_disp_l, float* d_disp_r) {
  if (int rc = planes_check(h, "pm_planes_finish", true)) return rc;
  if (!d_disp_l || (h->params.left_right_check && !d_disp_r)) {
    set_err(h, "pm_planes_finish: null output");
    return PM_ERR_INVALID_ARG;
  }
  return planes_finish(h, d_disp_l, d_disp_r);
}

}  // extern "C"
