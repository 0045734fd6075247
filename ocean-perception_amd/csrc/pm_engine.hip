// pm_engine.hip -- C-ABI implementation (include/pm/patchmatch.h) of the gfx950 PatchMatch
// stereo engine: handle, device memory plan, kernel launches, per-kernel timing.
//
// There is deliberately NO CPU fallback in this file: without a usable HIP device pm_create
// fails with PM_ERR_NO_DEVICE.  `file:line` citations are relative to the reference tree.
#include "pm/patchmatch.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "pm_kernels.hpp"
#include "pm_sweeps.hpp"
#include "pm_hostcopy.hpp"
#include "pm_internal.hpp"
#include "pm_seed.hpp"
#include "pm_planes.hpp"

using namespace pm;

namespace {

constexpr int kMaxEvents = 16384;  // event pairs kept before a forced drain

struct EventRec {
  hipEvent_t start, stop;
  int klass;
};

}  // namespace

struct pm_handle {
  pm_params params;
  int device = 0;
  int max_rows = 0, max_cols = 0, max_batch = 0;
  int max_pitch = 0;
  hipStream_t stream = nullptr;

  // engine planes (see pm::PlaneSet)
  uint8_t* img8 = nullptr;
  float* g32 = nullptr;
  uint8_t* g8 = nullptr;
  uint8_t* timg8 = nullptr;  // transposed copies for the column sweeps
  float* tg32 = nullptr;
  uint8_t* tg8 = nullptr;
  uint16_t* pk16 = nullptr;
  uint16_t* tpk16 = nullptr;
  float* rpg = nullptr;      // row / column PAIR planes of the run engine (pm::PlaneSet)
  uint32_t* rqk = nullptr;
  float* cpg = nullptr;
  float* disp = nullptr;
  float* cost = nullptr;
  float* noise = nullptr;
  unsigned long long* counters = nullptr;  // device, 8 words
  bool counters_on = false;                // same-address atomics serialise: opt-in only
  int noise_rows = 0, noise_cols = 0, noise_pitch = 0;

  // PM_MODE_PLANES: [max_batch][2 views][a, b, z, cost][rows][pitch], f32 or f16 (pm_planes.hpp)
  void* planes_state = nullptr;
  int pl_rows = 0, pl_cols = 0, pl_n = 0;  // what pm_planes_begin last prepared
  bool pl_on = false;

  SeedScratch seed{};   // scratch of the device seeder (pm_seed.hpp)
  SeedScratch seed2{};  // second set for the right view's seeder (allocated on first use; per-view streams)
  bool need_seed[2] = {false, false};  // set by pm_match_device: views whose seed map the device computes

  // row-tiled mode (pm_tile_*)
  bool tile_on = false;
  pm_tile tile{};
  int tile_band_rows = 0, tile_cols = 0;
  float* snap_disp = nullptr;  // snapshot of the disparity / cost planes (2 views)
  float* snap_cost = nullptr;
  size_t noise_capacity = 0;   // floats allocated for the noise table

  // staging for the host-buffer entry points: tightly packed [B][rows][cols]
  uint8_t* st_left = nullptr;
  uint8_t* st_right = nullptr;
  float* st_seed_l = nullptr;
  float* st_seed_r = nullptr;
  float* st_disp_l = nullptr;
  float* st_disp_r = nullptr;
  void* pinned = nullptr;  // host staging, pinned
  size_t pinned_bytes = 0;

  // pipelined host-buffer path (pm_submit_u8 / pm_collect): slot k of the staging buffers, uploads on
  // s_in, compute on `stream`, downloads on s_out
  struct PipeSlot {
    hipEvent_t in_done = nullptr, compute_done = nullptr, out_done = nullptr;
    uint64_t tag = 0;
    int rows = 0, cols = 0;
  };
  // per-view streams: the two views are independent until the cross-check, so their launch chains run on
  // two streams and one view's kernels fill the CUs the other view's kernel tails leave idle
  hipStream_t view_stream[2] = {nullptr, nullptr};
  hipEvent_t view_fork = nullptr, view_join[2] = {nullptr, nullptr};
  void* imaging_state = nullptr;  // owned by pm_imaging.hip (pm_internal.hpp)
  // pm_match_bgr_device: the next Match reads enhanced BGR inputs through k_prep_bgr instead of 8-bit gray images
  const pm::BgrSource* bgr = nullptr;
  hipGraphExec_t graph_exec = nullptr;  // pm_capture_* / pm_replay
  bool capturing = false;
  bool no_tiled = false;        // PM_NO_TILED (experiment knob), read once by pm_create
  hipEvent_t ext_fork = nullptr, ext_join = nullptr;  // pm_match_view_device: caller stream <-> handle stream
  hipEvent_t left_out = nullptr;  // pm_match_u8: the left map has arrived in the pinned buffer
  hipEvent_t right_out = nullptr;  // ... the right one
  pm::CopyPool* copy_pool = nullptr;  // host threads sharing the pack / unpack copies of the host-buffer entry points
  hipStream_t s_in = nullptr, s_out = nullptr;
  std::vector<PipeSlot> pipe;
  int pipe_head = 0, pipe_count = 0;

  // profiling
  bool profiling = false;
  std::vector<EventRec> ev_pool;
  int ev_used = 0;
  pm_profile prof{};

  char err[512] = {0};
};

namespace {

void set_err(pm_handle* h, const char* fmt, ...) {
  if (!h) return;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(h->err, sizeof(h->err), fmt, ap);
  va_end(ap);
}

#define PM_HIP(h, call)                                                                      \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      set_err((h), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return PM_ERR_HIP;                                                                     \
    }                                                                                        \
  } while (0)

inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

int check_patch(pm_handle* h, int pw, int ph) {
  if (pw < 3 || ph < 3 || pw > PM_MAX_PATCH || ph > PM_MAX_PATCH || (pw % 2) == 0 || (ph % 2) == 0) {
    // patchmatch.cpp:257-258 CHECKs oddness; 1x1 is excluded because its passes A/B and C/D visit
    // different pixel sets, which the engine's shared cost plane does not model.
    set_err(h, "patch %dx%d unsupported: sides must be odd and within [3, %d]", pw, ph, PM_MAX_PATCH);
    return PM_ERR_INVALID_ARG;
  }
  return PM_OK;
}

// cv::RNG(seed) + RNG::fill(CV_32F, UNIFORM, -1, 1) (OpenCV 3.4 modules/core/src/rand.cpp
// randf_32f): multiply-with-carry generator, out = (float)(int)next * 2^-31 + 0.  The generator
// is inherently sequential and the image depends only on (seed, rows, cols): it is built once per
// size on the host and uploaded, exactly as the reference does (patchmatch_gpu.cu:339-344).
void fill_unit_noise(float* dst, int rows, int cols, int pitch, uint64_t seed) {
  uint64_t state = seed ? seed : 0xffffffffu;
  const float scale = (float)(2.0 * 2.3283064365386962890625e-10);
  for (int y = 0; y < rows; ++y) {
    float* row = dst + (size_t)y * pitch;
    for (int x = 0; x < cols; ++x) {
      state = (uint64_t)(uint32_t)state * 4164903690u + (uint32_t)(state >> 32);
      const float t = (float)(int32_t)(uint32_t)state;
      const float m = t * scale;
      row[x] = m + 0.0f;
    }
    for (int x = cols; x < pitch; ++x) row[x] = 0.f;
  }
}

PlaneSet plane_set(const pm_handle* h, int rows, int cols, int n_views) {
  PlaneSet ps;
  ps.img8 = h->img8;
  ps.g32 = h->g32;
  ps.g8 = h->g8;
  ps.timg8 = h->timg8;
  ps.tg32 = h->tg32;
  ps.tg8 = h->tg8;
  ps.pitch_t = align_up(rows, 64);
  ps.plane_t = (size_t)(cols + kTransPad) * ps.pitch_t;
  ps.pk16 = h->pk16;
  ps.tpk16 = h->tpk16;
  ps.rpg = h->rpg;
  ps.rqk = h->rqk;
  ps.cpg = h->cpg;
  ps.nrl = rows + 2;
  ps.ncl = cols + kTransPad + 2;
  ps.disp = h->disp;
  ps.cost = h->cost;
  ps.noise = h->noise;
  ps.counters = h->counters_on ? h->counters : nullptr;
  ps.chain_mask = nullptr;
  ps.rows = rows;
  ps.cols = cols;
  ps.pitch = align_up(cols, 64);
  ps.n_views = n_views;
  ps.view_fixed = -1;
  ps.plane = (size_t)rows * ps.pitch;
  return ps;
}

CostParams cost_params(const pm_params& p, int pw, int ph) {
  CostParams cp;
  cp.semantics = p.semantics;
  cp.pw = p.semantics == PM_SEM_CPU ? pw : 3;
  cp.ph = p.semantics == PM_SEM_CPU ? ph : 3;
  cp.alpha = p.functor_alpha;
  cp.one_minus_alpha = 1.f - p.functor_alpha;
  cp.tau_color = p.functor_tau_color;
  cp.tau_grad = p.functor_tau_grad;
  cp.inv_n = 1. / (double)(cp.pw * cp.ph);
  cp.inv_n_hi = (float)cp.inv_n;
  cp.inv_n_lo = (float)(cp.inv_n - (double)cp.inv_n_hi);
  cp.g_alpha = p.cost_alpha;
  cp.g_one_minus_alpha = 1.f - p.cost_alpha;
  return cp;
}

// Pixels the sweeps visit.
//  PM_SEM_CPU (patchmatch.cpp:264-310): every pass skips y < ph/2, x < pw/2, y > h-ph/2-1,
//  x > w-pw/2-1; with sides >= 3 the loop bounds 1 / h-2 / w-2 lie outside that set, so all four
//  passes visit exactly [pw/2, w-pw/2-1] x [ph/2, h-ph/2-1].
//  PM_SEM_GPU (patchmatch_gpu.cu:134,143-144,192,201-202 with radius 1): union of the four sweeps
//  = [1, W-2] x [1, H-2]; each sweep's exclusive loop end trims one position (see sweep_geom).
Interior interior(const pm_params& p, int rows, int cols, int pw, int ph) {
  Interior in;
  if (p.semantics == PM_SEM_CPU) {
    in.x_lo = pw / 2;
    in.x_hi = cols - pw / 2 - 1;
    in.y_lo = ph / 2;
    in.y_hi = rows - ph / 2 - 1;
  } else {
    in.x_lo = 1;
    in.x_hi = cols - 2;
    in.y_lo = 1;
    in.y_hi = rows - 2;
  }
  return in;
}

// k-th sweep of an iteration: 0 = row +1 (pass A), 1 = col +1 (B), 2 = row -1 (C), 3 = col -1 (D).
SweepGeom sweep_geom(const pm_params& p, const Interior& in, int k) {
  SweepGeom g;
  g.axis = k & 1;
  g.dir = k < 2 ? 1 : -1;
  const int lo = g.axis == 0 ? in.x_lo : in.y_lo, hi = g.axis == 0 ? in.x_hi : in.y_hi;
  g.c_lo = g.axis == 0 ? in.y_lo : in.x_lo;
  g.c_hi = g.axis == 0 ? in.y_hi : in.x_hi;
  if (p.semantics == PM_SEM_CPU) {
    g.s_first = g.dir > 0 ? lo : hi;
    g.s_last = g.dir > 0 ? hi : lo;
  } else {
    // `for (col = start; dir > 0 ? col < end : col > end; col += dir)` (patchmatch_gpu.cu:156)
    g.s_first = g.dir > 0 ? lo : hi;
    g.s_last = g.dir > 0 ? hi - 1 : lo + 1;
  }
  return g;
}

struct Launch {
  pm_handle* h;
  int klass;
  bool timed;
  EventRec* rec = nullptr;
  Launch(pm_handle* h_, int k) : h(h_), klass(k), timed(h_->profiling) {
    if (!timed) return;
    if (h->ev_used == (int)h->ev_pool.size()) {
      if ((int)h->ev_pool.size() >= kMaxEvents) {
        timed = false;  // drained by pm_profile_read; never block inside a launch path
        return;
      }
      EventRec r;
      r.klass = k;
      if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) {
        timed = false;
        return;
      }
      h->ev_pool.push_back(r);
    }
    rec = &h->ev_pool[h->ev_used++];
    rec->klass = k;
    (void)hipEventRecord(rec->start, h->stream);
  }
  ~Launch() {
    if (timed && rec) (void)hipEventRecord(rec->stop, h->stream);
  }
};

int launch_check(pm_handle* h, const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_err(h, "launch of %s failed: %s", what, hipGetErrorString(e));
    return PM_ERR_HIP;
  }
  return PM_OK;
}

dim3 pixel_grid(int cols, int rows, int z) { return dim3((unsigned)((cols + 255) / 256), (unsigned)rows, (unsigned)z); }

// k_prep, or k_prep_bgr when the call came in through pm_match_bgr_device (the gray images are then never stored)
void launch_prep(pm_handle* h, const PlaneSet& ps, const uint8_t* d_left, const uint8_t* d_right, int n, size_t stride,
                 int view = -1) {
  if (h->bgr)
    hipLaunchKernelGGL(k_prep_bgr, dim3((unsigned)((ps.cols + 63) / 64), (unsigned)((ps.rows + 3) / 4), (unsigned)n),
                       dim3(256), 0, h->stream, ps, *h->bgr);
  else
    hipLaunchKernelGGL(k_prep, pixel_grid(ps.cols, ps.rows, n), dim3(256), 0, h->stream, ps, d_left, d_right, stride,
                       view);
}

// The line-triple planes (about 80 B per pixel and pair) serve the fixed-window kernels of pm_run3.hpp only: PM_SEM_CPU,
// scalar mode, the run engine.  Other handles (PM_SEM_GPU, PM_MODE_PLANES, the serial / wave anchors) neither
// allocate nor build them; allocation happens on the first call that builds them.
bool pair_planes_wanted(const pm_handle* h) {
  const pm_params& p = h->params;
  if (p.semantics != PM_SEM_CPU || p.mode != PM_MODE_SCALAR) return false;
  if (p.engine != PM_ENGINE_AUTO && p.engine != PM_ENGINE_RUNBLK2) return false;
  return true;
}
int pair_planes_alloc(pm_handle* h) {
  if (h->rpg) return PM_OK;
  const size_t B = (size_t)h->max_batch;
  const size_t pitch_t = (size_t)align_up(h->max_rows, 64);
  const size_t nrp = B * 2 * (size_t)(h->max_rows + 2) * h->max_pitch + 64;
  const size_t ncp = B * 2 * (size_t)(h->max_cols + kTransPad + 2) * pitch_t + 64;
  PM_HIP(h, hipMalloc((void**)&h->rpg, sizeof(float) * 4 * nrp));
  PM_HIP(h, hipMalloc((void**)&h->rqk, sizeof(uint32_t) * 2 * nrp));
  PM_HIP(h, hipMalloc((void**)&h->cpg, sizeof(float) * 4 * ncp));
  // row padding behind `cols` / `rows` is read (with weight 0 or by lanes out of reach) and must be finite
  PM_HIP(h, hipMemsetAsync(h->rpg, 0, sizeof(float) * 4 * nrp, h->stream));
  PM_HIP(h, hipMemsetAsync(h->rqk, 0, sizeof(uint32_t) * 2 * nrp, h->stream));
  PM_HIP(h, hipMemsetAsync(h->cpg, 0, sizeof(float) * 4 * ncp, h->stream));
  return PM_OK;
}

// transposed copies of the 12 image-type planes of n pairs (run by every path that ran k_prep)
// view >= 0: the planes of that view only (per-view streams: each stream derives its own planes)
int run_transpose(pm_handle* h, const PlaneSet& ps, int n, int view = -1) {
  SetupGrid sg{};
  sg.view = view;
  sg.tx = (unsigned)((ps.cols + 63) / 64);
  sg.ty = (unsigned)((ps.rows + 63) / 64);
  sg.tz = (unsigned)(n * (view < 0 ? 4 : 2));
  sg.with_lines = pair_planes_wanted(h) ? 1 : 0;  // the line-triple / quad planes of the run engine (pm_run3.hpp)
  PlaneSet pp = ps;
  unsigned blocks0 = 4 * sg.tx * sg.ty * sg.tz;
  if (sg.with_lines) {
    if (int rc = pair_planes_alloc(h)) return rc;
    pp.rpg = h->rpg;
    pp.rqk = h->rqk;
    pp.cpg = h->cpg;
    sg.lx = (unsigned)((ps.cols + 255) / 256);
    sg.ly = (unsigned)ps.nrl;
    sg.lz = (unsigned)(n * (view < 0 ? 2 : 1));
    sg.cx = (unsigned)((ps.rows + 255) / 256);
    sg.cy = (unsigned)ps.ncl;
    sg.cz = (unsigned)(n * (view < 0 ? 2 : 1));
    blocks0 += 2 * sg.lx * sg.ly * sg.lz;
  }
  hipLaunchKernelGGL(k_setup, dim3(blocks0), dim3(256), 0, h->stream, pp, sg, 0);
  if (sg.with_lines)
    hipLaunchKernelGGL(k_setup, dim3(sg.cx * sg.cy * sg.cz), dim3(256), 0, h->stream, pp, sg, 1);
  return launch_check(h, "transpose");
}

SeedParams seed_params(const pm_params& p) {
  SeedParams sp;
  sp.max_features = p.max_features_per_frame;
  sp.min_distance = p.min_distance_btw_features;
  sp.block_size = p.gftt_block_size;
  sp.templ_cols = p.templ_cols;
  sp.templ_rows = p.templ_rows;
  sp.max_disp = p.max_disp;
  sp.quality_level = p.gftt_quality_level;
  sp.max_matching_cost = p.max_matching_cost;
  return sp;
}

// SparseInit for view `view` of pair `b` straight into its disparity plane.  View 1 is seeded on the
// mirrored pair (patchmatch_gpu.cu:362-365), whose map is already in the mirrored coordinates the plane uses.
int alloc_seed_scratch(pm_handle* h, SeedScratch& sc) {
  const size_t plane = (size_t)h->max_rows * h->max_pitch;
  // every pixel can be a candidate: the 3x3 test is not strict, so plateaus of EQUAL responses (periodic images) pass
  // whole; a capacity of a quarter of the pixels dropped candidates there in whatever order the atomics fell
  sc.cap = (int)(plane + 64);
  PM_HIP(h, hipMalloc((void**)&sc.eig, sizeof(float) * plane));
  PM_HIP(h, hipMalloc((void**)&sc.keys, sizeof(unsigned long long) * sc.cap));
  PM_HIP(h, hipMalloc((void**)&sc.keys_sorted, sizeof(unsigned long long) * sc.cap));
  PM_HIP(h, hipMalloc((void**)&sc.counters, sizeof(unsigned) * kSeedCounters));
  PM_HIP(h, hipMalloc((void**)&sc.kp_xy, sizeof(int) * 2 * kSeedMaxFeatures));
  PM_HIP(h, hipMalloc((void**)&sc.kp_d, sizeof(float) * kSeedMaxFeatures));
  sc.sort_tmp = nullptr;
  sc.sort_tmp_bytes = 0;
  PM_HIP(h, hipcub::DeviceRadixSort::SortKeysDescending(nullptr, sc.sort_tmp_bytes, sc.keys, sc.keys_sorted, sc.cap,
                                                        0, 64, h->stream));
  PM_HIP(h, hipMalloc(&sc.sort_tmp, sc.sort_tmp_bytes));
  return PM_OK;
}

int run_sparse_init(pm_handle* h, const PlaneSet& ps, int b, int view, int scratch = 0) {
  if (scratch == 1 && !h->seed2.eig)
    if (int rc = alloc_seed_scratch(h, h->seed2)) return rc;
  SeedScratch& sc = scratch == 1 ? h->seed2 : h->seed;
  const uint8_t* ref = ps.img8 + ((size_t)b * 4 + (view == 0 ? 0 : 3)) * ps.plane;
  const uint8_t* tgt = ps.img8 + ((size_t)b * 4 + (view == 0 ? 1 : 2)) * ps.plane;
  float* out = ps.disp + ((size_t)b * 2 + view) * ps.plane;
  if (h->params.cpu_initialize_factor == 1)  // Patchmatch::Initialize(il, ir, 1) (patchmatch_test.cpp:149-150): 5x5, / 2
    PM_HIP(h, seed_initialize(sc, seed_params(h->params), ref, tgt, ps.rows, ps.cols, ps.pitch, 1, out, ps.pitch,
                              h->stream));
  else
    PM_HIP(h, seed_sparse_init(sc, seed_params(h->params), ref, tgt, ps.rows, ps.cols, ps.pitch,
                               h->params.init_dilate_factor, out, ps.pitch, h->stream));
  return PM_OK;
}

// The noise table depends only on (seed, rows, cols).  In tiled mode `rows` is the height of the WHOLE
// image (every tile adds the slice of the same table that belongs to its rows).
int ensure_noise(pm_handle* h, int rows, int cols) {
  const int pitch = align_up(cols, 64);
  if (h->noise_rows == rows && h->noise_cols == cols && h->noise_pitch == pitch) return PM_OK;
  if (h->capturing) {  // building the table synchronises and copies: not capturable
    set_err(h, "the noise table for %dx%d does not exist yet: match this size once before capturing", cols, rows);
    return PM_ERR_BUSY;
  }
  const size_t count = (size_t)rows * pitch;
  PM_HIP(h, hipStreamSynchronize(h->stream));
  if (count > h->noise_capacity) {
    if (h->noise) PM_HIP(h, hipFree(h->noise));
    h->noise = nullptr;
    PM_HIP(h, hipMalloc((void**)&h->noise, sizeof(float) * (count + 64)));
    h->noise_capacity = count;
  }
  std::vector<float> host(count);
  fill_unit_noise(host.data(), rows, cols, pitch, h->params.noise_seed);
  PM_HIP(h, hipMemcpy(h->noise, host.data(), sizeof(float) * count, hipMemcpyHostToDevice));
  h->noise_rows = rows;
  h->noise_cols = cols;
  h->noise_pitch = pitch;
  return PM_OK;
}

int check_size(pm_handle* h, int rows, int cols, int n) {
  if (rows < 8 || cols < 8) {
    set_err(h, "image %dx%d too small (min 8x8)", cols, rows);
    return PM_ERR_INVALID_ARG;
  }
  if (rows > h->max_rows || cols > h->max_cols || n > h->max_batch || n < 1) {
    set_err(h, "request %d x (%dx%d) exceeds plan %d x (%dx%d)", n, cols, rows, h->max_batch, h->max_cols,
            h->max_rows);
    return PM_ERR_SIZE;
  }
  return PM_OK;
}

int run_sweep(pm_handle* h, const PlaneSet& ps, const CostParams& cp, const SweepGeom& g, int slots,
              float amp = 1e30f) {
  const int chains = g.c_hi - g.c_lo + 1;
  if (chains <= 0 || (g.s_last - g.s_first) * g.dir < 0) return PM_OK;
  Launch l(h, g.axis == 0 ? PM_K_SWEEP_ROW : PM_K_SWEEP_COL);
  launch_sweep(ps, cp, g, slots, h->params.engine, amp, h->stream);  // pm_sweeps.hip
  return launch_check(h, "sweep");
}

// noise + clamp + cost of the current disparity; PM_SEM_CPU square windows use the LDS-tiled kernel
void launch_noise_cost(pm_handle* h, const PlaneSet& ps, const CostParams& cp, const Interior& in, float amount,
                       int slots, int keep_zero) {
  const bool tiled = cp.semantics == PM_SEM_CPU && cp.pw == cp.ph && !h->no_tiled;
  const dim3 tgrid((unsigned)((ps.cols + kTileW - 1) / kTileW), (unsigned)((ps.rows + kTileH - 1) / kTileH),
                   (unsigned)slots);
  if (tiled && cp.pw == 3) {
    hipLaunchKernelGGL((k_noise_cost_tiled<3, 3>), tgrid, dim3(256), 0, h->stream, ps, cp, in, amount, keep_zero);
  } else if (tiled && cp.pw == 5) {
    hipLaunchKernelGGL((k_noise_cost_tiled<5, 5>), tgrid, dim3(256), 0, h->stream, ps, cp, in, amount, keep_zero);
  } else if (tiled && cp.pw == 7) {
    hipLaunchKernelGGL((k_noise_cost_tiled<7, 7>), tgrid, dim3(256), 0, h->stream, ps, cp, in, amount, keep_zero);
  } else if (tiled && cp.pw == 9) {
    hipLaunchKernelGGL((k_noise_cost_tiled<9, 9>), tgrid, dim3(256), 0, h->stream, ps, cp, in, amount, keep_zero);
  } else if (tiled && cp.pw == 11) {
    hipLaunchKernelGGL((k_noise_cost_tiled<11, 11>), tgrid, dim3(256), 0, h->stream, ps, cp, in, amount, keep_zero);
  } else {
    hipLaunchKernelGGL(k_noise_cost, pixel_grid(ps.cols, ps.rows, slots), dim3(256), 0, h->stream, ps, cp, in, amount);
  }
}

// RemoveBackground / MaskBackground; PM_SEM_CPU square windows use the LDS-tiled kernel
void launch_background(pm_handle* h, const PlaneSet& ps, const CostParams& cp, const Interior& in, float factor,
                       int cached, int slots) {
  const bool tiled = cp.semantics == PM_SEM_CPU && cp.pw == cp.ph && !h->no_tiled;
  const dim3 tgrid((unsigned)((ps.cols + kTileW - 1) / kTileW), (unsigned)((ps.rows + kTileH - 1) / kTileH),
                   (unsigned)slots);
#define PM_BG_CASE(W)                                                                                        \
  case W:                                                                                                    \
    hipLaunchKernelGGL((k_background_tiled<W, W>), tgrid, dim3(256), 0, h->stream, ps, cp, in, factor, cached); \
    return;
  if (tiled) {
    switch (cp.pw) {
      PM_BG_CASE(3)
      PM_BG_CASE(5)
      PM_BG_CASE(7)
      PM_BG_CASE(9)
      PM_BG_CASE(11)
      default: break;
    }
  }
#undef PM_BG_CASE
  hipLaunchKernelGGL(k_background, pixel_grid(ps.cols, ps.rows, slots), dim3(256), 0, h->stream, ps, cp, in, factor,
                     cached);
}

// iterations {noise, 4 sweeps} + background for all slots: PatchmatchGpu::Match(GpuMat...)
// (patchmatch_gpu.cu:379-411) / the recipe of patchmatch_test.cpp:173-183.
// `sets` plane sets are advanced in step, each on its own stream (one set on the handle's stream, or the two
// views on their view streams): the host enqueues launch k of EVERY set before launch k + 1 of any, so all
// streams have work from the first microsecond on.  (Enqueuing one view's whole chain of ~45 launches first left
// the other stream empty for the 0.2-0.4 ms that takes: visible in the rocprofv3 kernel trace.)
int run_view_sets(pm_handle* h, const PlaneSet* pss, hipStream_t* streams, int sets, int slots) {
  const pm_params& p = h->params;
  hipStream_t keep = h->stream;
  struct Restore {
    pm_handle* h;
    hipStream_t s;
    ~Restore() { h->stream = s; }
  } restore{h, keep};
  CostParams cp{};
  int last_pw = 0, last_ph = 0;
  for (int it = 0; it < p.patchmatch_iters; ++it) {
    const int pw = p.patch_w[it], ph = p.patch_h[it];
    cp = cost_params(p, pw, ph);
    const Interior in = interior(p, pss[0].rows, pss[0].cols, cp.pw, cp.ph);
    for (int s = 0; s < sets; ++s) {
      h->stream = streams[s];  // every launch helper enqueues on h->stream
      {
        Launch l(h, PM_K_NOISE);
        // from the second iteration on the cost plane is valid for this window if the window is unchanged
        const int keep_zero = (it > 0 && cp.pw == last_pw && cp.ph == last_ph) ? 1 : 0;
        launch_noise_cost(h, pss[s], cp, in, p.noise_amp[it], slots, keep_zero);
      }
      if (int rc = launch_check(h, "noise_cost")) return rc;
    }
    for (int k = 0; k < 4; ++k)
      for (int s = 0; s < sets; ++s) {
        h->stream = streams[s];
        if (int rc = run_sweep(h, pss[s], cp, sweep_geom(p, in, k), slots, p.noise_amp[it])) return rc;
      }
    last_pw = cp.pw;
    last_ph = cp.ph;
  }
  for (int s = 0; s < sets; ++s) {
    h->stream = streams[s];
    const CostParams bcp = cost_params(p, p.bg_patch_w, p.bg_patch_h);
    const Interior in = interior(p, pss[s].rows, pss[s].cols, bcp.pw, bcp.ph);
    const int cached = (p.patchmatch_iters > 0 && bcp.pw == last_pw && bcp.ph == last_ph) ? 1 : 0;
    const float factor = p.semantics == PM_SEM_CPU ? p.win_by_factor : p.cost_improve_factor;
    {
      Launch l(h, PM_K_BACKGROUND);
      launch_background(h, pss[s], bcp, in, factor, cached, slots);
    }
    if (int rc = launch_check(h, "background")) return rc;
  }
  return PM_OK;
}
int run_one_view_set(pm_handle* h, const PlaneSet& ps, int slots) {
  hipStream_t s = h->stream;
  return run_view_sets(h, &ps, &s, 1, slots);
}

bool view_streams_enabled() {
  static bool v = [] {
    const char* e = getenv("PM_VIEW_STREAMS");
    return e ? atoi(e) != 0 : true;
  }();
  return v;
}

int seed_views(pm_handle* h, const PlaneSet& ps, int n_pairs, int view, int scratch) {
  if (!h->need_seed[view]) return PM_OK;
  Launch l(h, PM_K_SEED);
  for (int b = 0; b < n_pairs; ++b)
    if (int rc = run_sparse_init(h, ps, b, view, scratch)) return rc;
  return PM_OK;
}

// What a view stream needs to prepare its own planes (match_device_impl): with the views on their own streams the
// prep / transpose / line-plane / seed kernels of a view run at the head of that view's stream, so the two halves of
// the setup run side by side and neither view waits for the other's (round 2: eight launches in a row on the main
// stream, then a cross-stream event in front of each view).
struct ViewSetup {
  const uint8_t* d_left;
  const uint8_t* d_right;
  const float* d_seed_l;
  const float* d_seed_r;
  int n;
};

int run_views(pm_handle* h, const PlaneSet& ps, int slots, const ViewSetup* setup = nullptr) {
  if (ps.n_views != 2 || !view_streams_enabled()) {
    for (int v = 0; v < ps.n_views; ++v)
      if (int rc = seed_views(h, ps, slots / ps.n_views, v, 0)) return rc;
    return run_one_view_set(h, ps, slots);
  }
  if (!h->view_fork) {
    PM_HIP(h, hipEventCreateWithFlags(&h->view_fork, hipEventDisableTiming));
    for (int v = 0; v < 2; ++v) {
      PM_HIP(h, hipStreamCreateWithFlags(&h->view_stream[v], hipStreamNonBlocking));
      PM_HIP(h, hipEventCreateWithFlags(&h->view_join[v], hipEventDisableTiming));
    }
  }
  hipStream_t main_stream = h->stream;
  PM_HIP(h, hipEventRecord(h->view_fork, main_stream));
  int rc = PM_OK;
  PlaneSet pv[2] = {ps, ps};
  for (int v = 0; v < 2 && rc == PM_OK; ++v) {
    pv[v].view_fixed = v;
    if (hipStreamWaitEvent(h->view_stream[v], h->view_fork, 0) != hipSuccess) {
      rc = PM_ERR_HIP;
      break;
    }
    h->stream = h->view_stream[v];
    if (setup) {
      {
        Launch l(h, PM_K_PREP);
        launch_prep(h, ps, setup->d_left, setup->d_right, setup->n, (size_t)ps.cols, v);
        rc = launch_check(h, "prep");
        if (rc == PM_OK) rc = run_transpose(h, ps, setup->n, v);
      }
      if (rc == PM_OK) {
        Launch l(h, PM_K_SEED);
        hipLaunchKernelGGL(k_seed, pixel_grid(ps.cols, ps.rows, setup->n), dim3(256), 0, h->stream, ps, setup->d_seed_l,
                           setup->d_seed_r, (size_t)ps.cols, v);
        rc = launch_check(h, "seed");
      }
    }
    if (rc == PM_OK) rc = seed_views(h, ps, slots / 2, v, v);
    h->stream = main_stream;
  }
  if (rc == PM_OK) rc = run_view_sets(h, pv, h->view_stream, 2, slots / 2);
  for (int v = 0; v < 2 && rc == PM_OK; ++v)
    if (hipEventRecord(h->view_join[v], h->view_stream[v]) != hipSuccess) rc = PM_ERR_HIP;
  if (rc != PM_OK) {
    if (rc == PM_ERR_HIP && !h->err[0]) set_err(h, "per-view stream setup failed");
    return rc;
  }
  for (int v = 0; v < 2; ++v) PM_HIP(h, hipStreamWaitEvent(main_stream, h->view_join[v], 0));
  return PM_OK;
}

int validate_params(pm_handle* h, const pm_params& p) {
  if (p.struct_size != sizeof(pm_params) || p.abi_version != PM_ABI_VERSION) {
    set_err(h, "pm_params size/version mismatch (got %u/%u, want %zu/%d)", p.struct_size, p.abi_version,
            sizeof(pm_params), PM_ABI_VERSION);
    return PM_ERR_INVALID_ARG;
  }
  if (p.semantics != PM_SEM_CPU && p.semantics != PM_SEM_GPU) {
    set_err(h, "unknown semantics %d", p.semantics);
    return PM_ERR_INVALID_ARG;
  }
  if (p.engine != PM_ENGINE_AUTO && p.engine != PM_ENGINE_SERIAL && p.engine != PM_ENGINE_WAVE &&
      p.engine != PM_ENGINE_RUNBLK2) {
    set_err(h, "unknown engine %d", p.engine);
    return PM_ERR_INVALID_ARG;
  }
  if (p.patchmatch_iters < 0 || p.patchmatch_iters > PM_MAX_ITERS) {
    set_err(h, "patchmatch_iters %d outside [0, %d]", p.patchmatch_iters, PM_MAX_ITERS);
    return PM_ERR_INVALID_ARG;
  }
  if (p.semantics == PM_SEM_CPU) {
    for (int i = 0; i < p.patchmatch_iters; ++i)
      if (int rc = check_patch(h, p.patch_w[i], p.patch_h[i])) return rc;
    if (int rc = check_patch(h, p.bg_patch_w, p.bg_patch_h)) return rc;
    if (!(p.win_by_factor > 0.f)) {
      set_err(h, "win_by_factor must be > 0");
      return PM_ERR_INVALID_ARG;
    }
  }
  if (p.max_features_per_frame < 0 || p.max_features_per_frame > kSeedMaxFeatures || p.gftt_block_size < 1 ||
      (p.gftt_block_size % 2) == 0 || p.gftt_block_size > 15 || p.templ_cols < 1 || p.templ_rows < 1 ||
      ((p.mode != PM_MODE_PLANES || p.sparse_init) && p.max_disp < p.templ_cols) || p.init_dilate_factor < 0 ||
      p.init_dilate_factor > 8 ||
      // the template matcher keeps template + stripe in LDS and its sums in 32 bits
      (long long)p.templ_rows * p.templ_cols > 4096 ||
      (long long)p.templ_rows * p.templ_cols + (long long)(p.templ_rows + 2) * p.max_disp > 60 * 1024) {
    set_err(h, "seeder parameters out of range");
    return PM_ERR_INVALID_ARG;
  }
  for (int i = 0; i < p.patchmatch_iters; ++i)
    if (!(p.noise_amp[i] >= 0.f)) {
      set_err(h, "noise_amp[%d] must be >= 0", i);
      return PM_ERR_INVALID_ARG;
    }
  if (p.cpu_initialize_factor != 0 && p.cpu_initialize_factor != 1) {
    // Initialize(f > 1) also shrinks the map; Match() works at the image size, so only f = 1 (the reference's own
    // call, patchmatch_test.cpp:149) can seed it.  pm_initialize offers every factor as a stage.
    set_err(h, "cpu_initialize_factor must be 0 (SparseInit seeding) or 1 (Patchmatch::Initialize(il, ir, 1))");
    return PM_ERR_INVALID_ARG;
  }
  if (p.mode != PM_MODE_SCALAR && p.mode != PM_MODE_PLANES) {
    set_err(h, "unknown mode %d", p.mode);
    return PM_ERR_INVALID_ARG;
  }
  if (p.mode == PM_MODE_PLANES) {
    const int w = p.patch_w[0];
    if (w < 3 || w > PM_MAX_PATCH || (w % 2) == 0 || p.patch_h[0] != w) {
      set_err(h, "PM_MODE_PLANES: window patch_w[0] x patch_h[0] must be square, odd and within [3, %d]", PM_MAX_PATCH);
      return PM_ERR_INVALID_ARG;
    }
    if ((p.state_dtype != PM_STATE_F32 && p.state_dtype != PM_STATE_F16) || p.plane_refine_steps < 0 ||
        p.plane_refine_steps > 16 || !(p.plane_slope_max > 0.f) || !(p.plane_slope_max <= 4.f) ||
        !(p.plane_slope_init >= 0.f) || !(p.plane_slope_init <= p.plane_slope_max) ||
        !(p.plane_slope_per_disp >= 0.f) || !(p.plane_lr_tol >= 0.f) || p.max_disp < 1 || p.max_disp > 1024) {
      set_err(h, "PM_MODE_PLANES: plane parameters out of range");
      return PM_ERR_INVALID_ARG;
    }
  }
  return PM_OK;
}

// ---- PM_MODE_PLANES ------------------------------------------------------------------------------------------

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
  pp.inv_n = 1.0f / (float)(pp.patch * pp.patch);
  pp.lr_tol = p.plane_lr_tol;
  pp.seed = p.noise_seed;
  pp.n_views = p.left_right_check ? 2 : 1;
  return pp;
}

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

template <int STAGE>
int planes_stage(pm_handle* h, const PlaneSet& ps, const PlArgs& ar, int slots, int klass, const char* what) {
  Launch l(h, klass);
  const hipError_t e = pl_launch<STAGE>(ps, h->planes_state, h->params.state_dtype == PM_STATE_F16,
                                        planes_params(h->params), ar, slots, h->stream);
  if (e != hipSuccess) {
    set_err(h, "launch of planes %s failed: %s", what, hipGetErrorString(e));
    return PM_ERR_HIP;
  }
  return PM_OK;
}

int planes_step(pm_handle* h, const PlaneSet& ps, int n, int stage, int arg) {
  const int nv = ps.n_views;
  PlArgs ar{};
  ar.stage = stage;
  ar.arg = arg;
  ar.view_fixed = -1;
  switch (stage) {
    case PM_PL_SPATIAL:
      return planes_stage<PL_SPATIAL>(h, ps, ar, n * nv, PM_K_PL_SPATIAL, "spatial propagation");
    case PM_PL_VIEW:
      if (nv < 2) return PM_OK;
      ar.view_fixed = arg;
      return planes_stage<PL_VIEW>(h, ps, ar, n, PM_K_PL_VIEW, "view propagation");
    case PM_PL_REFINE:
      ar.refine_amp = h->params.noise_amp[arg];
      return planes_stage<PL_REFINE>(h, ps, ar, n * nv, PM_K_PL_REFINE, "refinement");
    case PM_PL_VIEW_REFINE: {  // arg = iteration * 2 + view: view propagation into `view`, then its refinement
      const int view = arg & 1, it = arg >> 1;
      if (view >= nv) return PM_OK;
      ar.arg = it;
      ar.view_fixed = view;
      ar.refine_amp = h->params.noise_amp[it];
      if (nv < 2) return planes_stage<PL_REFINE>(h, ps, ar, n, PM_K_PL_REFINE, "refinement");
      return planes_stage<PL_VIEW_REFINE>(h, ps, ar, n, PM_K_PL_VIEW_REFINE, "view propagation + refinement");
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

// The whole schedule of oracle/pm_planes_oracle.c::pmo_planes_match.
int planes_match(pm_handle* h, int n, const uint8_t* d_left, const uint8_t* d_right, int rows, int cols,
                 const float* d_seed_l, const float* d_seed_r, float* d_disp_l, float* d_disp_r) {
  if (int rc = planes_begin(h, n, d_left, d_right, rows, cols, d_seed_l, d_seed_r)) return rc;
  const int nv = h->params.left_right_check ? 2 : 1;
  const PlaneSet ps = plane_set(h, rows, cols, nv);
  for (int it = 0; it < h->params.patchmatch_iters; ++it) {
    if (int rc = planes_step(h, ps, n, PM_PL_SPATIAL, 0)) return rc;
    if (int rc = planes_step(h, ps, n, PM_PL_SPATIAL, 1)) return rc;
    // per view: view propagation then refinement, fused in one launch (one tile fill for 1 + R candidates)
    for (int v = 0; v < nv; ++v)
      if (int rc = planes_step(h, ps, n, PM_PL_VIEW_REFINE, it * 2 + v)) return rc;
  }
  return planes_finish(h, d_disp_l, d_disp_r);
}

}  // namespace

namespace {
// Ends a capture in progress and throws the partial graph away (error paths, pm_destroy).
void abort_capture(pm_handle* h) {
  if (!h->capturing) return;
  h->capturing = false;
  hipGraph_t graph = nullptr;
  (void)hipStreamEndCapture(h->stream, &graph);
  if (graph) (void)hipGraphDestroy(graph);
  (void)hipGetLastError();
}
int refuse_while_capturing(pm_handle* h, const char* what) {
  if (!h->capturing) return PM_OK;
  set_err(h, "%s: not allowed between pm_capture_begin and pm_capture_end", what);
  return PM_ERR_BUSY;
}
}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================

extern "C" {

void pm_params_default(pm_params* p, int semantics) {
  if (!p) return;
  std::memset(p, 0, sizeof(*p));
  p->struct_size = (uint32_t)sizeof(pm_params);
  p->abi_version = PM_ABI_VERSION;
  p->cost_alpha = 0.9f;           // patchmatch_gpu.h:85
  p->patchmatch_iters = 3;        // patchmatch_gpu.h:86
  p->init_dilate_factor = 4;      // patchmatch_gpu.h:87
  p->cost_improve_factor = 0.8f;  // patchmatch_gpu.h:88
  p->semantics = semantics;
  p->engine = PM_ENGINE_AUTO;
  for (int i = 0; i < PM_MAX_ITERS; ++i) {
    p->noise_amp[i] = (float)(32.0 / std::pow(2.0, (double)(float)i));  // patchmatch_gpu.cu:395
    p->patch_w[i] = 3;
    p->patch_h[i] = 3;
  }
  p->bg_patch_w = 3;  // patchmatch_test.cpp:183
  p->bg_patch_h = 3;
  p->win_by_factor = 1.5f;        // patchmatch_test.cpp:183
  p->functor_alpha = 0.7f;        // patchmatch_test.cpp:35
  p->functor_tau_color = 50.0f;   // patchmatch_test.cpp:36
  p->functor_tau_grad = 20.0f;    // patchmatch_test.cpp:37
  p->noise_seed = 123;            // patchmatch.cpp:146, patchmatch_gpu.cu:341
  p->left_right_check = 1;
  p->sparse_init = 0;
  p->max_features_per_frame = 200;   // feature_detector.hpp:28
  p->min_distance_btw_features = 20; // :31
  p->gftt_block_size = 5;            // :33
  p->gftt_quality_level = 0.01;      // :32
  p->templ_cols = 31;                // stereo_matcher.hpp:21
  p->templ_rows = 11;                // :22
  p->max_disp = 128;                 // :23
  p->max_matching_cost = 0.15;       // :24
  p->cpu_initialize_factor = 0;
  p->mode = PM_MODE_SCALAR;
  p->state_dtype = PM_STATE_F32;
  p->plane_refine_steps = 3;         // oracle/pm_planes_oracle.c: pmo_planes_params_default
  p->plane_slope_max = 1.0f;
  p->plane_slope_init = 0.25f;
  p->plane_slope_per_disp = 1.0f / 64.0f;
  p->plane_lr_tol = 1.0f;
}

const char* pm_status_string(int status) {
  switch (status) {
    case PM_OK: return "ok";
    case PM_ERR_INVALID_ARG: return "invalid argument";
    case PM_ERR_SIZE: return "size exceeds the handle's plan";
    case PM_ERR_HIP: return "HIP runtime error";
    case PM_ERR_NO_DEVICE: return "no usable HIP device";
    case PM_ERR_NOMEM: return "out of memory";
    case PM_ERR_BUSY: return "pipeline full / nothing to collect";
    default: return "unknown status";
  }
}

const char* pm_kernel_name(int k) {
  static const char* names[PM_K_COUNT] = {"prep", "seed", "noise_cost", "sweep_row", "sweep_col", "background",
                                          "finalize", "planes_init", "planes_spatial", "planes_view",
                                          "planes_refine", "planes_view_refine"};
  return (k >= 0 && k < PM_K_COUNT) ? names[k] : "?";
}

const char* pm_last_error(const pm_handle* h) { return h ? h->err : "null handle"; }

void pm_destroy(pm_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  abort_capture(h);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  pm_internal::release_imaging(h);
  if (h->ext_fork) (void)hipEventDestroy(h->ext_fork);
  if (h->ext_join) (void)hipEventDestroy(h->ext_join);
  if (h->left_out) (void)hipEventDestroy(h->left_out);
  if (h->right_out) (void)hipEventDestroy(h->right_out);
  if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
  for (auto& r : h->ev_pool) {
    (void)hipEventDestroy(r.start);
    (void)hipEventDestroy(r.stop);
  }
  void* dev[] = {h->rpg, h->rqk, h->cpg, h->img8, h->g32, h->g8, h->timg8, h->tg32, h->tg8, h->pk16, h->tpk16, h->disp, h->cost, h->noise, h->counters, h->st_left, h->st_right,
                 h->st_seed_l, h->st_seed_r, h->st_disp_l, h->st_disp_r, h->seed.eig, h->seed.keys, h->seed.keys_sorted,
                 h->seed.counters, h->seed.kp_xy, h->seed.kp_d, h->seed.sort_tmp, h->seed2.eig, h->seed2.keys, h->seed2.keys_sorted, h->seed2.counters, h->seed2.kp_xy, h->seed2.kp_d,
                 h->seed2.sort_tmp, h->snap_disp,
                 h->snap_cost, h->planes_state};
  for (void* p : dev)
    if (p) (void)hipFree(p);
  delete h->copy_pool;
  if (h->pinned) (void)hipHostFree(h->pinned);
  for (int v = 0; v < 2; ++v) {
    if (h->view_stream[v]) (void)hipStreamSynchronize(h->view_stream[v]);
    if (h->view_join[v]) (void)hipEventDestroy(h->view_join[v]);
    if (h->view_stream[v]) (void)hipStreamDestroy(h->view_stream[v]);
  }
  if (h->view_fork) (void)hipEventDestroy(h->view_fork);
  if (h->s_in) (void)hipStreamSynchronize(h->s_in);
  if (h->s_out) (void)hipStreamSynchronize(h->s_out);
  for (auto& sl : h->pipe) {
    if (sl.in_done) (void)hipEventDestroy(sl.in_done);
    if (sl.compute_done) (void)hipEventDestroy(sl.compute_done);
    if (sl.out_done) (void)hipEventDestroy(sl.out_done);
  }
  if (h->s_in) (void)hipStreamDestroy(h->s_in);
  if (h->s_out) (void)hipStreamDestroy(h->s_out);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int pm_create(const pm_params* params, int device, int max_rows, int max_cols, int max_batch, pm_handle** out) {
  if (!out) return PM_ERR_INVALID_ARG;
  *out = nullptr;
  if (!params || max_rows < 8 || max_cols < 8 || max_batch < 1) return PM_ERR_INVALID_ARG;
  pm_handle* h = new (std::nothrow) pm_handle();
  if (!h) return PM_ERR_NOMEM;
  // On failure the handle is still handed back so the caller can read pm_last_error(); such a
  // handle is good for pm_last_error / pm_destroy only.
  *out = h;
  h->params = *params;
  h->device = device;
  h->no_tiled = getenv("PM_NO_TILED") != nullptr;
  if (int rc = validate_params(h, *params)) return rc;
  // the sweep kernels address a view's planes with 32-bit byte offsets (12 bytes per pair element at most)
  if ((size_t)(max_rows + 64) * (size_t)(max_cols + 128) >= ((size_t)1 << 28)) {
    set_err(h, "plan of %dx%d exceeds the 2^28 pixels per view the kernels address", max_cols, max_rows);
    return PM_ERR_SIZE;
  }

  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    set_err(h, "no HIP device available (%s); this engine has no CPU fallback",
            e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return PM_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= count) {
    set_err(h, "device %d out of range (0..%d)", device, count - 1);
    return PM_ERR_NO_DEVICE;
  }
  PM_HIP(h, hipSetDevice(device));
  PM_HIP(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));

  h->max_rows = max_rows;
  h->max_cols = max_cols;
  h->max_batch = max_batch;
  h->max_pitch = align_up(max_cols, 64);
  const size_t plane = (size_t)max_rows * h->max_pitch;
  const size_t B = (size_t)max_batch;
  // +256 B of slack after the last plane: window loops may prefetch one element past a row end.
  PM_HIP(h, hipMalloc((void**)&h->img8, B * 4 * plane + 256));
  PM_HIP(h, hipMalloc((void**)&h->g32, sizeof(float) * (B * 4 * plane + 64)));
  PM_HIP(h, hipMalloc((void**)&h->g8, B * 4 * plane + 256));
  const size_t plane_t = (size_t)(max_cols + kTransPad) * align_up(max_rows, 64);
  PM_HIP(h, hipMalloc((void**)&h->timg8, B * 4 * plane_t + 256));
  PM_HIP(h, hipMalloc((void**)&h->tg32, sizeof(float) * (B * 4 * plane_t + 64)));
  PM_HIP(h, hipMalloc((void**)&h->tg8, B * 4 * plane_t + 256));
  // Row padding ([cols, pitch)) and the slack behind the last plane are read (with weight 0) by the
  // paired bilinear loads and never written afterwards: they must hold finite values.
  PM_HIP(h, hipMemsetAsync(h->img8, 0, B * 4 * plane + 256, h->stream));
  PM_HIP(h, hipMemsetAsync(h->g32, 0, sizeof(float) * (B * 4 * plane + 64), h->stream));
  PM_HIP(h, hipMemsetAsync(h->g8, 0, B * 4 * plane + 256, h->stream));
  PM_HIP(h, hipMemsetAsync(h->timg8, 0, B * 4 * plane_t + 256, h->stream));
  PM_HIP(h, hipMemsetAsync(h->tg32, 0, sizeof(float) * (B * 4 * plane_t + 64), h->stream));
  PM_HIP(h, hipMemsetAsync(h->tg8, 0, B * 4 * plane_t + 256, h->stream));
  PM_HIP(h, hipMalloc((void**)&h->pk16, sizeof(uint16_t) * (B * 4 * plane + 128)));
  PM_HIP(h, hipMalloc((void**)&h->tpk16, sizeof(uint16_t) * (B * 4 * plane_t + 128)));
  PM_HIP(h, hipMemsetAsync(h->pk16, 0, sizeof(uint16_t) * (B * 4 * plane + 128), h->stream));
  PM_HIP(h, hipMemsetAsync(h->tpk16, 0, sizeof(uint16_t) * (B * 4 * plane_t + 128), h->stream));
  if (pair_planes_wanted(h))
    if (int rc = pair_planes_alloc(h)) return rc;
  PM_HIP(h, hipMalloc((void**)&h->disp, sizeof(float) * (B * 2 * plane + 64)));
  PM_HIP(h, hipMalloc((void**)&h->cost, sizeof(float) * (B * 2 * plane + 64)));
  PM_HIP(h, hipMalloc((void**)&h->noise, sizeof(float) * (plane + 64)));
  h->noise_capacity = plane;
  PM_HIP(h, hipMalloc((void**)&h->counters, sizeof(unsigned long long) * 16));  // [8..13]: timing builds only
  PM_HIP(h, hipMemsetAsync(h->counters, 0, sizeof(unsigned long long) * 16, h->stream));
  if (int rc = alloc_seed_scratch(h, h->seed)) return rc;
  if (params->mode == PM_MODE_PLANES)
    if (int rc = planes_alloc(h)) return rc;
  const size_t tight = (size_t)max_rows * max_cols;
  PM_HIP(h, hipMalloc((void**)&h->st_left, B * tight));
  PM_HIP(h, hipMalloc((void**)&h->st_right, B * tight));
  PM_HIP(h, hipMalloc((void**)&h->st_seed_l, sizeof(float) * B * tight));
  PM_HIP(h, hipMalloc((void**)&h->st_seed_r, sizeof(float) * B * tight));
  PM_HIP(h, hipMalloc((void**)&h->st_disp_l, sizeof(float) * B * tight));
  PM_HIP(h, hipMalloc((void**)&h->st_disp_r, sizeof(float) * B * tight));
  // pinned host staging: per pair 2 u8 images + 2 seeds + 2 outputs (also used for the noise table)
  h->pinned_bytes = B * tight * (2 + 4 * sizeof(float));
  const size_t noise_bytes = sizeof(float) * plane;
  if (h->pinned_bytes < noise_bytes) h->pinned_bytes = noise_bytes;
  PM_HIP(h, hipHostMalloc(&h->pinned, h->pinned_bytes, hipHostMallocDefault));
  // the cost planes are read only where the noise kernel wrote them; clear once so that tools that
  // scan whole planes never see uninitialised memory
  PM_HIP(h, hipMemsetAsync(h->cost, 0, sizeof(float) * (B * 2 * plane + 64), h->stream));
  PM_HIP(h, hipMemsetAsync(h->disp, 0, sizeof(float) * (B * 2 * plane + 64), h->stream));
  PM_HIP(h, hipStreamSynchronize(h->stream));
  return PM_OK;
}

int pm_synchronize(pm_handle* h) {
  if (!h) return PM_ERR_INVALID_ARG;
  if (int rc = refuse_while_capturing(h, "pm_synchronize")) return rc;
  PM_HIP(h, hipSetDevice(h->device));
  PM_HIP(h, hipStreamSynchronize(h->stream));
  return PM_OK;
}

void* pm_stream(pm_handle* h) { return h ? (void*)h->stream : nullptr; }

// ---- HIP-graph replay of a recorded call sequence -----------------------------------------------------------
// Everything enqueued between pm_capture_begin and pm_capture_end (typically one pm_match_device with fixed
// device pointers) is recorded into a HIP graph instead of being executed -- the fork / join of the per-view
// streams included -- and pm_replay launches the whole DAG with one call.
int pm_capture_begin(pm_handle* h) {
  if (!h) return PM_ERR_INVALID_ARG;
  if (h->profiling) {
    set_err(h, "pm_capture_begin: per-kernel profiling must be off while capturing");
    return PM_ERR_INVALID_ARG;
  }
  if (h->capturing) {
    set_err(h, "pm_capture_begin: already capturing");
    return PM_ERR_BUSY;
  }
  if (h->pipe_count > 0) {
    set_err(h, "pm_capture_begin: pairs are in flight (pm_collect them first)");
    return PM_ERR_BUSY;
  }
  PM_HIP(h, hipSetDevice(h->device));
  PM_HIP(h, hipStreamSynchronize(h->stream));
  // lazily created resources must exist before the capture starts (creating them is not capturable)
  if (h->params.sparse_init && !h->seed2.eig)
    if (int rc = alloc_seed_scratch(h, h->seed2)) return rc;
  PM_HIP(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
  h->capturing = true;
  return PM_OK;
}

int pm_capture_end(pm_handle* h) {
  if (!h || !h->capturing) return PM_ERR_INVALID_ARG;
  h->capturing = false;
  hipGraph_t graph = nullptr;
  PM_HIP(h, hipStreamEndCapture(h->stream, &graph));
  if (h->graph_exec) {
    (void)hipGraphExecDestroy(h->graph_exec);
    h->graph_exec = nullptr;
  }
  const hipError_t e = hipGraphInstantiate(&h->graph_exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) {
    set_err(h, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
    return PM_ERR_HIP;
  }
  return PM_OK;
}

int pm_replay(pm_handle* h) {
  if (!h || !h->graph_exec) {
    if (h) set_err(h, "pm_replay: nothing captured");
    return PM_ERR_INVALID_ARG;
  }
  PM_HIP(h, hipSetDevice(h->device));
  PM_HIP(h, hipGraphLaunch(h->graph_exec, h->stream));
  return PM_OK;
}

static int match_device_impl(pm_handle* h, int n, const uint8_t* d_left, const uint8_t* d_right, int rows, int cols,
                             const float* d_seed_l, const float* d_seed_r, float* d_disp_l, float* d_disp_r) {
  if (!d_left || !d_right || !d_disp_l) {
    set_err(h, "pm_match_device: null image or output pointer");
    return PM_ERR_INVALID_ARG;
  }
  if (int rc = check_size(h, rows, cols, n)) return rc;
  PM_HIP(h, hipSetDevice(h->device));
  const int n_views = h->params.left_right_check ? 2 : 1;
  if (n_views == 2 && !d_disp_r) {
    set_err(h, "pm_match_device: disp_r required when left_right_check is set");
    return PM_ERR_INVALID_ARG;
  }
  if (h->params.mode == PM_MODE_PLANES)
    return planes_match(h, n, d_left, d_right, rows, cols, d_seed_l, d_seed_r, d_disp_l, d_disp_r);
  if (int rc = ensure_noise(h, rows, cols)) return rc;
  PlaneSet ps = plane_set(h, rows, cols, n_views);
  // a missing seed map is computed on the device, as the reference's Match() does (inside run_views, so that
  // the two views' seeders overlap on their own streams)
  h->need_seed[0] = h->params.sparse_init && !d_seed_l;
  h->need_seed[1] = h->params.sparse_init && !d_seed_r && n_views > 1;
  // two views on their own streams: each stream prepares its own planes (run_views); otherwise here
  const bool per_view_setup = n_views == 2 && view_streams_enabled() && !h->bgr;
  if (per_view_setup) {
    const ViewSetup vs{d_left, d_right, d_seed_l, d_seed_r, n};
    if (int rc = run_views(h, ps, n * n_views, &vs)) return rc;
  } else {
    {
      Launch l(h, PM_K_PREP);
      launch_prep(h, ps, d_left, d_right, n, (size_t)cols);
    }
    if (int rc = launch_check(h, "prep")) return rc;
    {
      Launch l(h, PM_K_PREP);
      if (int rc = run_transpose(h, ps, n)) return rc;
    }
    {
      Launch l(h, PM_K_SEED);
      hipLaunchKernelGGL(k_seed, pixel_grid(cols, rows, n), dim3(256), 0, h->stream, ps, d_seed_l, d_seed_r,
                         (size_t)cols, -1);
    }
    if (int rc = launch_check(h, "seed")) return rc;
    if (int rc = run_views(h, ps, n * n_views)) return rc;
  }
  {
    Launch l(h, PM_K_FINALIZE);
    hipLaunchKernelGGL(k_finalize, pixel_grid(cols, rows, n), dim3(256), 0, h->stream, ps, d_disp_l, d_disp_r,
                       (size_t)cols);
  }
  return launch_check(h, "finalize");
}

int pm_match_device(pm_handle* h, int n, const uint8_t* d_left, const uint8_t* d_right, int rows, int cols,
                    const float* d_seed_l, const float* d_seed_r, float* d_disp_l, float* d_disp_r) {
  if (!h) return PM_ERR_INVALID_ARG;
  const int rc = match_device_impl(h, n, d_left, d_right, rows, cols, d_seed_l, d_seed_r, d_disp_l, d_disp_r);
  if (rc != PM_OK) abort_capture(h);  // a failed call must not leave the stream in capture mode
  return rc;
}

int pm_match_view_device(pm_handle* h, const float* d_iml, const float* d_imr, const float* d_Gl, const float* d_Gr,
                         int rows, int cols, size_t step, float* d_disp, size_t disp_step, void* stream) {
  if (!h) return PM_ERR_INVALID_ARG;
  if (int rc = refuse_while_capturing(h, "pm_match_view_device")) return rc;
  if (!d_iml || !d_imr || !d_Gl || !d_Gr || !d_disp) {
    set_err(h, "pm_match_view_device: null pointer");
    return PM_ERR_INVALID_ARG;
  }
  if (h->params.mode != PM_MODE_SCALAR) {
    set_err(h, "pm_match_view_device: scalar-disparity mode only");
    return PM_ERR_INVALID_ARG;
  }
  if (int rc = check_size(h, rows, cols, 1)) return rc;
  if (step == 0) step = sizeof(float) * (size_t)cols;
  if (disp_step == 0) disp_step = sizeof(float) * (size_t)cols;
  if (step < sizeof(float) * (size_t)cols || disp_step < sizeof(float) * (size_t)cols || (step % sizeof(float)) ||
      (disp_step % sizeof(float))) {
    set_err(h, "pm_match_view_device: a row step is smaller than a row or not a multiple of 4");
    return PM_ERR_INVALID_ARG;
  }
  PM_HIP(h, hipSetDevice(h->device));
  if (int rc = ensure_noise(h, rows, cols)) return rc;
  hipStream_t user = (hipStream_t)stream;
  const bool foreign = user != nullptr && user != h->stream;
  if (foreign) {
    if (!h->ext_fork) {
      PM_HIP(h, hipEventCreateWithFlags(&h->ext_fork, hipEventDisableTiming));
      PM_HIP(h, hipEventCreateWithFlags(&h->ext_join, hipEventDisableTiming));
    }
    PM_HIP(h, hipEventRecord(h->ext_fork, user));
    PM_HIP(h, hipStreamWaitEvent(h->stream, h->ext_fork, 0));
  }
  PlaneSet ps = plane_set(h, rows, cols, 1);
  {
    Launch l(h, PM_K_PREP);
    hipLaunchKernelGGL(k_prep_view, pixel_grid(cols, rows, 1), dim3(256), 0, h->stream, ps, d_iml, d_imr, d_Gl, d_Gr,
                       step / sizeof(float));
  }
  if (int rc = launch_check(h, "prep_view")) return rc;
  {
    Launch l(h, PM_K_PREP);
    if (int rc = run_transpose(h, ps, 1)) return rc;
  }
  {
    Launch l(h, PM_K_SEED);
    hipLaunchKernelGGL(k_copy_disp_strided, pixel_grid(cols, rows, 1), dim3(256), 0, h->stream, ps, d_disp,
                       disp_step / sizeof(float), 0);
  }
  if (int rc = launch_check(h, "seed")) return rc;
  h->need_seed[0] = h->need_seed[1] = false;
  if (int rc = run_one_view_set(h, ps, 1)) return rc;
  {
    Launch l(h, PM_K_FINALIZE);
    hipLaunchKernelGGL(k_copy_disp_strided, pixel_grid(cols, rows, 1), dim3(256), 0, h->stream, ps, d_disp,
                       disp_step / sizeof(float), 1);
  }
  if (int rc = launch_check(h, "copy out")) return rc;
  if (foreign) {
    PM_HIP(h, hipEventRecord(h->ext_join, h->stream));
    PM_HIP(h, hipStreamWaitEvent(user, h->ext_join, 0));
  }
  return PM_OK;
}

int pm_set_unit_noise(pm_handle* h, const float* noise, int rows, int cols) {
  if (!h) return PM_ERR_INVALID_ARG;
  if (int rc = refuse_while_capturing(h, "pm_set_unit_noise")) return rc;
  if (!noise) {
    set_err(h, "pm_set_unit_noise: null pointer");
    return PM_ERR_INVALID_ARG;
  }
  if (int rc = check_size(h, rows, cols, 1)) return rc;
  PM_HIP(h, hipSetDevice(h->device));
  if (int rc = ensure_noise(h, rows, cols)) return rc;  // allocation + bookkeeping for this size
  const int pitch = align_up(cols, 64);
  PM_HIP(h, hipStreamSynchronize(h->stream));
  PM_HIP(h, hipMemcpy2D(h->noise, sizeof(float) * (size_t)pitch, noise, sizeof(float) * (size_t)cols,
                        sizeof(float) * (size_t)cols, (size_t)rows, hipMemcpyHostToDevice));
  return PM_OK;
}

int pm_match_batch_u8(pm_handle* h, int n, const uint8_t* const* left, const uint8_t* const* right, int rows,
                      int cols, const float* const* seed_l, const float* const* seed_r, float* const* disp_l,
                      float* const* disp_r) {
  if (!h) return PM_ERR_INVALID_ARG;
  if (int rc = refuse_while_capturing(h, "pm_match_batch_u8")) return rc;
  if (!left || !right || !disp_l) {
    set_err(h, "pm_match_batch_u8: null pointer array");
    return PM_ERR_INVALID_ARG;
  }
  if (int rc = check_size(h, rows, cols, n)) return rc;
  const bool lr = h->params.left_right_check != 0;
  if (lr && !disp_r) {
    set_err(h, "pm_match_batch_u8: disp_r required when left_right_check is set");
    return PM_ERR_INVALID_ARG;
  }
  PM_HIP(h, hipSetDevice(h->device));
  if (int rc = ensure_noise(h, rows, cols)) return rc;  // uses the pinned buffer: before staging inputs
  const size_t px = (size_t)rows * cols;
  float* psl = (float*)h->pinned;  // floats first so every sub-buffer stays 4-byte aligned
  float* psr = psl + n * px;
  float* pdl = psr + n * px;
  float* pdr = pdl + n * px;
  uint8_t* pl = (uint8_t*)(pdr + n * px);
  uint8_t* pr = pl + n * px;
  bool any_sl = false, any_sr = false;
  // With sparse_init a missing seed map means "seed this view on the device", which is decided per call, not per
  // pair: a batch must give the seed map of a view for every pair or for none.
  if (h->params.sparse_init) {
    int nl = 0, nr = 0;
    for (int i = 0; i < n; ++i) {
      nl += (seed_l && seed_l[i]) ? 1 : 0;
      nr += (seed_r && seed_r[i]) ? 1 : 0;
    }
    if ((nl != 0 && nl != n) || (nr != 0 && nr != n)) {
      set_err(h, "pm_match_batch_u8: with sparse_init a view's seed maps must be given for all pairs or for none");
      return PM_ERR_INVALID_ARG;
    }
  }
  for (int i = 0; i < n; ++i) {
    if (!left[i] || !right[i] || !disp_l[i] || (lr && !disp_r[i])) {
      set_err(h, "pm_match_batch_u8: null pointer for pair %d", i);
      return PM_ERR_INVALID_ARG;
    }
    std::memcpy(pl + i * px, left[i], px);
    std::memcpy(pr + i * px, right[i], px);
    if (seed_l && seed_l[i]) {
      std::memcpy(psl + i * px, seed_l[i], sizeof(float) * px);
      any_sl = true;
    } else {
      std::memset(psl + i * px, 0, sizeof(float) * px);
    }
    if (seed_r && seed_r[i]) {
      std::memcpy(psr + i * px, seed_r[i], sizeof(float) * px);
      any_sr = true;
    } else {
      std::memset(psr + i * px, 0, sizeof(float) * px);
    }
  }
  PM_HIP(h, hipMemcpyAsync(h->st_left, pl, n * px, hipMemcpyHostToDevice, h->stream));
  PM_HIP(h, hipMemcpyAsync(h->st_right, pr, n * px, hipMemcpyHostToDevice, h->stream));
  if (any_sl) PM_HIP(h, hipMemcpyAsync(h->st_seed_l, psl, sizeof(float) * n * px, hipMemcpyHostToDevice, h->stream));
  if (any_sr) PM_HIP(h, hipMemcpyAsync(h->st_seed_r, psr, sizeof(float) * n * px, hipMemcpyHostToDevice, h->stream));
  if (int rc = match_device_impl(h, n, h->st_left, h->st_right, rows, cols, any_sl ? h->st_seed_l : nullptr,
                               any_sr ? h->st_seed_r : nullptr, h->st_disp_l, lr ? h->st_disp_r : nullptr))
    return rc;
  PM_HIP(h, hipMemcpyAsync(pdl, h->st_disp_l, sizeof(float) * n * px, hipMemcpyDeviceToHost, h->stream));
  if (lr) PM_HIP(h, hipMemcpyAsync(pdr, h->st_disp_r, sizeof(float) * n * px, hipMemcpyDeviceToHost, h->stream));
  PM_HIP(h, hipStreamSynchronize(h->stream));
  for (int i = 0; i < n; ++i) {
    std::memcpy(disp_l[i], pdl + i * px, sizeof(float) * px);
    if (lr) std::memcpy(disp_r[i], pdr + i * px, sizeof(float) * px);
  }
  return PM_OK;
}

int pm_match_u8(pm_handle* h, const uint8_t* left, const uint8_t* right, int rows, int cols, size_t image_step,
                const float* seed_l, const float* seed_r, size_t seed_step, float* disp_l, float* disp_r,
                size_t disp_step) {
  if (!h) return PM_ERR_INVALID_ARG;
  if (int rc = refuse_while_capturing(h, "pm_match_u8")) return rc;
  if (!left || !right || !disp_l) {
    set_err(h, "pm_match_u8: null image or output pointer");
    return PM_ERR_INVALID_ARG;
  }
  if (int rc = check_size(h, rows, cols, 1)) return rc;
  const bool lr = h->params.left_right_check != 0;
  if (lr && !disp_r) {
    set_err(h, "pm_match_u8: disp_r required when left_right_check is set");
    return PM_ERR_INVALID_ARG;
  }
  if (image_step == 0) image_step = (size_t)cols;
  if (seed_step == 0) seed_step = sizeof(float) * (size_t)cols;
  if (disp_step == 0) disp_step = sizeof(float) * (size_t)cols;
  if (image_step < (size_t)cols || seed_step < sizeof(float) * (size_t)cols ||
      disp_step < sizeof(float) * (size_t)cols) {
    set_err(h, "pm_match_u8: a row step is smaller than a row");
    return PM_ERR_INVALID_ARG;
  }
  PM_HIP(h, hipSetDevice(h->device));
  if (int rc = ensure_noise(h, rows, cols)) return rc;
  const size_t px = (size_t)rows * cols;
  float* psl = (float*)h->pinned;
  float* psr = psl + px;
  float* pdl = psr + px;
  float* pdr = pdl + px;
  uint8_t* pl = (uint8_t*)(pdr + px);
  uint8_t* pr = pl + px;
  // every plane is packed into the pinned buffer (a few host threads share each copy, pm_hostcopy.hpp) and its upload
  // enqueued at once: the DMA of one plane runs while the host packs the next
  if (!h->copy_pool) h->copy_pool = new pm::CopyPool();
  auto pack = [&](void* dst, const void* src, size_t step, size_t row_bytes) {
    h->copy_pool->Copy2D(dst, row_bytes, src, step, row_bytes, rows);
  };
  pack(pl, left, image_step, (size_t)cols);
  PM_HIP(h, hipMemcpyAsync(h->st_left, pl, px, hipMemcpyHostToDevice, h->stream));
  pack(pr, right, image_step, (size_t)cols);
  PM_HIP(h, hipMemcpyAsync(h->st_right, pr, px, hipMemcpyHostToDevice, h->stream));
  if (seed_l) {
    pack(psl, seed_l, seed_step, sizeof(float) * (size_t)cols);
    PM_HIP(h, hipMemcpyAsync(h->st_seed_l, psl, sizeof(float) * px, hipMemcpyHostToDevice, h->stream));
  }
  if (seed_r) {
    pack(psr, seed_r, seed_step, sizeof(float) * (size_t)cols);
    PM_HIP(h, hipMemcpyAsync(h->st_seed_r, psr, sizeof(float) * px, hipMemcpyHostToDevice, h->stream));
  }
  if (int rc = match_device_impl(h, 1, h->st_left, h->st_right, rows, cols, seed_l ? h->st_seed_l : nullptr,
                               seed_r ? h->st_seed_r : nullptr, h->st_disp_l, lr ? h->st_disp_r : nullptr))
    return rc;
  // the left map is unpacked into the caller's buffer while the right one is still on the bus
  if (!h->left_out) {
    PM_HIP(h, hipEventCreateWithFlags(&h->left_out, hipEventDisableTiming));
    PM_HIP(h, hipEventCreateWithFlags(&h->right_out, hipEventDisableTiming));
  }
  PM_HIP(h, hipMemcpyAsync(pdl, h->st_disp_l, sizeof(float) * px, hipMemcpyDeviceToHost, h->stream));
  PM_HIP(h, hipEventRecord(h->left_out, h->stream));
  if (lr) {
    PM_HIP(h, hipMemcpyAsync(pdr, h->st_disp_r, sizeof(float) * px, hipMemcpyDeviceToHost, h->stream));
    PM_HIP(h, hipEventRecord(h->right_out, h->stream));
  }
  const size_t row_bytes = sizeof(float) * (size_t)cols;
  PM_HIP(h, hipEventSynchronize(h->left_out));
  h->copy_pool->Copy2D(disp_l, disp_step, pdl, row_bytes, row_bytes, rows);
  if (lr) {
    PM_HIP(h, hipEventSynchronize(h->right_out));
    h->copy_pool->Copy2D(disp_r, disp_step, pdr, row_bytes, row_bytes, rows);
  }
  return PM_OK;
}

// ---- pipelined host-buffer path ---------------------------------------------------------------------
// What the Sequence caller of the reference does frame by frame (patchmatch_gpu_test.cpp:118-128) with
// the copies taken off the critical path: while pair k is matched, pair k+1 is packed and uploaded and
// pair k-1 is downloaded.  Depth = max_batch of the plan.
namespace {

int pipe_init(pm_handle* h) {
  if (!h->pipe.empty()) return PM_OK;
  PM_HIP(h, hipStreamCreateWithFlags(&h->s_in, hipStreamNonBlocking));
  PM_HIP(h, hipStreamCreateWithFlags(&h->s_out, hipStreamNonBlocking));
  h->pipe.resize((size_t)h->max_batch);
  for (auto& sl : h->pipe) {
    PM_HIP(h, hipEventCreateWithFlags(&sl.in_done, hipEventDisableTiming));
    PM_HIP(h, hipEventCreateWithFlags(&sl.compute_done, hipEventDisableTiming));
    PM_HIP(h, hipEventCreateWithFlags(&sl.out_done, hipEventDisableTiming));
  }
  return PM_OK;
}

struct PinnedSlot {
  float *sl, *sr, *dl, *dr;
  uint8_t *l, *r;
};
PinnedSlot pinned_slot(pm_handle* h, int slot, size_t px) {
  const size_t tight = (size_t)h->max_rows * h->max_cols;
  char* base = (char*)h->pinned + (size_t)slot * tight * (2 + 4 * sizeof(float));
  PinnedSlot p;
  p.sl = (float*)base;
  p.sr = p.sl + px;
  p.dl = p.sr + px;
  p.dr = p.dl + px;
  p.l = (uint8_t*)(p.dr + px);
  p.r = p.l + px;
  return p;
}

}  // namespace

int pm_submit_u8(pm_handle* h, const uint8_t* left, const uint8_t* right, int rows, int cols, size_t image_step,
                 const float* seed_l, const float* seed_r, size_t seed_step, uint64_t tag) {
  if (!h) return PM_ERR_INVALID_ARG;
  if (int rc = refuse_while_capturing(h, "pm_submit_u8")) return rc;
  if (!left || !right) {
    set_err(h, "pm_submit_u8: null image pointer");
    return PM_ERR_INVALID_ARG;
  }
  if (int rc = check_size(h, rows, cols, 1)) return rc;
  if (image_step == 0) image_step = (size_t)cols;
  if (seed_step == 0) seed_step = sizeof(float) * (size_t)cols;
  if (image_step < (size_t)cols || seed_step < sizeof(float) * (size_t)cols) {
    set_err(h, "pm_submit_u8: a row step is smaller than a row");
    return PM_ERR_INVALID_ARG;
  }
  PM_HIP(h, hipSetDevice(h->device));
  if (int rc = pipe_init(h)) return rc;
  if (h->pipe_count == h->max_batch) {
    set_err(h, "pm_submit_u8: %d pairs in flight (the plan's max_batch); collect one first", h->pipe_count);
    return PM_ERR_BUSY;
  }
  if (h->noise_rows != rows || h->noise_cols != cols) {
    // the noise table is staged through the pinned buffer the slots live in
    if (h->pipe_count > 0) {
      set_err(h, "pm_submit_u8: image size changed with pairs in flight; collect them first");
      return PM_ERR_BUSY;
    }
    if (int rc = ensure_noise(h, rows, cols)) return rc;
    PM_HIP(h, hipStreamSynchronize(h->stream));
  }
  const int slot = (h->pipe_head + h->pipe_count) % h->max_batch;
  pm_handle::PipeSlot& sl = h->pipe[(size_t)slot];
  const size_t px = (size_t)rows * cols;
  const size_t tight = (size_t)h->max_rows * h->max_cols;
  const PinnedSlot ps = pinned_slot(h, slot, px);
  if (!h->copy_pool) h->copy_pool = new pm::CopyPool();
  h->copy_pool->Copy2D(ps.l, (size_t)cols, left, image_step, (size_t)cols, rows);
  h->copy_pool->Copy2D(ps.r, (size_t)cols, right, image_step, (size_t)cols, rows);
  if (seed_l) h->copy_pool->Copy2D(ps.sl, sizeof(float) * cols, seed_l, seed_step, sizeof(float) * cols, rows);
  if (seed_r) h->copy_pool->Copy2D(ps.sr, sizeof(float) * cols, seed_r, seed_step, sizeof(float) * cols, rows);
  uint8_t* dl8 = h->st_left + slot * tight;
  uint8_t* dr8 = h->st_right + slot * tight;
  float* dsl = h->st_seed_l + slot * tight;
  float* dsr = h->st_seed_r + slot * tight;
  float* ddl = h->st_disp_l + slot * tight;
  float* ddr = h->st_disp_r + slot * tight;
  const bool lr = h->params.left_right_check != 0;
  PM_HIP(h, hipMemcpyAsync(dl8, ps.l, px, hipMemcpyHostToDevice, h->s_in));
  PM_HIP(h, hipMemcpyAsync(dr8, ps.r, px, hipMemcpyHostToDevice, h->s_in));
  if (seed_l) PM_HIP(h, hipMemcpyAsync(dsl, ps.sl, sizeof(float) * px, hipMemcpyHostToDevice, h->s_in));
  if (seed_r) PM_HIP(h, hipMemcpyAsync(dsr, ps.sr, sizeof(float) * px, hipMemcpyHostToDevice, h->s_in));
  PM_HIP(h, hipEventRecord(sl.in_done, h->s_in));
  PM_HIP(h, hipStreamWaitEvent(h->stream, sl.in_done, 0));
  if (int rc = match_device_impl(h, 1, dl8, dr8, rows, cols, seed_l ? dsl : nullptr, seed_r ? dsr : nullptr, ddl,
                               lr ? ddr : nullptr))
    return rc;
  PM_HIP(h, hipEventRecord(sl.compute_done, h->stream));
  PM_HIP(h, hipStreamWaitEvent(h->s_out, sl.compute_done, 0));
  PM_HIP(h, hipMemcpyAsync(ps.dl, ddl, sizeof(float) * px, hipMemcpyDeviceToHost, h->s_out));
  if (lr) PM_HIP(h, hipMemcpyAsync(ps.dr, ddr, sizeof(float) * px, hipMemcpyDeviceToHost, h->s_out));
  PM_HIP(h, hipEventRecord(sl.out_done, h->s_out));
  sl.tag = tag;
  sl.rows = rows;
  sl.cols = cols;
  ++h->pipe_count;
  return PM_OK;
}

int pm_collect(pm_handle* h, float* disp_l, float* disp_r, size_t disp_step, uint64_t* tag) {
  if (!h) return PM_ERR_INVALID_ARG;
  if (int rc = refuse_while_capturing(h, "pm_collect")) return rc;
  if (h->pipe_count == 0) {
    set_err(h, "pm_collect: nothing in flight");
    return PM_ERR_BUSY;
  }
  const bool lr = h->params.left_right_check != 0;
  if (!disp_l || (lr && !disp_r)) {
    set_err(h, "pm_collect: null output pointer");
    return PM_ERR_INVALID_ARG;
  }
  pm_handle::PipeSlot& sl = h->pipe[(size_t)h->pipe_head];
  const int rows = sl.rows, cols = sl.cols;
  if (disp_step == 0) disp_step = sizeof(float) * (size_t)cols;
  if (disp_step < sizeof(float) * (size_t)cols) {
    set_err(h, "pm_collect: disp_step is smaller than a row");
    return PM_ERR_INVALID_ARG;
  }
  PM_HIP(h, hipSetDevice(h->device));
  PM_HIP(h, hipEventSynchronize(sl.out_done));
  const PinnedSlot ps = pinned_slot(h, h->pipe_head, (size_t)rows * cols);
  if (!h->copy_pool) h->copy_pool = new pm::CopyPool();
  h->copy_pool->Copy2D(disp_l, disp_step, ps.dl, sizeof(float) * cols, sizeof(float) * cols, rows);
  if (lr) h->copy_pool->Copy2D(disp_r, disp_step, ps.dr, sizeof(float) * cols, sizeof(float) * cols, rows);
  if (tag) *tag = sl.tag;
  h->pipe_head = (h->pipe_head + 1) % h->max_batch;
  --h->pipe_count;
  return PM_OK;
}

int pm_in_flight(const pm_handle* h) { return h ? h->pipe_count : 0; }

// ---- single stages ----------------------------------------------------------------------------

namespace {

// uploads a tightly packed pair into staging and runs prep for one pair / one view
int stage_prep(pm_handle* h, const uint8_t* left, const uint8_t* right, int rows, int cols, PlaneSet* ps_out) {
  if (int rc = check_size(h, rows, cols, 1)) return rc;
  PM_HIP(h, hipSetDevice(h->device));
  if (int rc = ensure_noise(h, rows, cols)) return rc;
  const size_t px = (size_t)rows * cols;
  PM_HIP(h, hipMemcpyAsync(h->st_left, left, px, hipMemcpyHostToDevice, h->stream));
  PM_HIP(h, hipMemcpyAsync(h->st_right, right ? right : left, px, hipMemcpyHostToDevice, h->stream));
  const PlaneSet ps = plane_set(h, rows, cols, 1);
  hipLaunchKernelGGL(k_prep, pixel_grid(cols, rows, 1), dim3(256), 0, h->stream, ps, h->st_left, h->st_right,
                     (size_t)cols, -1);
  if (int rc = launch_check(h, "prep")) return rc;
  if (int rc = run_transpose(h, ps, 1)) return rc;
  *ps_out = ps;
  return PM_OK;
}

int stage_disp_in(pm_handle* h, const PlaneSet& ps, const float* disp) {
  const size_t px = (size_t)ps.rows * ps.cols;
  PM_HIP(h, hipMemcpyAsync(h->st_disp_l, disp, sizeof(float) * px, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_copy_in, pixel_grid(ps.cols, ps.rows, 1), dim3(256), 0, h->stream, ps, h->st_disp_l);
  return launch_check(h, "copy_in");
}

int stage_out(pm_handle* h, const PlaneSet& ps, float* dst, int which) {
  const size_t px = (size_t)ps.rows * ps.cols;
  hipLaunchKernelGGL(k_copy_out, pixel_grid(ps.cols, ps.rows, 1), dim3(256), 0, h->stream, ps, h->st_disp_l, which);
  if (int rc = launch_check(h, "copy_out")) return rc;
  PM_HIP(h, hipMemcpyAsync(dst, h->st_disp_l, sizeof(float) * px, hipMemcpyDeviceToHost, h->stream));
  PM_HIP(h, hipStreamSynchronize(h->stream));
  return PM_OK;
}

}  // namespace

int pm_gradient_magnitude(pm_handle* h, const uint8_t* image, int rows, int cols, float* grad) {
  if (!h) return PM_ERR_INVALID_ARG;
  if (int rc = refuse_while_capturing(h, "pm_gradient_magnitude")) return rc;
  if (!image || !grad) {
    set_err(h, "pm_gradient_magnitude: null pointer");
    return PM_ERR_INVALID_ARG;
  }
  PlaneSet ps;
  if (int rc = stage_prep(h, image, nullptr, rows, cols, &ps)) return rc;
  return stage_out(h, ps, grad, 1);
}

int pm_unit_noise(pm_handle* h, int rows, int cols, float* noise) {
  if (!h) return PM_ERR_INVALID_ARG;
  if (int rc = refuse_while_capturing(h, "pm_unit_noise")) return rc;
  if (!noise) {
    set_err(h, "pm_unit_noise: null pointer");
    return PM_ERR_INVALID_ARG;
  }
  if (int rc = check_size(h, rows, cols, 1)) return rc;
  PM_HIP(h, hipSetDevice(h->device));
  if (int rc = ensure_noise(h, rows, cols)) return rc;
  return stage_out(h, plane_set(h, rows, cols, 1), noise, 2);
}

int pm_add_noise(pm_handle* h, float* disp, int rows, int cols, float amount) {
  if (!h) return PM_ERR_INVALID_ARG;
  if (int rc = refuse_while_capturing(h, "pm_add_noise")) return rc;
  if (!disp || !(amount >= 0.f)) {
    set_err(h, "pm_add_noise: null pointer or negative amount");
    return PM_ERR_INVALID_ARG;
  }
  if (int rc = check_size(h, rows, cols, 1)) return rc;
  PM_HIP(h, hipSetDevice(h->device));
  if (int rc = ensure_noise(h, rows, cols)) return rc;
  const PlaneSet ps = plane_set(h, rows, cols, 1);
  if (int rc = stage_disp_in(h, ps, disp)) return rc;
  CostParams cp = cost_params(h->params, 3, 3);
  const Interior none{1, 0, 1, 0};  // empty: noise only, no clamp / cost
  hipLaunchKernelGGL(k_noise_cost, pixel_grid(cols, rows, 1), dim3(256), 0, h->stream, ps, cp, none, amount);
  if (int rc = launch_check(h, "noise")) return rc;
  return stage_out(h, ps, disp, 0);
}

int pm_propagate(pm_handle* h, const uint8_t* left, const uint8_t* right, int rows, int cols, float* disp,
                 int patch_h, int patch_w, int pass_mask) {
  if (!h) return PM_ERR_INVALID_ARG;
  if (int rc = refuse_while_capturing(h, "pm_propagate")) return rc;
  if (!left || !right || !disp) {
    set_err(h, "pm_propagate: null pointer");
    return PM_ERR_INVALID_ARG;
  }
  if (h->params.semantics == PM_SEM_CPU)
    if (int rc = check_patch(h, patch_w, patch_h)) return rc;
  PlaneSet ps;
  if (int rc = stage_prep(h, left, right, rows, cols, &ps)) return rc;
  if (int rc = stage_disp_in(h, ps, disp)) return rc;
  const CostParams cp = cost_params(h->params, patch_w, patch_h);
  const Interior in = interior(h->params, rows, cols, cp.pw, cp.ph);
  launch_noise_cost(h, ps, cp, in, -1.f, 1, 0);
  if (int rc = launch_check(h, "cost")) return rc;
  for (int k = 0; k < 4; ++k)
    if (pass_mask & (1 << k))
      if (int rc = run_sweep(h, ps, cp, sweep_geom(h->params, in, k), 1)) return rc;
  return stage_out(h, ps, disp, 0);
}

int pm_remove_background(pm_handle* h, const uint8_t* left, const uint8_t* right, int rows, int cols, float* disp,
                         int patch_h, int patch_w, float factor) {
  if (!h) return PM_ERR_INVALID_ARG;
  if (int rc = refuse_while_capturing(h, "pm_remove_background")) return rc;
  if (!left || !right || !disp || !(factor > 0.f)) {
    set_err(h, "pm_remove_background: null pointer or non-positive factor");
    return PM_ERR_INVALID_ARG;
  }
  if (h->params.semantics == PM_SEM_CPU)
    if (int rc = check_patch(h, patch_w, patch_h)) return rc;
  PlaneSet ps;
  if (int rc = stage_prep(h, left, right, rows, cols, &ps)) return rc;
  if (int rc = stage_disp_in(h, ps, disp)) return rc;
  const CostParams cp = cost_params(h->params, patch_w, patch_h);
  const Interior in = interior(h->params, rows, cols, cp.pw, cp.ph);
  launch_background(h, ps, cp, in, factor, 0, 1);
  if (int rc = launch_check(h, "background")) return rc;
  return stage_out(h, ps, disp, 0);
}

int pm_sparse_init(pm_handle* h, const uint8_t* left, const uint8_t* right, int rows, int cols, int dilate_factor,
                   float* seed) {
  if (!h) return PM_ERR_INVALID_ARG;
  if (int rc = refuse_while_capturing(h, "pm_sparse_init")) return rc;
  if (!left || !right || !seed || dilate_factor < 0 || dilate_factor > 8) {
    set_err(h, "pm_sparse_init: null pointer or dilate_factor outside [0, 8]");
    return PM_ERR_INVALID_ARG;
  }
  PlaneSet ps;
  if (int rc = stage_prep(h, left, right, rows, cols, &ps)) return rc;
  PM_HIP(h, seed_sparse_init(h->seed, seed_params(h->params), ps.img8, ps.img8 + ps.plane, rows, cols, ps.pitch,
                             dilate_factor, ps.disp, ps.pitch, h->stream));
  return stage_out(h, ps, seed, 0);
}

int pm_initialize(pm_handle* h, const uint8_t* left, const uint8_t* right, int rows, int cols, int downsample_factor,
                  float* seed) {
  if (!h) return PM_ERR_INVALID_ARG;
  if (int rc = refuse_while_capturing(h, "pm_initialize")) return rc;
  if (!left || !right || !seed || downsample_factor < 1 || downsample_factor > 8 || rows / downsample_factor < 1 ||
      cols / downsample_factor < 1) {
    set_err(h, "pm_initialize: null pointer or downsample_factor outside [1, 8]");
    return PM_ERR_INVALID_ARG;
  }
  PlaneSet ps;
  if (int rc = stage_prep(h, left, right, rows, cols, &ps)) return rc;
  const int orows = rows / downsample_factor, ocols = cols / downsample_factor;
  PM_HIP(h, seed_initialize(h->seed, seed_params(h->params), ps.img8, ps.img8 + ps.plane, rows, cols, ps.pitch,
                            downsample_factor, h->st_disp_l, ocols, h->stream));
  PM_HIP(h, hipMemcpyAsync(seed, h->st_disp_l, sizeof(float) * (size_t)orows * ocols, hipMemcpyDeviceToHost, h->stream));
  PM_HIP(h, hipStreamSynchronize(h->stream));
  return PM_OK;
}

int pm_mask_occlusions(pm_handle* h, float* disp_l, const float* disp_r, int rows, int cols) {
  if (!h) return PM_ERR_INVALID_ARG;
  if (int rc = refuse_while_capturing(h, "pm_mask_occlusions")) return rc;
  if (!disp_l || !disp_r) {
    set_err(h, "pm_mask_occlusions: null pointer");
    return PM_ERR_INVALID_ARG;
  }
  if (int rc = check_size(h, rows, cols, 1)) return rc;
  PM_HIP(h, hipSetDevice(h->device));
  const size_t px = (size_t)rows * cols;
  PM_HIP(h, hipMemcpyAsync(h->st_disp_l, disp_l, sizeof(float) * px, hipMemcpyHostToDevice, h->stream));
  PM_HIP(h, hipMemcpyAsync(h->st_disp_r, disp_r, sizeof(float) * px, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_mask_occlusions, pixel_grid(cols, rows, 1), dim3(256), 0, h->stream, h->st_disp_l,
                     h->st_disp_r, rows, cols);
  if (int rc = launch_check(h, "mask_occlusions")) return rc;
  PM_HIP(h, hipMemcpyAsync(disp_l, h->st_disp_l, sizeof(float) * px, hipMemcpyDeviceToHost, h->stream));
  PM_HIP(h, hipStreamSynchronize(h->stream));
  return PM_OK;
}


// ---- row-tiled mode --------------------------------------------------------------------------------------

namespace {

// PlaneSet of the band with the noise pointer moved to the band's slice of the whole-image table.
PlaneSet tile_plane_set(pm_handle* h) {
  const int n_views = h->params.left_right_check ? 2 : 1;
  PlaneSet ps = plane_set(h, h->tile_band_rows, h->tile_cols, n_views);
  ps.noise = h->noise + (size_t)h->tile.band_row0 * ps.pitch;
  return ps;
}

// Rows this tile sweeps: owned rows that the sweeps of the whole image visit, in band coordinates.
Interior tile_interior(pm_handle* h, int pw, int ph) {
  const pm_tile& t = h->tile;
  Interior in = interior(h->params, t.global_rows, h->tile_cols, pw, ph);  // whole-image rows / columns
  const int lo = in.y_lo > t.own_row0 ? in.y_lo : t.own_row0;
  const int hi = in.y_hi < t.own_row0 + t.own_rows - 1 ? in.y_hi : t.own_row0 + t.own_rows - 1;
  in.y_lo = lo - t.band_row0;
  in.y_hi = hi - t.band_row0;
  return in;
}

int tile_check(pm_handle* h, const char* what) {
  if (!h) return PM_ERR_INVALID_ARG;
  if (!h->tile_on) {
    set_err(h, "%s: call pm_tile_begin first", what);
    return PM_ERR_INVALID_ARG;
  }
  PM_HIP(h, hipSetDevice(h->device));
  return PM_OK;
}

}  // namespace

int pm_tile_begin(pm_handle* h, const pm_tile* tile, const uint8_t* d_left_band, const uint8_t* d_right_band,
                  int band_rows, int cols, const float* d_seed_l_band, const float* d_seed_r_band) {
  if (!h) return PM_ERR_INVALID_ARG;
  if (!tile || !d_left_band || !d_right_band) {
    set_err(h, "pm_tile_begin: null pointer");
    return PM_ERR_INVALID_ARG;
  }
  if (int rc = check_size(h, band_rows, cols, 1)) return rc;
  const pm_params& p = h->params;
  int max_ph = p.bg_patch_h;
  for (int i = 0; i < p.patchmatch_iters; ++i) max_ph = p.patch_h[i] > max_ph ? p.patch_h[i] : max_ph;
  const int halo = (p.semantics == PM_SEM_CPU ? max_ph / 2 : 1) + 1;  // window rows + one more for the Sobel
  const int need_top = tile->own_row0 - halo > 0 ? tile->own_row0 - halo : 0;
  const int own_end = tile->own_row0 + tile->own_rows;
  const int need_end = own_end + halo < tile->global_rows ? own_end + halo : tile->global_rows;
  if (tile->own_rows < 1 || tile->own_row0 < 0 || own_end > tile->global_rows || tile->band_row0 < 0 ||
      tile->band_row0 > need_top || tile->band_row0 + band_rows < need_end ||
      tile->band_row0 + band_rows > tile->global_rows) {
    set_err(h, "pm_tile_begin: band [%d, %d) must cover rows [%d, %d) (owned [%d, %d) + %d halo rows) of %d",
            tile->band_row0, tile->band_row0 + band_rows, need_top, need_end, tile->own_row0, own_end, halo,
            tile->global_rows);
    return PM_ERR_INVALID_ARG;
  }
  PM_HIP(h, hipSetDevice(h->device));
  if (int rc = ensure_noise(h, tile->global_rows, cols)) return rc;
  h->tile = *tile;
  h->tile_band_rows = band_rows;
  h->tile_cols = cols;
  h->tile_on = true;
  const PlaneSet ps = tile_plane_set(h);
  hipLaunchKernelGGL(k_prep, pixel_grid(cols, band_rows, 1), dim3(256), 0, h->stream, ps, d_left_band, d_right_band,
                     (size_t)cols, -1);
  if (int rc = launch_check(h, "prep")) return rc;
  if (int rc = run_transpose(h, ps, 1)) return rc;
  hipLaunchKernelGGL(k_seed, pixel_grid(cols, band_rows, 1), dim3(256), 0, h->stream, ps, d_seed_l_band,
                     d_seed_r_band, (size_t)cols, -1);
  return launch_check(h, "seed");
}

int pm_tile_noise(pm_handle* h, int it) {
  if (int rc = tile_check(h, "pm_tile_noise")) return rc;
  const pm_params& p = h->params;
  if (it < 0 || it >= p.patchmatch_iters) {
    set_err(h, "pm_tile_noise: iteration %d out of range", it);
    return PM_ERR_INVALID_ARG;
  }
  const PlaneSet ps = tile_plane_set(h);
  const CostParams cp = cost_params(p, p.patch_w[it], p.patch_h[it]);
  const Interior in = tile_interior(h, cp.pw, cp.ph);
  launch_noise_cost(h, ps, cp, in, p.noise_amp[it], ps.n_views, 0);
  return launch_check(h, "noise_cost");
}

static int tile_sweep(pm_handle* h, int it, int k, const int* d_mask);
int pm_tile_sweep(pm_handle* h, int it, int k) { return tile_sweep(h, it, k, nullptr); }
int pm_tile_sweep_masked(pm_handle* h, int it, int k, const int* d_mask) {
  if (!h) return PM_ERR_INVALID_ARG;
  if (!d_mask) {
    set_err(h, "pm_tile_sweep_masked: null mask");
    return PM_ERR_INVALID_ARG;
  }
  return tile_sweep(h, it, k, d_mask);
}

static int tile_sweep(pm_handle* h, int it, int k, const int* d_mask) {
  if (int rc = tile_check(h, "pm_tile_sweep")) return rc;
  const pm_params& p = h->params;
  if (it < 0 || it >= p.patchmatch_iters || k < 0 || k > 3) {
    set_err(h, "pm_tile_sweep: iteration %d / sweep %d out of range", it, k);
    return PM_ERR_INVALID_ARG;
  }
  PlaneSet ps = tile_plane_set(h);
  if (d_mask && (k & 1) == 0) {
    set_err(h, "pm_tile_sweep_masked: the column mask applies to the vertical sweeps (k = 1, 3)");
    return PM_ERR_INVALID_ARG;
  }
  ps.chain_mask = d_mask;
  const CostParams cp = cost_params(p, p.patch_w[it], p.patch_h[it]);
  // Geometry of the sweep on the WHOLE image (PM_SEM_GPU trims one position at the far end of each sweep,
  // patchmatch_gpu.cu:156,214 -- that end is an end of the image, not of a band), then cut to the owned rows.
  const pm_tile& t = h->tile;
  SweepGeom g = sweep_geom(p, interior(p, t.global_rows, h->tile_cols, cp.pw, cp.ph), k);
  const int own_lo = t.own_row0, own_hi = t.own_row0 + t.own_rows - 1;
  if (g.axis == 0) {  // chains are rows
    g.c_lo = (g.c_lo > own_lo ? g.c_lo : own_lo) - t.band_row0;
    g.c_hi = (g.c_hi < own_hi ? g.c_hi : own_hi) - t.band_row0;
    if (g.c_hi < g.c_lo) return PM_OK;
  } else {  // positions along a chain are rows
    if (g.dir > 0) {
      g.s_first = g.s_first > own_lo ? g.s_first : own_lo;
      g.s_last = g.s_last < own_hi ? g.s_last : own_hi;
      if (g.s_last < g.s_first) return PM_OK;
    } else {
      g.s_first = g.s_first < own_hi ? g.s_first : own_hi;
      g.s_last = g.s_last > own_lo ? g.s_last : own_lo;
      if (g.s_last > g.s_first) return PM_OK;
    }
    g.s_first -= t.band_row0;
    g.s_last -= t.band_row0;
  }
  return run_sweep(h, ps, cp, g, ps.n_views, p.noise_amp[it]);
}

int pm_tile_snapshot(pm_handle* h) {
  if (int rc = tile_check(h, "pm_tile_snapshot")) return rc;
  const size_t plane = (size_t)h->max_rows * h->max_pitch;
  if (!h->snap_disp) {
    PM_HIP(h, hipMalloc((void**)&h->snap_disp, sizeof(float) * 2 * plane));
    PM_HIP(h, hipMalloc((void**)&h->snap_cost, sizeof(float) * 2 * plane));
  }
  const PlaneSet ps = tile_plane_set(h);
  const size_t bytes = sizeof(float) * ps.plane * ps.n_views;
  PM_HIP(h, hipMemcpyAsync(h->snap_disp, h->disp, bytes, hipMemcpyDeviceToDevice, h->stream));
  PM_HIP(h, hipMemcpyAsync(h->snap_cost, h->cost, bytes, hipMemcpyDeviceToDevice, h->stream));
  return PM_OK;
}

int pm_tile_restore(pm_handle* h) {
  if (int rc = tile_check(h, "pm_tile_restore")) return rc;
  if (!h->snap_disp) {
    set_err(h, "pm_tile_restore: no snapshot");
    return PM_ERR_INVALID_ARG;
  }
  const PlaneSet ps = tile_plane_set(h);
  const size_t bytes = sizeof(float) * ps.plane * ps.n_views;
  PM_HIP(h, hipMemcpyAsync(h->disp, h->snap_disp, bytes, hipMemcpyDeviceToDevice, h->stream));
  PM_HIP(h, hipMemcpyAsync(h->cost, h->snap_cost, bytes, hipMemcpyDeviceToDevice, h->stream));
  return PM_OK;
}

int pm_tile_restore_cols(pm_handle* h, const int* d_mask) {
  if (int rc = tile_check(h, "pm_tile_restore_cols")) return rc;
  if (!h->snap_disp || !d_mask) {
    set_err(h, "pm_tile_restore_cols: no snapshot or null mask");
    return PM_ERR_INVALID_ARG;
  }
  const PlaneSet ps = tile_plane_set(h);
  hipLaunchKernelGGL(k_restore_cols, pixel_grid(ps.cols, ps.rows, ps.n_views), dim3(256), 0, h->stream, ps,
                     (const float*)h->snap_disp, (const float*)h->snap_cost, d_mask);
  return launch_check(h, "restore_cols");
}

static int tile_row_copy(pm_handle* h, int image_row, float* d_dst, const float* d_src, const char* what) {
  if (int rc = tile_check(h, what)) return rc;
  const int r = image_row - h->tile.band_row0;
  if (r < 0 || r >= h->tile_band_rows || (!d_dst && !d_src)) {
    set_err(h, "%s: row %d outside the band or null pointer", what, image_row);
    return PM_ERR_INVALID_ARG;
  }
  const PlaneSet ps = tile_plane_set(h);
  for (int v = 0; v < ps.n_views; ++v) {
    float* plane_row = h->disp + (size_t)v * ps.plane + (size_t)r * ps.pitch;
    if (d_dst)
      PM_HIP(h, hipMemcpyAsync(d_dst + (size_t)v * ps.cols, plane_row, sizeof(float) * ps.cols,
                               hipMemcpyDeviceToDevice, h->stream));
    else
      PM_HIP(h, hipMemcpyAsync(plane_row, d_src + (size_t)v * ps.cols, sizeof(float) * ps.cols,
                               hipMemcpyDeviceToDevice, h->stream));
  }
  return PM_OK;
}

int pm_tile_get_row(pm_handle* h, int image_row, float* d_dst) {
  return tile_row_copy(h, image_row, d_dst, nullptr, "pm_tile_get_row");
}
int pm_tile_set_row(pm_handle* h, int image_row, const float* d_src) {
  return tile_row_copy(h, image_row, nullptr, d_src, "pm_tile_set_row");
}

int pm_tile_background(pm_handle* h) {
  if (int rc = tile_check(h, "pm_tile_background")) return rc;
  const pm_params& p = h->params;
  const PlaneSet ps = tile_plane_set(h);
  const CostParams bcp = cost_params(p, p.bg_patch_w, p.bg_patch_h);
  const Interior in = tile_interior(h, bcp.pw, bcp.ph);
  const int last = p.patchmatch_iters - 1;
  const int cached = (last >= 0 && bcp.pw == (p.semantics == PM_SEM_CPU ? p.patch_w[last] : 3) &&
                      bcp.ph == (p.semantics == PM_SEM_CPU ? p.patch_h[last] : 3)) ? 1 : 0;
  const float factor = p.semantics == PM_SEM_CPU ? p.win_by_factor : p.cost_improve_factor;
  launch_background(h, ps, bcp, in, factor, cached, ps.n_views);
  return launch_check(h, "background");
}

int pm_tile_finish(pm_handle* h, float* d_disp_l_own, float* d_disp_r_own) {
  if (int rc = tile_check(h, "pm_tile_finish")) return rc;
  const PlaneSet ps = tile_plane_set(h);
  if (!d_disp_l_own || (ps.n_views > 1 && !d_disp_r_own)) {
    set_err(h, "pm_tile_finish: null output");
    return PM_ERR_INVALID_ARG;
  }
  hipLaunchKernelGGL(k_finalize, pixel_grid(ps.cols, ps.rows, 1), dim3(256), 0, h->stream, ps, h->st_disp_l,
                     ps.n_views > 1 ? h->st_disp_r : nullptr, (size_t)ps.cols);
  if (int rc = launch_check(h, "finalize")) return rc;
  const size_t ofs = (size_t)(h->tile.own_row0 - h->tile.band_row0) * ps.cols;
  const size_t bytes = sizeof(float) * (size_t)h->tile.own_rows * ps.cols;
  PM_HIP(h, hipMemcpyAsync(d_disp_l_own, h->st_disp_l + ofs, bytes, hipMemcpyDeviceToDevice, h->stream));
  if (ps.n_views > 1)
    PM_HIP(h, hipMemcpyAsync(d_disp_r_own, h->st_disp_r + ofs, bytes, hipMemcpyDeviceToDevice, h->stream));
  h->tile_on = false;
  return PM_OK;
}

// ---- PM_MODE_PLANES, stage by stage ------------------------------------------------------------------------

namespace {
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
  return PM_OK;
}
}  // namespace

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
  const bool ok = (stage == PM_PL_SPATIAL && (arg == 0 || arg == 1)) ||
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

// ---- profiling ----------------------------------------------------------------------------------

int pm_debug_counters_enable(pm_handle* h, int on) {
  if (!h) return PM_ERR_INVALID_ARG;
  h->counters_on = on != 0;
  return PM_OK;
}

int pm_debug_counters(pm_handle* h, uint64_t out[8]) {
  if (!h || !out) return PM_ERR_INVALID_ARG;
  PM_HIP(h, hipSetDevice(h->device));
  PM_HIP(h, hipStreamSynchronize(h->stream));
  PM_HIP(h, hipMemcpy(out, h->counters, sizeof(uint64_t) * 8, hipMemcpyDeviceToHost));
#ifdef PM_RUN3_STATS
  {
    uint64_t t[8];
    PM_HIP(h, hipMemcpy(t, h->counters + 8, sizeof(t), hipMemcpyDeviceToHost));
    if (t[3]) {
      const double blocks = (double)t[3], waves = blocks * 4.0;  // (4 wavefronts per workgroup: the default)
      fprintf(stderr, "run3 stats: workgroups %.0f; round-1 steps of the slowest wavefront per workgroup %.1f, of the average "
                      "wavefront %.1f, of the average group %.1f; fix-up steps of the slowest wavefront %.1f; clock ticks per "
                      "workgroup: round 1 %.0f, fix-up rounds %.0f\n",
              blocks, (double)t[0] / blocks, (double)(out[0] + out[4]) / waves, (double)t[2] / (double)t[6],
              (double)t[1] / blocks, (double)t[4] / blocks, (double)t[5] / blocks);
    }
  }
#endif
  PM_HIP(h, hipMemset(h->counters, 0, sizeof(uint64_t) * 16));
  return PM_OK;
}

int pm_profile_enable(pm_handle* h, int on) {
  if (!h) return PM_ERR_INVALID_ARG;
  if (on)
    if (int rc = refuse_while_capturing(h, "pm_profile_enable")) return rc;  // events would be recorded into the graph
  h->profiling = on != 0;
  return PM_OK;
}

int pm_profile_read(pm_handle* h, pm_profile* out) {
  if (!h || !out) return PM_ERR_INVALID_ARG;
  if (int rc = refuse_while_capturing(h, "pm_profile_read")) return rc;
  PM_HIP(h, hipSetDevice(h->device));
  PM_HIP(h, hipStreamSynchronize(h->stream));
  for (int i = 0; i < h->ev_used; ++i) {
    float ms = 0.f;
    const EventRec& r = h->ev_pool[i];
    if (hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) {
      h->prof.launches[r.klass] += 1;
      h->prof.total_ms[r.klass] += (double)ms;
    }
  }
  h->ev_used = 0;
  *out = h->prof;
  std::memset(&h->prof, 0, sizeof(h->prof));
  return PM_OK;
}

}  // extern "C"


// ---- the narrow interface pm_imaging.hip works through (pm_internal.hpp) ---------------------------------------
namespace pm_internal {
int device(const pm_handle* h) { return h->device; }
hipStream_t stream(pm_handle* h) { return h->stream; }
const pm_params& params(const pm_handle* h) { return h->params; }
void plan_size(const pm_handle* h, int* max_rows, int* max_cols) {
  *max_rows = h->max_rows;
  *max_cols = h->max_cols;
}
void** imaging_slot(pm_handle* h) { return &h->imaging_state; }
void set_bgr_source(pm_handle* h, const pm::BgrSource* src) { h->bgr = src; }
void set_error(pm_handle* h, const char* fmt, ...) {
  if (!h) return;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(h->err, sizeof(h->err), fmt, ap);
  va_end(ap);
}
}  // namespace pm_internal
