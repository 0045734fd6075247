// pm_engine.hip -- C-ABI implementation (include/pm/patchmatch.h) of the gfx950 PatchMatch stereo engine: handle
// lifecycle, device memory plan, parameter checks, the Match() schedule on device buffers (per-view streams,
// iterations, background, cross-check), HIP-graph capture / replay, per-kernel timing.  The kernels live in the other
// units (pm_handle.hpp lists them); this one is host logic over their launch functions.
//
// There is deliberately NO CPU fallback in this library: without a usable HIP device pm_create fails with
// PM_ERR_NO_DEVICE.  `file:line` citations are relative to the reference tree.
#include "pm/patchmatch.h"
#include "pm/testing.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "pm_handle.hpp"
#include "pm_internal.hpp"

using namespace pm;
using namespace pm::eng;

namespace pm {
namespace eng {

void set_err(pm_handle* h, const char* fmt, ...) {
  if (!h) return;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(h->err, sizeof(h->err), fmt, ap);
  va_end(ap);
}

hipError_t create_stream(hipStream_t* s, int kind, int prio_class) {
  // ONE class for every stream of the engine: high priority -- not for the priority, but because streams of different
  // classes never share a hardware queue, so a class of their own keeps the engine's streams off the queues of whatever
  // else the process creates (torch's side streams, RCCL), and the engine's own streams do not slow each other down
  // the way default-priority streams do while a stream of another class exists (round 3 had only the view streams
  // high: eight band handles 49 -> 117 ms and pm_match_u8 314 -> 218 pairs/s as soon as a single-pair handle lived in
  // the same process).  Measured with every handle kind alive in one process (tools/stream_matrix.py,
  // profiles/r04_stream_matrix.txt): single 384, batch 429, pm_match_u8 310-323, eight bands 47.6 ms -- each within
  // 1 % of the same leg in a process of its own.  The tuning build reads PM_STREAM_PRIO = "main,view,copy,lane"
  // (1 high, 0 default, -1 low; one number = all four).
  static const struct Prio {
    int v[4];
    Prio() {
      v[0] = v[1] = v[2] = v[3] = 1;
      const char* e = pm::tune_env("PM_STREAM_PRIO");
      if (e) {
        int a = 1, b = 1, c = 1, d = 1;
        const int n = sscanf(e, "%d,%d,%d,%d", &a, &b, &c, &d);
        if (n == 1) b = c = d = a;
        v[0] = a; v[1] = b; v[2] = c; v[3] = d;
      }
    }
  } prio;
  // pm_params.stream_priority selects the class of a handle's streams; the tuning build's PM_STREAM_PRIO overrides it
  const int p = pm::tune_env("PM_STREAM_PRIO") ? prio.v[kind & 3] : prio_class;
  if (p == 0) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
  if (p == 2) {
    // a stream with a CU mask never shares its hardware queue (the runtime keeps such queues out of the shared pools);
    // with every CU enabled the mask restricts nothing
    hipDeviceProp_t prop;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return e;
    const int words = (prop.multiProcessorCount + 31) / 32;
    std::vector<uint32_t> mask((size_t)words, 0xffffffffu);
    if (prop.multiProcessorCount % 32) mask.back() = (1u << (prop.multiProcessorCount % 32)) - 1u;
    return hipExtStreamCreateWithCUMask(s, (uint32_t)words, mask.data());
  }
  int lo = 0, hi = 0;
  const hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);
  if (e != hipSuccess) return e;
  return hipStreamCreateWithPriority(s, hipStreamNonBlocking, p > 0 ? hi : lo);
}

// The four streams of a handle, created TOGETHER and in a rotating order.  A process has four hardware queues per
// priority class (GPU_MAX_HW_QUEUES); the runtime binds a stream to a queue when the stream is created: a new queue
// while the class has fewer than four, else the queue with the fewest streams, the first such in a fixed order
// (rocprofv3 --kernel-trace shows the queue of every dispatch: profiles/r04_queue_assignment.txt).  Streams created one
// by one on first use therefore collide as history dictates -- the round-4 trace has a batch handle whose two view
// streams shared one queue because the class filled up between them (430 -> 329 pairs/s), and a fifth stream of a handle
// always shares.  Four streams created back to back land on four different queues whenever the class is balanced,
// and every handle adds one stream to every queue, so it stays balanced as handles come and go.  The rotation (handle k
// creates role r at position (r + k) mod 4) spreads the SAME role of different handles over the queues: eight band
// handles of a row-tiled image, which only ever use their own stream, take the four queues in turn instead of one.
int create_handle_streams(pm_handle* h) {
  static std::atomic<unsigned> handles_created{0};
  const unsigned k = handles_created.fetch_add(1u);
  hipStream_t grp[4] = {nullptr, nullptr, nullptr, nullptr};
  const int kinds[4] = {kStreamMain, kStreamView, kStreamCopy, kStreamCopy};
  hipStream_t* roles[4] = {&h->stream, &h->view1_stream, &h->s_out, &h->s_in};
  int role_at[4];
  for (int pos = 0; pos < 4; ++pos) {
    role_at[pos] = (int)((pos + 4u - (k & 3u)) & 3u);  // the role created at this position: (role + k) % 4 == pos
    const hipError_t e = create_stream(&grp[pos], kinds[role_at[pos]], h->params.stream_priority);
    if (e != hipSuccess) {
      // the handle's fields are assigned only once all four streams exist: pm_destroy (which the caller must still
      // run on the handle pm_create hands back with the error) never sees a destroyed stream
      for (hipStream_t st : grp)
        if (st) (void)hipStreamDestroy(st);
      set_err(h, "stream creation failed: %s", hipGetErrorString(e));
      return PM_ERR_HIP;
    }
  }
  for (int pos = 0; pos < 4; ++pos) *roles[role_at[pos]] = grp[pos];
  return PM_OK;
}

int check_patch(pm_handle* h, int pw, int ph) {
  if (pw < 3 || ph < 3 || pw > PM_MAX_PATCH || ph > PM_MAX_PATCH || (pw % 2) == 0 || (ph % 2) == 0) {
    // patchmatch.cpp:257-258 CHECKs oddness; 1x1 is excluded because its passes A/B and C/D visit
    // different pixel sets, which the engine's shared cost plane does not model.
    set_err(h, "patch %dx%d unsupported: sides must be odd and within [3, %d]", pw, ph, PM_MAX_PATCH);
    return PM_ERR_INVALID_ARG;
  }
  return PM_OK;
}

namespace {
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
}  // namespace

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
  ps.splane = (size_t)align_up(rows, 4) * ps.pitch;
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

int launch_check(pm_handle* h, const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_err(h, "launch of %s failed: %s", what, hipGetErrorString(e));
    return PM_ERR_HIP;
  }
  return PM_OK;
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
  sp.use_harris = p.gftt_use_harris;
  sp.harris_k = p.gftt_k;
  sp.subpixel_corners = p.subpixel_corners;
  sp.subpix_winsize = p.subpix_winsize;
  sp.subpix_zerozone = p.subpix_zerozone;
  sp.subpix_maxiters = p.subpix_maxiters;
  sp.subpix_epsilon = p.subpix_epsilon;
  sp.subpixel_refinement = p.subpixel_refinement;
  return sp;
}

int alloc_seed_scratch(pm_handle* h, SeedScratch& sc) {
  PM_HIP(h, seed_scratch_alloc(sc, (size_t)h->max_rows * h->max_pitch, h->stream));
  PM_HIP(h, seed_subpix_prepare(sc, seed_params(h->params), h->stream));
  return PM_OK;
}

// SparseInit for view `view` of pair `b` straight into its disparity plane.  View 1 is seeded on the
// mirrored pair (patchmatch_gpu.cu:362-365), whose map is already in the mirrored coordinates the plane uses.
int run_sparse_init(pm_handle* h, const PlaneSet& ps, int b, int view, int scratch, unsigned stages) {
  SeedScratch& sc = h->seeds[scratch];
  if (!sc.eig) {
    if (h->capturing) {
      set_err(h, "the seeder scratch of this lane does not exist yet: run this call once before capturing it");
      return PM_ERR_BUSY;
    }
    if (int rc = alloc_seed_scratch(h, sc)) return rc;
  }
  const uint8_t* ref = ps.img8 + ((size_t)b * 4 + (view == 0 ? 0 : 3)) * ps.plane;
  const uint8_t* tgt = ps.img8 + ((size_t)b * 4 + (view == 0 ? 1 : 2)) * ps.plane;
  float* out = ps.disp + ((size_t)b * 2 + view) * ps.splane;  // a state plane: rows interleaved (out_pitch < 0 below)
  if (h->params.cpu_initialize_factor == 1)  // Patchmatch::Initialize(il, ir, 1) (patchmatch_test.cpp:149-150): 5x5, / 2
    PM_HIP(h, seed_initialize(sc, seed_params(h->params), ref, tgt, ps.rows, ps.cols, ps.pitch, 1, out, -ps.pitch,
                              h->stream, stages));
  else
    PM_HIP(h, seed_sparse_init(sc, seed_params(h->params), ref, tgt, ps.rows, ps.cols, ps.pitch,
                               h->params.init_dilate_factor, out, -ps.pitch, h->stream, stages));
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

namespace {

// Launch number `idx` of a view's schedule -- per iteration {noise + cost, sweeps A B C D}, then the background mask:
// PatchmatchGpu::Match(GpuMat...) (patchmatch_gpu.cu:379-411) / the recipe of patchmatch_test.cpp:173-183 -- enqueued
// on h->stream for all `slots` of plane set `ps`.
int view_op(pm_handle* h, const PlaneSet& ps, int idx, int slots) {
  const pm_params& p = h->params;
  const int it = idx / 5, k = idx % 5;
  if (it < p.patchmatch_iters) {
    const CostParams cp = cost_params(p, p.patch_w[it], p.patch_h[it]);
    const Interior in = interior(p, ps.rows, ps.cols, cp.pw, cp.ph);
    if (k > 0) return run_sweep(h, ps, cp, sweep_geom(p, in, k - 1), slots, p.noise_amp[it]);
    {
      Launch l(h, PM_K_NOISE);
      // from the second iteration on the cost plane is valid for this window if the window is unchanged
      const CostParams prev = it > 0 ? cost_params(p, p.patch_w[it - 1], p.patch_h[it - 1]) : cp;
      const int keep_zero = (it > 0 && cp.pw == prev.pw && cp.ph == prev.ph) ? 1 : 0;
      launch_noise_cost(h, ps, cp, in, p.noise_amp[it], slots, keep_zero);
    }
    return launch_check(h, "noise_cost");
  }
  const CostParams bcp = cost_params(p, p.bg_patch_w, p.bg_patch_h);
  const Interior in = interior(p, ps.rows, ps.cols, bcp.pw, bcp.ph);
  int cached = 0;
  if (p.patchmatch_iters > 0) {
    const CostParams last = cost_params(p, p.patch_w[p.patchmatch_iters - 1], p.patch_h[p.patchmatch_iters - 1]);
    cached = (bcp.pw == last.pw && bcp.ph == last.ph) ? 1 : 0;
  }
  const float factor = p.semantics == PM_SEM_CPU ? p.win_by_factor : p.cost_improve_factor;
  {
    Launch l(h, PM_K_BACKGROUND);
    launch_background(h, ps, bcp, in, factor, cached, slots);
  }
  return launch_check(h, "background");
}

// `sets` plane sets are advanced side by side, each on its own stream (one set on the handle's stream, or the two
// views on their view streams): the host enqueues launch k of EVERY set before launch k + 1 of any, so all
// streams have work from the first microsecond on.  (Enqueuing one view's whole chain of ~45 launches first left
// the other stream empty for the 0.2-0.4 ms that takes: visible in the rocprofv3 kernel trace.)
int run_view_sets(pm_handle* h, const PlaneSet* pss, hipStream_t* streams, int sets, int slots) {
  hipStream_t keep = h->stream;
  struct Restore {
    pm_handle* h;
    hipStream_t s;
    ~Restore() { h->stream = s; }
  } restore{h, keep};
  const int n_ops = 5 * h->params.patchmatch_iters + 1;
  for (int idx = 0; idx < n_ops; ++idx)
    for (int s = 0; s < sets; ++s) {
      h->stream = streams[s];  // every launch helper enqueues on h->stream
      if (int rc = view_op(h, pss[s], idx, slots)) return rc;
    }
  return PM_OK;
}

}  // namespace

int run_one_view_set(pm_handle* h, const PlaneSet& ps, int slots) {
  hipStream_t s = h->stream;
  return run_view_sets(h, &ps, &s, 1, slots);
}

namespace {

bool view_streams_enabled() {
  static bool v = [] {
    const char* e = pm::tune_env("PM_VIEW_STREAMS");
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

// Fork / join bookkeeping.  While a capture is open, every stream work was forked onto is remembered until the
// handle's stream has waited for an event recorded behind that work: hipStreamEndCapture on a capture with an unjoined
// fork does not return an error on this runtime, it faults (gpurun_out/r03/crash.log), so pm_capture_end checks first.
void mark_forked(pm_handle* h, hipStream_t s) {
  if (!h->capturing || s == h->stream) return;
  for (hipStream_t t : h->cap_unjoined)
    if (t == s) return;
  h->cap_unjoined.push_back(s);
}
void mark_joined(pm_handle* h, hipStream_t s) {
  for (size_t i = 0; i < h->cap_unjoined.size(); ++i)
    if (h->cap_unjoined[i] == s) {
      h->cap_unjoined[i] = h->cap_unjoined.back();
      h->cap_unjoined.pop_back();
      return;
    }
}
// `onto` waits for everything `from` holds now
int join_stream(pm_handle* h, hipStream_t from, hipEvent_t ev, hipStream_t onto) {
  PM_HIP(h, hipEventRecord(ev, from));
  PM_HIP(h, hipStreamWaitEvent(onto, ev, 0));
  if (onto == h->stream) mark_joined(h, from);
  prof_break(h, onto);
  return PM_OK;
}

}  // namespace

// The roles of a handle's four streams (create_handle_streams):
//   stream        the handle's own: every single-pair call, the FIRST view of every chunk
//   view1_stream  the second view of a single pair and of every chunk
//   s_out         chunks only: the cross-check of a chunk behind both of its views, then the downloads
//   s_in          host-buffer sequences only: uploads
// A fifth stream would share the hardware queue of one of the four (with five, a frame sequence ran at 313-331 pairs/s in
// a process of its own and at 400-435 beside other handles, depending on whether a view stream ended up behind the
// download stream's waits: profiles/r04_stream_matrix.txt).  This creates the events the view streams fork and join on.
int view_streams_create(pm_handle* h) {
  if (h->view_fork && h->view1_join && h->out_join && h->in_join && h->view_end[0] && h->view_end[1]) return PM_OK;
  if (h->capturing) {
    set_err(h, "the view events do not exist yet: run this call once before capturing it");
    return PM_ERR_BUSY;
  }
  if (!h->view_fork) PM_HIP(h, hipEventCreateWithFlags(&h->view_fork, hipEventDisableTiming));
  if (!h->view1_join) PM_HIP(h, hipEventCreateWithFlags(&h->view1_join, hipEventDisableTiming));
  if (!h->out_join) PM_HIP(h, hipEventCreateWithFlags(&h->out_join, hipEventDisableTiming));
  if (!h->in_join) PM_HIP(h, hipEventCreateWithFlags(&h->in_join, hipEventDisableTiming));
  for (int v = 0; v < 2; ++v)
    if (!h->view_end[v]) PM_HIP(h, hipEventCreate(&h->view_end[v]));
  return PM_OK;
}

// The per-slot events of chunks and frame sequences (h->pipe[b]: batch / ring slot b).
int seq_events_create(pm_handle* h) {
  if (!h->pipe.empty()) return PM_OK;
  if (h->capturing) {
    set_err(h, "the chunk events do not exist yet: run this call once before capturing it");
    return PM_ERR_BUSY;
  }
  std::vector<pm_handle::PipeSlot> slots((size_t)h->max_batch);
  for (auto& sl : slots) {
    hipEvent_t* evs[] = {&sl.in_done, &sl.head_done, &sl.v_done[0], &sl.v_done[1], &sl.fin_done, &sl.out_done};
    for (hipEvent_t* e : evs) PM_HIP(h, hipEventCreateWithFlags(e, hipEventDisableTiming));
  }
  h->pipe.swap(slots);
  return PM_OK;
}

namespace {

// Both views of the pairs of `ps` on vstream[0] / vstream[1]; a stream other than the handle's (the handle's too with
// wait_on_main) first waits for the non-null events of `waits`.  Enqueue only; the caller joins.  `scratch` = seeder scratch set of view 0 (view 1: + 1).
int run_views_on(pm_handle* h, const PlaneSet& ps, int slots, const ViewSetup* setup, hipStream_t vstream[2],
                 const hipEvent_t* waits, int n_waits, bool wait_on_main = false, int scratch = 0) {
  hipStream_t main_stream = h->stream;
  int rc = PM_OK;
  PlaneSet pv[2] = {ps, ps};
  for (int v = 0; v < 2 && rc == PM_OK; ++v) {
    pv[v].view_fixed = v;
    if (vstream[v] != main_stream || wait_on_main) {
      for (int w = 0; w < n_waits && rc == PM_OK; ++w)
        if (waits[w] && hipStreamWaitEvent(vstream[v], waits[w], 0) != hipSuccess) rc = PM_ERR_HIP;
      if (rc != PM_OK) break;
      mark_forked(h, vstream[v]);
      prof_break(h, vstream[v]);
    }
  }
  // The head of a view -- prep, transposed planes, the seeder's seven launches -- goes out like the sweeps behind it:
  // launch k of BOTH views before launch k + 1 of either.  (One view's whole head first left the other view's stream
  // empty for the ~55 us the host needs for nine launches, and the second view then ends that much later: a tenth of
  // the reference's own 376x240 call, profiles/r06_reference_call_timeline.txt.)
  for (int op = 0; op < 2 && setup && rc == PM_OK; ++op)
    for (int v = 0; v < 2 && rc == PM_OK; ++v) {
      h->stream = vstream[v];
      Launch l(h, PM_K_PREP);
      if (op == 0) {
        const PrepSeedMaps seeds{setup->d_seed_l, setup->d_seed_r};  // the seed copy rides along (one launch less)
        launch_prep(h, ps, setup->d_left, setup->d_right, setup->n, (size_t)ps.cols, v, &seeds);
        rc = launch_check(h, "prep");
      } else {
        rc = run_transpose(h, ps, setup->n, v);
      }
    }
  for (int b = 0; b < slots / 2 && rc == PM_OK; ++b)  // a view's pairs share its scratch: pair by pair
    for (int st = 0; st < kSeedStages && rc == PM_OK; ++st)
      for (int v = 0; v < 2 && rc == PM_OK; ++v) {
        if (!h->need_seed[v]) continue;
        h->stream = vstream[v];
        Launch l(h, PM_K_SEED);
        rc = run_sparse_init(h, ps, b, v, scratch + v, 1u << st);
      }
  h->stream = main_stream;
  if (rc == PM_OK) rc = run_view_sets(h, pv, vstream, 2, slots / 2);
  if (rc == PM_ERR_HIP && !h->err[0]) set_err(h, "per-view stream setup failed");
  return rc;
}

int run_views(pm_handle* h, const PlaneSet& ps, int slots, const ViewSetup* setup = nullptr) {
  if (ps.n_views != 2 || !view_streams_enabled()) {
    for (int v = 0; v < ps.n_views; ++v)
      if (int rc = seed_views(h, ps, slots / ps.n_views, v, 0)) return rc;
    return run_one_view_set(h, ps, slots);
  }
  // one view stays on the caller's stream, the other forks off
  if (int rc = view_streams_create(h)) return rc;
  PM_HIP(h, hipEventRecord(h->view_fork, h->stream));
  // the view that ended last the time before stays on the handle's stream, the other one forks off (pm_handle::late_view)
  if (!h->capturing && h->view_end_recorded) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->view_end[0], h->view_end[1]) == hipSuccess) {
      h->late_view = ms >= 0.f ? 1 : 0;
      h->view_end_recorded = false;
    } else {
      (void)hipGetLastError();  // that call is still running: ask again next time
    }
  }
  static const int force_late = [] { const char* e = pm::tune_env("PM_LATE_VIEW"); return e ? atoi(e) : -1; }();  // tuning builds
  if (force_late == 0 || force_late == 1) h->late_view = force_late;
  hipStream_t vs[2];
  vs[h->late_view] = h->stream;
  vs[1 - h->late_view] = h->view1_stream;
  if (int rc = run_views_on(h, ps, slots, setup, vs, &h->view_fork, 1)) return rc;
  // sampled: a timed event is a command of its own on the stream, in front of the join of every call it is recorded in
  if (!h->capturing && !h->view_end_recorded && (h->view_calls++ % pm_handle::kViewEndEvery) == 0) {
    for (int v = 0; v < 2; ++v) PM_HIP(h, hipEventRecord(h->view_end[v], vs[v]));
    h->view_end_recorded = true;
  }
  return join_stream(h, h->view1_stream, h->view1_join, h->stream);
}

// The plane set of pair b alone (every per-pair array advanced to that pair; see make_view for the strides).
PlaneSet plane_set_of_pair(const PlaneSet& ps, int b) {
  PlaneSet q = ps;
  const size_t b4 = (size_t)b * 4, b2 = (size_t)b * 2;
  q.img8 += b4 * ps.plane;
  q.g32 += b4 * ps.plane;
  q.g8 += b4 * ps.plane;
  q.pk16 += b4 * ps.plane;
  q.timg8 += b4 * ps.plane_t;
  q.tg32 += b4 * ps.plane_t;
  q.tg8 += b4 * ps.plane_t;
  q.tpk16 += b4 * ps.plane_t;
  if (q.rpg) q.rpg += b2 * (size_t)ps.nrl * ps.pitch * 4;
  if (q.rqk) q.rqk += b2 * (size_t)ps.nrl * ps.pitch * 2;
  if (q.cpg) q.cpg += b2 * (size_t)ps.ncl * ps.pitch_t * 4;
  q.disp += b2 * ps.splane;
  q.cost += b2 * ps.splane;
  return q;
}

// pairs per chunk (PM_PAIR_CHUNK in the tuning build).  Two pairs advanced through every launch together run at 408
// pairs/s where one runs at 373 -- a single pair leaves SIMDs short of wavefronts -- while 4 or 32 in lockstep fall back
// to 368 / 357 (working sets): profiles/r03_pair_lanes.txt.
int pair_chunk() {
  static const int v = [] {
    const char* e = pm::tune_env("PM_PAIR_CHUNK");
    const int x = e ? atoi(e) : 2;
    return x < 1 ? 1 : x;
  }();
  return v;
}

}  // namespace

int seq_chunk_pairs() { return pair_chunk(); }

PlaneSet pair_plane_set(const PlaneSet& ps, int b) { return plane_set_of_pair(ps, b); }

// A second lane for half of a batch (plane mode, pm_planes_host.hip): view1_stream runs behind everything the handle's
// stream holds now; lane_join makes the handle's stream wait for the lane again.
int lane_fork(pm_handle* h) {
  if (int rc = view_streams_create(h)) return rc;
  PM_HIP(h, hipEventRecord(h->view_fork, h->stream));
  PM_HIP(h, hipStreamWaitEvent(h->view1_stream, h->view_fork, 0));
  mark_forked(h, h->view1_stream);
  prof_break(h, h->view1_stream);
  return PM_OK;
}
int lane_join(pm_handle* h) { return join_stream(h, h->view1_stream, h->view1_join, h->stream); }

bool seq_pipelined(const pm_handle* h) {
  return h->params.mode == PM_MODE_SCALAR && h->params.left_right_check != 0 && view_streams_enabled() && !h->bgr;
}

// One CHUNK of a batch or of a frame sequence: pairs [b, b + c) of the plan.
//   the HEAD of the chunk, behind the non-null events `ready` (the chunk's inputs are in device memory) and `slot_free`
//                 (whoever last used these plane slots is done with them): images, gradients, transposes and line
//                 planes of both views, and the device seeder where a view seeds itself (a chain of latency-bound
//                 launches, 0.11 ms per view and pair).  A chunk that seeds itself runs its head on s_in -- the stream
//                 the uploads ran on -- BESIDE the sweeps of the chunk in front instead of between two chunks on the
//                 view streams; the others keep it at the head of their view streams (see below);
//   the handle's stream / view1_stream   the iterations of the first / second view (behind event head_done);
//   s_out         behind both views (events v_done[0 / 1]): the cross-check / un-mirror into d_disp_l / d_disp_r
//                 ([c][rows][cols]).
// Chunks enqueued one after the other run back to back on the two view streams, nothing forks or joins in between.
// Enqueue only; the caller orders what follows behind s_out.
int seq_enqueue_chunk(pm_handle* h, int b, int c, const uint8_t* d_left, const uint8_t* d_right, int rows, int cols,
                      const float* d_seed_l, const float* d_seed_r, float* d_disp_l, float* d_disp_r, hipEvent_t ready,
                      hipEvent_t slot_free, hipEvent_t v_done[2], hipEvent_t head_done) {
  if (int rc = view_streams_create(h)) return rc;
  const PlaneSet ps = plane_set(h, rows, cols, 2);
  const PlaneSet pb = plane_set_of_pair(ps, b);
  h->need_seed[0] = h->params.sparse_init && !d_seed_l;
  h->need_seed[1] = h->params.sparse_init && !d_seed_r;
  hipStream_t keep = h->stream;
  struct Restore {
    pm_handle* h;
    hipStream_t s;
    ~Restore() { h->stream = s; }
  } restore{h, keep};
  // ---- head: on s_in when a view seeds itself (measured, frame sequence at 720p: self-seeded 396 -> 401 pairs/s with the
  // head aside, seeded 410 -> 402 -- without the seeder the head is too short to pay for one more cross-queue wait)
  const bool head_aside = h->need_seed[0] || h->need_seed[1];
  hipStream_t vs[2] = {keep, h->view1_stream};
  if (head_aside) {
    if (ready) PM_HIP(h, hipStreamWaitEvent(h->s_in, ready, 0));
    if (slot_free) PM_HIP(h, hipStreamWaitEvent(h->s_in, slot_free, 0));
    mark_forked(h, h->s_in);
    prof_break(h, h->s_in);
    h->stream = h->s_in;
    for (int v = 0; v < 2; ++v) {
      {
        Launch l(h, PM_K_PREP);
        const PrepSeedMaps seeds{d_seed_l, d_seed_r};  // the seed copy rides along (one launch less)
        launch_prep(h, pb, d_left, d_right, c, (size_t)cols, v, &seeds);
        if (int rc = launch_check(h, "prep")) return rc;
      }
      {
        Launch l(h, PM_K_PREP);
        if (int rc = run_transpose(h, pb, c, v)) return rc;
      }
    }
    for (int v = 0; v < 2; ++v) {
      PlaneSet pv = pb;
      pv.view_fixed = v;
      if (int rc = seed_views(h, pv, c, v, v)) return rc;
    }
    PM_HIP(h, hipEventRecord(head_done, h->s_in));
    h->stream = keep;
    h->need_seed[0] = h->need_seed[1] = false;  // done in the head: the view streams start with the first noise + cost
    if (int rc = run_views_on(h, pb, 2 * c, nullptr, vs, &head_done, 1, true)) return rc;
  } else {
    const ViewSetup sb{d_left, d_right, d_seed_l, d_seed_r, c};
    const hipEvent_t waits[2] = {ready, slot_free};
    if (int rc = run_views_on(h, pb, 2 * c, &sb, vs, waits, 2, true)) return rc;
  }
  mark_joined(h, h->s_in);
  for (int v = 0; v < 2; ++v) {
    PM_HIP(h, hipEventRecord(v_done[v], vs[v]));
    PM_HIP(h, hipStreamWaitEvent(h->s_out, v_done[v], 0));
    mark_joined(h, vs[v]);
  }
  mark_forked(h, h->s_out);
  prof_break(h, h->s_out);
  h->stream = h->s_out;
  {
    Launch l(h, PM_K_FINALIZE);
    launch_finalize(h, pb, d_disp_l, d_disp_r, c);
    if (int rc = launch_check(h, "finalize")) return rc;
  }
  return PM_OK;
}

namespace {

// A batch of pairs as chunks of pair_chunk() pairs, one after the other on the two view streams, every chunk's
// cross-check on s_out as soon as its two views are through; the handle's stream joins s_out at the end of the call.
int run_pairs_as_chunks(pm_handle* h, int n, const ViewSetup& vs, int rows, int cols, float* d_disp_l, float* d_disp_r) {
  const int chunk = pair_chunk();
  if (int rc = view_streams_create(h)) return rc;
  if (int rc = seq_events_create(h)) return rc;
  PM_HIP(h, hipEventRecord(h->view_fork, h->stream));
  const size_t px = (size_t)rows * cols;
  for (int b = 0; b < n; b += chunk) {
    const int c = n - b < chunk ? n - b : chunk;
    if (int rc = seq_enqueue_chunk(h, b, c, vs.d_left + b * px, vs.d_right + b * px, rows, cols,
                                   vs.d_seed_l ? vs.d_seed_l + b * px : nullptr,
                                   vs.d_seed_r ? vs.d_seed_r + b * px : nullptr, d_disp_l + b * px,
                                   d_disp_r ? d_disp_r + b * px : nullptr, b == 0 ? h->view_fork : nullptr, nullptr,
                                   h->pipe[(size_t)b].v_done, h->pipe[(size_t)b].head_done))
      return rc;
  }
  return join_stream(h, h->s_out, h->out_join, h->stream);
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
  if ((p.gftt_use_harris != 0 && p.gftt_use_harris != 1) || !(p.gftt_k >= 0.0) || !(p.gftt_k <= 1.0)) {
    set_err(h, "gftt_use_harris must be 0 or 1 and gftt_k within [0, 1]");
    return PM_ERR_INVALID_ARG;
  }
  if ((p.subpixel_corners != 0 && p.subpixel_corners != 1) || (p.subpixel_refinement != 0 && p.subpixel_refinement != 1) ||
      p.subpix_winsize < 1 || p.subpix_winsize > kSubpixMaxWin || p.subpix_maxiters < 1 || !(p.subpix_epsilon >= 0.f)) {
    set_err(h, "cv::cornerSubPix parameters out of range (subpixel_corners / subpixel_refinement 0 or 1, subpix_winsize "
               "within [1, %d], subpix_maxiters >= 1, subpix_epsilon >= 0)", kSubpixMaxWin);
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
        !(p.plane_slope_per_disp >= 0.f) || !(p.plane_lr_tol >= 0.f) || p.max_disp < 1 || p.max_disp > 1024 ||
        (p.plane_window != PM_PL_WINDOW_FULL && p.plane_window != PM_PL_WINDOW_CHECKER) ||
        (p.plane_neighbours != PM_PL_NEIGH_FOUR && p.plane_neighbours != PM_PL_NEIGH_TWO)) {
      set_err(h, "PM_MODE_PLANES: plane parameters out of range");
      return PM_ERR_INVALID_ARG;
    }
  }
  if (p.stream_priority < PM_STREAM_PRIO_LOW || p.stream_priority > PM_STREAM_PRIO_HIGH) {
    set_err(h, "stream_priority must be one of pm_stream_priority (-1, 0, 1)");
    return PM_ERR_INVALID_ARG;
  }
  return PM_OK;
}

}  // namespace

// Ends a capture in progress and throws the partial graph away (error paths, pm_destroy).
void abort_capture(pm_handle* h) {
  if (!h->capturing) return;
  for (hipStream_t st : h->cap_unjoined) {  // see pm_capture_end: an unjoined fork must not reach hipStreamEndCapture
    hipEvent_t ev = st == h->s_out ? h->out_join : (st == h->s_in ? h->in_join : h->view1_join);
    if (ev && hipEventRecord(ev, st) == hipSuccess) (void)hipStreamWaitEvent(h->stream, ev, 0);
  }
  h->cap_unjoined.clear();
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

// Opens a capture on the handle's stream (thread-local mode).  Lazily created resources must exist before it starts:
// creating streams, events or device memory is not capturable.
int capture_open(pm_handle* h) {
  if (h->params.left_right_check && h->params.mode == PM_MODE_SCALAR) {
    if (int rc = view_streams_create(h)) return rc;
    if (int rc = seq_events_create(h)) return rc;
  }
  if (h->params.sparse_init)
    for (int i = 1; i < 2; ++i)
      if (!h->seeds[i].eig)
        if (int rc = alloc_seed_scratch(h, h->seeds[i])) return rc;
  h->cap_unjoined.clear();
  PM_HIP(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
  h->capturing = true;
  return PM_OK;
}

// Ends the capture and instantiates what was recorded.  Every stream the recorded calls forked work onto must have
// been joined back into the handle's stream: ending a capture with an unjoined fork is an error the runtime answers with
// a fault, not a status (gpurun_out/r03/crash.log: a schedule experiment that left a side stream forked).  The forks are
// joined here so that the capture can be ended at all, the graph is thrown away, and the caller gets PM_ERR_STATE.
int capture_close(pm_handle* h, hipGraphExec_t* exec, const char* what) {
  *exec = nullptr;
  const size_t unjoined = h->cap_unjoined.size();
  if (unjoined) {
    std::vector<hipStream_t> open_streams = h->cap_unjoined;
    for (hipStream_t st : open_streams) {
      hipEvent_t ev = st == h->s_out ? h->out_join : (st == h->s_in ? h->in_join : h->view1_join);
      if (hipEventRecord(ev, st) == hipSuccess) (void)hipStreamWaitEvent(h->stream, ev, 0);
      mark_joined(h, st);
    }
  }
  h->capturing = false;
  hipGraph_t graph = nullptr;
  PM_HIP(h, hipStreamEndCapture(h->stream, &graph));
  if (unjoined) {
    if (graph) (void)hipGraphDestroy(graph);
    set_err(h, "%s: %zu stream(s) the recorded calls forked work onto were never joined back; the capture "
               "was discarded", what, unjoined);
    return PM_ERR_STATE;
  }
  const hipError_t e = hipGraphInstantiate(exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) {
    *exec = nullptr;
    set_err(h, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
    return PM_ERR_HIP;
  }
  return PM_OK;
}

int match_device_impl(pm_handle* h, int n, const uint8_t* d_left, const uint8_t* d_right, int rows, int cols,
                      const float* d_seed_l, const float* d_seed_r, float* d_disp_l, float* d_disp_r) {
  if (!d_left || !d_right || !d_disp_l) {
    set_err(h, "pm_match_device: null image or output pointer");
    return PM_ERR_INVALID_ARG;
  }
  if (int rc = check_size(h, rows, cols, n)) return rc;
  PM_HIP(h, hipSetDevice(h->device));
  prof_break_all(h);  // whatever sits between two calls on the streams is not a kernel's time
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
  if (per_view_setup && n > pair_chunk()) {
    // chunks of pairs, each with its own cross-check launch (run_pairs_as_chunks)
    const ViewSetup vs{d_left, d_right, d_seed_l, d_seed_r, n};
    return run_pairs_as_chunks(h, n, vs, rows, cols, d_disp_l, d_disp_r);
  }
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
      launch_seed(h, ps, d_seed_l, d_seed_r, n);
    }
    if (int rc = launch_check(h, "seed")) return rc;
    if (int rc = run_views(h, ps, n * n_views)) return rc;
  }
  {
    Launch l(h, PM_K_FINALIZE);
    launch_finalize(h, ps, d_disp_l, d_disp_r, n);
  }
  return launch_check(h, "finalize");
}

}  // namespace eng
}  // namespace pm

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
  p->gftt_use_harris = 0;            // feature_detector.hpp:34
  p->gftt_k = 0.04;                  // :35
  p->subpixel_corners = 0;           // :39
  p->subpix_winsize = 10;            // :40
  p->subpix_zerozone = -1;           // :41
  p->subpix_maxiters = 10;           // :42
  p->subpix_epsilon = 0.01f;         // :43
  p->subpixel_refinement = 0;        // stereo_matcher.hpp:26
  p->cpu_initialize_factor = 0;
  p->mode = PM_MODE_SCALAR;
  p->state_dtype = PM_STATE_F32;
  p->plane_refine_steps = 3;         // oracle/pm_planes_oracle.c: pmo_planes_params_default
  p->plane_slope_max = 1.0f;
  p->plane_slope_init = 0.25f;
  p->plane_slope_per_disp = 1.0f / 64.0f;
  p->plane_lr_tol = 1.0f;
  p->plane_window = PM_PL_WINDOW_CHECKER;
  p->plane_neighbours = PM_PL_NEIGH_FOUR;
  p->stream_priority = PM_STREAM_PRIO_HIGH;
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
    case PM_ERR_STATE: return "call not valid in the handle's current state";
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
  // frames of a sequence may still be running on streams that never join the handle's own: wait for all of them
  hipStream_t streams[] = {h->stream, h->view1_stream, h->s_in, h->s_out};
  for (hipStream_t st : streams)
    if (st) (void)hipStreamSynchronize(st);
  pm_internal::release_imaging(h);
  hipEvent_t events[] = {h->ext_fork, h->ext_join, h->left_out, h->right_out, h->view1_join, h->out_join, h->in_join, h->view_fork,
                         h->view_end[0], h->view_end[1]};
  for (hipEvent_t e : events)
    if (e) (void)hipEventDestroy(e);
  if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
  for (auto& r : h->ev_pool) {
    (void)hipEventDestroy(r.start);
    (void)hipEventDestroy(r.stop);
  }
  for (auto& sl : h->pipe) {
    hipEvent_t evs[] = {sl.in_done, sl.head_done, sl.v_done[0], sl.v_done[1], sl.fin_done, sl.out_done};
    for (hipEvent_t e : evs)
      if (e) (void)hipEventDestroy(e);
  }
  void* dev[] = {h->rpg,     h->rqk,      h->cpg,       h->img8,      h->g32,       h->g8,        h->timg8,
                 h->tg32,    h->tg8,      h->pk16,      h->tpk16,     h->disp,      h->cost,      h->noise,
                 h->counters, h->st_left, h->st_seed_l, h->st_seed_r, h->st_disp_l, h->st_disp_r,
                 h->snap_disp, h->snap_cost, h->planes_state, h->texmask_scratch};
  for (auto& sc : h->seeds) seed_scratch_free(sc);
  for (void* p : dev)
    if (p) (void)hipFree(p);
  delete h->copy_pool;
  if (h->pinned) (void)hipHostFree(h->pinned);
  // memory handed out by pm_host_alloc and still held, registrations still standing
  for (auto& r : h->host_ranges) {
    if (r.owned) (void)hipHostFree(r.base);
    else (void)hipHostUnregister(r.base);
  }
  for (hipStream_t st : streams)
    if (st) (void)hipStreamDestroy(st);
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
  h->no_tiled = pm::tune_env("PM_NO_TILED") != nullptr;
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
  if (int rc = create_handle_streams(h)) return rc;

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
  const size_t splane = (size_t)align_up(max_rows, 4) * h->max_pitch;  // state planes: four rows interleaved (pm_device.hpp)
  PM_HIP(h, hipMalloc((void**)&h->disp, sizeof(float) * (B * 2 * splane + 64)));
  PM_HIP(h, hipMalloc((void**)&h->cost, sizeof(float) * (B * 2 * splane + 64)));
  PM_HIP(h, hipMalloc((void**)&h->noise, sizeof(float) * (plane + 64)));
  h->noise_capacity = plane;
  PM_HIP(h, hipMalloc((void**)&h->counters, sizeof(unsigned long long) * 16));  // [8..13]: timing builds only
  PM_HIP(h, hipMemsetAsync(h->counters, 0, sizeof(unsigned long long) * 16, h->stream));
  if (int rc = alloc_seed_scratch(h, h->seeds[0])) return rc;
  if (params->mode == PM_MODE_PLANES)
    if (int rc = planes_alloc(h)) return rc;
  const size_t tight = (size_t)max_rows * max_cols;
  // one block: a single small pair goes up as ONE copy, left and right back to back (pm_hostpath.hip::pm_match_u8)
  PM_HIP(h, hipMalloc((void**)&h->st_left, 2 * B * tight + 64));
  h->st_right = h->st_left + B * tight;
  PM_HIP(h, hipMalloc((void**)&h->st_seed_l, sizeof(float) * B * tight));
  PM_HIP(h, hipMalloc((void**)&h->st_seed_r, sizeof(float) * B * tight));
  PM_HIP(h, hipMalloc((void**)&h->st_disp_l, sizeof(float) * B * tight));
  PM_HIP(h, hipMalloc((void**)&h->st_disp_r, sizeof(float) * B * tight));
  // pinned host staging: per pair 2 u8 images + 2 seeds + 2 outputs (also used for the noise table)
  h->pinned_bytes = B * tight * (2 + 4 * sizeof(float));
  const size_t noise_bytes = sizeof(float) * plane;
  if (h->pinned_bytes < noise_bytes) h->pinned_bytes = noise_bytes;
  PM_HIP(h, hipHostMalloc(&h->pinned, h->pinned_bytes, hipHostMallocDefault));
  {
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, h->pinned, 0) == hipSuccess) h->pinned_dev = (char*)dp;
    (void)hipGetLastError();
  }
  // the cost planes are read only where the noise kernel wrote them; clear once so that tools that
  // scan whole planes never see uninitialised memory
  PM_HIP(h, hipMemsetAsync(h->cost, 0, sizeof(float) * (B * 2 * splane + 64), h->stream));
  PM_HIP(h, hipMemsetAsync(h->disp, 0, sizeof(float) * (B * 2 * splane + 64), h->stream));
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
  return pm::eng::capture_open(h);
}

int pm_capture_end(pm_handle* h) {
  if (!h || !h->capturing) return PM_ERR_INVALID_ARG;
  hipGraphExec_t exec = nullptr;
  if (int rc = pm::eng::capture_close(h, &exec, "pm_capture_end")) return rc;
  if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
  h->graph_exec = exec;
  return PM_OK;
}

// Test hook for the guard above: while a capture is open, forks an (empty) dependency onto the second-view stream and
// does NOT join it -- what a broken schedule would do.
int pm_debug_capture_fork(pm_handle* h) {
  if (!h) return PM_ERR_INVALID_ARG;
  if (!h->capturing || !h->view1_stream) {
    set_err(h, "pm_debug_capture_fork: only between pm_capture_begin and pm_capture_end of a handle with view streams");
    return PM_ERR_STATE;
  }
  PM_HIP(h, hipEventRecord(h->view_fork, h->stream));
  PM_HIP(h, hipStreamWaitEvent(h->view1_stream, h->view_fork, 0));
  mark_forked(h, h->view1_stream);
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

int pm_match_device(pm_handle* h, int n, const uint8_t* d_left, const uint8_t* d_right, int rows, int cols,
                    const float* d_seed_l, const float* d_seed_r, float* d_disp_l, float* d_disp_r) {
  if (!h) return PM_ERR_INVALID_ARG;
  if (h->pipe_count > 0) {  // the frames of a sequence live in the same plane slots
    set_err(h, "pm_match_device: pairs are in flight (pm_collect them first)");
    return PM_ERR_BUSY;
  }
  const int rc = match_device_impl(h, n, d_left, d_right, rows, cols, d_seed_l, d_seed_r, d_disp_l, d_disp_r);
  if (rc != PM_OK) abort_capture(h);  // a failed call must not leave the stream in capture mode
  return rc;
}

int pm_match_view_device(pm_handle* h, const float* d_iml, const float* d_imr, const float* d_Gl, const float* d_Gr,
                         int rows, int cols, size_t step, float* d_disp, size_t disp_step, void* stream) {
  if (!h) return PM_ERR_INVALID_ARG;
  if (int rc = refuse_while_capturing(h, "pm_match_view_device")) return rc;
  if (h->pipe_count > 0) {
    set_err(h, "pm_match_view_device: pairs are in flight (pm_collect them first)");
    return PM_ERR_BUSY;
  }
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
  prof_break_all(h);
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
    launch_prep_view(h, ps, d_iml, d_imr, d_Gl, d_Gr, step / sizeof(float));
  }
  if (int rc = launch_check(h, "prep_view")) return rc;
  {
    Launch l(h, PM_K_PREP);
    if (int rc = run_transpose(h, ps, 1)) return rc;
  }
  {
    Launch l(h, PM_K_SEED);
    launch_copy_disp_strided(h, ps, d_disp, disp_step / sizeof(float), 0);
  }
  if (int rc = launch_check(h, "seed")) return rc;
  h->need_seed[0] = h->need_seed[1] = false;
  if (int rc = run_one_view_set(h, ps, 1)) return rc;
  {
    Launch l(h, PM_K_FINALIZE);
    launch_copy_disp_strided(h, ps, d_disp, disp_step / sizeof(float), 1);
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
  prof_break_all(h);
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
    const hipEvent_t start = r.start_ref >= 0 ? h->ev_pool[r.start_ref].stop : r.start;
    if (hipEventElapsedTime(&ms, start, r.stop) == hipSuccess) {
      h->prof.launches[r.klass] += 1;
      h->prof.total_ms[r.klass] += (double)ms;
    }
  }
  prof_break_all(h);
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
