// pm_hostpath.hip -- the entry points that take HOST buffers (include/pm/patchmatch.h): pm_match_u8 /
// pm_match_batch_u8 (what PatchmatchGpu::Match(cv::Mat...) does, patchmatch_gpu.cu:322-376), the pipelined
// pm_submit_u8 / pm_collect, and the single-stage functions the parity tests drive one by one.  Staging, copies
// and synchronisation only: every kernel is reached through the launch functions of pm_handle.hpp.
#include <cstring>

#include "pm_handle.hpp"

using namespace pm;
using namespace pm::eng;

extern "C" {

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
  // every pair is packed into the pinned buffer by a few host threads (pm_hostcopy.hpp) and its upload enqueued at
  // once: the DMA of pair i runs while the host packs pair i + 1
  if (!h->copy_pool) h->copy_pool = new pm::CopyPool();
  const size_t frow = sizeof(float) * (size_t)cols;
  for (int i = 0; i < n; ++i) {
    if (!left[i] || !right[i] || !disp_l[i] || (lr && !disp_r[i])) {
      set_err(h, "pm_match_batch_u8: null pointer for pair %d", i);
      return PM_ERR_INVALID_ARG;
    }
    h->copy_pool->Copy2D(pl + i * px, (size_t)cols, left[i], (size_t)cols, (size_t)cols, rows);
    PM_HIP(h, hipMemcpyAsync(h->st_left + i * px, pl + i * px, px, hipMemcpyHostToDevice, h->stream));
    h->copy_pool->Copy2D(pr + i * px, (size_t)cols, right[i], (size_t)cols, (size_t)cols, rows);
    PM_HIP(h, hipMemcpyAsync(h->st_right + i * px, pr + i * px, px, hipMemcpyHostToDevice, h->stream));
    if (seed_l && seed_l[i]) {
      h->copy_pool->Copy2D(psl + i * px, frow, seed_l[i], frow, frow, rows);
      PM_HIP(h, hipMemcpyAsync(h->st_seed_l + i * px, psl + i * px, sizeof(float) * px, hipMemcpyHostToDevice, h->stream));
      any_sl = true;
    }
    if (seed_r && seed_r[i]) {
      h->copy_pool->Copy2D(psr + i * px, frow, seed_r[i], frow, frow, rows);
      PM_HIP(h, hipMemcpyAsync(h->st_seed_r + i * px, psr + i * px, sizeof(float) * px, hipMemcpyHostToDevice, h->stream));
      any_sr = true;
    }
  }
  // a view whose maps were given for some pairs only (allowed without sparse_init): the others start from zeros
  for (int i = 0; i < n; ++i) {
    if (any_sl && !(seed_l && seed_l[i])) PM_HIP(h, hipMemsetAsync(h->st_seed_l + i * px, 0, sizeof(float) * px, h->stream));
    if (any_sr && !(seed_r && seed_r[i])) PM_HIP(h, hipMemsetAsync(h->st_seed_r + i * px, 0, sizeof(float) * px, h->stream));
  }
  if (int rc = match_device_impl(h, n, h->st_left, h->st_right, rows, cols, any_sl ? h->st_seed_l : nullptr,
                               any_sr ? h->st_seed_r : nullptr, h->st_disp_l, lr ? h->st_disp_r : nullptr))
    return rc;
  // the left maps are unpacked into the caller's buffers while the right ones are still on the bus
  if (!h->left_out) {
    PM_HIP(h, hipEventCreateWithFlags(&h->left_out, hipEventDisableTiming));
    PM_HIP(h, hipEventCreateWithFlags(&h->right_out, hipEventDisableTiming));
  }
  PM_HIP(h, hipMemcpyAsync(pdl, h->st_disp_l, sizeof(float) * n * px, hipMemcpyDeviceToHost, h->stream));
  PM_HIP(h, hipEventRecord(h->left_out, h->stream));
  if (lr) {
    PM_HIP(h, hipMemcpyAsync(pdr, h->st_disp_r, sizeof(float) * n * px, hipMemcpyDeviceToHost, h->stream));
    PM_HIP(h, hipEventRecord(h->right_out, h->stream));
  }
  PM_HIP(h, hipEventSynchronize(h->left_out));
  for (int i = 0; i < n; ++i)
    h->copy_pool->Copy2D(disp_l[i], frow, pdl + i * px, frow, frow, rows);
  if (lr) {
    PM_HIP(h, hipEventSynchronize(h->right_out));
    for (int i = 0; i < n; ++i)
      h->copy_pool->Copy2D(disp_r[i], frow, pdr + i * px, frow, frow, rows);
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
}  // extern "C"

namespace {

int pipe_init(pm_handle* h) {
  if (!h->pipe.empty()) return PM_OK;
  PM_HIP(h, create_stream(&h->s_in, kStreamCopy));
  PM_HIP(h, create_stream(&h->s_out, kStreamCopy));
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

extern "C" {

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
  // Every frame computes on the handle's stream.  (Frames on lanes of their own -- what helps device-resident callers,
  // tools/multi_handle.py -- lose here: 368 -> 350 pairs/s with one to three lanes, tools/pipe_timing.py.)
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

}  // extern "C"

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
  launch_prep(h, ps, h->st_left, h->st_right, 1, (size_t)cols);
  if (int rc = launch_check(h, "prep")) return rc;
  if (int rc = run_transpose(h, ps, 1)) return rc;
  *ps_out = ps;
  return PM_OK;
}

int stage_disp_in(pm_handle* h, const PlaneSet& ps, const float* disp) {
  const size_t px = (size_t)ps.rows * ps.cols;
  PM_HIP(h, hipMemcpyAsync(h->st_disp_l, disp, sizeof(float) * px, hipMemcpyHostToDevice, h->stream));
  launch_copy_in(h, ps, h->st_disp_l);
  return launch_check(h, "copy_in");
}

int stage_out(pm_handle* h, const PlaneSet& ps, float* dst, int which) {
  const size_t px = (size_t)ps.rows * ps.cols;
  launch_copy_out(h, ps, h->st_disp_l, which);
  if (int rc = launch_check(h, "copy_out")) return rc;
  PM_HIP(h, hipMemcpyAsync(dst, h->st_disp_l, sizeof(float) * px, hipMemcpyDeviceToHost, h->stream));
  PM_HIP(h, hipStreamSynchronize(h->stream));
  return PM_OK;
}

}  // namespace

extern "C" {

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
  launch_noise_only(h, ps, cp, amount);
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
  PM_HIP(h, seed_sparse_init(h->seeds[0], seed_params(h->params), ps.img8, ps.img8 + ps.plane, rows, cols, ps.pitch,
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
  PM_HIP(h, seed_initialize(h->seeds[0], seed_params(h->params), ps.img8, ps.img8 + ps.plane, rows, cols, ps.pitch,
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
  launch_mask_occlusions(h, h->st_disp_l, h->st_disp_r, rows, cols);
  if (int rc = launch_check(h, "mask_occlusions")) return rc;
  PM_HIP(h, hipMemcpyAsync(disp_l, h->st_disp_l, sizeof(float) * px, hipMemcpyDeviceToHost, h->stream));
  PM_HIP(h, hipStreamSynchronize(h->stream));
  return PM_OK;
}

}  // extern "C"
