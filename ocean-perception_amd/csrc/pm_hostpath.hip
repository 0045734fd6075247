// pm_hostpath.hip -- the entry points that take HOST buffers (include/pm/patchmatch.h): pm_match_u8 /
// pm_match_batch_u8 (what PatchmatchGpu::Match(cv::Mat...) does, patchmatch_gpu.cu:322-376), the pipelined
// pm_submit_u8 / pm_collect, and the single-stage functions the parity tests drive one by one.  Staging, copies
// and synchronisation only: every kernel is reached through the launch functions of pm_handle.hpp.
#include <cstring>

#include "pm_handle.hpp"

using namespace pm;
using namespace pm::eng;

namespace {

// ---- caller memory known to be page-locked (pm_host_alloc / pm_host_register) ---------------------------------------
bool host_pinned(const pm_handle* h, const void* p, size_t span) {
  const char* c = (const char*)p;
  for (const auto& r : h->host_ranges)
    if (c >= r.base && c + span <= r.base + r.bytes) return true;
  return false;
}
size_t span_bytes(int rows, size_t step, size_t row_bytes) { return (size_t)(rows - 1) * step + row_bytes; }

// rows x row_bytes from host memory (row stride `step`) into tight device memory on `stream`: by DMA straight from the
// caller's buffer when it is page-locked, else packed into `slab` (page-locked, the handle's) by the copy pool first.
int upload_plane(pm_handle* h, void* d_dst, const void* src, size_t step, size_t row_bytes, int rows, void* slab,
                 hipStream_t stream) {
  if (host_pinned(h, src, span_bytes(rows, step, row_bytes))) {
    if (step == row_bytes)
      PM_HIP(h, hipMemcpyAsync(d_dst, src, row_bytes * (size_t)rows, hipMemcpyHostToDevice, stream));
    else
      PM_HIP(h, hipMemcpy2DAsync(d_dst, row_bytes, src, step, row_bytes, (size_t)rows, hipMemcpyHostToDevice, stream));
    return PM_OK;
  }
  h->copy_pool->Copy2D(slab, row_bytes, src, step, row_bytes, rows);
  PM_HIP(h, hipMemcpyAsync(d_dst, slab, row_bytes * (size_t)rows, hipMemcpyHostToDevice, stream));
  return PM_OK;
}
// the device's address of page-locked host memory, or null
char* host_dev_address(const pm_handle* h, const void* p, size_t span) {
  const char* c = (const char*)p;
  if (h->pinned_dev && c >= (const char*)h->pinned && c + span <= (const char*)h->pinned + h->pinned_bytes)
    return h->pinned_dev + (c - (const char*)h->pinned);
  for (const auto& r : h->host_ranges)
    if (r.dev_base && c >= r.base && c + span <= r.base + r.bytes) return r.dev_base + (c - r.base);
  return nullptr;
}

// The other way, for float maps: *direct = the map goes straight into `dst` (else it waits in `slab` for the unpack).
// Page-locked targets the device can address are written by k_download (a handful of wavefronts), not by the runtime's
// device-to-host copy, which on a busy stream is a full-size blit kernel (pm_kernels.hpp).
int download_plane(pm_handle* h, void* dst, size_t step, const void* d_src, size_t row_bytes, int rows, void* slab,
                   hipStream_t stream, bool* direct) {
  *direct = dst && host_pinned(h, dst, span_bytes(rows, step, row_bytes));
  void* target = *direct ? dst : slab;
  const size_t tstep = *direct ? step : row_bytes;
  char* dev = (tstep % sizeof(float)) == 0 ? host_dev_address(h, target, span_bytes(rows, tstep, row_bytes)) : nullptr;
  if (dev) {
    launch_download(h, (float*)dev, tstep / sizeof(float), (const float*)d_src, rows, (int)(row_bytes / sizeof(float)), stream);
    return launch_check(h, "download");
  }
  if (tstep == row_bytes)
    PM_HIP(h, hipMemcpyAsync(target, d_src, row_bytes * (size_t)rows, hipMemcpyDeviceToHost, stream));
  else
    PM_HIP(h, hipMemcpy2DAsync(target, tstep, d_src, row_bytes, row_bytes, (size_t)rows, hipMemcpyDeviceToHost, stream));
  return PM_OK;
}

constexpr size_t kSmallPairBytes = 1u << 20;  // both images of a pair that goes up as one kernel copy (pm_match_u8)

struct PinnedSlot {
  float *sl, *sr, *dl, *dr;
  uint8_t *l, *r;
};
PinnedSlot pinned_slot(pm_handle* h, int slot, size_t px) {
  const size_t tight = (size_t)h->max_rows * h->max_cols;
  char* base = (char*)h->pinned + (size_t)slot * tight * (2 + 4 * sizeof(float));
  PinnedSlot p;
  p.sl = (float*)base;  // floats first so every sub-buffer stays 4-byte aligned
  p.sr = p.sl + px;
  p.dl = p.sr + px;
  p.dr = p.dl + px;
  p.l = (uint8_t*)(p.dr + px);
  p.r = p.l + px;
  return p;
}

int pipe_init(pm_handle* h) {
  if (!h->copy_pool) h->copy_pool = new pm::CopyPool();
  if (int rc = seq_events_create(h)) return rc;
  if (seq_pipelined(h))
    if (int rc = view_streams_create(h)) return rc;
  return PM_OK;
}

// Frames [b, b + c) of the ring -- uploaded (or device resident) -- become ONE chunk of the sequence: both views on the
// view streams behind the later frame's upload, cross-check and downloads on s_out.  Handles that cannot run as chunks
// (plane mode, one view only) run the frame on the handle's stream instead, one frame after the other.
int enqueue_frames(pm_handle* h, int b, int c) {
  pm_handle::PipeSlot& f0 = h->pipe[(size_t)b];
  const int rows = f0.rows, cols = f0.cols;
  const size_t px = (size_t)rows * cols;
  const bool lr = h->params.left_right_check != 0;
  // host frames: the later frame's upload (s_in runs in order).  Device-resident frames: the caller's "inputs are
  // complete" events (pm_submit_device_after), one per frame of the chunk, in front of both views and the head
  hipEvent_t ready = f0.device_io ? f0.ready_ext : h->pipe[(size_t)(b + c - 1)].in_done;
  hipEvent_t ready2 = (f0.device_io && c > 1) ? h->pipe[(size_t)(b + c - 1)].ready_ext : nullptr;
  if (seq_pipelined(h)) {
    if (int rc = seq_enqueue_chunk(h, b, c, f0.d_left, f0.d_right, rows, cols, f0.d_seed_l, f0.d_seed_r, f0.d_out_l,
                                   f0.d_out_r, ready, ready2, f0.v_done, f0.head_done))
      return rc;
  } else {
    if (ready) PM_HIP(h, hipStreamWaitEvent(h->stream, ready, 0));
    if (ready2) PM_HIP(h, hipStreamWaitEvent(h->stream, ready2, 0));
    if (c > 1 && h->params.mode == PM_MODE_PLANES) {
      // plane mode: the frames of a chunk are neighbours in memory (can_gang) and run as ONE batch -- two lanes on two
      // streams, each filling the other's launch tails (pm_planes_host.hip::planes_match)
      if (int rc = match_device_impl(h, c, f0.d_left, f0.d_right, rows, cols, f0.d_seed_l, f0.d_seed_r, f0.d_out_l,
                                     lr ? f0.d_out_r : nullptr))
        return rc;
    } else {
      for (int i = 0; i < c; ++i) {
        pm_handle::PipeSlot& f = h->pipe[(size_t)(b + i)];
        if (int rc = match_device_impl(h, 1, f.d_left, f.d_right, rows, cols, f.d_seed_l, f.d_seed_r, f.d_out_l,
                                       lr ? f.d_out_r : nullptr))
          return rc;
      }
    }
    PM_HIP(h, hipEventRecord(f0.v_done[0], h->stream));
    PM_HIP(h, hipStreamWaitEvent(h->s_out, f0.v_done[0], 0));
  }
  for (int i = 0; i < c; ++i) {
    pm_handle::PipeSlot& f = h->pipe[(size_t)(b + i)];
    if (!f.device_io) {
      const PinnedSlot ps = pinned_slot(h, b + i, px);
      const size_t row_bytes = sizeof(float) * (size_t)cols;
      const size_t step = f.out_step ? f.out_step : row_bytes;
      if (int rc = download_plane(h, f.out_l, step, f.d_out_l, row_bytes, rows, ps.dl, h->s_out, &f.direct_l)) return rc;
      if (lr)
        if (int rc = download_plane(h, f.out_r, step, f.d_out_r, row_bytes, rows, ps.dr, h->s_out, &f.direct_r)) return rc;
    }
    PM_HIP(h, hipEventRecord(f.out_done, h->s_out));
    f.state = 2;
  }
  h->seq_last = b + c - 1;
  return PM_OK;
}

// Is the device still busy with the chunk enqueued last?  (Only then does holding a frame for a partner cost nothing.)
bool device_busy(const pm_handle* h) {
  if (h->seq_last < 0) return false;
  const pm_handle::PipeSlot& f = h->pipe[(size_t)h->seq_last];
  return f.state == 2 && hipEventQuery(f.out_done) == hipErrorNotReady;
}

// the ring slot being held for a partner, or -1
int held_slot(const pm_handle* h) {
  for (int i = 0; i < h->pipe_count; ++i) {
    const int s = (h->pipe_head + i) % h->max_batch;
    if (h->pipe[(size_t)s].state == 1) return s;
  }
  return -1;
}

// two frames can be advanced through every launch together if they are neighbours in device memory
bool can_gang(const pm_handle* h, const pm_handle::PipeSlot& a, const pm_handle::PipeSlot& b, int sa, int sb) {
  if (sb != sa + 1 || a.rows != b.rows || a.cols != b.cols || a.has_sl != b.has_sl || a.has_sr != b.has_sr ||
      a.device_io != b.device_io)
    return false;
  const size_t px = (size_t)a.rows * a.cols;
  return b.d_left == a.d_left + px && b.d_right == a.d_right + px && (!a.has_sl || b.d_seed_l == a.d_seed_l + px) &&
         (!a.has_sr || b.d_seed_r == a.d_seed_r + px) && b.d_out_l == a.d_out_l + px &&
         (!h->params.left_right_check || b.d_out_r == a.d_out_r + px);
}

// a new frame starts at once on an idle device and is held for a partner while the device is busy with earlier chunks
// anyway and a neighbour slot exists
int enqueue_or_hold(pm_handle* h, int slot) {
  // (plane mode is not pipelined over the view streams, but two of its frames make a batch of two lanes)
  const bool gangs = seq_pipelined(h) || (h->params.mode == PM_MODE_PLANES && !h->bgr);
  const bool hold = gangs && seq_chunk_pairs() >= 2 && slot + 1 < h->max_batch && device_busy(h);
  return hold ? PM_OK : enqueue_frames(h, slot, 1);
}

struct SubmitArgs {
  const uint8_t *left, *right;
  int rows, cols;
  size_t image_step;
  const float *seed_l, *seed_r;
  size_t seed_step;
  float *out_l, *out_r;  // host maps bound at submit, or (device_io) the device maps
  size_t out_step;
  uint64_t tag;
  bool device_io;
  hipEvent_t ready = nullptr;  // device_io: the caller's event behind the producer of the inputs, or null
};

int submit_impl(pm_handle* h, const SubmitArgs& a, const char* what) {
  if (int rc = refuse_while_capturing(h, what)) return rc;
  if (!a.left || !a.right) {
    set_err(h, "%s: null image pointer", what);
    return PM_ERR_INVALID_ARG;
  }
  const int rows = a.rows, cols = a.cols;
  if (int rc = check_size(h, rows, cols, 1)) return rc;
  const size_t image_step = a.image_step ? a.image_step : (size_t)cols;
  const size_t frow = sizeof(float) * (size_t)cols;
  const size_t seed_step = a.seed_step ? a.seed_step : frow;
  const size_t out_step = a.out_step ? a.out_step : frow;
  if (image_step < (size_t)cols || seed_step < frow || out_step < frow) {
    set_err(h, "%s: a row step is smaller than a row", what);
    return PM_ERR_INVALID_ARG;
  }
  const bool lr = h->params.left_right_check != 0;
  if (a.device_io && (!a.out_l || (lr && !a.out_r))) {
    set_err(h, "%s: null output pointer", what);
    return PM_ERR_INVALID_ARG;
  }
  if (!a.device_io && a.out_l && lr && !a.out_r) {
    set_err(h, "%s: disp_r required when left_right_check is set", what);
    return PM_ERR_INVALID_ARG;
  }
  PM_HIP(h, hipSetDevice(h->device));
  if (int rc = pipe_init(h)) return rc;
  if (h->pipe_count == h->max_batch) {
    set_err(h, "%s: %d pairs in flight (the plan's max_batch); collect one first", what, h->pipe_count);
    return PM_ERR_BUSY;
  }
  if (h->pipe_count > 0) {
    const pm_handle::PipeSlot& first = h->pipe[(size_t)h->pipe_head];
    if (first.rows != rows || first.cols != cols) {  // the frames in flight share the noise table and the slot layout
      set_err(h, "%s: image size changed with pairs in flight; collect them first", what);
      return PM_ERR_BUSY;
    }
  } else if (h->params.mode == PM_MODE_SCALAR) {
    if (int rc = ensure_noise(h, rows, cols)) return rc;
  }
  const int slot = (h->pipe_head + h->pipe_count) % h->max_batch;
  pm_handle::PipeSlot& sl = h->pipe[(size_t)slot];
  const size_t px = (size_t)rows * cols;
  sl.tag = a.tag;
  sl.rows = rows;
  sl.cols = cols;
  sl.has_sl = a.seed_l != nullptr;
  sl.has_sr = a.seed_r != nullptr;
  sl.device_io = a.device_io;
  sl.direct_l = sl.direct_r = false;
  if (a.device_io) {
    sl.d_left = a.left;
    sl.d_right = a.right;
    sl.d_seed_l = a.seed_l;
    sl.d_seed_r = a.seed_r;
    sl.d_out_l = a.out_l;
    sl.d_out_r = a.out_r;
    sl.out_l = sl.out_r = nullptr;
    sl.out_step = 0;
    // The caller's "inputs are complete" event is CONSUMED here: s_in waits for it and the slot's own event is recorded
    // behind it.  A held frame is only enqueued by a later submit / collect / flush -- by then the caller may have
    // re-recorded or destroyed its event (a torch.cuda.Event dropped after the call); the slot's event is ours.
    sl.ready_ext = nullptr;
    if (a.ready) {
      PM_HIP(h, hipStreamWaitEvent(h->s_in, a.ready, 0));
      PM_HIP(h, hipEventRecord(sl.in_done, h->s_in));
      sl.ready_ext = sl.in_done;
    }
  } else {
    // ring slot k keeps its inputs and outputs at offset k * px of the staging arrays: frames in flight share one size
    uint8_t* dl8 = h->st_left + (size_t)slot * px;
    uint8_t* dr8 = h->st_right + (size_t)slot * px;
    float* dsl = h->st_seed_l + (size_t)slot * px;
    float* dsr = h->st_seed_r + (size_t)slot * px;
    const PinnedSlot ps = pinned_slot(h, slot, px);
    if (int rc = upload_plane(h, dl8, a.left, image_step, (size_t)cols, rows, ps.l, h->s_in)) return rc;
    if (int rc = upload_plane(h, dr8, a.right, image_step, (size_t)cols, rows, ps.r, h->s_in)) return rc;
    if (a.seed_l)
      if (int rc = upload_plane(h, dsl, a.seed_l, seed_step, frow, rows, ps.sl, h->s_in)) return rc;
    if (a.seed_r)
      if (int rc = upload_plane(h, dsr, a.seed_r, seed_step, frow, rows, ps.sr, h->s_in)) return rc;
    PM_HIP(h, hipEventRecord(sl.in_done, h->s_in));
    sl.d_left = dl8;
    sl.d_right = dr8;
    sl.d_seed_l = a.seed_l ? dsl : nullptr;
    sl.d_seed_r = a.seed_r ? dsr : nullptr;
    sl.d_out_l = h->st_disp_l + (size_t)slot * px;
    sl.d_out_r = h->st_disp_r + (size_t)slot * px;
    sl.out_l = a.out_l;
    sl.out_r = a.out_r;
    sl.out_step = a.out_l ? out_step : 0;
  }
  sl.state = 1;
  ++h->pipe_count;
  // Chunks: a frame that is being held takes this one as its partner if the two are neighbours in memory, and goes alone
  // otherwise.  This frame is held in turn while the device is busy with earlier chunks anyway and a neighbour slot
  // exists for a partner; with an idle device it starts at once.
  const int held = held_slot(h) == slot ? -1 : held_slot(h);
  int rc = PM_OK;
  if (held >= 0) {
    if (can_gang(h, h->pipe[(size_t)held], sl, held, slot))
      rc = enqueue_frames(h, held, 2);
    else if ((rc = enqueue_frames(h, held, 1)) == PM_OK)
      rc = enqueue_or_hold(h, slot);
  } else {
    rc = enqueue_or_hold(h, slot);
  }
  if (rc != PM_OK) {
    // An enqueue that failed midway may have left launches of this frame (or of the partner it was ganged with) on the
    // streams.  Nothing of them may outlive this call: the caller is told "not submitted" and is free to reuse its
    // buffers.  After the wait, a partner that is still marked as held is simply enqueued again by the next submit /
    // collect -- a Match() is a pure function of its inputs, a second run rewrites the same maps.
    for (hipStream_t q : {h->stream, h->view1_stream, h->s_in, h->s_out})
      if (q) (void)hipStreamSynchronize(q);
    (void)hipGetLastError();
    if (sl.state == 1) {
      // the frame leaves the ring again, so that the caller's count of frames in flight (an error return = nothing
      // submitted) stays right
      sl.state = 0;
      --h->pipe_count;
    }
  }
  return rc;
}

}  // namespace

extern "C" {

int pm_host_alloc(pm_handle* h, size_t bytes, void** ptr) {
  if (!h || !ptr || bytes == 0) return PM_ERR_INVALID_ARG;
  *ptr = nullptr;
  PM_HIP(h, hipSetDevice(h->device));
  void* p = nullptr;
  PM_HIP(h, hipHostMalloc(&p, bytes, hipHostMallocDefault));
  void* dp = nullptr;
  if (hipHostGetDevicePointer(&dp, p, 0) != hipSuccess) dp = nullptr;
  (void)hipGetLastError();
  h->host_ranges.push_back({(char*)p, bytes, true, (char*)dp});
  *ptr = p;
  return PM_OK;
}

int pm_host_register(pm_handle* h, void* ptr, size_t bytes) {
  if (!h || !ptr || bytes == 0) return PM_ERR_INVALID_ARG;
  PM_HIP(h, hipSetDevice(h->device));
  PM_HIP(h, hipHostRegister(ptr, bytes, hipHostRegisterMapped));
  void* dp = nullptr;
  if (hipHostGetDevicePointer(&dp, ptr, 0) != hipSuccess) dp = nullptr;
  (void)hipGetLastError();
  h->host_ranges.push_back({(char*)ptr, bytes, false, (char*)dp});
  return PM_OK;
}

static int host_release(pm_handle* h, void* ptr, bool owned, const char* what) {
  if (!h || !ptr) return PM_ERR_INVALID_ARG;
  if (h->pipe_count > 0) {
    set_err(h, "%s: pairs are in flight (they may be reading or writing this memory); collect them first", what);
    return PM_ERR_BUSY;
  }
  for (size_t i = 0; i < h->host_ranges.size(); ++i) {
    if (h->host_ranges[i].base != (char*)ptr || h->host_ranges[i].owned != owned) continue;
    PM_HIP(h, hipSetDevice(h->device));
    // a synchronous entry point may have left a DMA into this range running only if it returned an error; be safe
    PM_HIP(h, hipStreamSynchronize(h->stream));
    h->host_ranges.erase(h->host_ranges.begin() + (long)i);
    if (owned)
      PM_HIP(h, hipHostFree(ptr));
    else
      PM_HIP(h, hipHostUnregister(ptr));
    return PM_OK;
  }
  set_err(h, "%s: %p is not the start of a range this handle %s", what, ptr, owned ? "allocated" : "registered");
  return PM_ERR_INVALID_ARG;
}
int pm_host_free(pm_handle* h, void* ptr) { return host_release(h, ptr, true, "pm_host_free"); }
int pm_host_unregister(pm_handle* h, void* ptr) { return host_release(h, ptr, false, "pm_host_unregister"); }

// n pairs per call as chunks of the frame sequence: the uploads of pair i + 1 run while pair i is matched, the maps of
// finished chunks come back (and are unpacked by this thread) while later chunks are matched.
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
  if (h->pipe_count > 0) {
    set_err(h, "pm_match_batch_u8: pairs are in flight (pm_collect them first)");
    return PM_ERR_BUSY;
  }
  int nl = 0, nr = 0;
  for (int i = 0; i < n; ++i) {
    if (!left[i] || !right[i] || !disp_l[i] || (lr && !disp_r[i])) {
      set_err(h, "pm_match_batch_u8: null pointer for pair %d", i);
      return PM_ERR_INVALID_ARG;
    }
    nl += (seed_l && seed_l[i]) ? 1 : 0;
    nr += (seed_r && seed_r[i]) ? 1 : 0;
  }
  // With sparse_init a missing seed map means "seed this view on the device", which is decided per call, not per
  // pair: a batch must give the seed map of a view for every pair or for none.
  if (h->params.sparse_init && ((nl != 0 && nl != n) || (nr != 0 && nr != n))) {
    set_err(h, "pm_match_batch_u8: with sparse_init a view's seed maps must be given for all pairs or for none");
    return PM_ERR_INVALID_ARG;
  }
  PM_HIP(h, hipSetDevice(h->device));
  if (int rc = pipe_init(h)) return rc;
  if (h->params.mode == PM_MODE_SCALAR)
    if (int rc = ensure_noise(h, rows, cols)) return rc;
  const size_t px = (size_t)rows * cols;
  const size_t frow = sizeof(float) * (size_t)cols;
  const bool any_sl = nl > 0, any_sr = nr > 0;
  const bool chunks = seq_pipelined(h);
  const int chunk = chunks ? seq_chunk_pairs() : n;
  h->pipe_head = 0;
  for (int b = 0; b < n; b += chunk) {
    const int c = n - b < chunk ? n - b : chunk;
    for (int i = b; i < b + c; ++i) {
      pm_handle::PipeSlot& sl = h->pipe[(size_t)i];
      const PinnedSlot ps = pinned_slot(h, i, px);
      uint8_t* dl8 = h->st_left + (size_t)i * px;
      uint8_t* dr8 = h->st_right + (size_t)i * px;
      float* dsl = h->st_seed_l + (size_t)i * px;
      float* dsr = h->st_seed_r + (size_t)i * px;
      if (int rc = upload_plane(h, dl8, left[i], (size_t)cols, (size_t)cols, rows, ps.l, h->s_in)) return rc;
      if (int rc = upload_plane(h, dr8, right[i], (size_t)cols, (size_t)cols, rows, ps.r, h->s_in)) return rc;
      // a view whose maps were given for some pairs only (allowed without sparse_init): the others start from zeros
      if (seed_l && seed_l[i]) {
        if (int rc = upload_plane(h, dsl, seed_l[i], frow, frow, rows, ps.sl, h->s_in)) return rc;
      } else if (any_sl) {
        PM_HIP(h, hipMemsetAsync(dsl, 0, sizeof(float) * px, h->s_in));
      }
      if (seed_r && seed_r[i]) {
        if (int rc = upload_plane(h, dsr, seed_r[i], frow, frow, rows, ps.sr, h->s_in)) return rc;
      } else if (any_sr) {
        PM_HIP(h, hipMemsetAsync(dsr, 0, sizeof(float) * px, h->s_in));
      }
      PM_HIP(h, hipEventRecord(sl.in_done, h->s_in));
      sl.tag = 0;
      sl.rows = rows;
      sl.cols = cols;
      sl.has_sl = any_sl;
      sl.has_sr = any_sr;
      sl.device_io = false;
      sl.d_left = dl8;
      sl.d_right = dr8;
      sl.d_seed_l = any_sl ? dsl : nullptr;
      sl.d_seed_r = any_sr ? dsr : nullptr;
      sl.d_out_l = h->st_disp_l + (size_t)i * px;
      sl.d_out_r = h->st_disp_r + (size_t)i * px;
      sl.out_l = disp_l[i];
      sl.out_r = lr ? disp_r[i] : nullptr;
      sl.out_step = frow;
      sl.state = 1;
    }
    if (chunks) {
      if (int rc = enqueue_frames(h, b, c)) return rc;
    } else {
      // all pairs through every launch together on the handle's stream (plane mode, single view)
      PM_HIP(h, hipStreamWaitEvent(h->stream, h->pipe[(size_t)(n - 1)].in_done, 0));
      if (int rc = match_device_impl(h, n, h->st_left, h->st_right, rows, cols, any_sl ? h->st_seed_l : nullptr,
                                     any_sr ? h->st_seed_r : nullptr, h->st_disp_l, lr ? h->st_disp_r : nullptr))
        return rc;
      PM_HIP(h, hipEventRecord(h->pipe[0].v_done[0], h->stream));
      PM_HIP(h, hipStreamWaitEvent(h->s_out, h->pipe[0].v_done[0], 0));
      for (int i = 0; i < n; ++i) {
        pm_handle::PipeSlot& f = h->pipe[(size_t)i];
        const PinnedSlot ps = pinned_slot(h, i, px);
        if (int rc = download_plane(h, f.out_l, frow, f.d_out_l, frow, rows, ps.dl, h->s_out, &f.direct_l)) return rc;
        if (lr)
          if (int rc = download_plane(h, f.out_r, frow, f.d_out_r, frow, rows, ps.dr, h->s_out, &f.direct_r)) return rc;
        PM_HIP(h, hipEventRecord(f.out_done, h->s_out));
      }
    }
  }
  int rc_all = PM_OK;
  for (int i = 0; i < n; ++i) {
    pm_handle::PipeSlot& f = h->pipe[(size_t)i];
    f.state = 0;
    if (hipEventSynchronize(f.out_done) != hipSuccess) {
      set_err(h, "pm_match_batch_u8: waiting for pair %d failed", i);
      rc_all = PM_ERR_HIP;
      continue;
    }
    const PinnedSlot ps = pinned_slot(h, i, px);
    if (!f.direct_l) h->copy_pool->Copy2D(disp_l[i], frow, ps.dl, frow, frow, rows);
    if (lr && !f.direct_r) h->copy_pool->Copy2D(disp_r[i], frow, ps.dr, frow, frow, rows);
  }
  h->seq_last = -1;
  return rc_all;
}

#ifdef PM_HOST_PHASES  // analysis builds: where the HOST spends a pm_match_u8 call (stderr, every 25th call)
#include <chrono>
namespace {
struct HostPhases {
  static constexpr int kN = 8;
  double sum[kN] = {};
  std::chrono::steady_clock::time_point last, exit_of_last;
  bool have_exit = false;
  int calls = 0;
  void begin() {
    last = std::chrono::steady_clock::now();
    if (have_exit) sum[0] += std::chrono::duration<double, std::micro>(last - exit_of_last).count();
  }
  void mark(int k) {
    const auto t = std::chrono::steady_clock::now();
    sum[k] += std::chrono::duration<double, std::micro>(t - last).count();
    last = t;
  }
  void end() {
    exit_of_last = std::chrono::steady_clock::now();
    have_exit = true;
    if (++calls % 25 == 0) {
      const char* names[kN] = {"between calls", "checks + noise", "uploads", "enqueue", "downloads enqueued", "wait left", "unpack left", "wait + unpack right"};
      fprintf(stderr, "pm_match_u8 host phases, us per call:");
      for (int k = 0; k < kN; ++k) fprintf(stderr, " %s %.1f;", names[k], sum[k] / 25.0), sum[k] = 0.0;
      fprintf(stderr, "\n");
    }
  }
};
HostPhases g_hp;
}  // namespace
#define HP(x) g_hp.x
#else
#define HP(x) (void)0
#endif

int pm_match_u8(pm_handle* h, const uint8_t* left, const uint8_t* right, int rows, int cols, size_t image_step,
                const float* seed_l, const float* seed_r, size_t seed_step, float* disp_l, float* disp_r,
                size_t disp_step) {
  if (!h) return PM_ERR_INVALID_ARG;
  HP(begin());
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
  if (h->pipe_count > 0) {
    set_err(h, "pm_match_u8: pairs are in flight (pm_collect them first)");
    return PM_ERR_BUSY;
  }
  const size_t frow = sizeof(float) * (size_t)cols;
  if (image_step == 0) image_step = (size_t)cols;
  if (seed_step == 0) seed_step = frow;
  if (disp_step == 0) disp_step = frow;
  if (image_step < (size_t)cols || seed_step < frow || disp_step < frow) {
    set_err(h, "pm_match_u8: a row step is smaller than a row");
    return PM_ERR_INVALID_ARG;
  }
  PM_HIP(h, hipSetDevice(h->device));
  if (int rc = ensure_noise(h, rows, cols)) return rc;
  const size_t px = (size_t)rows * cols;
  const PinnedSlot ps = pinned_slot(h, 0, px);
  // every plane goes up as soon as it is ready -- by DMA from the caller's buffer if that is page-locked, else packed
  // into the pinned slab by a few host threads (pm_hostcopy.hpp): the DMA of one plane runs while the host packs the next
  if (!h->copy_pool) h->copy_pool = new pm::CopyPool();
  HP(mark(1));
  // A small pair in ordinary (not page-locked) memory: both images are packed back to back into the pinned slab and go up
  // as ONE copy, made by a few wavefronts reading the slab through its device address -- a DMA copy per image costs
  // ~10 us of engine latency each, a quarter of what the device then needs for the reference's own 376 x 240 pair.
  // Larger pairs keep the copy engines: the DMA of one plane overlaps the packing of the next, and no CU waits on the bus.
  uint8_t* d_right = h->st_right;
  const size_t span = span_bytes(rows, image_step, (size_t)cols);
  char* slab_dev = (px % 16 == 0 && 2 * px <= kSmallPairBytes && !host_pinned(h, left, span) && !host_pinned(h, right, span))
                       ? host_dev_address(h, ps.l, 2 * px) : nullptr;
  if (slab_dev && ((uintptr_t)slab_dev % 16) == 0 && ((uintptr_t)h->st_left % 16) == 0) {
    h->copy_pool->Copy2D(ps.l, (size_t)cols, left, image_step, (size_t)cols, rows);
    h->copy_pool->Copy2D(ps.r, (size_t)cols, right, image_step, (size_t)cols, rows);
    d_right = h->st_left + px;
    const int words = (int)(2 * px / sizeof(float));
    launch_upload(h, (float*)h->st_left, (const float*)slab_dev, words, h->stream);
    if (int rc = launch_check(h, "upload")) return rc;
  } else {
    if (int rc = upload_plane(h, h->st_left, left, image_step, (size_t)cols, rows, ps.l, h->stream)) return rc;
    if (int rc = upload_plane(h, h->st_right, right, image_step, (size_t)cols, rows, ps.r, h->stream)) return rc;
  }
  if (seed_l)
    if (int rc = upload_plane(h, h->st_seed_l, seed_l, seed_step, frow, rows, ps.sl, h->stream)) return rc;
  if (seed_r)
    if (int rc = upload_plane(h, h->st_seed_r, seed_r, seed_step, frow, rows, ps.sr, h->stream)) return rc;
  HP(mark(2));
  if (int rc = match_device_impl(h, 1, h->st_left, d_right, rows, cols, seed_l ? h->st_seed_l : nullptr,
                               seed_r ? h->st_seed_r : nullptr, h->st_disp_l, lr ? h->st_disp_r : nullptr))
    return rc;
  HP(mark(3));
  // the left map is unpacked into the caller's buffer while the right one is still on the bus
  if (!h->left_out) {
    PM_HIP(h, hipEventCreateWithFlags(&h->left_out, hipEventDisableTiming));
    PM_HIP(h, hipEventCreateWithFlags(&h->right_out, hipEventDisableTiming));
  }
  bool direct_l = false, direct_r = false;
  if (int rc = download_plane(h, disp_l, disp_step, h->st_disp_l, frow, rows, ps.dl, h->stream, &direct_l)) return rc;
  PM_HIP(h, hipEventRecord(h->left_out, h->stream));
  if (lr) {
    if (int rc = download_plane(h, disp_r, disp_step, h->st_disp_r, frow, rows, ps.dr, h->stream, &direct_r)) return rc;
    PM_HIP(h, hipEventRecord(h->right_out, h->stream));
  }
  HP(mark(4));
  PM_HIP(h, hipEventSynchronize(h->left_out));
  HP(mark(5));
  if (!direct_l) h->copy_pool->Copy2D(disp_l, disp_step, ps.dl, frow, frow, rows);
  HP(mark(6));
  if (lr) {
    PM_HIP(h, hipEventSynchronize(h->right_out));
    if (!direct_r) h->copy_pool->Copy2D(disp_r, disp_step, ps.dr, frow, frow, rows);
  }
  HP(mark(7));
  HP(end());
  return PM_OK;
}

// ---- the frame sequence -------------------------------------------------------------------------------------------
// What the Sequence caller of the reference does frame by frame (patchmatch_gpu_test.cpp:118-128), with the copies off
// the critical path and consecutive frames overlapping on the device (pm_handle::PipeSlot, seq_enqueue_chunk).

int pm_submit_u8(pm_handle* h, const uint8_t* left, const uint8_t* right, int rows, int cols, size_t image_step,
                 const float* seed_l, const float* seed_r, size_t seed_step, uint64_t tag) {
  if (!h) return PM_ERR_INVALID_ARG;
  const SubmitArgs a{left, right, rows, cols, image_step, seed_l, seed_r, seed_step, nullptr, nullptr, 0, tag, false};
  return submit_impl(h, a, "pm_submit_u8");
}

int pm_submit_bound_u8(pm_handle* h, const uint8_t* left, const uint8_t* right, int rows, int cols, size_t image_step,
                       const float* seed_l, const float* seed_r, size_t seed_step, float* disp_l, float* disp_r,
                       size_t disp_step, uint64_t tag) {
  if (!h) return PM_ERR_INVALID_ARG;
  if (!disp_l) {
    set_err(h, "pm_submit_bound_u8: null output pointer");
    return PM_ERR_INVALID_ARG;
  }
  const SubmitArgs a{left, right, rows, cols, image_step, seed_l, seed_r, seed_step, disp_l, disp_r, disp_step, tag, false};
  return submit_impl(h, a, "pm_submit_bound_u8");
}

int pm_submit_device(pm_handle* h, const uint8_t* d_left, const uint8_t* d_right, int rows, int cols,
                     const float* d_seed_l, const float* d_seed_r, float* d_disp_l, float* d_disp_r, uint64_t tag) {
  if (!h) return PM_ERR_INVALID_ARG;
  const SubmitArgs a{d_left, d_right, rows, cols, 0, d_seed_l, d_seed_r, 0, d_disp_l, d_disp_r, 0, tag, true, nullptr};
  return submit_impl(h, a, "pm_submit_device");
}

int pm_submit_device_after(pm_handle* h, const uint8_t* d_left, const uint8_t* d_right, int rows, int cols,
                           const float* d_seed_l, const float* d_seed_r, float* d_disp_l, float* d_disp_r, uint64_t tag,
                           void* ready_event) {
  if (!h) return PM_ERR_INVALID_ARG;
  const SubmitArgs a{d_left, d_right, rows, cols, 0, d_seed_l, d_seed_r, 0, d_disp_l, d_disp_r, 0, tag, true,
                     (hipEvent_t)ready_event};
  return submit_impl(h, a, "pm_submit_device_after");
}

int pm_flush(pm_handle* h) {
  if (!h) return PM_ERR_INVALID_ARG;
  if (int rc = refuse_while_capturing(h, "pm_flush")) return rc;
  const int held = held_slot(h);
  if (held < 0) return PM_OK;
  PM_HIP(h, hipSetDevice(h->device));
  return enqueue_frames(h, held, 1);
}

int pm_collect(pm_handle* h, float* disp_l, float* disp_r, size_t disp_step, uint64_t* tag) {
  if (!h) return PM_ERR_INVALID_ARG;
  if (int rc = refuse_while_capturing(h, "pm_collect")) return rc;
  if (h->pipe_count == 0) {
    set_err(h, "pm_collect: nothing in flight");
    return PM_ERR_BUSY;
  }
  const bool lr = h->params.left_right_check != 0;
  pm_handle::PipeSlot& sl = h->pipe[(size_t)h->pipe_head];
  const int rows = sl.rows, cols = sl.cols;
  const size_t frow = sizeof(float) * (size_t)cols;
  // where the maps go: the buffers bound at submit, else the ones given here (a device-resident frame has neither)
  float* out_l = sl.out_l ? sl.out_l : disp_l;
  float* out_r = sl.out_l ? sl.out_r : disp_r;
  size_t out_step = sl.out_l ? sl.out_step : (disp_step ? disp_step : frow);
  if (!sl.device_io) {
    if (!out_l || (lr && !out_r)) {
      set_err(h, "pm_collect: null output pointer (no maps were bound when the pair was submitted)");
      return PM_ERR_INVALID_ARG;
    }
    if (sl.out_l && ((disp_l && disp_l != sl.out_l) || (disp_r && disp_r != sl.out_r))) {
      set_err(h, "pm_collect: other maps than the ones bound when the pair was submitted");
      return PM_ERR_INVALID_ARG;
    }
    if (out_step < frow) {
      set_err(h, "pm_collect: disp_step is smaller than a row");
      return PM_ERR_INVALID_ARG;
    }
  }
  PM_HIP(h, hipSetDevice(h->device));
  if (sl.state == 1)  // still held for a partner that never came
    if (int rc = enqueue_frames(h, h->pipe_head, 1)) return rc;
  // a LATER frame that is being held starts as soon as the device has nothing else to do: before the wait if the
  // device went idle since it was submitted, after it if the frame collected here was what kept the device busy (the
  // loop submit(k + 1); collect(k) would otherwise leave the device idle until the next call)
  for (int pass = 0; pass < 2; ++pass) {
    const int held = held_slot(h);
    if (held >= 0 && !device_busy(h))
      if (int rc = enqueue_frames(h, held, 1)) return rc;
    if (pass == 0) PM_HIP(h, hipEventSynchronize(sl.out_done));
  }
  if (!sl.device_io) {
    const PinnedSlot ps = pinned_slot(h, h->pipe_head, (size_t)rows * cols);
    if (!sl.direct_l) h->copy_pool->Copy2D(out_l, out_step, ps.dl, frow, frow, rows);
    if (lr && !sl.direct_r) h->copy_pool->Copy2D(out_r, out_step, ps.dr, frow, frow, rows);
  }
  if (tag) *tag = sl.tag;
  sl.state = 0;
  h->pipe_head = (h->pipe_head + 1) % h->max_batch;
  --h->pipe_count;
  if (h->pipe_count == 0) h->seq_last = -1;
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
                             dilate_factor, ps.disp, -ps.pitch, h->stream));  // into the state plane (rows interleaved)
  return stage_out(h, ps, seed, 0);
}

int pm_corner_subpix(pm_handle* h, const uint8_t* image, int rows, int cols, float* xs, float* ys, int n) {
  if (!h) return PM_ERR_INVALID_ARG;
  if (int rc = refuse_while_capturing(h, "pm_corner_subpix")) return rc;
  if (!image || !xs || !ys || n < 0 || n > kSeedMaxFeatures) {
    set_err(h, "pm_corner_subpix: null pointer, or more than %d points", kSeedMaxFeatures);
    return PM_ERR_INVALID_ARG;
  }
  PlaneSet ps;
  if (int rc = stage_prep(h, image, nullptr, rows, cols, &ps)) return rc;
  SeedParams sp = seed_params(h->params);
  sp.subpixel_corners = 1;  // the stage always needs the masks and the neighbourhood buffer
  PM_HIP(h, seed_subpix_prepare(h->seeds[0], sp, h->stream));
  float* d_x = h->st_disp_l;
  float* d_y = h->st_disp_l + kSeedMaxFeatures;
  PM_HIP(h, hipMemcpyAsync(d_x, xs, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, h->stream));
  PM_HIP(h, hipMemcpyAsync(d_y, ys, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, h->stream));
  PM_HIP(h, seed_corner_subpix(h->seeds[0], sp, ps.img8, rows, cols, ps.pitch, d_x, d_y, n, h->stream));
  PM_HIP(h, hipMemcpyAsync(xs, d_x, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
  PM_HIP(h, hipMemcpyAsync(ys, d_y, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
  PM_HIP(h, hipStreamSynchronize(h->stream));
  return PM_OK;
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
