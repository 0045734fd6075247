// pm_tiled.hip -- pm_tiled_*: the row-tiled large image (BASELINE configs[3]) driven from ONE process over n band
// handles on up to n devices, through the public pm_tile_* stages.  Protocol and exactness argument: include/pm/patchmatch.h
// (pm_tiled_match_u8) and ocean-perception_amd/python/tiled.py, which runs the same steps over torch.distributed.
// Everything a band does is ordered on its handle's stream; what crosses bands is one row of disparities per vertical
// sweep and round, copied by hipMemcpyPeerAsync on the RECEIVER's stream behind an event of the sender's stream.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "pm/patchmatch.h"
#include "pm_internal.hpp"

namespace {

struct Band {
  pm_handle* h = nullptr;
  int dev = 0;
  hipStream_t stream = nullptr;
  pm_tile tile{};
  int band_rows = 0;
  uint8_t *d_left = nullptr, *d_right = nullptr;
  float *d_seed_l = nullptr, *d_seed_r = nullptr, *d_out_l = nullptr, *d_out_r = nullptr;
  float* sent[2] = {nullptr, nullptr};  // the boundary row this band hands to its successor (double-buffered)
  float *used = nullptr, *incoming = nullptr;
  int *mask = nullptr, *flag = nullptr;
  hipEvent_t ev_sent[2] = {nullptr, nullptr};  // sent[i] is written (this band's stream)
  hipEvent_t ev_read[2] = {nullptr, nullptr};  // the successor has copied sent[i] (recorded on ITS stream)
  bool read_pending[2] = {false, false};
  int last = 0;  // the buffer of `sent` that holds the row this band published last
  // kernels of this band may read the `sent` buffers of band k - 1 / k + 1 where they lie: same device, or peer access
  // enabled in both directions (pm_tiled_create); otherwise the row is copied over first (hipMemcpyPeerAsync)
  bool direct_prev = false, direct_next = false;
};

}  // namespace

struct pm_tiled_plan {
  std::vector<Band> bands;
  int rows = 0, cols = 0, n_views = 1, halo = 0;
  bool resident = false, have_seed_l = false, have_seed_r = false;  // a pair has been uploaded (pm_tiled_upload_u8)
  int device_boundaries = 0;  // neighbouring bands that live on DIFFERENT devices (their rows cross by peer copy)
  int peer_links = 0;         // ... of which direct peer access could be enabled (the others are staged by the runtime)
  char err[512] = {0};
};

namespace {

int fail(pm_tiled_plan* p, int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(p->err, sizeof(p->err), fmt, ap);
  va_end(ap);
  return code;
}
#define TL_HIP(p, call)                                                                              \
  do {                                                                                               \
    hipError_t e_ = (call);                                                                          \
    if (e_ != hipSuccess) return fail((p), PM_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)
#define TL_PM(p, b, call)                                                                            \
  do {                                                                                               \
    int rc_ = (call);                                                                                \
    if (rc_ != PM_OK) return fail((p), rc_, "%s: %s", #call, pm_last_error((b).h));                  \
  } while (0)

int halo_rows(const pm_params& p) {
  if (p.semantics != PM_SEM_CPU) return 2;
  int ph = p.bg_patch_h;
  for (int i = 0; i < p.patchmatch_iters; ++i) ph = p.patch_h[i] > ph ? p.patch_h[i] : ph;
  return ph / 2 + 1;
}
void band_of(int k, int n, int rows, int halo, pm_tile* t, int* band_rows) {
  const int base = rows / n, rem = rows % n;
  t->global_rows = rows;
  t->own_row0 = k * base + (k < rem ? k : rem);
  t->own_rows = base + (k < rem ? 1 : 0);
  t->band_row0 = t->own_row0 - halo > 0 ? t->own_row0 - halo : 0;
  const int end = t->own_row0 + t->own_rows + halo < rows ? t->own_row0 + t->own_rows + halo : rows;
  *band_rows = end - t->band_row0;
}

// the band images and seed maps of a pair -> the bands' devices (stream-ordered, on every band's stream)
int upload(pm_tiled_plan* p, const uint8_t* left, const uint8_t* right, size_t image_step, const float* seed_l,
           const float* seed_r, size_t seed_step) {
  const int cols = p->cols;
  for (Band& b : p->bands) {
    TL_HIP(p, hipSetDevice(b.dev));
    const size_t w = (size_t)cols;
    TL_HIP(p, hipMemcpy2DAsync(b.d_left, w, left + (size_t)b.tile.band_row0 * image_step, image_step, w, (size_t)b.band_rows,
                               hipMemcpyHostToDevice, b.stream));
    TL_HIP(p, hipMemcpy2DAsync(b.d_right, w, right + (size_t)b.tile.band_row0 * image_step, image_step, w,
                               (size_t)b.band_rows, hipMemcpyHostToDevice, b.stream));
    if (seed_l)
      TL_HIP(p, hipMemcpy2DAsync(b.d_seed_l, w * 4, (const char*)seed_l + (size_t)b.tile.band_row0 * seed_step, seed_step,
                                 w * 4, (size_t)b.band_rows, hipMemcpyHostToDevice, b.stream));
    if (seed_r)
      TL_HIP(p, hipMemcpy2DAsync(b.d_seed_r, w * 4, (const char*)seed_r + (size_t)b.tile.band_row0 * seed_step, seed_step,
                                 w * 4, (size_t)b.band_rows, hipMemcpyHostToDevice, b.stream));
  }
  p->have_seed_l = seed_l != nullptr;
  p->have_seed_r = seed_r != nullptr;
  p->resident = true;
  return PM_OK;
}

// the owned rows of every band -> the caller's maps; waits for all bands
int download(pm_tiled_plan* p, float* disp_l, float* disp_r, size_t disp_step) {
  const int cols = p->cols, nv = p->n_views;
  for (Band& b : p->bands) {
    TL_HIP(p, hipSetDevice(b.dev));
    const size_t w4 = (size_t)cols * 4;
    TL_HIP(p, hipMemcpy2DAsync((char*)disp_l + (size_t)b.tile.own_row0 * disp_step, disp_step, b.d_out_l, w4, w4,
                               (size_t)b.tile.own_rows, hipMemcpyDeviceToHost, b.stream));
    if (nv > 1 && disp_r)
      TL_HIP(p, hipMemcpy2DAsync((char*)disp_r + (size_t)b.tile.own_row0 * disp_step, disp_step, b.d_out_r, w4, w4,
                                 (size_t)b.tile.own_rows, hipMemcpyDeviceToHost, b.stream));
  }
  for (Band& b : p->bands) {
    TL_HIP(p, hipSetDevice(b.dev));
    TL_HIP(p, hipStreamSynchronize(b.stream));
  }
  return PM_OK;
}

// one attempt on the resident pair with `rounds` exchange rounds per vertical sweep, results into the bands' output
// buffers; *moved = some band's boundary row still changed.  Waits for the flags (the one host read per attempt).
int attempt(pm_tiled_plan* p, int rounds, bool* moved, int* exchanges) {
  const int n = (int)p->bands.size(), cols = p->cols, nv = p->n_views;
  const pm_params& prm = pm_internal::params(p->bands[0].h);
  const int row_n = nv * cols;
  const size_t row_bytes = sizeof(float) * (size_t)row_n;
  for (Band& b : p->bands) {
    TL_HIP(p, hipSetDevice(b.dev));
    TL_HIP(p, hipMemsetAsync(b.flag, 0, sizeof(int), b.stream));
    TL_PM(p, b, pm_tile_begin(b.h, &b.tile, b.d_left, b.d_right, b.band_rows, cols, p->have_seed_l ? b.d_seed_l : nullptr,
                              p->have_seed_r ? b.d_seed_r : nullptr));
  }
  int cur = 0;
  // band k's boundary row -> sent[cur] (behind the successor's last read of that buffer)
  auto publish = [&](int k, int out_row) -> int {
    Band& b = p->bands[(size_t)k];
    TL_HIP(p, hipSetDevice(b.dev));
    if (b.read_pending[cur]) {
      TL_HIP(p, hipStreamWaitEvent(b.stream, b.ev_read[cur], 0));
      b.read_pending[cur] = false;
    }
    TL_PM(p, b, pm_tile_get_row(b.h, out_row, b.sent[cur]));
    TL_HIP(p, hipEventRecord(b.ev_sent[cur], b.stream));
    b.last = cur;
    return PM_OK;
  };
  // band k copies its predecessor's published row into dst (on k's stream, behind the predecessor's event)
  auto fetch = [&](int k, int pred, float* dst) -> int {
    Band& b = p->bands[(size_t)k];
    Band& s = p->bands[(size_t)pred];
    TL_HIP(p, hipSetDevice(b.dev));
    TL_HIP(p, hipStreamWaitEvent(b.stream, s.ev_sent[cur], 0));
    TL_HIP(p, hipMemcpyPeerAsync(dst, b.dev, s.sent[cur], s.dev, row_bytes, b.stream));
    TL_HIP(p, hipEventRecord(s.ev_read[cur], b.stream));
    s.read_pending[cur] = true;
    ++*exchanges;
    return PM_OK;
  };
  for (int it = 0; it < prm.patchmatch_iters; ++it) {
    for (Band& b : p->bands) TL_PM(p, b, pm_tile_noise(b.h, it));
    for (int k = 0; k < 4; ++k) {
      if (k == 0 || k == 2) {  // horizontal sweeps never leave the band
        for (Band& b : p->bands) TL_PM(p, b, pm_tile_sweep(b.h, it, k));
        continue;
      }
      const bool down = k == 1;
      auto out_row = [&](const Band& b) { return down ? b.tile.own_row0 + b.tile.own_rows - 1 : b.tile.own_row0; };
      auto pred_row = [&](const Band& b) { return down ? b.tile.own_row0 - 1 : b.tile.own_row0 + b.tile.own_rows; };
      auto pred_of = [&](int j) { return down ? j - 1 : j + 1; };
      // the neighbour's value BEFORE the sweep: the guess
      for (int j = 0; j < n; ++j)
        if (int rc = publish(j, out_row(p->bands[(size_t)j]))) return rc;
      for (int j = 0; j < n; ++j) {
        Band& b = p->bands[(size_t)j];
        const int pr = pred_of(j);
        const bool has_pred = pr >= 0 && pr < n;
        if (has_pred)
          if (int rc = fetch(j, pr, b.used)) return rc;
        // the guess into the planes and the snapshot in one launch (round 4: a row store and two runtime copies)
        TL_PM(p, b, pm_tile_presweep(b.h, pred_row(b), has_pred ? b.used : nullptr));
        TL_PM(p, b, pm_tile_sweep(b.h, it, k));
      }
      // Round r hands band j the row its predecessor held after round r - 1.  The first band of the sweep direction has
      // no predecessor: its row is final after the first sweep; its successor is final after round 0, and so on -- the
      // band at position `pos` (0 = first in sweep direction) is final after round pos - 1.  From round pos on it
      // would receive the row it already has, restore nothing and re-sweep nothing: it skips those rounds (and its
      // predecessor does not publish for it).  n (n - 1) / 2 band-rounds instead of (n - 1)^2, same result.
      auto pos_of = [&](int j) { return down ? j : n - 1 - j; };
      for (int r = 0; r < rounds; ++r) {
        cur ^= 1;
        for (int j = 0; j < n; ++j) {
          const int succ = down ? j + 1 : j - 1;
          if (succ < 0 || succ >= n || pos_of(succ) <= r) continue;  // nobody reads this band's row in this round
          if (int rc = publish(j, out_row(p->bands[(size_t)j]))) return rc;
        }
        for (int j = 0; j < n; ++j) {
          Band& b = p->bands[(size_t)j];
          const int pr = pred_of(j);
          if (pr < 0 || pr >= n || pos_of(j) <= r) continue;  // no predecessor, or final since round pos - 1
          // compare + restore + row store in one launch, then the masked sweep (round 4: a copy and four launches);
          // the incoming row is read where the predecessor published it when this band's kernels can reach it
          Band& s = p->bands[(size_t)pr];
          if (pr < j ? b.direct_prev : b.direct_next) {
            TL_HIP(p, hipSetDevice(b.dev));
            TL_HIP(p, hipStreamWaitEvent(b.stream, s.ev_sent[cur], 0));
            TL_PM(p, b, pm_tile_exchange_round(b.h, it, k, pred_row(b), s.sent[cur], b.used, b.incoming, b.mask));
            TL_HIP(p, hipEventRecord(s.ev_read[cur], b.stream));
            s.read_pending[cur] = true;
            ++*exchanges;
          } else {
            if (int rc = fetch(j, pr, b.incoming)) return rc;
            TL_PM(p, b, pm_tile_exchange_round(b.h, it, k, pred_row(b), b.incoming, b.used, b.incoming, b.mask));
          }
          float* t = b.used;
          b.used = b.incoming;
          b.incoming = t;
        }
      }
      // did my boundary row move after the last row I sent?  then my successor is stale (only bands that HAVE one ask)
      for (int j = 0; j < n; ++j) {
        Band& b = p->bands[(size_t)j];
        const int succ = down ? j + 1 : j - 1;
        if (succ < 0 || succ >= n) continue;
        TL_HIP(p, hipSetDevice(b.dev));
        TL_PM(p, b, pm_tile_row_moved(b.h, out_row(b), b.sent[b.last], b.flag));
      }
    }
  }
  for (Band& b : p->bands) {
    TL_PM(p, b, pm_tile_background(b.h));
    TL_PM(p, b, pm_tile_finish(b.h, b.d_out_l, nv > 1 ? b.d_out_r : nullptr));
  }
  *moved = false;
  for (Band& b : p->bands) {
    TL_HIP(p, hipSetDevice(b.dev));
    int f = 0;
    TL_HIP(p, hipMemcpyAsync(&f, b.flag, sizeof(int), hipMemcpyDeviceToHost, b.stream));
    TL_HIP(p, hipStreamSynchronize(b.stream));
    *moved = *moved || f != 0;
    b.read_pending[0] = b.read_pending[1] = false;  // every stream is idle: nothing is pending any more
  }
  return PM_OK;
}

}  // namespace

extern "C" {

int pm_tiled_band_rows(const pm_params* params, int global_rows, int n_bands) {
  if (!params || n_bands < 1 || global_rows < n_bands) return PM_ERR_INVALID_ARG;
  const int halo = halo_rows(*params);
  int worst = 0;
  for (int k = 0; k < n_bands; ++k) {
    pm_tile t;
    int br;
    band_of(k, n_bands, global_rows, halo, &t, &br);
    worst = br > worst ? br : worst;
  }
  return worst;
}

const char* pm_tiled_last_error(const pm_tiled_plan* plan) { return plan ? plan->err : "null plan"; }

int pm_tiled_topology(const pm_tiled_plan* plan, int* device_boundaries, int* peer_links) {
  if (!plan) return PM_ERR_INVALID_ARG;
  if (device_boundaries) *device_boundaries = plan->device_boundaries;
  if (peer_links) *peer_links = plan->peer_links;
  return PM_OK;
}

void pm_tiled_destroy(pm_tiled_plan* plan) {
  if (!plan) return;
  for (Band& b : plan->bands) {
    (void)hipSetDevice(b.dev);
    if (b.stream) (void)hipStreamSynchronize(b.stream);
    void* bufs[] = {b.d_left, b.d_right, b.d_seed_l, b.d_seed_r, b.d_out_l, b.d_out_r, b.sent[0], b.sent[1],
                    b.used,   b.incoming, b.mask,    b.flag};
    for (void* q : bufs)
      if (q) (void)hipFree(q);
    for (int i = 0; i < 2; ++i) {
      if (b.ev_sent[i]) (void)hipEventDestroy(b.ev_sent[i]);
      if (b.ev_read[i]) (void)hipEventDestroy(b.ev_read[i]);
    }
  }
  delete plan;
}

int pm_tiled_create(pm_handle* const* bands, int n_bands, int rows, int cols, pm_tiled_plan** out) {
  if (!out) return PM_ERR_INVALID_ARG;
  *out = nullptr;
  if (!bands || n_bands < 1 || rows < n_bands || cols < 8) return PM_ERR_INVALID_ARG;
  pm_tiled_plan* p = new pm_tiled_plan;
  *out = p;  // also on failure: the plan then carries the message and is good for pm_tiled_destroy only
  for (int k = 0; k < n_bands; ++k)
    if (!bands[k]) return fail(p, PM_ERR_INVALID_ARG, "band %d is a null handle", k);
  const pm_params& prm = pm_internal::params(bands[0]);
  if (prm.mode != PM_MODE_SCALAR) return fail(p, PM_ERR_INVALID_ARG, "the row-tiled driver runs the scalar mode only");
  for (int k = 1; k < n_bands; ++k)
    if (std::memcmp(&pm_internal::params(bands[k]), &prm, sizeof(pm_params)) != 0)
      return fail(p, PM_ERR_INVALID_ARG, "band %d was created with other parameters than band 0", k);
  p->rows = rows;
  p->cols = cols;
  p->n_views = prm.left_right_check ? 2 : 1;
  p->halo = halo_rows(prm);
  p->bands.resize((size_t)n_bands);
  for (int k = 0; k < n_bands; ++k) {
    Band& b = p->bands[(size_t)k];
    b.h = bands[k];
    b.dev = pm_internal::device(b.h);
    b.stream = pm_internal::stream(b.h);
    band_of(k, n_bands, rows, p->halo, &b.tile, &b.band_rows);
    int mr, mc;
    pm_internal::plan_size(b.h, &mr, &mc);
    if (b.band_rows > mr || cols > mc)
      return fail(p, PM_ERR_SIZE, "band %d needs a plan of %d x %d, its handle has %d x %d (pm_tiled_band_rows)", k, cols,
                  b.band_rows, mc, mr);
    TL_HIP(p, hipSetDevice(b.dev));
    const size_t px = (size_t)b.band_rows * cols, own = (size_t)b.tile.own_rows * cols;
    const size_t row_bytes = sizeof(float) * (size_t)p->n_views * cols;
    TL_HIP(p, hipMalloc((void**)&b.d_left, px));
    TL_HIP(p, hipMalloc((void**)&b.d_right, px));
    TL_HIP(p, hipMalloc((void**)&b.d_seed_l, px * 4));
    TL_HIP(p, hipMalloc((void**)&b.d_seed_r, px * 4));
    TL_HIP(p, hipMalloc((void**)&b.d_out_l, own * 4));
    TL_HIP(p, hipMalloc((void**)&b.d_out_r, own * 4));
    for (int i = 0; i < 2; ++i) {
      TL_HIP(p, hipMalloc((void**)&b.sent[i], row_bytes));
      TL_HIP(p, hipEventCreateWithFlags(&b.ev_sent[i], hipEventDisableTiming));
      TL_HIP(p, hipEventCreateWithFlags(&b.ev_read[i], hipEventDisableTiming));
    }
    TL_HIP(p, hipMalloc((void**)&b.used, row_bytes));
    TL_HIP(p, hipMalloc((void**)&b.incoming, row_bytes));
    TL_HIP(p, hipMalloc((void**)&b.mask, sizeof(int) * (size_t)p->n_views * cols));
    TL_HIP(p, hipMalloc((void**)&b.flag, sizeof(int)));
  }
  // neighbouring bands on different devices: direct peer copies over xGMI where the platform allows them
  for (int k = 0; k + 1 < n_bands; ++k) {
    const int a = p->bands[(size_t)k].dev, c = p->bands[(size_t)k + 1].dev;
    if (a == c) {  // bands sharing a device: hipMemcpyPeerAsync is then a plain device copy, nothing to enable
      p->bands[(size_t)k].direct_next = p->bands[(size_t)k + 1].direct_prev = true;
      continue;
    }
    ++p->device_boundaries;
    int ok = 0;
    if (hipDeviceCanAccessPeer(&ok, a, c) == hipSuccess && ok) {
      (void)hipSetDevice(a);
      const hipError_t e1 = hipDeviceEnablePeerAccess(c, 0);
      (void)hipSetDevice(c);
      const hipError_t e2 = hipDeviceEnablePeerAccess(a, 0);
      const auto fine = [](hipError_t e) { return e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled; };
      if (fine(e1) && fine(e2)) {
        ++p->peer_links;
        p->bands[(size_t)k].direct_next = p->bands[(size_t)k + 1].direct_prev = true;
      }
    }
    (void)hipGetLastError();
  }
  return PM_OK;
}

int pm_tiled_upload_u8(pm_tiled_plan* plan, const uint8_t* left, const uint8_t* right, size_t image_step,
                       const float* seed_l, const float* seed_r, size_t seed_step) {
  if (!plan) return PM_ERR_INVALID_ARG;
  if (!left || !right) return fail(plan, PM_ERR_INVALID_ARG, "pm_tiled_upload_u8: null image pointer");
  if (image_step == 0) image_step = (size_t)plan->cols;
  if (seed_step == 0) seed_step = (size_t)plan->cols * 4;
  if (image_step < (size_t)plan->cols || seed_step < (size_t)plan->cols * 4)
    return fail(plan, PM_ERR_INVALID_ARG, "pm_tiled_upload_u8: a row step is smaller than a row");
  return upload(plan, left, right, image_step, seed_l, seed_r, seed_step);
}

int pm_tiled_run(pm_tiled_plan* plan, int rounds, pm_tiled_info* info) {
  if (!plan) return PM_ERR_INVALID_ARG;
  if (!plan->resident) return fail(plan, PM_ERR_INVALID_ARG, "pm_tiled_run: no pair uploaded (pm_tiled_upload_u8)");
  const int n = (int)plan->bands.size();
  if (rounds < 0 || rounds > n - 1) rounds = n - 1;  // negative: the exact count, no repeat possible
  bool moved = false;
  int exchanges = 0;
  if (int rc = attempt(plan, rounds, &moved, &exchanges)) return rc;
  int repeated = 0;
  if (moved && rounds < n - 1) {  // band k is final after round k + 1: n - 1 rounds are always enough
    rounds = n - 1;
    repeated = 1;
    if (int rc = attempt(plan, rounds, &moved, &exchanges)) return rc;
  }
  if (info) {
    info->rounds_used = rounds;
    info->repeated = repeated;
    info->exchanges = exchanges;
  }
  return PM_OK;
}

int pm_tiled_download(pm_tiled_plan* plan, float* disp_l, float* disp_r, size_t disp_step) {
  if (!plan) return PM_ERR_INVALID_ARG;
  if (!disp_l) return fail(plan, PM_ERR_INVALID_ARG, "pm_tiled_download: null output pointer");
  if (disp_step == 0) disp_step = (size_t)plan->cols * 4;
  if (disp_step < (size_t)plan->cols * 4) return fail(plan, PM_ERR_INVALID_ARG, "pm_tiled_download: a row step is smaller than a row");
  return download(plan, disp_l, disp_r, disp_step);
}

int pm_tiled_match_u8(pm_tiled_plan* plan, const uint8_t* left, const uint8_t* right, size_t image_step,
                      const float* seed_l, const float* seed_r, size_t seed_step, float* disp_l, float* disp_r,
                      size_t disp_step, int rounds, pm_tiled_info* info) {
  if (int rc = pm_tiled_upload_u8(plan, left, right, image_step, seed_l, seed_r, seed_step)) return rc;
  if (int rc = pm_tiled_run(plan, rounds, info)) return rc;
  return pm_tiled_download(plan, disp_l, disp_r, disp_step);
}

}  // extern "C"
