// pm_tiled.hip -- pm_tiled_*: the row-tiled large image (BASELINE configs[3]) driven from ONE process over n band
// handles on up to n devices, through the public pm_tile_* stages.  Protocol and exactness argument: include/pm/patchmatch.h
// (pm_tiled_match_u8) and ocean-perception_amd/python/tiled.py, which runs the same steps over torch.distributed.
// Everything a band does is ordered on its handle's stream; what crosses bands is one row of disparities per vertical
// sweep and round, read by the RECEIVER (a peer copy on its stream, or its kernel reading the row in place) behind an
// event of the sender's stream.
//
// Device discipline (round 6).  With one band per GPU every runtime object belongs to ONE device, and the rules are:
//   * an event is created on its owner's device and recorded ONLY on a stream of that device (hipEventRecord with an
//     event and a stream of different devices is an error); waiting for an event of another device is legal;
//   * a stream is used -- copies, memsets, launches, synchronisation -- with its device current;
//   * an allocation is made with its band's device current;
//   * a kernel of a band reads memory of another device only where that is the declared intent (`direct` exchange).
// So the SENDER of a row owns "the row is written" (ev_sent, recorded on its stream) and the READER owns "the row is
// consumed" (ev_done, recorded on the reader's stream; the sender waits for it before it overwrites the buffer).  A band
// has a different reader per sweep direction, hence two pairs of ev_done.  Every HIP call of this file goes through the
// rt_* functions below, which -- when the plan was made by pm_tiled_create_logical (include/pm/testing.h) -- keep a log
// of (call, current device, device of the stream / event / pointers) in LOGICAL device ids, so that the discipline can
// be proven on a box with ONE GPU: bands get distinct logical devices, all mapped onto the physical device of their
// handle.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <initializer_list>
#include <unordered_map>
#include <vector>

#include "pm/patchmatch.h"
#include "pm/testing.h"
#include "pm_internal.hpp"

namespace {

struct Band {
  pm_handle* h = nullptr;
  int index = 0;
  int dev = 0;   // the physical device of the handle
  int ldev = 0;  // the device the band is accounted to: = dev, or a logical id given to pm_tiled_create_logical
  hipStream_t stream = nullptr;
  pm_tile tile{};
  int band_rows = 0;
  uint8_t *d_left = nullptr, *d_right = nullptr;
  float *d_seed_l = nullptr, *d_seed_r = nullptr, *d_out_l = nullptr, *d_out_r = nullptr;
  float* sent[2] = {nullptr, nullptr};  // the boundary row this band hands to its successor (double-buffered)
  float *used = nullptr, *incoming = nullptr;
  int *mask = nullptr, *flag = nullptr;
  hipEvent_t ev_sent[2] = {nullptr, nullptr};  // sent[i] is written (this band's event, this band's stream)
  // this band, as the READER of its predecessor's sent[i], has consumed it: [sweep direction: 0 down, 1 up][i].
  // Created on THIS band's device and recorded on THIS band's stream; the predecessor waits for it.
  hipEvent_t ev_done[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
  hipEvent_t unread[2] = {nullptr, nullptr};  // the reader's ev_done that guards sent[i], or null: nobody is reading it
  hipEvent_t ev_probe = nullptr;  // logical-device plans only: an event nobody waits for (pm_tiled_debug_inject)
  int last = 0;  // the buffer of `sent` that holds the row this band published last
  // peer access is enabled in both directions between this band's device and its neighbour's (pm_tiled_create)
  bool peer_prev = false, peer_next = false;
};

struct AuditRange {
  const char* base;
  size_t bytes;
  int ldev;
};
struct Audit {
  int cur = -1;  // the logical device that is current
  std::vector<pm_tiled_audit_record> log;
  std::vector<AuditRange> mem;
  std::unordered_map<const void*, int> obj;  // events and streams -> logical device
  int violations = 0;
  int simulate_peer = 0;
  int inject = 0;  // pm_tiled_debug_inject: deliberate breaches of the discipline, to show that the log catches them
};

}  // namespace

struct pm_tiled_plan {
  std::vector<Band> bands;
  int rows = 0, cols = 0, n_views = 1, halo = 0;
  bool resident = false, have_seed_l = false, have_seed_r = false;  // a pair has been uploaded (pm_tiled_upload_u8)
  int device_boundaries = 0;  // neighbouring bands that live on DIFFERENT devices (their rows cross by peer copy)
  int peer_links = 0;         // ... of which direct peer access could be enabled (the others are staged by the runtime)
  int exchange = PM_TILED_EXCHANGE_AUTO;
  int schedule = PM_TILED_SCHEDULE_PIPELINED;
  Audit* audit = nullptr;  // only plans of pm_tiled_create_logical keep a log
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

// ---- the runtime layer: the ONLY place of this file that calls HIP; logs in logical device ids when asked to ---------

int au_mem(const Audit& a, const void* q) {  // the logical device an address belongs to; -1 = not one of the plan's
  for (const AuditRange& r : a.mem)
    if ((const char*)q >= r.base && (const char*)q < r.base + r.bytes) return r.ldev;
  return -1;
}
int au_obj(const Audit& a, const void* o) {
  auto it = a.obj.find(o);
  return it == a.obj.end() ? -1 : it->second;
}
// one record; `stream`, `object`, `source`: logical devices or -1 (the call has no such argument).  Violations:
//   1 the stream argument is not a stream of the current device      2 an event is recorded on a stream of another device
//   4 the object (allocation, event being created, destination) does not belong to the current device
//   8 the source lies on another device and the call is not one that may read there
//   16 an argument is not known to the plan at all
void au_log(pm_tiled_plan* p, int call, int band, int detail, bool has_stream, int stream, bool has_object, int object,
            bool has_source, int source, bool foreign_ok) {
  Audit& a = *p->audit;
  pm_tiled_audit_record r{};
  r.call = call;
  r.band = band;
  r.detail = detail;
  r.current_device = a.cur;
  r.stream_device = has_stream ? stream : -1;
  r.object_device = has_object ? object : -1;
  r.source_device = has_source ? source : -1;
  r.foreign_allowed = foreign_ok ? 1 : 0;
  int v = 0;
  if (has_stream && stream < 0) v |= 16;
  if (has_object && object < 0) v |= 16;
  if (has_source && source < 0) v |= 16;
  if (has_stream && stream >= 0 && stream != a.cur) v |= 1;
  if (call == PM_TILED_CALL_EVENT_RECORD && has_stream && has_object && stream != object) v |= 2;
  if (call != PM_TILED_CALL_STREAM_WAIT_EVENT && has_object && object >= 0 && object != a.cur) v |= 4;
  if (has_source && source >= 0 && source != a.cur && !foreign_ok) v |= 8;
  r.violation = v;
  if (v) ++a.violations;
  a.log.push_back(r);
}

// binds band b's device (physical) and accounts for it (logical)
hipError_t rt_use(pm_tiled_plan* p, const Band& b) {
  const hipError_t e = hipSetDevice(b.dev);
  if (p->audit) {
    p->audit->cur = b.ldev;
    au_log(p, PM_TILED_CALL_SET_DEVICE, b.index, 0, false, -1, true, b.ldev, false, -1, false);
  }
  return e;
}
hipError_t rt_malloc(pm_tiled_plan* p, const Band& b, void** out, size_t bytes) {
  const hipError_t e = hipMalloc(out, bytes);
  if (p->audit) {
    if (e == hipSuccess) p->audit->mem.push_back({(const char*)*out, bytes, p->audit->cur});
    au_log(p, PM_TILED_CALL_MALLOC, b.index, 0, false, -1, true, b.ldev, false, -1, false);
  }
  return e;
}
hipError_t rt_event_create(pm_tiled_plan* p, const Band& b, hipEvent_t* ev) {
  const hipError_t e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
  if (p->audit) {
    if (e == hipSuccess) p->audit->obj[(const void*)*ev] = p->audit->cur;  // an event belongs to the device current at its creation
    au_log(p, PM_TILED_CALL_EVENT_CREATE, b.index, 0, false, -1, true, b.ldev, false, -1, false);
  }
  return e;
}
hipError_t rt_event_record(pm_tiled_plan* p, const Band& b, hipEvent_t ev) {
  if (p->audit)
    au_log(p, PM_TILED_CALL_EVENT_RECORD, b.index, 0, true, au_obj(*p->audit, b.stream), true, au_obj(*p->audit, ev), false,
           -1, false);
  return hipEventRecord(ev, b.stream);
}
hipError_t rt_wait_event(pm_tiled_plan* p, const Band& b, hipEvent_t ev) {  // the event may belong to any device
  if (p->audit)
    au_log(p, PM_TILED_CALL_STREAM_WAIT_EVENT, b.index, 0, true, au_obj(*p->audit, b.stream), true, au_obj(*p->audit, ev),
           false, -1, false);
  return hipStreamWaitEvent(b.stream, ev, 0);
}
hipError_t rt_sync(pm_tiled_plan* p, const Band& b) {
  if (p->audit)
    au_log(p, PM_TILED_CALL_STREAM_SYNC, b.index, 0, true, au_obj(*p->audit, b.stream), false, -1, false, -1, false);
  return hipStreamSynchronize(b.stream);
}
hipError_t rt_memset(pm_tiled_plan* p, const Band& b, void* dst, int value, size_t bytes) {
  if (p->audit)
    au_log(p, PM_TILED_CALL_MEMSET, b.index, 0, true, au_obj(*p->audit, b.stream), true, au_mem(*p->audit, dst), false, -1,
           false);
  return hipMemsetAsync(dst, value, bytes, b.stream);
}
hipError_t rt_h2d_2d(pm_tiled_plan* p, const Band& b, void* dst, size_t dpitch, const void* src, size_t spitch, size_t w,
                     size_t h) {
  if (p->audit)
    au_log(p, PM_TILED_CALL_COPY_H2D, b.index, 0, true, au_obj(*p->audit, b.stream), true, au_mem(*p->audit, dst), false, -1,
           false);
  return hipMemcpy2DAsync(dst, dpitch, src, spitch, w, h, hipMemcpyHostToDevice, b.stream);
}
hipError_t rt_d2h_2d(pm_tiled_plan* p, const Band& b, void* dst, size_t dpitch, const void* src, size_t spitch, size_t w,
                     size_t h) {
  if (p->audit)
    au_log(p, PM_TILED_CALL_COPY_D2H, b.index, 0, true, au_obj(*p->audit, b.stream), false, -1, true,
           au_mem(*p->audit, src), false);
  return hipMemcpy2DAsync(dst, dpitch, src, spitch, w, h, hipMemcpyDeviceToHost, b.stream);
}
hipError_t rt_d2h(pm_tiled_plan* p, const Band& b, void* dst, const void* src, size_t bytes) {
  if (p->audit)
    au_log(p, PM_TILED_CALL_COPY_D2H, b.index, 0, true, au_obj(*p->audit, b.stream), false, -1, true,
           au_mem(*p->audit, src), false);
  return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, b.stream);
}
// a row of band s -> memory of band b, on b's stream: the one call that names both devices
hipError_t rt_copy_peer(pm_tiled_plan* p, const Band& b, void* dst, const Band& s, const void* src, size_t bytes) {
  if (p->audit)
    au_log(p, PM_TILED_CALL_COPY_PEER, b.index, s.index, true, au_obj(*p->audit, b.stream), true, au_mem(*p->audit, dst),
           true, au_mem(*p->audit, src), true);
  return hipMemcpyPeerAsync(dst, b.dev, src, s.dev, bytes, b.stream);
}
// A pm_tile_* stage of band b: the entry point binds b's device itself (pm_tile.hip::tile_check) and launches on b's
// stream; what is checked here is where its pointer arguments live.
struct StageArg {
  const void* ptr;
  bool foreign_ok;  // the kernel may read this address in another device's memory (direct exchange)
};
void rt_stage(pm_tiled_plan* p, const Band& b, int stage, std::initializer_list<StageArg> args) {
  if (!p->audit) return;
  Audit& a = *p->audit;
  a.cur = b.ldev;
  int i = 0;
  au_log(p, PM_TILED_CALL_STAGE, b.index, stage * 16, true, au_obj(a, b.stream), false, -1, false, -1, false);
  for (const StageArg& g : args) {
    ++i;
    if (!g.ptr) continue;
    au_log(p, PM_TILED_CALL_STAGE_ARG, b.index, stage * 16 + i, false, -1, false, -1, true, au_mem(a, g.ptr), g.foreign_ok);
  }
}
#define TL_STAGE(p, b, stage, call, ...)                                                             \
  do {                                                                                               \
    rt_stage((p), (b), (stage), {__VA_ARGS__});                                                      \
    int rc_ = (call);                                                                                \
    if (rc_ != PM_OK) return fail((p), rc_, "%s: %s", #call, pm_last_error((b).h));                  \
  } while (0)
enum {
  ST_BEGIN = 1,
  ST_NOISE,
  ST_SWEEP,
  ST_GET_ROW,
  ST_PRESWEEP,
  ST_EXCHANGE_ROUND,
  ST_ROW_MOVED,
  ST_BACKGROUND,
  ST_FINISH,
  ST_SET_ROW
};

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

// may band b's kernels read the published row of its neighbour s where it lies?
bool direct(const pm_tiled_plan* p, const Band& b, const Band& s) {
  if (p->exchange == PM_TILED_EXCHANGE_COPY) return false;
  if (b.ldev == s.ldev) return true;  // one device: the neighbour's buffer is ordinary device memory
  if (p->exchange != PM_TILED_EXCHANGE_DIRECT) return false;
  return s.index < b.index ? b.peer_prev : b.peer_next;
}

// the band images and seed maps of a pair -> the bands' devices (stream-ordered, on every band's stream)
int upload(pm_tiled_plan* p, const uint8_t* left, const uint8_t* right, size_t image_step, const float* seed_l,
           const float* seed_r, size_t seed_step) {
  const int cols = p->cols;
  for (Band& b : p->bands) {
    TL_HIP(p, rt_use(p, b));
    const size_t w = (size_t)cols;
    TL_HIP(p, rt_h2d_2d(p, b, b.d_left, w, left + (size_t)b.tile.band_row0 * image_step, image_step, w, (size_t)b.band_rows));
    TL_HIP(p, rt_h2d_2d(p, b, b.d_right, w, right + (size_t)b.tile.band_row0 * image_step, image_step, w, (size_t)b.band_rows));
    if (seed_l)
      TL_HIP(p, rt_h2d_2d(p, b, b.d_seed_l, w * 4, (const char*)seed_l + (size_t)b.tile.band_row0 * seed_step, seed_step,
                          w * 4, (size_t)b.band_rows));
    if (seed_r)
      TL_HIP(p, rt_h2d_2d(p, b, b.d_seed_r, w * 4, (const char*)seed_r + (size_t)b.tile.band_row0 * seed_step, seed_step,
                          w * 4, (size_t)b.band_rows));
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
    TL_HIP(p, rt_use(p, b));
    const size_t w4 = (size_t)cols * 4;
    TL_HIP(p, rt_d2h_2d(p, b, (char*)disp_l + (size_t)b.tile.own_row0 * disp_step, disp_step, b.d_out_l, w4, w4,
                        (size_t)b.tile.own_rows));
    if (nv > 1 && disp_r)
      TL_HIP(p, rt_d2h_2d(p, b, (char*)disp_r + (size_t)b.tile.own_row0 * disp_step, disp_step, b.d_out_r, w4, w4,
                          (size_t)b.tile.own_rows));
  }
  for (Band& b : p->bands) {
    TL_HIP(p, rt_use(p, b));
    TL_HIP(p, rt_sync(p, b));
  }
  return PM_OK;
}

// Test hook (pm_tiled_debug_inject), called where a reader marks a row as consumed: what round 5's driver did -- an event of
// the PUBLISHER's device recorded on the reader's stream -- and a stream used while another band's device is current.
// Both are harmless on the one physical device a logical plan runs on (nobody waits for ev_probe), and both must show up
// in the log.
int inject_breach(pm_tiled_plan* p, Band& b, Band& s) {
  if (!p->audit || !p->audit->inject) return PM_OK;
  if (p->audit->inject & 1) TL_HIP(p, rt_event_record(p, b, s.ev_probe));
  if (p->audit->inject & 2) {
    TL_HIP(p, rt_use(p, s));
    TL_HIP(p, rt_wait_event(p, b, s.ev_probe));
    TL_HIP(p, rt_use(p, b));
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
    TL_HIP(p, rt_use(p, b));
    TL_HIP(p, rt_memset(p, b, b.flag, 0, sizeof(int)));
    TL_STAGE(p, b, ST_BEGIN,
             pm_tile_begin(b.h, &b.tile, b.d_left, b.d_right, b.band_rows, cols, p->have_seed_l ? b.d_seed_l : nullptr,
                           p->have_seed_r ? b.d_seed_r : nullptr),
             {b.d_left, false}, {b.d_right, false}, {p->have_seed_l ? b.d_seed_l : nullptr, false},
             {p->have_seed_r ? b.d_seed_r : nullptr, false});
  }
  int cur = 0;
  int dir = 0;  // 0: the sweep runs down (a band reads from the band above it), 1: up
  // band k's boundary row -> sent[cur] (behind its reader's last use of that buffer)
  auto publish = [&](int k, int out_row) -> int {
    Band& b = p->bands[(size_t)k];
    TL_HIP(p, rt_use(p, b));
    if (b.unread[cur]) {
      TL_HIP(p, rt_wait_event(p, b, b.unread[cur]));  // the reader's event, possibly of another device: waiting is legal
      b.unread[cur] = nullptr;
    }
    TL_STAGE(p, b, ST_GET_ROW, pm_tile_get_row(b.h, out_row, b.sent[cur]), {b.sent[cur], false});
    TL_HIP(p, rt_event_record(p, b, b.ev_sent[cur]));
    b.last = cur;
    return PM_OK;
  };
  // band k, with its device current, marks its predecessor's sent[cur] as consumed: ITS event on ITS stream
  auto consumed = [&](Band& b, Band& s) -> int {
    if (int rc = inject_breach(p, b, s)) return rc;
    TL_HIP(p, rt_event_record(p, b, b.ev_done[dir][cur]));
    s.unread[cur] = b.ev_done[dir][cur];
    ++*exchanges;
    return PM_OK;
  };
  // band k copies its predecessor's published row into dst (on k's stream, behind the predecessor's event)
  auto fetch = [&](int k, int pred, float* dst) -> int {
    Band& b = p->bands[(size_t)k];
    Band& s = p->bands[(size_t)pred];
    TL_HIP(p, rt_use(p, b));
    TL_HIP(p, rt_wait_event(p, b, s.ev_sent[cur]));
    TL_HIP(p, rt_copy_peer(p, b, dst, s, s.sent[cur], row_bytes));
    return consumed(b, s);
  };
  for (int it = 0; it < prm.patchmatch_iters; ++it) {
    for (Band& b : p->bands) TL_STAGE(p, b, ST_NOISE, pm_tile_noise(b.h, it));
    for (int k = 0; k < 4; ++k) {
      if (k == 0 || k == 2) {  // horizontal sweeps never leave the band
        for (Band& b : p->bands) TL_STAGE(p, b, ST_SWEEP, pm_tile_sweep(b.h, it, k));
        continue;
      }
      const bool down = k == 1;
      dir = down ? 0 : 1;
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
        TL_STAGE(p, b, ST_PRESWEEP, pm_tile_presweep(b.h, pred_row(b), has_pred ? b.used : nullptr),
                 {has_pred ? b.used : nullptr, false});
        TL_STAGE(p, b, ST_SWEEP, pm_tile_sweep(b.h, it, k));
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
          if (direct(p, b, s)) {
            TL_HIP(p, rt_use(p, b));
            TL_HIP(p, rt_wait_event(p, b, s.ev_sent[cur]));
            TL_STAGE(p, b, ST_EXCHANGE_ROUND,
                     pm_tile_exchange_round(b.h, it, k, pred_row(b), s.sent[cur], b.used, b.incoming, b.mask),
                     {s.sent[cur], true}, {b.used, false}, {b.incoming, false}, {b.mask, false});
            if (int rc = consumed(b, s)) return rc;
          } else {
            if (int rc = fetch(j, pr, b.incoming)) return rc;
            TL_STAGE(p, b, ST_EXCHANGE_ROUND,
                     pm_tile_exchange_round(b.h, it, k, pred_row(b), b.incoming, b.used, b.incoming, b.mask),
                     {b.incoming, false}, {b.used, false}, {b.incoming, false}, {b.mask, false});
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
        TL_STAGE(p, b, ST_ROW_MOVED, pm_tile_row_moved(b.h, out_row(b), b.sent[b.last], b.flag), {b.sent[b.last], false},
                 {b.flag, false});
      }
    }
  }
  for (Band& b : p->bands) {
    TL_STAGE(p, b, ST_BACKGROUND, pm_tile_background(b.h));
    TL_STAGE(p, b, ST_FINISH, pm_tile_finish(b.h, b.d_out_l, nv > 1 ? b.d_out_r : nullptr), {b.d_out_l, false},
             {nv > 1 ? b.d_out_r : nullptr, false});
  }
  *moved = false;
  for (Band& b : p->bands) {
    TL_HIP(p, rt_use(p, b));
    int f = 0;
    TL_HIP(p, rt_d2h(p, b, &f, b.flag, sizeof(int)));
    TL_HIP(p, rt_sync(p, b));
    *moved = *moved || f != 0;
  }
  for (Band& b : p->bands) b.unread[0] = b.unread[1] = nullptr;  // every stream is idle: nothing is being read any more
  return PM_OK;
}

// PM_TILED_SCHEDULE_PIPELINED: a vertical sweep runs through the bands IN ORDER -- the first band of the sweep direction
// sweeps its rows, publishes its boundary row, the next band stores that row in front of its chains and sweeps, and so
// on.  Nothing is guessed, so nothing is snapshot, compared, restored or swept again: the result is the sequential sweep by
// construction.  The bands' streams still overlap everything that does not cross a boundary (noise / cost, the horizontal
// sweeps; band 0 is in its next horizontal sweep while band n - 1 still finishes the vertical one).  Same objects and rules
// as the speculative schedule: the publisher records ev_sent on its stream, the reader waits for it, reads the row (peer
// copy or in place) and records ITS ev_done, which the publisher waits for before it overwrites the buffer.
int attempt_pipelined(pm_tiled_plan* p, int* exchanges) {
  const int n = (int)p->bands.size(), cols = p->cols, nv = p->n_views;
  const pm_params& prm = pm_internal::params(p->bands[0].h);
  const size_t row_bytes = sizeof(float) * (size_t)nv * cols;
  for (Band& b : p->bands) {
    TL_STAGE(p, b, ST_BEGIN,
             pm_tile_begin(b.h, &b.tile, b.d_left, b.d_right, b.band_rows, cols, p->have_seed_l ? b.d_seed_l : nullptr,
                           p->have_seed_r ? b.d_seed_r : nullptr),
             {b.d_left, false}, {b.d_right, false}, {p->have_seed_l ? b.d_seed_l : nullptr, false},
             {p->have_seed_r ? b.d_seed_r : nullptr, false});
  }
  int cur = 0;
  for (int it = 0; it < prm.patchmatch_iters; ++it) {
    for (Band& b : p->bands) TL_STAGE(p, b, ST_NOISE, pm_tile_noise(b.h, it));
    for (int k = 0; k < 4; ++k) {
      if (k == 0 || k == 2) {
        for (Band& b : p->bands) TL_STAGE(p, b, ST_SWEEP, pm_tile_sweep(b.h, it, k));
        continue;
      }
      const bool down = k == 1;
      const int dir = down ? 0 : 1;
      cur ^= 1;  // one buffer per vertical sweep; its reader of two sweeps ago has long been waited for
      for (int pos = 0; pos < n; ++pos) {
        const int j = down ? pos : n - 1 - pos;
        Band& b = p->bands[(size_t)j];
        if (pos > 0) {  // the predecessor's final boundary row in front of this band's chains
          Band& s = p->bands[(size_t)(down ? j - 1 : j + 1)];
          const int pred_row = down ? b.tile.own_row0 - 1 : b.tile.own_row0 + b.tile.own_rows;
          TL_HIP(p, rt_use(p, b));
          TL_HIP(p, rt_wait_event(p, b, s.ev_sent[cur]));
          if (direct(p, b, s)) {
            TL_STAGE(p, b, ST_SET_ROW, pm_tile_set_row(b.h, pred_row, s.sent[cur]), {s.sent[cur], true});
          } else {
            TL_HIP(p, rt_copy_peer(p, b, b.incoming, s, s.sent[cur], row_bytes));
            TL_STAGE(p, b, ST_SET_ROW, pm_tile_set_row(b.h, pred_row, b.incoming), {b.incoming, false});
          }
          if (int rc = inject_breach(p, b, s)) return rc;
          TL_HIP(p, rt_event_record(p, b, b.ev_done[dir][cur]));
          s.unread[cur] = b.ev_done[dir][cur];
          ++*exchanges;
        }
        TL_STAGE(p, b, ST_SWEEP, pm_tile_sweep(b.h, it, k));
        if (pos + 1 < n) {  // somebody continues from this band's last row
          const int out_row = down ? b.tile.own_row0 + b.tile.own_rows - 1 : b.tile.own_row0;
          if (b.unread[cur]) {
            TL_HIP(p, rt_wait_event(p, b, b.unread[cur]));
            b.unread[cur] = nullptr;
          }
          TL_STAGE(p, b, ST_GET_ROW, pm_tile_get_row(b.h, out_row, b.sent[cur]), {b.sent[cur], false});
          TL_HIP(p, rt_event_record(p, b, b.ev_sent[cur]));
        }
      }
    }
  }
  for (Band& b : p->bands) {
    TL_STAGE(p, b, ST_BACKGROUND, pm_tile_background(b.h));
    TL_STAGE(p, b, ST_FINISH, pm_tile_finish(b.h, b.d_out_l, nv > 1 ? b.d_out_r : nullptr), {b.d_out_l, false},
             {nv > 1 ? b.d_out_r : nullptr, false});
  }
  for (Band& b : p->bands) {
    TL_HIP(p, rt_use(p, b));
    TL_HIP(p, rt_sync(p, b));
  }
  for (Band& b : p->bands) b.unread[0] = b.unread[1] = nullptr;
  return PM_OK;
}

int create(pm_handle* const* bands, int n_bands, int rows, int cols, const int* logical, int simulate_peer,
           pm_tiled_plan** out) {
  if (!out) return PM_ERR_INVALID_ARG;
  *out = nullptr;
  if (!bands || n_bands < 1 || rows < n_bands || cols < 8) return PM_ERR_INVALID_ARG;
  pm_tiled_plan* p = new pm_tiled_plan;
  *out = p;  // also on failure: the plan then carries the message and is good for pm_tiled_destroy only
  if (logical) {
    p->audit = new Audit;
    p->audit->simulate_peer = simulate_peer;
  }
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
    b.index = k;
    b.dev = pm_internal::device(b.h);
    b.ldev = logical ? logical[k] : b.dev;
    if (b.ldev < 0) return fail(p, PM_ERR_INVALID_ARG, "band %d: negative logical device", k);
    b.stream = pm_internal::stream(b.h);
    if (p->audit) p->audit->obj[(const void*)b.stream] = b.ldev;  // the handle's stream lives on the handle's device
    band_of(k, n_bands, rows, p->halo, &b.tile, &b.band_rows);
    int mr, mc;
    pm_internal::plan_size(b.h, &mr, &mc);
    if (b.band_rows > mr || cols > mc)
      return fail(p, PM_ERR_SIZE, "band %d needs a plan of %d x %d, its handle has %d x %d (pm_tiled_band_rows)", k, cols,
                  b.band_rows, mc, mr);
    TL_HIP(p, rt_use(p, b));
    const size_t px = (size_t)b.band_rows * cols, own = (size_t)b.tile.own_rows * cols;
    const size_t row_bytes = sizeof(float) * (size_t)p->n_views * cols;
    TL_HIP(p, rt_malloc(p, b, (void**)&b.d_left, px));
    TL_HIP(p, rt_malloc(p, b, (void**)&b.d_right, px));
    TL_HIP(p, rt_malloc(p, b, (void**)&b.d_seed_l, px * 4));
    TL_HIP(p, rt_malloc(p, b, (void**)&b.d_seed_r, px * 4));
    TL_HIP(p, rt_malloc(p, b, (void**)&b.d_out_l, own * 4));
    TL_HIP(p, rt_malloc(p, b, (void**)&b.d_out_r, own * 4));
    for (int i = 0; i < 2; ++i) {
      TL_HIP(p, rt_malloc(p, b, (void**)&b.sent[i], row_bytes));
      TL_HIP(p, rt_event_create(p, b, &b.ev_sent[i]));
      TL_HIP(p, rt_event_create(p, b, &b.ev_done[0][i]));
      TL_HIP(p, rt_event_create(p, b, &b.ev_done[1][i]));
    }
    TL_HIP(p, rt_malloc(p, b, (void**)&b.used, row_bytes));
    TL_HIP(p, rt_malloc(p, b, (void**)&b.incoming, row_bytes));
    TL_HIP(p, rt_malloc(p, b, (void**)&b.mask, sizeof(int) * (size_t)p->n_views * cols));
    TL_HIP(p, rt_malloc(p, b, (void**)&b.flag, sizeof(int)));
    if (p->audit) {  // (not counted among a band's events by the tests: created outside the audited layer)
      TL_HIP(p, hipEventCreateWithFlags(&b.ev_probe, hipEventDisableTiming));
      p->audit->obj[(const void*)b.ev_probe] = b.ldev;
      TL_HIP(p, hipEventRecord(b.ev_probe, b.stream));
    }
  }
  // neighbouring bands on different devices: peer access over xGMI where the platform allows it (hipMemcpyPeerAsync is
  // then a direct copy; with PM_TILED_EXCHANGE_DIRECT the receiving band's kernel reads the row across the link)
  for (int k = 0; k + 1 < n_bands; ++k) {
    Band& a = p->bands[(size_t)k];
    Band& c = p->bands[(size_t)k + 1];
    if (a.ldev == c.ldev) continue;  // bands sharing a device: nothing to enable
    ++p->device_boundaries;
    bool linked = false;
    if (a.dev == c.dev) {
      linked = p->audit && p->audit->simulate_peer;  // logical devices on one physical device: as the test asks
    } else {
      int ok = 0;
      if (hipDeviceCanAccessPeer(&ok, a.dev, c.dev) == hipSuccess && ok) {
        (void)rt_use(p, a);
        const hipError_t e1 = hipDeviceEnablePeerAccess(c.dev, 0);
        (void)rt_use(p, c);
        const hipError_t e2 = hipDeviceEnablePeerAccess(a.dev, 0);
        const auto fine = [](hipError_t e) { return e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled; };
        linked = fine(e1) && fine(e2);
      }
      (void)hipGetLastError();
    }
    if (linked) {
      ++p->peer_links;
      a.peer_next = c.peer_prev = true;
    }
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

int pm_tiled_set_exchange(pm_tiled_plan* plan, int mode) {
  if (!plan) return PM_ERR_INVALID_ARG;
  if (mode != PM_TILED_EXCHANGE_AUTO && mode != PM_TILED_EXCHANGE_COPY && mode != PM_TILED_EXCHANGE_DIRECT)
    return fail(plan, PM_ERR_INVALID_ARG, "pm_tiled_set_exchange: unknown mode %d", mode);
  plan->exchange = mode;
  return PM_OK;
}

int pm_tiled_set_schedule(pm_tiled_plan* plan, int schedule) {
  if (!plan) return PM_ERR_INVALID_ARG;
  if (schedule != PM_TILED_SCHEDULE_SPECULATIVE && schedule != PM_TILED_SCHEDULE_PIPELINED)
    return fail(plan, PM_ERR_INVALID_ARG, "pm_tiled_set_schedule: unknown schedule %d", schedule);
  plan->schedule = schedule;
  return PM_OK;
}

void pm_tiled_destroy(pm_tiled_plan* plan) {
  if (!plan) return;
  for (Band& b : plan->bands) {
    (void)rt_use(plan, b);
    if (b.stream) (void)rt_sync(plan, b);
    void* bufs[] = {b.d_left, b.d_right, b.d_seed_l, b.d_seed_r, b.d_out_l, b.d_out_r, b.sent[0], b.sent[1],
                    b.used,   b.incoming, b.mask,    b.flag};
    for (void* q : bufs)
      if (q) (void)hipFree(q);
    for (int i = 0; i < 2; ++i) {
      if (b.ev_sent[i]) (void)hipEventDestroy(b.ev_sent[i]);
      for (int d = 0; d < 2; ++d)
        if (b.ev_done[d][i]) (void)hipEventDestroy(b.ev_done[d][i]);
    }
    if (b.ev_probe) (void)hipEventDestroy(b.ev_probe);
  }
  delete plan->audit;
  delete plan;
}

int pm_tiled_create(pm_handle* const* bands, int n_bands, int rows, int cols, pm_tiled_plan** out) {
  return create(bands, n_bands, rows, cols, nullptr, 0, out);
}

// ---- include/pm/testing.h -----------------------------------------------------------------------------------------
int pm_tiled_create_logical(pm_handle* const* bands, int n_bands, int rows, int cols, const int* logical_devices,
                            int simulate_peer_access, pm_tiled_plan** out) {
  if (!logical_devices) return PM_ERR_INVALID_ARG;
  return create(bands, n_bands, rows, cols, logical_devices, simulate_peer_access, out);
}

int pm_tiled_audit(const pm_tiled_plan* plan, pm_tiled_audit_record* records, int capacity, int* total, int* violations) {
  if (!plan || !plan->audit) return PM_ERR_INVALID_ARG;
  const Audit& a = *plan->audit;
  if (total) *total = (int)a.log.size();
  if (violations) *violations = a.violations;
  if (records)
    for (int i = 0; i < capacity && i < (int)a.log.size(); ++i) records[i] = a.log[(size_t)i];
  return PM_OK;
}

int pm_tiled_debug_inject(pm_tiled_plan* plan, int what) {
  if (!plan || !plan->audit || what < 0 || what > 3) return PM_ERR_INVALID_ARG;
  plan->audit->inject = what;
  return PM_OK;
}

int pm_tiled_audit_reset(pm_tiled_plan* plan) {
  if (!plan || !plan->audit) return PM_ERR_INVALID_ARG;
  plan->audit->log.clear();
  plan->audit->violations = 0;
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
  if (plan->schedule == PM_TILED_SCHEDULE_PIPELINED) {  // in order: exact without rounds
    int ex = 0;
    if (int rc = attempt_pipelined(plan, &ex)) return rc;
    if (info) {
      info->rounds_used = 0;
      info->repeated = 0;
      info->exchanges = ex;
    }
    return PM_OK;
  }
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
