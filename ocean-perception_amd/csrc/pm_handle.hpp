// pm_handle.hpp -- the engine handle and the host-side helpers its translation units share.  Not part of the ABI.
//
//   pm_engine.hip        handle lifecycle, parameter checks, Match() on device buffers, capture / replay, profiling
//   pm_launch.hip        the ONLY unit that includes the scalar-mode kernels (pm_kernels.hpp): one launch function each
//   pm_sweeps.hip        the directional sweep kernels (pm_run3.hpp, pm_run2.hpp, pm_wave.hpp, pm_serial.hpp)
//   pm_seed.hip          the device seeder (pm_seed.hpp)
//   pm_planes_host.hip   PM_MODE_PLANES: kernels (pm_planes.hpp), schedule, pm_planes_* entry points
//   pm_hostpath.hip      host-buffer entry points (pm_match_u8, submit / collect, the single-stage functions)
//   pm_tile.hip          the row-tiled phase API pm_tile_*
//   pm_tiled.hip         pm_tiled_*: n band handles of one process driven over that API
//   pm_imaging.hip       the imaging rows; sees the handle through pm_internal.hpp only
// Every __global__ kernel lives in exactly one unit; the others reach it through the launch functions declared here.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "pm/patchmatch.h"
#include "pm_color.hpp"
#include "pm_device.hpp"
#include "pm_hostcopy.hpp"
#include "pm_seed_api.hpp"
#include "pm_sweep_defs.hpp"
#include "pm_tune.hpp"

namespace pm {
namespace eng {

constexpr int kMaxEvents = 16384;  // event pairs kept before a forced drain

struct EventRec {
  hipEvent_t start, stop;
  int klass;
  int start_ref;  // >= 0: the stop event of that record is this one's start (the launch before it on the same stream)
};

}  // namespace eng
}  // namespace pm


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

  // scratch of the device seeder (pm_seed.hpp), one set per (lane, view): index lane * 2 + view.  Set 0 exists from
  // pm_create on, the others are allocated on first use (seeders of different views / lanes run side by side).
  pm::SeedScratch seeds[8] = {};
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
  char* pinned_dev = nullptr;  // the device's address of the staging slab (k_download writes the maps through it)

  // The frame SEQUENCE (pm_submit_* / pm_collect, pm_match_batch_u8; pm_engine.hip::seq_enqueue_chunk): frame k lives in
  // ring slot k % max_batch -- plane slot, device staging slot and slab slot of that index.  Uploads run on s_in, the two
  // views of a chunk (one frame, or two frames advanced through every launch together) on the two sequence streams,
  // the cross-check and the download on s_out; the handle's own stream is not involved, so nothing joins and nothing
  // forks per frame.
  struct PipeSlot {
    hipEvent_t in_done = nullptr;   // s_in: the slot's inputs are in device memory
    hipEvent_t head_done = nullptr; // s_in: images, gradients, line planes and seeds of the chunk that STARTS here are ready
    hipEvent_t v_done[2] = {nullptr, nullptr};  // view stream v: the chunk that STARTS at this slot has run
    hipEvent_t fin_done = nullptr;  // s_out: cross-check done, the slot's planes and input staging are free again
    hipEvent_t out_done = nullptr;  // s_out: the slot's maps have arrived on the host
    uint64_t tag = 0;
    int rows = 0, cols = 0;
    int state = 0;                  // 0 free, 1 uploaded and held for a partner, 2 enqueued
    bool has_sl = false, has_sr = false;
    const uint8_t *d_left = nullptr, *d_right = nullptr;  // where the chunk reads this frame (staging or caller memory)
    const float *d_seed_l = nullptr, *d_seed_r = nullptr;
    float *d_out_l = nullptr, *d_out_r = nullptr;         // where the cross-check writes (staging or caller memory)
    bool device_io = false;         // pm_submit_device: no copies at all
    hipEvent_t ready_ext = nullptr; // pm_submit_device_after: the caller's event behind the producer of the inputs
    float *out_l = nullptr, *out_r = nullptr;  // host maps bound at submit (null: handed to pm_collect)
    size_t out_step = 0;
    bool direct_l = false, direct_r = false;   // the download goes straight into the bound (registered) host map
  };
  // caller memory the engine may DMA from / into without staging (pm_host_alloc, pm_host_register)
  struct HostRange {
    char* base;
    size_t bytes;
    bool owned;
    char* dev_base;  // the device's address of `base` (null: not mapped -- downloads into it go through hipMemcpyAsync)
  };
  std::vector<HostRange> host_ranges;
  // streams a capture forked work onto and has not joined back yet (pm_capture_end refuses to end such a capture)
  std::vector<hipStream_t> cap_unjoined;
  // Per-view streams: the two views are independent until the cross-check, so their launch chains run on two streams and
  // one view's kernels fill the CUs the other view's kernel tails leave idle.  A single pair keeps its first view on the
  // handle's stream and forks the second onto view1_stream; the chunks of a batch or of a frame sequence run on the same
  // two streams, one chunk behind the other, with their cross-checks on s_out (pm_engine.hip::view_streams_create,
  // seq_enqueue_chunk).
  hipEvent_t view_fork = nullptr;
  hipStream_t view1_stream = nullptr;
  hipEvent_t view1_join = nullptr;
  // Which view of a single pair ended last the time before (view_end: timed events behind the views' last launches).
  // That view goes onto the handle's stream: a join the waiting stream reaches AFTER its event has fired costs nothing,
  // one it reaches before costs a cross-queue wake-up (~12 us of the reference's own 0.46 ms call).
  hipEvent_t view_end[2] = {nullptr, nullptr};
  bool view_end_recorded = false;
  static constexpr unsigned kViewEndEvery = 16;  // the order is sampled in every 16th single-pair call
  unsigned view_calls = 0;
  int late_view = 1;
  hipEvent_t out_join = nullptr;  // s_out -> the handle's stream at the end of a batch
  hipEvent_t in_join = nullptr;   // s_in -> the handle's stream (only when a capture is ended with the head stream unjoined)
  void* imaging_state = nullptr;  // owned by pm_imaging.hip (pm_internal.hpp)
  void* texmask_scratch = nullptr;  // pm_foreground_texture_mask: four byte planes, allocated on first use
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
  int seq_last = -1;  // ring slot of the frame enqueued last (is the device still busy with it?), -1: none

  // profiling
  bool profiling = false;
  std::vector<pm::eng::EventRec> ev_pool;
  int ev_used = 0;
  static constexpr int kProfTails = 16;
  struct ProfTail {
    hipStream_t stream;
    int rec;
  };
  ProfTail prof_tail[kProfTails];  // per stream: the last bracket, whose stop event can start the next (pm::eng::Launch)
  int n_prof_tail = 0;
  pm_profile prof{};

  char err[512] = {0};
};

namespace pm {
namespace eng __attribute__((visibility("hidden"))) {

void set_err(pm_handle* h, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#define PM_HIP(h, call)                                                                               \
  do {                                                                                                \
    hipError_t e_ = (call);                                                                           \
    if (e_ != hipSuccess) {                                                                           \
      pm::eng::set_err((h), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return PM_ERR_HIP;                                                                              \
    }                                                                                                 \
  } while (0)

inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

// Every stream of the engine is created through this.  PM_STREAM_PRIO = "main,view,copy,lane" (read once; one number
// = all four): the priority class of the handle's stream, the second view's stream, the upload / download streams and
// the lanes' streams -- 1 high, 0 default, -1 low.  Default "0,1,0,1": the streams the views of a pair and the
// pipelines of a batch run on are HIGH, not for the priority but because streams of different classes never share a
// hardware queue and a class of their own keeps them off the queues of whatever else the process creates (with
// default-priority streams, where a stream lands depends on the process's other queue-owning streams, and some
// constellations put both views of a pair on ONE queue: 384 -> 275 pairs/s; taking the queues in a fixed order in
// pm_create does not prevent that -- the runtime reassigns).  The handle's own stream and the copy streams stay
// default: copies on streams of a non-default class are slower (pm_match_u8 x 4: 12.3 -> 14.4 ms).  The price:
// default-priority streams that work side by side slow down while a stream of another class exists in the process
// (pm_submit_u8's upload / download streams beside the compute stream: 343 -> 280-300 pairs/s; eight band handles on
// one device 49 -> 117 ms); "0" restores one class for everything (tools/multi_handle.py, batch_after.py, pipe_timing.py).
enum StreamKind { kStreamMain = 0, kStreamView = 1, kStreamCopy = 2, kStreamLane = 3 };
hipError_t create_stream(hipStream_t* s, int kind, int prio_class);
int create_handle_streams(pm_handle* h);  // the handle's four streams, together (pm_engine.hip)

// Brackets the launches of one kernel class with a pair of events while the handle is profiling.
// While the handle is profiling, the launches of one kernel class are bracketed by events on the current stream.  An
// in-stream event before a launch fires when the launch in front of it has finished: the stop event of the previous
// bracket on the same stream IS the start of the next one, so a chain of launches costs one event per launch, not two
// (each record is a serialising packet in the queue: with two per launch a profiled Match ran 7 % longer).  prof_break()
// ends a chain where something else was put on the stream (an event wait, the boundary of an API call).
inline void prof_break(pm_handle* h, hipStream_t stream) {
  for (int i = 0; i < h->n_prof_tail; ++i)
    if (h->prof_tail[i].stream == stream) {
      h->prof_tail[i] = h->prof_tail[--h->n_prof_tail];
      return;
    }
}
inline void prof_break_all(pm_handle* h) { h->n_prof_tail = 0; }

struct Launch {
  pm_handle* h;
  int klass;
  bool timed;
  int rec = -1;
  Launch(pm_handle* h_, int k) : h(h_), klass(k), timed(h_->profiling) {
    if (!timed) return;
    if (h->ev_used == (int)h->ev_pool.size()) {
      if ((int)h->ev_pool.size() >= kMaxEvents) {
        timed = false;  // drained by pm_profile_read; never block inside a launch path
        prof_break(h, h->stream);
        return;
      }
      EventRec r;
      r.klass = k;
      r.start_ref = -1;
      if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) {
        timed = false;
        return;
      }
      h->ev_pool.push_back(r);
    }
    rec = h->ev_used++;
    EventRec& r = h->ev_pool[rec];
    r.klass = k;
    r.start_ref = -1;
    for (int i = 0; i < h->n_prof_tail; ++i)
      if (h->prof_tail[i].stream == h->stream) r.start_ref = h->prof_tail[i].rec;
    if (r.start_ref < 0) (void)hipEventRecord(r.start, h->stream);
  }
  ~Launch() {
    if (!timed || rec < 0) return;
    (void)hipEventRecord(h->ev_pool[rec].stop, h->stream);
    for (int i = 0; i < h->n_prof_tail; ++i)
      if (h->prof_tail[i].stream == h->stream) {
        h->prof_tail[i].rec = rec;
        return;
      }
    if (h->n_prof_tail < pm_handle::kProfTails) h->prof_tail[h->n_prof_tail++] = {h->stream, rec};
  }
};

// ---- pm_engine.hip: geometry, checks, the Match() schedule -------------------------------------------------------
int check_patch(pm_handle* h, int pw, int ph);
int check_size(pm_handle* h, int rows, int cols, int n);
PlaneSet plane_set(const pm_handle* h, int rows, int cols, int n_views);
CostParams cost_params(const pm_params& p, int pw, int ph);
Interior interior(const pm_params& p, int rows, int cols, int pw, int ph);
SweepGeom sweep_geom(const pm_params& p, const Interior& in, int k);  // k: 0 = row +1, 1 = col +1, 2 = row -1, 3 = col -1
int launch_check(pm_handle* h, const char* what);
int ensure_noise(pm_handle* h, int rows, int cols);
void abort_capture(pm_handle* h);
int refuse_while_capturing(pm_handle* h, const char* what);
int capture_open(pm_handle* h);                                             // pm_engine.hip
int capture_close(pm_handle* h, hipGraphExec_t* exec, const char* what);
bool pair_planes_wanted(const pm_handle* h);
int pair_planes_alloc(pm_handle* h);
int run_one_view_set(pm_handle* h, const PlaneSet& ps, int slots);
int match_device_impl(pm_handle* h, int n, const uint8_t* d_left, const uint8_t* d_right, int rows, int cols,
                      const float* d_seed_l, const float* d_seed_r, float* d_disp_l, float* d_disp_r);
// view streams, chunk events, and one chunk of a batch / frame sequence (pm_engine.hip)
int view_streams_create(pm_handle* h);
PlaneSet pair_plane_set(const PlaneSet& ps, int b);  // the plane set of the pairs from b on
int lane_fork(pm_handle* h);
int lane_join(pm_handle* h);
int seq_events_create(pm_handle* h);
int seq_chunk_pairs();
bool seq_pipelined(const pm_handle* h);
int seq_enqueue_chunk(pm_handle* h, int b, int c, const uint8_t* d_left, const uint8_t* d_right, int rows, int cols,
                      const float* d_seed_l, const float* d_seed_r, float* d_disp_l, float* d_disp_r, hipEvent_t ready,
                      hipEvent_t slot_free, hipEvent_t v_done[2], hipEvent_t head_done);
SeedParams seed_params(const pm_params& p);
int alloc_seed_scratch(pm_handle* h, SeedScratch& sc);
// SparseInit (or Patchmatch::Initialize(.., 1)) for view `view` of pair `b` straight into its disparity plane
int run_sparse_init(pm_handle* h, const PlaneSet& ps, int b, int view, int scratch = 0, unsigned stages = kSeedAllStages);

// ---- pm_launch.hip: one function per scalar-mode kernel, enqueued on h->stream -----------------------------------
// k_prep, or k_prep_bgr when the call came in through pm_match_bgr_device (the gray images are then never stored)
// seeds != null: the launch also copies the seed maps (null members = all background) into the disparity planes of its
// view(s), i.e. what launch_seed does
struct PrepSeedMaps {
  const float* l;
  const float* r;
};
void launch_prep(pm_handle* h, const PlaneSet& ps, const uint8_t* d_left, const uint8_t* d_right, int n, size_t stride,
                 int view = -1, const PrepSeedMaps* seeds = nullptr);
void launch_prep_view(pm_handle* h, const PlaneSet& ps, const float* d_iml, const float* d_imr, const float* d_Gl,
                      const float* d_Gr, size_t stride);
// transposed copies + line-triple / quad planes of n pairs (run by every path that ran a prep kernel);
// view >= 0: the planes of that view only (per-view streams: each stream derives its own planes)
int run_transpose(pm_handle* h, const PlaneSet& ps, int n, int view = -1);
void launch_seed(pm_handle* h, const PlaneSet& ps, const float* d_seed_l, const float* d_seed_r, int n, int view = -1);
// noise + clamp + cost of the current disparity (amount < 0: cost only); RemoveBackground / MaskBackground
void launch_noise_cost(pm_handle* h, const PlaneSet& ps, const CostParams& cp, const Interior& in, float amount,
                       int slots, int keep_zero);
void launch_noise_only(pm_handle* h, const PlaneSet& ps, const CostParams& cp, float amount);
void launch_background(pm_handle* h, const PlaneSet& ps, const CostParams& cp, const Interior& in, float factor,
                       int cached, int slots);
int run_sweep(pm_handle* h, const PlaneSet& ps, const CostParams& cp, const SweepGeom& g, int slots, float amp = 1e30f);
void launch_finalize(pm_handle* h, const PlaneSet& ps, float* d_disp_l, float* d_disp_r, int n);
void launch_mask_occlusions(pm_handle* h, float* d_disp_l, const float* d_disp_r, int rows, int cols);
void launch_state_row(pm_handle* h, const PlaneSet& ps, int r, float* d_buf, int to_buf);
void launch_restore_cols(pm_handle* h, const PlaneSet& ps, const float* snap_disp, const float* snap_cost,
                         const int* d_mask);
void launch_tile_round(pm_handle* h, const PlaneSet& ps, const float* snap_disp, const float* snap_cost,
                       const float* d_incoming, const float* d_used, float* d_used_next, int* d_mask, int pred_r,
                       int y_lo, int y_hi);
void launch_state_row_moved(pm_handle* h, const PlaneSet& ps, int r, const float* d_ref, int* d_flag);
void launch_tile_presweep(pm_handle* h, const PlaneSet& ps, float* snap_disp, float* snap_cost, const float* d_row, int pred_r);
// rows x cols floats from tight device memory into page-locked host memory (its DEVICE address), row stride in floats
void launch_download(pm_handle* h, float* dst_dev, size_t dst_step_floats, const float* d_src, int rows, int cols,
                     hipStream_t stream);
void launch_upload(pm_handle* h, float* d_dst, const float* src_dev, int words, hipStream_t stream);
void launch_copy_in(pm_handle* h, const PlaneSet& ps, const float* d_src);
void launch_copy_out(pm_handle* h, const PlaneSet& ps, float* d_dst, int which);
void launch_copy_disp_strided(pm_handle* h, const PlaneSet& ps, float* d_buf, size_t stride, int to_buf);

// ---- pm_planes_host.hip ------------------------------------------------------------------------------------------
int planes_alloc(pm_handle* h);
int planes_match(pm_handle* h, int n, const uint8_t* d_left, const uint8_t* d_right, int rows, int cols,
                 const float* d_seed_l, const float* d_seed_r, float* d_disp_l, float* d_disp_r);

}  // namespace eng
}  // namespace pm
