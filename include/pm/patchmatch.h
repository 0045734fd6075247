/*
 * pm/patchmatch.h -- C ABI of the MI355X (gfx950) PatchMatch stereo engine.
 *
 * This is the drop-in boundary for the reference module src/vehicle/patchmatch_gpu
 * (library target `vehicle_pm_gpu`, src/vehicle/patchmatch_gpu/CMakeLists.txt:1,10).
 * Host code (C++ wrapper bm::pm::PatchmatchGpu in ocean-perception_amd/host/, or any FFI)
 * sees only plain pointers and sizes; no HIP, torch or OpenCV type crosses this header.
 * All `file:line` citations are relative to the reference tree.
 *
 * Conventions
 *   - every entry point returns PM_OK (0) or a negative pm_status; nothing throws or aborts
 *     (the reference aborts through glog CHECK / cv::Exception and never checks CUDA errors,
 *     patchmatch_gpu.cu:331-411);
 *   - images are single channel, row major; `*_step` arguments are row strides in BYTES
 *     (cv::Mat::step), 0 meaning "tightly packed";
 *   - disparity maps are float32, 0 = background / unknown (patchmatch_gpu.cu:422-423);
 *   - one handle = one device + one HIP stream + preallocated scratch; a handle is not
 *     re-entrant (one in-flight call), different handles are independent.  Every entry point
 *     binds the handle's device first, so it may be called from any host thread (the reference's
 *     `Sequence` caller invokes Match from a playback worker thread,
 *     test/stereo_matching/patchmatch_gpu_test.cpp:122-136).
 */
#ifndef PM_PATCHMATCH_H_
#define PM_PATCHMATCH_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PM_ABI_VERSION 6
#define PM_MAX_ITERS 16
#define PM_MAX_PATCH 15 /* largest supported window side (odd) */

typedef struct pm_handle pm_handle;

typedef enum pm_status {
  PM_OK = 0,
  PM_ERR_INVALID_ARG = -1, /* null pointer, even / oversized patch, bad enum ...            */
  PM_ERR_SIZE = -2,        /* image larger than the handle was planned for, batch too large */
  PM_ERR_HIP = -3,         /* a HIP runtime call failed; pm_last_error() has the text       */
  PM_ERR_NO_DEVICE = -4,   /* no usable gfx950 device (the engine has NO CPU fallback)      */
  PM_ERR_NOMEM = -5,
  PM_ERR_BUSY = -6,        /* pm_submit_u8 with max_batch pairs in flight / pm_collect with none */
  PM_ERR_STATE = -7        /* the call is not valid in the handle's current state (e.g. pm_capture_end on a capture
                              whose forked streams never joined back)                                   */
} pm_status;

/* Which reference code the sweeps/cost reproduce bit for bit. */
typedef enum pm_semantics {
  /* src/vehicle/stereo_matching/patchmatch.cpp (AddNoise :143-155, PropagateNeighbors :158-196,
   * Propagate :248-311, RemoveBackground :314-360) with the cost functor of
   * test/stereo_matching/patchmatch_test.cpp:30-45 on patch_w x patch_h windows.
   * This is the parity target named by BASELINE.json. */
  PM_SEM_CPU = 0,
  /* src/vehicle/patchmatch_gpu/patchmatch_gpu.cu kernels (L1GradientCost3x3 :72-114,
   * PropagateRow :116-172, PropagateCol :175-230, MaskBackground :233-270) run race-free as one
   * stripe per row / column. */
  PM_SEM_GPU = 1
} pm_semantics;

/* What a pixel's state is.
 *   PM_MODE_SCALAR  one disparity per pixel: the reference's algorithm (semantics above), the parity target.
 *   PM_MODE_PLANES  a slanted plane per pixel (a, b, z) + its cost: random plane initialisation, red-black
 *                   spatial propagation, view propagation, random plane refinement, P x P windowed cost --
 *                   the kernels BASELINE.json's north_star names.  The reference has no counterpart
 *                   (patchmatch_gpu.cu:379-411 is scalar); the algorithm is defined by
 *                   oracle/pm_planes_oracle.h and reproduced bit for bit on the same seeded random numbers. */
typedef enum pm_mode { PM_MODE_SCALAR = 0, PM_MODE_PLANES = 1 } pm_mode;
/* Storage type of the plane / cost planes in PM_MODE_PLANES (BASELINE configs[4]: fp16 planes/cost). */
typedef enum pm_state_dtype { PM_STATE_F32 = 0, PM_STATE_F16 = 1 } pm_state_dtype;

/* How the directional sweeps are executed on the device (results are identical). */
typedef enum pm_engine {
  PM_ENGINE_AUTO = 0,   /* = PM_ENGINE_RUNBLK2 for both semantics                                              */
  PM_ENGINE_SERIAL = 1, /* one lane per row/column chain, strictly sequential: correctness anchor         */
  PM_ENGINE_WAVE = 2,   /* one wavefront per chain, window taps spread over the 64 lanes (PM_SEM_GPU: one  */
                        /* lane per chain segment): second, independent anchor                             */
  /* 3, 4: one-segment-per-wavefront run engines of round 1, retired (PM_ERR_INVALID_ARG)                  */
  PM_ENGINE_RUNBLK2 = 5 /* workgroup per chain, a whole adoption run per step, two or four chain segments  */
                        /* per wavefront, in-kernel fix-up to a fixpoint: the product engine               */
} pm_engine;

/*
 * Mirrors bm::pm::PatchmatchGpu::Params (patchmatch_gpu.h:79-92): the four scalar fields keep
 * their names and defaults.  The nested detector_params / matcher_params of the reference
 * configure the CPU seeder (SparseInit, patchmatch_gpu.cu:414-442); seeds are an explicit input
 * at this boundary, so they live in the C++ wrapper, not here.
 * The remaining fields expose what the reference hard-codes per call site.
 */
typedef struct pm_params {
  uint32_t struct_size; /* = sizeof(pm_params), checked by pm_create */
  uint32_t abi_version; /* = PM_ABI_VERSION */

  /* --- reference fields ------------------------------------------------------------------ */
  float cost_alpha;          /* 0.9  PM_SEM_GPU cost weight (patchmatch_gpu.h:85)              */
  int patchmatch_iters;      /* 3    iterations of {noise, 4 sweeps} (patchmatch_gpu.h:86)      */
  int init_dilate_factor;    /* 4    seed dilation 2*(2^f+1)+1 (patchmatch_gpu.h:87, .cu:436)   */
  float cost_improve_factor; /* 0.8  PM_SEM_GPU MaskBackground (patchmatch_gpu.h:88, .cu:267)   */

  /* --- what the reference hard-codes ------------------------------------------------------ */
  int semantics;                  /* pm_semantics */
  int engine;                     /* pm_engine */
  float noise_amp[PM_MAX_ITERS];  /* amplitude of iteration i; default 32/2^i (patchmatch_gpu.cu:395);
                                     the CPU test uses 32, 8, 2, 0.5 (patchmatch_test.cpp:173-179) */
  int patch_w[PM_MAX_ITERS];      /* PM_SEM_CPU window of iteration i (odd, <= PM_MAX_PATCH)      */
  int patch_h[PM_MAX_ITERS];
  int bg_patch_w, bg_patch_h;     /* PM_SEM_CPU RemoveBackground window (patchmatch_test.cpp:183) */
  float win_by_factor;            /* PM_SEM_CPU RemoveBackground factor, 1.5 (patchmatch_test.cpp:183) */
  float functor_alpha;            /* 0.7  L1GradientCostFunction (patchmatch_test.cpp:35)          */
  float functor_tau_color;        /* 50   (patchmatch_test.cpp:36)                                 */
  float functor_tau_grad;         /* 20   (patchmatch_test.cpp:37)                                 */
  uint64_t noise_seed;            /* 123  cv::RNG seed (patchmatch.cpp:146, patchmatch_gpu.cu:341) */
  int left_right_check;           /* 1    right view + MaskOcclusions (patchmatch_gpu.cu:357-372)  */

  /* --- sparse seeding: PatchmatchGpu::SparseInit (patchmatch_gpu.cu:414-442) on the device ------ */
  int sparse_init;                /* 0: a NULL seed map means "all background"; 1: a NULL seed map is
                                     computed by SparseInit (what the reference's Match() always does) */
  int max_features_per_frame;     /* 200   ft::FeatureDetector::Params (feature_detector.hpp:28)       */
  int min_distance_btw_features;  /* 20    min_distance_btw_tracked_and_detected_features (:31)        */
  int gftt_block_size;            /* 5     (:33)                                                        */
  double gftt_quality_level;      /* 0.01  (:32)                                                        */
  int templ_cols;                 /* 31    ft::StereoMatcher::Params (stereo_matcher.hpp:21)            */
  int templ_rows;                 /* 11    (:22)                                                        */
  int max_disp;                   /* 128   (:23)                                                        */
  double max_matching_cost;       /* 0.15  (:24)                                                        */
  int gftt_use_harris;            /* 0     gftt_use_harris_corner_detector (feature_detector.hpp:34): corner response
                                           det(M) - k trace(M)^2 instead of the smaller eigenvalue               */
  double gftt_k;                  /* 0.04  gftt_k (:35), used with gftt_use_harris only                          */
  int subpixel_corners;           /* 0     cv::cornerSubPix on the detected corners (feature_detector.cpp:110-120)    */
  int subpix_winsize;             /* 10    half window of that refinement (feature_detector.hpp:40), <= 15            */
  int subpix_zerozone;            /* -1    half size of its dead zone, -1 = none (:41)                                */
  int subpix_maxiters;            /* 10    (:42)                                                                      */
  float subpix_epsilon;           /* 0.01  (:43)                                                                      */
  int subpixel_refinement;        /* 0     cv::cornerSubPix on the match in the right image, window 10, 40 steps, 0.001
                                           (stereo_matcher.cpp:94-103): the seed disparities become fractional        */
  int cpu_initialize_factor;      /* 0: a self-seeded Match() seeds with SparseInit (patchmatch_gpu.cu:414-442);
                                     1: with Patchmatch::Initialize(il, ir, 1) as the CPU recipe does
                                        (patchmatch.cpp:52-87 called at patchmatch_test.cpp:149-150): dilation
                                        2*(2^(f-1)+1)+1 = 5x5 and the seeds divided by 2^f = 2      */

  /* --- PM_MODE_PLANES (no reference counterpart; defaults of oracle/pm_planes_oracle.c) ------------------
   * Shared with the scalar mode: patchmatch_iters, patch_w[0] (square window), noise_amp[i] (dz of the first
   * refinement step of iteration i), functor_*, noise_seed, max_disp (disparities live in [0, max_disp]),
   * left_right_check, sparse_init / seed maps (a seed > 0 fixes a pixel's initial disparity). */
  int mode;                       /* pm_mode, 0                                                          */
  int state_dtype;                /* pm_state_dtype, 0                                                   */
  int plane_refine_steps;         /* 3     candidates per pixel and iteration, ranges halving             */
  float plane_slope_max;          /* 1.0   |a|, |b| bound                                                 */
  float plane_slope_init;         /* 0.25  initial slopes uniform in +-this                               */
  float plane_slope_per_disp;     /* 1/64  slope range of a refinement step = dz * this                   */
  float plane_lr_tol;             /* 1.0   |dl - dr| above which the left disparity is zeroed             */
  int plane_window;               /* 1     pm_plane_window: which taps of the window count                        */
  int plane_neighbours;           /* 0     pm_plane_neighbours: the spatial stage's candidates                    */

  /* --- how the handle sits in the host process (ABI 6; no reference counterpart: the reference runs everything on the
   * default stream with device-wide synchronisation, patchmatch_gpu.cu:396-410) ---------------------------------- */
  int stream_priority;            /* 1     pm_stream_priority: the class of the handle's four streams              */
} pm_params;
/* The priority class ALL streams of a handle are created in.  HIGH (default): not for the priority but because streams
 * of different classes never share a hardware queue, which keeps the matcher's two view streams off the queues of
 * whatever else the process creates (a framework's side streams, RCCL) -- measured 384 -> 275 pairs/s when both views
 * land on one queue.  The price: the matcher's kernels are scheduled AHEAD of the host application's default-class work.
 * An application that must keep its own kernels in front picks DEFAULT (the matcher then shares the default class's four
 * queues with the application's streams; its rate depends on what else owns queues) or LOW (a class of its own again,
 * behind everything else). */
typedef enum pm_stream_priority { PM_STREAM_PRIO_LOW = -1, PM_STREAM_PRIO_DEFAULT = 0, PM_STREAM_PRIO_HIGH = 1 } pm_stream_priority;
/* PM_MODE_PLANES window.  CHECKER (default since ABI 5): tap (i, j) counts iff i + j is even -- the centre and every other
 * tap in both directions, 61 of 121 for 11 x 11; the mean divides by the taps that count.  Half the arithmetic of the
 * full window at the same quality on the benchmark pairs (99.86 % of the valid pixels within 1 px either way). */
typedef enum pm_plane_window { PM_PL_WINDOW_FULL = 0, PM_PL_WINDOW_CHECKER = 1 } pm_plane_window;
/* PM_MODE_PLANES spatial stage.  FOUR (default): a pixel is offered the planes of its left, right, upper and lower
 * neighbour.  TWO: the left and the upper one in the colour passes of an even iteration, the right and the lower one in
 * those of an odd iteration -- half the evaluations of the stage (the whole mode ~20 % faster); on the benchmark pairs
 * the same validity, 99.84-99.88 % of the valid pixels within 1 px (FOUR: 99.86-99.89 %), mean absolute error + 12-13 %. */
typedef enum pm_plane_neighbours { PM_PL_NEIGH_FOUR = 0, PM_PL_NEIGH_TWO = 1 } pm_plane_neighbours;

/* Fills *p with the reference defaults for the given semantics. */
void pm_params_default(pm_params* p, int semantics);

/* Replaces the constructor PatchmatchGpu::PatchmatchGpu(const Params&) (patchmatch_gpu.cu:322-328)
 * and the lazy GpuMat scratch of patchmatch_gpu.h:120-123: all device memory for up to `max_batch`
 * pairs of `max_rows` x `max_cols` is allocated here, and the unit-noise image the reference
 * creates on first use (patchmatch_gpu.cu:339-344, Q3/Q20) is generated on the device per size. */
int pm_create(const pm_params* params, int device, int max_rows, int max_cols, int max_batch,
              pm_handle** out);
void pm_destroy(pm_handle* h);
const char* pm_last_error(const pm_handle* h);
/* Text for a status code (valid without a handle). */
const char* pm_status_string(int status);

/* Replaces void PatchmatchGpu::Match(const Image1b& iml, const Image1b& imr, Image1f& disp,
 * Image1f& dispr) (patchmatch_gpu.h:99-102, patchmatch_gpu.cu:331-376) with host buffers.
 * seed_l / seed_r: sparse-init disparity maps (what SparseInit returns, .cu:414-442) in left /
 * right image coordinates, or NULL for "all background".  disp_r may be NULL when
 * left_right_check == 0. */
int pm_match_u8(pm_handle* h, const uint8_t* left, const uint8_t* right, int rows, int cols,
                size_t image_step, const float* seed_l, const float* seed_r, size_t seed_step,
                float* disp_l, float* disp_r, size_t disp_step);

/* n independent pairs per call (pair i -> batch slot i); arrays of n pointers, tightly packed. */
int pm_match_batch_u8(pm_handle* h, int n, const uint8_t* const* left, const uint8_t* const* right,
                      int rows, int cols, const float* const* seed_l, const float* const* seed_r,
                      float* const* disp_l, float* const* disp_r);

/* The same Match() for a SEQUENCE of pairs (the per-frame callback loop of
 * test/stereo_matching/patchmatch_gpu_test.cpp:118-128) with the copies off the critical path and the frames
 * overlapping on the device: pm_submit_u8 uploads a pair and enqueues its match without waiting; pm_collect waits for
 * the OLDEST submitted pair and hands its maps out.  Up to max_batch pairs may be in flight (then pm_submit_u8 returns
 * PM_ERR_BUSY); results are those of pm_match_u8, in submission order.  `tag` is handed back by the matching
 * pm_collect.  The input buffers may be reused as soon as pm_submit_u8 returns UNLESS they lie in memory made known
 * through pm_host_alloc / pm_host_register (below): such buffers are read by DMA and must stay untouched until the
 * frame has been collected.  While the device is busy with earlier frames a submitted frame may be HELD until the
 * next pm_submit (two frames advanced through every launch together run 10 % faster than one after the other);
 * pm_flush enqueues a held frame at once; pm_collect enqueues the frame it is asked for at once, and a LATER held
 * frame as soon as the device has nothing else to do (before its wait, or right after it when the frame being
 * collected was what kept the device busy -- the loop submit(k + 1); collect(k) never leaves the device idle).  The
 * other host-buffer entry points must not be called while pairs are in flight (they share the staging buffers). */
int pm_submit_u8(pm_handle* h, const uint8_t* left, const uint8_t* right, int rows, int cols,
                 size_t image_step, const float* seed_l, const float* seed_r, size_t seed_step,
                 uint64_t tag);
/* pm_submit_u8 with the output maps bound at submission: the matching pm_collect may pass NULL maps (or the same
 * pointers).  If the maps lie in pm_host_alloc / pm_host_register memory the download goes straight into them -- no
 * staging copy on either side of the frame. */
int pm_submit_bound_u8(pm_handle* h, const uint8_t* left, const uint8_t* right, int rows, int cols,
                       size_t image_step, const float* seed_l, const float* seed_r, size_t seed_step,
                       float* disp_l, float* disp_r, size_t disp_step, uint64_t tag);
/* The same sequence for callers whose frames are DEVICE resident (tightly packed rows x cols planes, as for
 * pm_match_device): nothing is copied; inputs must stay untouched and outputs unread until the frame is collected
 * (pm_collect with NULL maps waits for it).  ORDERING: unlike pm_match_device, a frame of the sequence is NOT ordered
 * behind what pm_stream(h) holds -- its second view and a self-seeding head run on internal streams as soon as the
 * frames in front of them allow (ordering every frame behind the handle's stream, on which the first view of the
 * previous frame runs, would tie the two view streams together at every frame boundary: measured 486 -> 463 pairs/s).
 * The inputs must therefore be COMPLETE in device memory when pm_submit_device is called (the producer was
 * synchronised), or the producer's completion is handed over as an event: */
int pm_submit_device(pm_handle* h, const uint8_t* d_left, const uint8_t* d_right, int rows, int cols,
                     const float* d_seed_l, const float* d_seed_r, float* d_disp_l, float* d_disp_r, uint64_t tag);
/* pm_submit_device for inputs that are still being produced on the device: `ready_event` is a hipEvent_t (as void*)
 * the caller recorded behind the producer of THIS frame's inputs, on whatever stream that work runs; both views and the
 * head of the frame wait for it on the device, nothing waits on the host.  The event is consumed INSIDE the call (an
 * internal stream is made to wait for it and the frame waits for that stream), also when the frame is held for a
 * partner: the caller may re-record or destroy the event as soon as the call returns.  NULL = pm_submit_device. */
int pm_submit_device_after(pm_handle* h, const uint8_t* d_left, const uint8_t* d_right, int rows, int cols,
                           const float* d_seed_l, const float* d_seed_r, float* d_disp_l, float* d_disp_r, uint64_t tag,
                           void* ready_event);
int pm_collect(pm_handle* h, float* disp_l, float* disp_r, size_t disp_step, uint64_t* tag);
/* enqueues a frame that is being held for a partner (see pm_submit_u8); never needed for correctness */
int pm_flush(pm_handle* h);
int pm_in_flight(const pm_handle* h);

/* Caller memory the host-buffer entry points may DMA from and into WITHOUT the staging copy through the handle's
 * pinned slab: pm_host_alloc hands out page-locked memory (for the cv::Mat / Image buffers of a capture loop),
 * pm_host_register page-locks a range the caller already owns (hipHostRegister; fails with PM_ERR_HIP where the
 * platform refuses).  Every image, seed map or output map that lies ENTIRELY inside such a range is transferred
 * straight from / to it (strided buffers by a 2-D copy); anything else is staged as before.  The reference's host
 * Match() uploads and downloads pageable cv::Mat memory synchronously (patchmatch_gpu.cu:343-375). */
int pm_host_alloc(pm_handle* h, size_t bytes, void** ptr);
int pm_host_free(pm_handle* h, void* ptr);
int pm_host_register(pm_handle* h, void* ptr, size_t bytes);
int pm_host_unregister(pm_handle* h, void* ptr);

/* Replaces void PatchmatchGpu::Match(const cu::GpuMat& ...) (patchmatch_gpu.h:104-108,
 * patchmatch_gpu.cu:379-411) widened to whole pairs: all pointers are DEVICE memory holding n
 * tightly packed planes ([n][rows][cols]); nothing is copied through the host and the call only
 * enqueues work on the handle's stream (pm_synchronize to wait).  Gradients are computed inside. */
int pm_match_device(pm_handle* h, int n, const uint8_t* d_left, const uint8_t* d_right, int rows,
                    int cols, const float* d_seed_l, const float* d_seed_r, float* d_disp_l,
                    float* d_disp_r);
/* Replaces void PatchmatchGpu::Match(const cu::GpuMat& iml, const cu::GpuMat& imr, const cu::GpuMat& Gl,
 * const cu::GpuMat& Gr, cu::GpuMat& disp) (patchmatch_gpu.h:104-108, patchmatch_gpu.cu:379-411) as it stands: ONE
 * view, caller-supplied gradients, `disp` holds the sparse-init map on entry and the result on return.
 * All pointers are DEVICE memory of CV_32F layout: rows x cols floats, `step` = row stride in BYTES (GpuMat::step,
 * 0 = tightly packed); d_iml / d_imr hold the 8-bit image values as floats (convertTo, patchmatch_gpu.cu:346-349).
 * `stream` (a hipStream_t, or NULL for the handle's own stream) is the caller's stream: the work is ordered after
 * what that stream holds at the time of the call, and work the caller enqueues afterwards is ordered after it.
 * Iterations {noise, 4 sweeps} + background mask under the handle's semantics; no cross-check (one view).
 * PM_MODE_SCALAR only. */
int pm_match_view_device(pm_handle* h, const float* d_iml, const float* d_imr, const float* d_Gl,
                         const float* d_Gr, int rows, int cols, size_t step, float* d_disp, size_t disp_step,
                         void* stream);
/* Replaces the unit-noise image the reference's host Match() creates on first use and its device Match()
 * silently depends on (patchmatch_gpu.cu:339-344 vs :395): installs a caller-supplied HOST table of
 * rows x cols floats (tightly packed) in place of cv::RNG(noise_seed) uniform [-1, 1).  It stays until a call
 * with another image size regenerates the default table. */
int pm_set_unit_noise(pm_handle* h, const float* noise, int rows, int cols);
int pm_synchronize(pm_handle* h);
/* Optional: record the calls made between pm_capture_begin and pm_capture_end (e.g. one pm_match_device with fixed
 * device pointers and image size; the handle must already have matched that size once) into a HIP graph instead
 * of executing them, then launch the whole DAG -- both view streams -- with pm_replay.  Results are those of the
 * recorded calls on whatever the device buffers hold at replay time.  Profiling must be off while capturing. */
int pm_capture_begin(pm_handle* h);
int pm_capture_end(pm_handle* h);
int pm_replay(pm_handle* h);
/* The hipStream_t the handle enqueues on (as void*), for event timing by the caller. */
void* pm_stream(pm_handle* h);

/* ---- single stages, host buffers: one entry point per reference function ------------------ */

/* GradientMagnitude (patchmatch_gpu.cu:307-319) / ComputeGradient (patchmatch_test.cpp:48-64). */
int pm_gradient_magnitude(pm_handle* h, const uint8_t* image, int rows, int cols, float* grad);
/* The cv::RNG(seed) uniform [-1,1) image of patchmatch_gpu.cu:339-344, as generated on the device. */
int pm_unit_noise(pm_handle* h, int rows, int cols, float* noise);
/* AddForegroundNoise (patchmatch_gpu.cu:298-304) == Patchmatch::AddNoise(disp, amount, disp > 0)
 * (patchmatch.cpp:143-155 as called at patchmatch_test.cpp:173-179). */
int pm_add_noise(pm_handle* h, float* disp, int rows, int cols, float amount);
/* Patchmatch::Propagate (patchmatch.cpp:248-311) for PM_SEM_CPU, or the PropagateRow(+1),
 * PropagateCol(+1), PropagateRow(-1), PropagateCol(-1) sequence (patchmatch_gpu.cu:397-403) for
 * PM_SEM_GPU.  pass_mask bit k enables the k-th of those four sweeps.  disp is updated in place. */
int pm_propagate(pm_handle* h, const uint8_t* left, const uint8_t* right, int rows, int cols,
                 float* disp, int patch_h, int patch_w, int pass_mask);
/* Patchmatch::RemoveBackground (patchmatch.cpp:314-360) / MaskBackground (patchmatch_gpu.cu:233-270). */
int pm_remove_background(pm_handle* h, const uint8_t* left, const uint8_t* right, int rows, int cols,
                         float* disp, int patch_h, int patch_w, float factor);
/* PatchmatchGpu::SparseInit(iml, imr, dilate_factor) (patchmatch_gpu.h:110-112, patchmatch_gpu.cu:414-442):
 * GFTT corners, rectified template matching, scatter, (2*(2^f+1)+1)^2 dilation -- all on the device. */
int pm_sparse_init(pm_handle* h, const uint8_t* left, const uint8_t* right, int rows, int cols, int dilate_factor,
                   float* seed);
/* cv::cornerSubPix as FeatureDetector::Detect applies it (feature_detector.cpp:110-120): n <= 1024 points of an 8-bit
 * image refined in place with the handle's subpix_winsize / subpix_zerozone / subpix_maxiters / subpix_epsilon. */
int pm_corner_subpix(pm_handle* h, const uint8_t* image, int rows, int cols, float* xs, float* ys, int n);
/* Patchmatch::Initialize(iml, imr, downsample_factor) (stereo_matching/patchmatch.hpp:24, patchmatch.cpp:52-87):
 * as SparseInit with dilation 2*(2^(f-1)+1)+1, then cv::resize(INTER_NEAREST) to (rows/f) x (cols/f) and division by
 * 2^f (not f: SURVEY Q1).  `seed` holds (rows/f) * (cols/f) floats. */
int pm_initialize(pm_handle* h, const uint8_t* left, const uint8_t* right, int rows, int cols, int downsample_factor,
                  float* seed);
/* ForegroundTextureMask(gray, mask, ksize, min_grad, downsize) (src/vehicle/stereo_matching/patchmatch.hpp:11-16,
 * patchmatch.cpp:19-49): 255 where the morphological gradient of the gray image over a (2 k + 1)^2 rectangle,
 * k = ksize / downsize, exceeds min_grad; with downsize > 1 the gradient is taken on the image shrunk with
 * cv::resize(INTER_LINEAR) and the thresholded mask is blown up again the same way (values between 0 and 255 at its
 * edges, as in the reference).  DEVICE images, tightly packed, on the handle's stream.  Nothing in the reference calls
 * this function; it completes stereo_matching/patchmatch.{hpp,cpp}. */
int pm_foreground_texture_mask(pm_handle* h, const uint8_t* d_gray, int rows, int cols, int ksize, double min_grad,
                               int downsize, uint8_t* d_mask);
/* MaskOcclusions (patchmatch_gpu.cu:273-295). */
int pm_mask_occlusions(pm_handle* h, float* disp_l, const float* disp_r, int rows, int cols);

/* ---- one large image row-tiled over several handles / GPUs (BASELINE config 4) ------------------
 * A tile owns `own_rows` consecutive image rows and is handed a BAND = its rows plus halo rows of image
 * data (at least patch_h/2 + 1 where the image continues, so that windows and Sobel gradients of the
 * owned rows see true neighbours).  Everything except the two vertical sweeps of an iteration is local
 * to a tile (rectified stereo only looks along rows).  A vertical sweep carries one value per column
 * across the tile boundary: the caller moves that ROW of disparities between neighbouring tiles with
 * pm_tile_get_row / pm_tile_set_row (hipMemcpyPeer, RCCL send/recv, ...) and repeats the sweep from
 * pm_tile_snapshot's state until no tile's incoming row changes; the fixpoint is the untiled result
 * (driver: ocean-perception_amd/python/tiled.py).  All pointers are DEVICE memory; n = 1 pair. */
typedef struct pm_tile {
  int global_rows; /* rows of the whole image                     */
  int band_row0;   /* image row of the band's first row           */
  int own_row0;    /* image row of the first row this tile owns   */
  int own_rows;    /* rows this tile owns                         */
} pm_tile;
int pm_tile_begin(pm_handle* h, const pm_tile* tile, const uint8_t* d_left_band, const uint8_t* d_right_band,
                  int band_rows, int cols, const float* d_seed_l_band, const float* d_seed_r_band);
int pm_tile_noise(pm_handle* h, int iteration);         /* noise + clamp + cost of iteration `iteration` */
int pm_tile_sweep(pm_handle* h, int iteration, int k);  /* k: 0 row+, 1 col+, 2 row-, 3 col- (owned rows) */
int pm_tile_snapshot(pm_handle* h);                     /* save disparity + cost planes                  */
int pm_tile_restore(pm_handle* h);
/* pm_tile_set_row(pred_image_row, d_row) -- if d_row is not NULL -- followed by pm_tile_snapshot, in ONE launch: what
 * stands in front of every vertical sweep of a band (the neighbour's boundary row as the guess, then the snapshot). */
int pm_tile_presweep(pm_handle* h, int pred_image_row, const float* d_row);
/* Re-sweep after a boundary exchange without leaving the device timeline: d_mask is a DEVICE array of
 * [n_views][cols] ints (plane columns: view 1 in mirrored coordinates, as pm_tile_get_row delivers its rows);
 * pm_tile_restore_cols puts the flagged columns back to the snapshot, pm_tile_sweep_masked runs a vertical sweep
 * (k = 1 or 3) on the flagged columns only.  Columns of a vertical sweep are independent chains, so re-sweeping
 * the changed ones is exact. */
int pm_tile_restore_cols(pm_handle* h, const int* d_mask);
int pm_tile_sweep_masked(pm_handle* h, int iteration, int k, const int* d_mask);
/* one image row of the disparity planes: [n_views][cols] floats */
int pm_tile_get_row(pm_handle* h, int image_row, float* d_dst);
int pm_tile_set_row(pm_handle* h, int image_row, const float* d_src);
/* One exchange round of vertical sweep k (1 or 3) in two launches instead of five: d_incoming is the neighbour's
 * boundary row as it stands now ([n_views][cols] floats; any device-readable address: the place the neighbour published
 * it, also in a peer device's memory when peer access is enabled, or a local copy), d_used the row the last sweep
 * used.  The columns where they differ are written to d_mask ([n_views][cols] ints), put back to the snapshot and
 * re-swept with the incoming value stored at image row pred_image_row (a row of the neighbour, not one of the band's
 * own); d_used_next (another buffer than d_used; may be d_incoming itself) receives the incoming row for the next
 * round's comparison.  = compare + pm_tile_restore_cols + pm_tile_set_row + pm_tile_sweep_masked. */
int pm_tile_exchange_round(pm_handle* h, int iteration, int k, int pred_image_row, const float* d_incoming,
                           const float* d_used, float* d_used_next, int* d_mask);
/* *d_flag |= 1 (device int) if image row `image_row` of the disparity planes differs from d_ref_row ([n_views][cols]):
 * "did my boundary row move after the last row I sent?" in one launch. */
int pm_tile_row_moved(pm_handle* h, int image_row, const float* d_ref_row, int* d_flag);
int pm_tile_background(pm_handle* h);
/* cross-check + un-mirror; writes the owned rows only: [own_rows][cols] each */
int pm_tile_finish(pm_handle* h, float* d_disp_l_own, float* d_disp_r_own);

/* ---- the row-tiled driver over the pm_tile_* stages: ONE process, n band handles on up to n devices ----------------
 * The C / C++ caller's form of BASELINE config 4 (the multi-process variant over torch.distributed / RCCL is
 * ocean-perception_amd/python/tiled.py; both run the same protocol).  Band k owns rows split as evenly as possible and
 * works on its rows plus patch_h/2 + 1 halo rows of image data.  Noise / cost, the horizontal sweeps, the background
 * mask and the cross-check are local to a band.  A vertical sweep carries one value per column across a band boundary
 * (one row of disparities, read by the receiving band behind an event of the sender's stream -- no host synchronisation,
 * no RCCL inside one process).  Two schedules give the same maps (pm_tiled_set_schedule below):
 *   PIPELINED (default)  the bands sweep in order along the sweep direction, each continuing from its predecessor's final
 *                        row: exact by construction, `rounds` is ignored;
 *   SPECULATIVE          every band sweeps with the neighbour's OLD boundary row, the new boundary rows travel one hop,
 *                        and a band re-sweeps exactly the columns whose incoming value changed, `rounds` times per sweep
 *                        (band k is final after round k + 1).  Whether that sufficed is one device flag per band ("my
 *                        boundary row still moved after the last row I sent"), read once per Match; only if it is set is
 *                        the Match repeated with n_bands - 1 rounds, which is always enough.
 * The result equals the untiled Match() bit for bit (tests/test_cpp_tiled.py, 4096x2160).
 * Replaces, for one large image, PatchmatchGpu::Match(const Image1b&, const Image1b&, Image1f&, Image1f&)
 * (src/vehicle/patchmatch_gpu/patchmatch_gpu.h:99-102). */
typedef struct pm_tiled_plan pm_tiled_plan;
typedef struct pm_tiled_info {
  int rounds_used;  /* exchange rounds per vertical sweep of the result that was returned (0: pipelined) */
  int repeated;     /* 1: a boundary row still moved after `rounds` rounds and the Match was repeated  */
  int exchanges;    /* boundary rows that travelled between bands (all bands, both attempts)           */
} pm_tiled_info;
/* rows a band handle must be planned for (pm_create max_rows) when `global_rows` are split over n_bands */
int pm_tiled_band_rows(const pm_params* params, int global_rows, int n_bands);
/* bands[k]: handles created with the SAME params, max_rows >= pm_tiled_band_rows(), max_cols >= cols, max_batch >= 1,
 * on any devices of this process (several bands may share a device).  The plan owns the per-band device buffers. */
int pm_tiled_create(pm_handle* const* bands, int n_bands, int rows, int cols, pm_tiled_plan** out);
void pm_tiled_destroy(pm_tiled_plan* plan);
/* host buffers as in pm_match_u8 (seed maps may be NULL; the device seeder works on whole images and is not available
 * here); rounds (PM_TILED_SCHEDULE_SPECULATIVE only) = exchange rounds per vertical sweep: negative (recommended) or >= n_bands - 1 = n_bands - 1 rounds, which
 * are always enough (band k is final after round k + 1) -- a round in which nothing changed is three empty launches per
 * band, a repeated Match costs a whole Match (measured at 4096x2160 / 8 bands: 49 ms with 7 rounds, 88 ms with 2 rounds
 * and the repeat they always end in, tools/tiled_rounds.py); fewer rounds only pay when values rarely cross bands */
int pm_tiled_match_u8(pm_tiled_plan* plan, const uint8_t* left, const uint8_t* right, size_t image_step,
                      const float* seed_l, const float* seed_r, size_t seed_step, float* disp_l, float* disp_r,
                      size_t disp_step, int rounds, pm_tiled_info* info);
/* The same in three steps, for callers that keep the pair resident (a timed region without PCIe, several runs of one
 * pair): upload = band images and seed maps to the bands' devices (stream-ordered); run = the Match on the resident
 * pair (returns after the one flag read that decides about a repeat; results stay in the bands' output buffers);
 * download = owned rows into the caller's maps (waits for every band). */
int pm_tiled_upload_u8(pm_tiled_plan* plan, const uint8_t* left, const uint8_t* right, size_t image_step,
                       const float* seed_l, const float* seed_r, size_t seed_step);
int pm_tiled_run(pm_tiled_plan* plan, int rounds, pm_tiled_info* info);
int pm_tiled_download(pm_tiled_plan* plan, float* disp_l, float* disp_r, size_t disp_step);
const char* pm_tiled_last_error(const pm_tiled_plan* plan);
/* How the bands are spread: neighbouring bands that live on different devices (their boundary rows cross devices), and how
 * many of those boundaries got direct peer access (hipDeviceEnablePeerAccess both ways; the rest is staged by the
 * runtime).  All bands on one device: 0 and 0 -- no peer access is requested at all. */
int pm_tiled_topology(const pm_tiled_plan* plan, int* device_boundaries, int* peer_links);
/* How a band gets at its neighbour's boundary row in an exchange round.
 *   AUTO (default)  neighbours on ONE device: the band's kernel reads the row where the neighbour published it;
 *                   neighbours on different devices: hipMemcpyPeerAsync on the receiving band's stream first
 *   COPY            always the copy
 *   DIRECT          as AUTO, and across devices too wherever peer access could be enabled (pm_tiled_topology): the
 *                   receiving band's kernel reads 32 KB across the link -- one copy less per round.  Kernel reads of peer
 *                   memory have not been timed on a multi-GPU node yet, which is why this is not what AUTO does.
 * Results are identical in every mode. */
typedef enum pm_tiled_exchange { PM_TILED_EXCHANGE_AUTO = 0, PM_TILED_EXCHANGE_COPY = 1, PM_TILED_EXCHANGE_DIRECT = 2 } pm_tiled_exchange;
int pm_tiled_set_exchange(pm_tiled_plan* plan, int mode);
/* How a vertical sweep crosses the band boundaries.
 *   PIPELINED (default since round 6)  the bands sweep IN ORDER along the sweep direction: a band stores its predecessor's
 *                          final row in front of its chains, sweeps once, publishes its own last row.  Nothing is guessed,
 *                          so there is no snapshot, no mask, no re-sweep and no repeat (`rounds` is ignored;
 *                          pm_tiled_info.rounds_used = 0); during a vertical sweep the bands take turns, everything that
 *                          does not cross a boundary overlaps -- eight bands on ONE device: 28.5 ms per 4096x2160 frame,
 *                          less than the untiled frame's 29.5 (neighbouring bands' phases fill each other's tails).
 *   SPECULATIVE            every band sweeps at once with its neighbour's OLD boundary row, the new rows travel one hop per
 *                          round, a band re-sweeps the columns whose incoming value changed (snapshot, mask, `rounds`,
 *                          the repeat rule: see above).  All bands work in every round -- and re-sweep: 35.7 ms on one
 *                          device.  Kept for boxes where several GPUs make the parallel rounds pay.
 * Same maps either way (the sequential sweep). */
typedef enum pm_tiled_schedule { PM_TILED_SCHEDULE_SPECULATIVE = 0, PM_TILED_SCHEDULE_PIPELINED = 1 } pm_tiled_schedule;
int pm_tiled_set_schedule(pm_tiled_plan* plan, int schedule);

/* ---- PM_MODE_PLANES stage by stage (device pointers; state stays resident in the handle) ------------------
 * pm_match_u8 / pm_match_batch_u8 / pm_submit_u8 / pm_match_device run the whole schedule
 *   begin; for it < patchmatch_iters: red, black (both views), then per view v = 0, 1: view propagation into v
 *   followed by v's refinement (one fused launch); finish
 * when params.mode == PM_MODE_PLANES.  These entry points run one stage each (tests, tools). */
enum { PM_PL_SPATIAL = 1, PM_PL_VIEW = 2, PM_PL_REFINE = 3, PM_PL_VIEW_REFINE = 4 };
/* prep + random plane initialisation (+ its cost) of n pairs; seeds as in pm_match_device */
int pm_planes_begin(pm_handle* h, int n, const uint8_t* d_left, const uint8_t* d_right, int rows, int cols,
                    const float* d_seed_l, const float* d_seed_r);
/* stage = PM_PL_SPATIAL: arg = colour + 2 * iteration (colour 0 red: x + y even, 1 black), both views; the iteration
 *                        only matters with PM_PL_NEIGH_TWO (which pair of neighbours);
 *         PM_PL_VIEW:    arg = the view that receives candidates from the other one;
 *         PM_PL_REFINE:  arg = iteration (selects noise_amp[arg] and the random numbers), both views;
 *         PM_PL_VIEW_REFINE: arg = iteration * 2 + view: PM_PL_VIEW for that view, then its PM_PL_REFINE, fused */
int pm_planes_step(pm_handle* h, int stage, int arg);
/* planes of (pair, view) as four tightly packed rows x cols float maps a, b, z, cost: HOST buffer of
 * 4 * rows * cols floats (view 1 is in mirrored coordinates, as the engine holds it) */
int pm_planes_read(pm_handle* h, int pair, int view, float* planes);
int pm_planes_write(pm_handle* h, int pair, int view, const float* planes);
/* disparity maps (right map un-mirrored) + consistency mask: [n][rows][cols] device floats */
int pm_planes_finish(pm_handle* h, float* d_disp_l, float* d_disp_r);

/* ---- per-kernel timing (hipEvents on the handle's stream) ---------------------------------- */

enum {
  PM_K_PREP = 0,     /* u8 -> mirrored copies + Sobel magnitude        */
  PM_K_SEED = 1,     /* seed upload / mirror                           */
  PM_K_NOISE = 2,    /* foreground noise + clamp + cost of current d   */
  PM_K_SWEEP_ROW = 3,
  PM_K_SWEEP_COL = 4,
  PM_K_BACKGROUND = 5,
  PM_K_FINALIZE = 6, /* un-mirror + MaskOcclusions                     */
  PM_K_PL_INIT = 7,    /* PM_MODE_PLANES: random plane initialisation + cost */
  PM_K_PL_SPATIAL = 8, /* red / black spatial propagation                    */
  PM_K_PL_VIEW = 9,    /* view propagation                                   */
  PM_K_PL_REFINE = 10, /* random plane refinement                            */
  PM_K_PL_VIEW_REFINE = 11, /* view propagation + refinement of one view in one launch (the default schedule) */
  PM_K_COUNT = 12
};
typedef struct pm_profile {
  uint64_t launches[PM_K_COUNT];
  double total_ms[PM_K_COUNT]; /* sum of hipEventElapsedTime over those launches */
} pm_profile;
/* on != 0: bracket every kernel with events (adds host overhead; use for roofline legs only). */
int pm_profile_enable(pm_handle* h, int on);
/* Waits for the stream, then accumulates pending events into *out and resets the counters. */
int pm_profile_read(pm_handle* h, pm_profile* out);
const char* pm_kernel_name(int kernel_class);
/* Work counters of the run engine since the last call (then reset): for row sweeps [0..3] and column
 * sweeps [4..7]: wave steps of the speculative round, wave steps of the fix-up rounds, fix-up rounds
 * summed over chains, positions swept. */
int pm_debug_counters(pm_handle* h, uint64_t out[8]);
/* Counting costs one same-address atomic set per wavefront (it serialises large grids): off by default. */
int pm_debug_counters_enable(pm_handle* h, int on);

#ifdef __cplusplus
}
#endif
#endif /* PM_PATCHMATCH_H_ */
