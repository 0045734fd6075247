/*
 * pm_oracle.h -- CPU restatement of the reference PatchMatch stereo path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: it may be
 * imported / linked / executed only by tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py, and there only as the checker or the reported
 * CPU baseline.  The product path (ocean-perception_amd/, include/) never calls it.
 *
 * PARITY UNPINNED.  The reference's tests for this path hold no assertions and no
 * golden vectors (test/stereo_matching/patchmatch_test.cpp, patchmatch_gpu_test.cpp:
 * imshow demos only), the reference cannot be compiled here (needs OpenCV 3.4.0 EXACT,
 * glog, Eigen, Boost, CUDA -- none present), and every arithmetic primitive of the path
 * lives in OpenCV 3.4.0 (CMakeLists.txt:28), which is not vendored.  This file restates
 * the reference's own code line by line and OpenCV 3.4's published algorithms
 * (getRectSubPix, Sobel, dilate, RNG, mean, saturate_cast) as documented per function;
 * it is pinned only by hand-computed known-answer tests and by an independent numpy
 * restatement (tests/pyref.py).  Floating point is evaluated without FMA contraction
 * (-ffp-contract=off), one IEEE-754 binary32 rounding per operation.
 *
 * Two semantics are restated:
 *   PMO_SEM_CPU  src/vehicle/stereo_matching/patchmatch.cpp + the cost functor and
 *                recipe of test/stereo_matching/patchmatch_test.cpp  (the parity target)
 *   PMO_SEM_GPU  src/vehicle/patchmatch_gpu/patchmatch_gpu.cu kernels, executed
 *                race-free as ONE stripe per row/column (the reference's 16 overlapping
 *                stripes race).
 */
#ifndef PM_ORACLE_H_
#define PM_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PMO_MAX_ITERS 16

enum { PMO_SEM_CPU = 0, PMO_SEM_GPU = 1 };

/* One rectified view pair, all planes contiguous rows*cols. */
typedef struct pmo_images {
  int rows, cols;
  const uint8_t* il;  /* reference ("left") image  */
  const uint8_t* ir;  /* target ("right") image    */
  const float* gl;    /* gradient magnitude of il  */
  const float* gr;    /* gradient magnitude of ir  */
} pmo_images;

/* Constants of L1GradientCostFunction, test/stereo_matching/patchmatch_test.cpp:30-45. */
typedef struct pmo_functor {
  float alpha;      /* 0.7 */
  float tau_color;  /* 50  */
  float tau_grad;   /* 20  */
} pmo_functor;

typedef struct pmo_params {
  int semantics;                   /* PMO_SEM_CPU / PMO_SEM_GPU */
  int n_iters;                     /* PatchmatchGpu::Params::patchmatch_iters */
  float noise_amp[PMO_MAX_ITERS];  /* per-iteration noise amplitude */
  int patch_w[PMO_MAX_ITERS];      /* per-iteration window (SEM_CPU); SEM_GPU: always the 5-tap 3x3 */
  int patch_h[PMO_MAX_ITERS];
  int bg_patch_w, bg_patch_h;      /* RemoveBackground window (SEM_CPU) */
  float bg_factor;                 /* SEM_CPU: win_by_factor; SEM_GPU: cost_improve_factor */
  float cost_alpha;                /* SEM_GPU: PatchmatchGpu::Params::cost_alpha */
  pmo_functor functor;             /* SEM_CPU */
  uint64_t noise_seed;             /* 123 */
  int left_right_check;            /* run the right view + MaskOcclusions */
  int literal;                     /* SEM_CPU: 1 = via getRectSubPix patches + functor (the reference's
                                      call structure), 0 = fused direct formula (same results) */
  int nthreads;                    /* rows/columns of a sweep are independent -> exact with any count */
} pmo_params;

void pmo_params_default(pmo_params* p, int semantics);

/* ---- OpenCV 3.4 primitives (restated) ------------------------------------------------ */

/* cv::RNG(seed) then RNG::fill(mat, UNIFORM, lo, hi) on a continuous CV_32F matrix. */
void pmo_rng_fill_uniform(float* dst, size_t n, double lo, double hi, uint64_t seed);
/* first n raw 32-bit outputs of cv::RNG(seed) (KATs). */
void pmo_rng_raw(uint32_t* dst, size_t n, uint64_t seed);

/* ComputeGradient (patchmatch_test.cpp:48-64) == GradientMagnitude (patchmatch_gpu.cu:307-319). */
void pmo_gradient_magnitude(const uint8_t* im, int rows, int cols, float* g);

/* cv::dilate with a (2k+1)x(2k+1) MORPH_RECT, anchor (k,k). */
void pmo_dilate_rect(const float* src, float* dst, int rows, int cols, int k);

/* cv::getRectSubPix, 8u->8u and 32f->32f, single channel. */
void pmo_get_rect_subpix_u8(const uint8_t* src, int rows, int cols, int pw, int ph,
                            float cx, float cy, uint8_t* dst);
void pmo_get_rect_subpix_f32(const float* src, int rows, int cols, int pw, int ph,
                             float cx, float cy, float* dst);

/* cv::resize(src, dst, Size(dcols, drows)) with the default INTER_LINEAR on 8-bit single-channel images -- what both
 * reference PatchMatch tests apply to their inputs (test/stereo_matching/patchmatch_test.cpp:131-133,
 * patchmatch_gpu_test.cpp:62-64: il.size() / 2).  OpenCV 3.4 imgproc/resize.cpp as remembered (parity unpinned):
 *   - an exact integer shrink by 2 in both axes is routed to the INTER_AREA fast path ("INTER_AREA (fast) also is equal
 *     to INTER_LINEAR"): dst = (a + b + c + d + 2) >> 2 over the 2x2 block;
 *   - otherwise: source coordinate fx = (dx + 0.5) * scale - 0.5, clamped at the borders, 11-bit fixed-point weights
 *     cvRound(w * 2048), horizontal pass in int, vertical pass ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2 >> 2. */
void pmo_resize_linear_u8(const uint8_t* src, int rows, int cols, uint8_t* dst, int drows, int dcols);
/* ForegroundTextureMask (stereo_matching/patchmatch.cpp:19-49); 0, or -1 where the reference CHECK-fails. */
int pmo_foreground_texture_mask(const uint8_t* gray, int rows, int cols, int ksize, double min_grad, int downsize,
                                uint8_t* mask);
void pmo_flip_h_u8(const uint8_t* src, uint8_t* dst, int rows, int cols);
void pmo_flip_h_f32(const float* src, float* dst, int rows, int cols);

/* ---- SEM_CPU: stereo_matching/patchmatch.cpp ------------------------------------------ */

/* L1GradientCostFunction on extracted patches (n = pw*ph); gl/gr are the f32 patches,
 * saturate-cast to u8 inside as the reference's implicit Image1f->Image1b conversion does. */
float pmo_cpu_functor(const uint8_t* pl, const uint8_t* pr, const float* gl, const float* gr,
                      int n, const pmo_functor* f);
/* cost of disparity d at integer pixel (x,y): literal (patches) and direct (fused). */
float pmo_cpu_cost_literal(const pmo_images* im, int pw, int ph, float x, float y, float d,
                           const pmo_functor* f);
float pmo_cpu_cost_direct(const pmo_images* im, int pw, int ph, int x, int y, float d,
                          const pmo_functor* f);

/* Patchmatch::AddNoise (patchmatch.cpp:143-155); mask may be NULL. */
void pmo_cpu_add_noise(float* disp, int rows, int cols, float amount, const uint8_t* mask,
                       uint64_t seed);
/* Patchmatch::Propagate (patchmatch.cpp:248-311); pass_mask bit0..3 = passes A..D. */
void pmo_cpu_propagate(const pmo_images* im, float* disp, int ph, int pw, const pmo_functor* f,
                       int pass_mask, int literal, int nthreads);
/* Patchmatch::RemoveBackground (patchmatch.cpp:314-360). */
void pmo_cpu_remove_background(const pmo_images* im, float* disp, int ph, int pw,
                               const pmo_functor* f, float win_by_factor, int literal,
                               int nthreads);

/* ---- SEM_GPU: patchmatch_gpu/patchmatch_gpu.cu ----------------------------------------- */

float pmo_gpu_get_subpixel(const float* im, int rows, int cols, float row, float col);
float pmo_gpu_cost5(const pmo_images* im, int yl, int xl, float yr, float xr, float alpha);
void pmo_gpu_add_foreground_noise(float* disp, const float* unit_noise, size_t n, float scale);
void pmo_gpu_propagate_row(const pmo_images* im, float* disp, int direction, int patch_size,
                           float alpha, int nthreads);
void pmo_gpu_propagate_col(const pmo_images* im, float* disp, int direction, int patch_size,
                           float alpha, int nthreads);
void pmo_gpu_mask_background(const pmo_images* im, float* disp, int patch_size, float alpha,
                             float improve_factor, int nthreads);
void pmo_gpu_mask_occlusions(float* displ, const float* dispr, int rows, int cols);

/* ---- pipelines -------------------------------------------------------------------------- */

/* One view: iterations {noise, 4 sweeps} + background mask, on `disp` (seed in, result out).
 * SEM_GPU == PatchmatchGpu::Match(GpuMat...) (patchmatch_gpu.cu:379-411);
 * SEM_CPU == the recipe of patchmatch_test.cpp:173-183 with p's schedule. */
void pmo_match_view(const pmo_params* p, const pmo_images* im, float* disp);

/* Both views + cross-check, structure of PatchmatchGpu::Match (patchmatch_gpu.cu:331-376).
 * seed_l / seed_r are given in left / right image coordinates (seed_r is mirrored
 * internally as the reference computes it on the mirrored pair); disp_r comes back in
 * right-image coordinates.  With left_right_check == 0 only disp_l is produced. */
void pmo_match(const pmo_params* p, const uint8_t* left, const uint8_t* right, int rows, int cols,
               const float* seed_l, const float* seed_r, float* disp_l, float* disp_r);

/* ---- sparse seeding: FeatureDetector::Detect + StereoMatcher::MatchRectified + SparseInit ----
 * The reference delegates to cv::goodFeaturesToTrack and cv::matchTemplate (OpenCV 3.4), whose float
 * pipelines (scaled Sobel, DFT-based correlation) cannot be reproduced bit for bit without OpenCV.  These
 * functions restate the ALGORITHMS (min-eigenvalue corners, quality threshold, 3x3 non-maximum suppression,
 * greedy minimum distance; normalised squared difference, first minimum) on exact integer sums; they are
 * this build's definition of the seeder, validated functionally (parity unpinned, as for the rest). */
typedef struct pmo_seed_params {
  int max_features;         /* 200   feature_detector.hpp:28 max_features_per_frame */
  int min_distance;         /* 20    :31 min_distance_btw_tracked_and_detected_features */
  double quality_level;     /* 0.01  :32 gftt_quality_level */
  int block_size;           /* 5     :33 gftt_block_size */
  int templ_cols;           /* 31    stereo_matcher.hpp:21 */
  int templ_rows;           /* 11    :22 */
  int max_disp;             /* 128   :23 */
  double max_matching_cost; /* 0.15  :24 */
  int use_harris;           /* 0     feature_detector.hpp:34 gftt_use_harris_corner_detector */
  double harris_k;          /* 0.04  :35 gftt_k */
  int subpixel_corners;     /* 0     :39 cv::cornerSubPix on the detected corners (feature_detector.cpp:110-120) */
  int subpix_winsize;       /* 10    :40 */
  int subpix_zerozone;      /* -1    :41 */
  int subpix_maxiters;      /* 10    :42 */
  float subpix_epsilon;     /* 0.01  :43 */
  int subpixel_refinement;  /* 0     stereo_matcher.hpp:26 cornerSubPix on the match (stereo_matcher.cpp:94-103) */
} pmo_seed_params;
/* cv::cornerSubPix on an 8-bit image (pm_seed_oracle.c); xs / ys in and out */
void pmo_corner_subpix(const uint8_t* img, int rows, int cols, float* xs, float* ys, int n, int win, int zero_zone,
                       int max_iters, double eps);
/* its window mask, (2 win + 1)^2 floats */
void pmo_subpix_mask(int win, int zero_zone, float* mask);

void pmo_seed_params_default(pmo_seed_params* p);
/* min-eigenvalue response (cv::cornerMinEigenVal, unscaled): eig = (a+c) - sqrt((a-c)^2 + b^2) in binary32 with
 * a = Sxx/2, b = Sxy, c = Syy/2, S** = block_size^2 box sums (REFLECT_101) of the Sobel products. */
void pmo_min_eig_map(const uint8_t* img, int rows, int cols, int block_size, float* eig);
/* the same with the Harris response (cv::cornerHarris, unscaled): (float)(a*c - b*b - k*(a+c)*(a+c)) with
 * a = Sxx, b = Sxy, c = Syy as binary32 values and a binary64 k, as calcHarris computes it */
void pmo_corner_response_map(const uint8_t* img, int rows, int cols, int block_size, int use_harris, double harris_k,
                             float* eig);
/* goodFeaturesToTrack as FeatureDetector::Detect configures it (feature_detector.cpp:44-57,89-122);
 * returns the number of corners written to xs/ys (strongest first). */
int pmo_gftt_detect(const uint8_t* img, int rows, int cols, const pmo_seed_params* p, int* xs, int* ys, int cap);
/* StereoMatcher::MatchRectified (stereo_matcher.cpp:22-116): disparity of one keypoint or -1. */
double pmo_match_rectified(const uint8_t* left, const uint8_t* right, int rows, int cols, float kx, float ky,
                           const pmo_seed_params* p);
/* PatchmatchGpu::SparseInit (patchmatch_gpu.cu:414-442): seed map, 0 = unknown. */
void pmo_sparse_init(const uint8_t* left, const uint8_t* right, int rows, int cols, int dilate_factor,
                     const pmo_seed_params* p, float* seed);

/* Patchmatch::Initialize (stereo_matching/patchmatch.cpp:52-87); out: (rows/f) x (cols/f) floats. */
void pmo_cpu_initialize(const uint8_t* left, const uint8_t* right, int rows, int cols, int downsample_factor,
                        const pmo_seed_params* p, float* out);

#ifdef __cplusplus
}
#endif
#endif  /* PM_ORACLE_H_ */
