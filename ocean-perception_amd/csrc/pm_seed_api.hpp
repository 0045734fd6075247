// pm_seed_api.hpp -- what the engine sees of the device seeder (pm_seed.hpp holds the kernels, pm_seed.hip is their
// translation unit): parameters, the scratch a handle owns, and the enqueue-only entry points.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace pm {

struct SeedParams {
  int max_features, min_distance, block_size;
  int templ_cols, templ_rows, max_disp;
  double quality_level, max_matching_cost;
  int use_harris;   // corner response det(M) - k trace(M)^2 (cv::cornerHarris) instead of the smaller eigenvalue
  double harris_k;
  // cv::cornerSubPix on the detected corners (feature_detector.cpp:110-120) / on the match (stereo_matcher.cpp:94-103)
  int subpixel_corners, subpix_winsize, subpix_zerozone, subpix_maxiters;
  float subpix_epsilon;
  int subpixel_refinement;
};
constexpr int kSubpixMaxWin = 15;       // largest half window of cornerSubPix the scratch is sized for
constexpr int kSubpixMatchWin = 10;     // StereoMatcher's fixed window (stereo_matcher.cpp:97)

constexpr int kSeedMaxFeatures = 1024;  // capacity of the accepted-corner list
constexpr int kSeedCounters = 8;        // SeedScratch::counters
constexpr int kSubpixMaskStride = 1024;  // floats between the two masks of SeedScratch::sp_mask (31 * 31 = 961)

// Scratch owned by the handle (sized for max_rows x max_cols).
struct SeedScratch {
  float* eig;                // [rows][pitch]
  unsigned long long* keys;  // [cap] candidates, then sorted
  unsigned long long* keys_sorted;
  unsigned* counters;        // [kSeedCounters]: [0] = max response bits, [1] = candidate count, [2] = accepted
                             // count, [3] = grid overflow flag
  int* kp_xy;                // [kSeedMaxFeatures][2]
  float* kp_d;               // [kSeedMaxFeatures] matched disparity of a corner, < 0 = no match
  float* kp_f;               // [kSeedMaxFeatures][2] sub-pixel corner positions (subpixel_corners)
  float* sp_buf;             // cornerSubPix neighbourhoods, [(2 * kSubpixMaxWin + 3)^2][kSeedMaxFeatures]
  float* sp_mask;            // window masks: detector's [(2 w + 1)^2] at 0, matcher's [21 * 21] at kSubpixMaskStride
  int sp_mask_win, sp_mask_zero;  // what the detector's mask was built for
  void* sort_tmp;
  size_t sort_tmp_bytes;
  int cap;
  // the selection kernels leave counters[0], [1] and [3] at zero for the next map; false after an allocation or an
  // enqueue that did not get as far as the selection: the next map then starts with a memset
  mutable bool counters_clean;
};

// Allocates the scratch for planes of `plane_elems` pixels (rows x pitch of the plan).  Every pixel can be a candidate:
// the 3x3 test is not strict, so plateaus of EQUAL responses (periodic images) pass whole; a capacity of a quarter of the
// pixels dropped candidates there in whatever order the atomics fell.
hipError_t seed_scratch_alloc(SeedScratch& sc, size_t plane_elems, hipStream_t stream);
void seed_scratch_free(SeedScratch& sc);
// the masks and the neighbourhood buffer of cv::cornerSubPix, for handles whose parameters ask for it (synchronises
// the stream when it has to upload the masks: call outside captures)
hipError_t seed_subpix_prepare(SeedScratch& sc, const SeedParams& sp, hipStream_t stream);

// cv::cornerSubPix with the detector's window parameters of `sp` on n device points (needs seed_subpix_prepare)
hipError_t seed_corner_subpix(const SeedScratch& sc, const SeedParams& sp, const uint8_t* img, int rows, int cols, int pitch,
                              float* d_xs, float* d_ys, int n, hipStream_t stream);
// The launch sequence of one map comes in kSeedStages parts (bit i of `stages` = part i; all by default).  Parts of one
// map go onto one stream in order; a caller with several maps on several streams may interleave them part by part.
constexpr int kSeedStages = 5;
constexpr unsigned kSeedAllStages = (1u << kSeedStages) - 1u;
// `out`: a row-major map with out_pitch elements per row, or -- out_pitch < 0 -- one of the engine's state planes (four
// rows interleaved, pm_device.hpp::state_at, pitch -out_pitch).
// PatchmatchGpu::SparseInit(iml, imr, f)  (patchmatch_gpu.cu:414-442): dilation half-width 2^f + 1, map at image size
hipError_t seed_sparse_init(const SeedScratch& sc, const SeedParams& sp, const uint8_t* left, const uint8_t* right,
                            int rows, int cols, int pitch, int dilate_factor, float* out, int out_pitch,
                            hipStream_t stream, unsigned stages = kSeedAllStages);
// Patchmatch::Initialize(iml, imr, f)  (patchmatch.cpp:60-84): half-width 2^(f-1) + 1, map of size / f, scaled by 2^-f;
// downsample_factor >= 1
hipError_t seed_initialize(const SeedScratch& sc, const SeedParams& sp, const uint8_t* left, const uint8_t* right,
                           int rows, int cols, int pitch, int downsample_factor, float* out, int out_pitch,
                           hipStream_t stream, unsigned stages = kSeedAllStages);

}  // namespace pm
