// patchmatch_gpu.hpp -- host-side C++ mirror of the reference's bm::pm::PatchmatchGpu
// (src/vehicle/patchmatch_gpu/patchmatch_gpu.h:77-124) on top of the C ABI in pm/patchmatch.h.
//
// Same namespace, class name, Params field names / defaults and Match() signatures, so a caller
// written against the reference (test/stereo_matching/patchmatch_gpu_test.cpp:68-88) compiles
// against this header after swapping the include.  No HIP header is included here: host code is
// plain C++17 and reaches the device only through the C ABI.
//
// Image types: the reference's Image1b / Image1f are cv::Mat_<uchar> / cv::Mat_<float>
// (src/vehicle/vision_core/cv_types.hpp:8-12).  OpenCV is not a dependency of this library; the
// minimal bm::core::Image<T> below carries the members the path uses (rows, cols, step, ptr(),
// at(), empty(), create()).  When <opencv2/core.hpp> is available, the template overloads of
// Match() accept cv::Mat_ directly (anything with rows/cols/step/data).
#pragma once

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "pm/patchmatch.h"

namespace bm {
namespace core {

template <typename T>
class Image {
 public:
  int rows = 0, cols = 0;
  size_t step = 0;  // bytes per row, as cv::Mat::step

  Image() = default;
  Image(int rows_, int cols_, T value = T()) { create(rows_, cols_, value); }

  void create(int rows_, int cols_, T value = T()) {
    rows = rows_;
    cols = cols_;
    step = sizeof(T) * (size_t)cols_;
    storage_.assign((size_t)rows_ * cols_, value);
  }
  bool empty() const { return storage_.empty(); }
  T* ptr(int r = 0) { return storage_.data() + (size_t)r * cols; }
  const T* ptr(int r = 0) const { return storage_.data() + (size_t)r * cols; }
  T& at(int r, int c) { return storage_[(size_t)r * cols + c]; }
  const T& at(int r, int c) const { return storage_[(size_t)r * cols + c]; }
  T* data() { return storage_.data(); }
  const T* data() const { return storage_.data(); }

 private:
  std::vector<T> storage_;
};

typedef Image<uint8_t> Image1b;
typedef Image<float> Image1f;

}  // namespace core

namespace ft {

// ft::FeatureDetector::Params (src/vehicle/feature_tracking/feature_detector.hpp:19-47) and
// ft::StereoMatcher::Params (src/vehicle/feature_tracking/stereo_matcher.hpp:15-29): the configuration of the
// sparse seeder, nested in their classes exactly as the reference nests them (PatchmatchGpu::Params holds a
// `ft::FeatureDetector::Params detector_params` and a `ft::StereoMatcher::Params matcher_params`,
// patchmatch_gpu.h:82-83).  The classes themselves carry no behaviour here: detection and matching run on the
// device inside the engine (pm_sparse_init).
class FeatureDetector final {
 public:
  struct Params final {
    int max_features_per_frame = 200;
    int min_distance_btw_tracked_and_detected_features = 20;
    double gftt_quality_level = 0.01;
    int gftt_block_size = 5;
    bool gftt_use_harris_corner_detector = false;
    double gftt_k = 0.04;
    bool subpixel_corners = false;
    int subpix_winsize = 10;
    int subpix_zerozone = -1;
    int subpix_maxiters = 10;
    float subpix_epsilon = 0.01f;
  };
};

class StereoMatcher final {
 public:
  struct Params final {
    int templ_cols = 31;
    int templ_rows = 11;
    int max_disp = 128;
    double max_matching_cost = 0.15;
    bool bidirectional = false;
    bool subpixel_refinement = false;
  };
};

// round-1 spellings
typedef FeatureDetector::Params FeatureDetectorParams;
typedef StereoMatcher::Params StereoMatcherParams;

}  // namespace ft

namespace pm {

using core::Image1b;
using core::Image1f;

class PatchmatchGpu final {
 public:
  struct Params final {
    // --- reference fields (patchmatch_gpu.h:82-88) ---
    ft::FeatureDetector::Params detector_params;
    ft::StereoMatcher::Params matcher_params;
    float cost_alpha = 0.9f;
    int patchmatch_iters = 3;
    int init_dilate_factor = 4;
    float cost_improve_factor = 0.8f;

    // --- engine fields (pm_params) ---
    int semantics = PM_SEM_GPU;  // which reference code is reproduced; PM_SEM_CPU = stereo_matching/patchmatch.cpp
    int engine = PM_ENGINE_AUTO;
    int patch_size = 3;          // PM_SEM_CPU window side for every iteration and the background mask
    int mode = PM_MODE_SCALAR;   // PM_MODE_PLANES: slanted-plane state (include/pm/patchmatch.h)
    int state_dtype = PM_STATE_F32;
    bool left_right_check = true;
    int device = 0;
    int max_rows = 0, max_cols = 0;  // 0: plan on the first Match()
    int max_batch = 1;
    // the priority class of the object's GPU streams (pm_stream_priority): HIGH keeps its two view streams off the
    // hardware queues of the host application's default-class streams; an application that must keep its own kernels
    // in front of the matcher's sets PM_STREAM_PRIO_DEFAULT or PM_STREAM_PRIO_LOW
    int stream_priority = PM_STREAM_PRIO_HIGH;

    // Fills a pm_params from these fields.
    pm_params ToC() const;
  };

  PatchmatchGpu(const PatchmatchGpu&) = delete;
  PatchmatchGpu& operator=(const PatchmatchGpu&) = delete;

  explicit PatchmatchGpu(const Params& params);
  ~PatchmatchGpu();

  // patchmatch_gpu.h:99-102.  Like the reference, Match() seeds itself with SparseInit on both views
  // (on the device); seed maps set through SetSeeds() take precedence.
  void Match(const Image1b& iml, const Image1b& imr, Image1f& disp, Image1f& dispr);

  // patchmatch_gpu.h:104-108 widened to a device-resident pair: raw device pointers to tightly
  // packed planes.  Gl/Gr of the reference are computed inside the engine.
  void Match(const uint8_t* d_iml, const uint8_t* d_imr, int rows, int cols, const float* d_seed_l,
             const float* d_seed_r, float* d_disp, float* d_dispr);

  // patchmatch_gpu.h:104-108 as it stands: one view, caller-supplied gradients, `disp` = sparse-init map in,
  // result out.  GpuImage1f is what the call needs of a cu::GpuMat (CV_32F): device pointer, size, step in bytes.
  struct GpuImage1f {
    float* data = nullptr;
    int rows = 0, cols = 0;
    size_t step = 0;
  };
  void Match(const GpuImage1f& iml, const GpuImage1f& imr, const GpuImage1f& Gl, const GpuImage1f& Gr,
             GpuImage1f& disp, void* stream = nullptr);

  // Several pairs of one size in ONE call (BASELINE configs[2]'s per-GPU share; at most Params::max_batch): the engine
  // runs them as pipelines side by side.  From host images the copies overlap the matching of neighbouring chunks:
  // 461-475 pairs/s at 720p (pair-by-pair Match() calls: 363-381); a caller whose pairs are already on the device
  // gets 480-484 through pm_match_device(handle(), n, ...) (round 4).  Every pair seeds itself like Match() does (seed maps set
  // through SetSeeds() are for single pairs: not allowed here).  Results equal Match()'s, pair by pair.
  void MatchBatch(const std::vector<Image1b>& imls, const std::vector<Image1b>& imrs, std::vector<Image1f>& disps,
                  std::vector<Image1f>& disprs);

  // Match() for a sequence of frames (the callback loop of patchmatch_gpu_test.cpp:118-128) with the
  // copies off the critical path: Submit() returns once the pair is packed and enqueued, Collect() waits
  // for the oldest submitted pair.  At most Params::max_batch pairs in flight (Submit() returns false
  // when full).  Results equal Match()'s.
  bool Submit(const Image1b& iml, const Image1b& imr, uint64_t tag = 0);
  bool Collect(Image1f& disp, Image1f& dispr, uint64_t* tag = nullptr);
  // The same with the output maps bound at submission (they must have the image size and stay alive until the frame is
  // collected): Collect(tag) then only waits.  With images and maps that were Register()ed nothing is staged on either
  // side of the frame: the DMA engines read the images and write the maps in place.
  bool Submit(const Image1b& iml, const Image1b& imr, Image1f& disp, Image1f& dispr, uint64_t tag = 0);
  bool Collect(uint64_t* tag = nullptr);
  // Page-locks the storage of an image (pm_host_register): the host-buffer entry points then move it by DMA without
  // the staging copy through the handle's pinned slab.  Register buffers that are reused from frame to frame (the
  // capture loop's images, the output maps); Unregister() before an image is resized or destroyed.
  template <typename T>
  void Register(core::Image<T>& im) {
    RegisterRange(im.data(), sizeof(T) * (size_t)im.rows * (size_t)im.cols);
  }
  template <typename T>
  void Unregister(core::Image<T>& im) {
    UnregisterRange(im.data());
  }
  int InFlight() const { return handle_ ? pm_in_flight(handle_) : 0; }

  // The sparse-init maps Match() starts from (what SparseInit returns, patchmatch_gpu.cu:414-442):
  // left-image and right-image coordinates.  Kept until replaced; pass empty images to clear.
  void SetSeeds(const Image1f& seed_l, const Image1f& seed_r);

  // patchmatch_gpu.h:110-112 -- GFTT corners + rectified template matching + dilation, on the device.
  Image1f SparseInit(const Image1b& iml, const Image1b& imr, int dilate_factor);

  // Anything that looks like a cv::Mat_ (rows, cols, step, data).
  template <typename MatB, typename MatF>
  void MatchMat(const MatB& iml, const MatB& imr, MatF& disp, MatF& dispr) {
    EnsurePlan(iml.rows, iml.cols);
    Check(pm_match_u8(handle_, (const uint8_t*)iml.data, (const uint8_t*)imr.data, iml.rows, iml.cols,
                      (size_t)iml.step, seed_l_.empty() ? nullptr : seed_l_.data(),
                      seed_r_.empty() ? nullptr : seed_r_.data(), 0, (float*)disp.data, (float*)dispr.data,
                      (size_t)disp.step),
          "pm_match_u8");
  }

  pm_handle* handle() { return handle_; }

 private:
  void EnsurePlan(int rows, int cols);
  void Check(int status, const char* what) const;
  void RegisterRange(void* ptr, size_t bytes);
  void UnregisterRange(void* ptr);
  std::vector<std::pair<void*, size_t>> registered_;

  Params params_;
  pm_handle* handle_ = nullptr;
  int plan_rows_ = 0, plan_cols_ = 0;
  Image1f seed_l_, seed_r_;
  std::vector<std::pair<int, int>> in_flight_sizes_;  // (rows, cols) of the submitted pairs, oldest first
};

// One large rectified pair matched by several band handles of this process -- the C++ form of pm_tiled_*
// (include/pm/patchmatch.h; BASELINE configs[3]).  Same construct-and-Match() shape as PatchmatchGpu
// (patchmatch_gpu.h:94-102); `devices[k]` is the HIP device of band k (bands may share a device).  The result equals
// PatchmatchGpu::Match() on the whole image bit for bit.  Seed maps are an input here (SetSeeds): the device seeder
// works on whole images.
class TiledPatchmatchGpu final {
 public:
  TiledPatchmatchGpu(const TiledPatchmatchGpu&) = delete;
  TiledPatchmatchGpu& operator=(const TiledPatchmatchGpu&) = delete;
  TiledPatchmatchGpu(const PatchmatchGpu::Params& params, int rows, int cols, const std::vector<int>& devices);
  ~TiledPatchmatchGpu();

  void SetSeeds(const Image1f& seed_l, const Image1f& seed_r);
  // rounds: boundary exchange rounds per vertical sweep; -1 = bands - 1, always exact without a repeat (pm_tiled_run)
  void Match(const Image1b& iml, const Image1b& imr, Image1f& disp, Image1f& dispr, int rounds = -1);
  const pm_tiled_info& LastInfo() const { return info_; }
  // neighbouring bands on different devices, and how many of those boundaries have direct peer access (pm_tiled_topology)
  std::pair<int, int> Topology() const {
    int a = 0, b = 0;
    pm_tiled_topology(plan_, &a, &b);
    return {a, b};
  }

 private:
  int rows_, cols_;
  std::vector<pm_handle*> bands_;
  pm_tiled_plan* plan_ = nullptr;
  pm_tiled_info info_{};
  Image1f seed_l_, seed_r_;
};

}  // namespace pm
}  // namespace bm
