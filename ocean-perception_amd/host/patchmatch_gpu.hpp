// patchmatch_gpu.hpp -- host-side C++ mirror of the reference's bm::pm::PatchmatchGpu
// (src/vehicle/patchmatch_gpu/patchmatch_gpu.h:77-124) on top of the C ABI in pm/patchmatch.h.
//
// Same namespace, class name, Params field names / defaults and Match() signatures, so a caller
// written against the reference (test/stereo_matching/patchmatch_gpu_test.cpp:68-88) compiles
// against this header unchanged (the reference's include line "patchmatch_gpu/patchmatch_gpu.h"
// resolves to a forwarding header beside this file).  No HIP header is included here: host code is
// plain C++17 and reaches the device only through the C ABI.
//
// Image types.  The reference's Image1b / Image1f are cv::Mat1b / cv::Mat1f
// (src/vehicle/vision_core/cv_types.hpp:8-12), and its callers hand those to Match().
//   * Where <opencv2/core.hpp> can be included (or PM_USE_OPENCV_TYPES is defined) this header declares
//     bm::core::Image1b / Image1f as exactly those typedefs -- a repeated identical typedef, so it sits beside
//     vision_core/cv_types.hpp in one translation unit -- and the three reference signatures take cv::Mat_:
//     Match() honours `step`, and (re)allocates the output maps with create(rows, cols) as GpuMat::download does
//     (patchmatch_gpu.cu:374-375).  PM_NO_OPENCV_TYPES switches this off.
//   * Elsewhere OpenCV is not a dependency: the minimal bm::core::Image<T> below carries the members the path
//     uses (rows, cols, step, ptr(), at(), empty(), create()) and Image1b / Image1f are Image<uint8_t> / Image<float>.
// libvehicle_pm_gpu.so is the same in both cases: what it exports of these classes takes plain views
// (pointer, rows, cols, step); the methods with image types are inline adapters over them.
#pragma once

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "pm/patchmatch.h"

#if !defined(PM_NO_OPENCV_TYPES) && !defined(PM_USE_OPENCV_TYPES) && defined(__has_include)
#if __has_include(<opencv2/core.hpp>)
#define PM_USE_OPENCV_TYPES 1
#endif
#endif
#if defined(PM_USE_OPENCV_TYPES) && !defined(PM_NO_OPENCV_TYPES)
#include <opencv2/core.hpp>
#define PM_HAVE_OPENCV_TYPES 1
#endif

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

#ifdef PM_HAVE_OPENCV_TYPES
typedef cv::Mat1b Image1b;  // src/vehicle/vision_core/cv_types.hpp:8
typedef cv::Mat1f Image1f;  // :12
#else
typedef Image<uint8_t> Image1b;
typedef Image<float> Image1f;
#endif

}  // namespace core

namespace pm {
// What the compiled library sees of an image, whatever class the caller holds it in.
template <typename T>
struct ImageView {
  T* data = nullptr;
  int rows = 0, cols = 0;
  size_t step = 0;  // bytes per row
  bool empty() const { return data == nullptr || rows <= 0 || cols <= 0; }
};
typedef ImageView<const uint8_t> View1b;
typedef ImageView<const float> ConstView1f;
typedef ImageView<float> View1f;

namespace detail {
template <typename T>
inline ImageView<const T> cview(const core::Image<T>& m) {
  return {m.data(), m.rows, m.cols, m.step};
}
template <typename T>
inline ImageView<T> mview(core::Image<T>& m) {
  return {m.data(), m.rows, m.cols, m.step};
}
// like GpuMat::download (patchmatch_gpu.cu:374-375) an output is (re)allocated to the image size
template <typename T>
inline void alloc(core::Image<T>& m, int rows, int cols) {
  if (m.rows != rows || m.cols != cols || m.empty()) m.create(rows, cols);
}
#ifdef PM_HAVE_OPENCV_TYPES
template <typename T>
inline ImageView<const T> cview(const cv::Mat_<T>& m) {
  return {reinterpret_cast<const T*>(m.data), m.rows, m.cols, (size_t)m.step};
}
template <typename T>
inline ImageView<T> mview(cv::Mat_<T>& m) {
  return {reinterpret_cast<T*>(m.data), m.rows, m.cols, (size_t)m.step};
}
template <typename T>
inline void alloc(cv::Mat_<T>& m, int rows, int cols) {
  m.create(rows, cols);  // cv::Mat::create keeps a matrix that already has this size and type
}
#endif
}  // namespace detail
}  // namespace pm

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
  // (on the device); seed maps set through SetSeeds() take precedence.  disp / dispr are (re)allocated to the
  // image size as GpuMat::download does (patchmatch_gpu.cu:374-375); the images' row steps are honoured.
  void Match(const Image1b& iml, const Image1b& imr, Image1f& disp, Image1f& dispr) { MatchMat(iml, imr, disp, dispr); }

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
                  std::vector<Image1f>& disprs) {
    const size_t n = imls.size();
    if (n == 0 || imrs.size() != n) throw std::invalid_argument("PatchmatchGpu::MatchBatch: no pairs, or left / right counts differ");
    disps.resize(n);
    disprs.resize(n);
    std::vector<View1b> vl(n), vr(n);
    std::vector<View1f> dl(n), dr(n);
    for (size_t i = 0; i < n; ++i) {
      detail::alloc(disps[i], imls[0].rows, imls[0].cols);
      detail::alloc(disprs[i], imls[0].rows, imls[0].cols);
      vl[i] = detail::cview(imls[i]);
      vr[i] = detail::cview(imrs[i]);
      dl[i] = detail::mview(disps[i]);
      dr[i] = detail::mview(disprs[i]);
    }
    MatchBatchViews(vl, vr, dl, dr);
  }

  // Match() for a sequence of frames (the callback loop of patchmatch_gpu_test.cpp:118-128) with the
  // copies off the critical path: Submit() returns once the pair is packed and enqueued, Collect() waits
  // for the oldest submitted pair.  At most Params::max_batch pairs in flight (Submit() returns false
  // when full).  Results equal Match()'s.
  bool Submit(const Image1b& iml, const Image1b& imr, uint64_t tag = 0) {
    return SubmitViews(detail::cview(iml), detail::cview(imr), tag);
  }
  bool Collect(Image1f& disp, Image1f& dispr, uint64_t* tag = nullptr) {
    int rows = 0, cols = 0;
    if (!NextCollectSize(&rows, &cols)) return false;
    detail::alloc(disp, rows, cols);
    detail::alloc(dispr, rows, cols);
    return CollectViews(detail::mview(disp), detail::mview(dispr), tag);
  }
  // The same with the output maps bound at submission (they must have the image size and stay alive until the frame is
  // collected): Collect(tag) then only waits.  With images and maps that were Register()ed nothing is staged on either
  // side of the frame: the DMA engines read the images and write the maps in place.
  bool Submit(const Image1b& iml, const Image1b& imr, Image1f& disp, Image1f& dispr, uint64_t tag = 0) {
    return SubmitBoundViews(detail::cview(iml), detail::cview(imr), detail::mview(disp), detail::mview(dispr), tag);
  }
  bool Collect(uint64_t* tag = nullptr);
  // Page-locks the storage of an image (pm_host_register): the host-buffer entry points then move it by DMA without
  // the staging copy through the handle's pinned slab.  Register buffers that are reused from frame to frame (the
  // capture loop's images, the output maps); Unregister() before an image is resized or destroyed.
  template <typename Mat>
  void Register(Mat& im) {
    const auto v = detail::mview(im);
    RegisterRange((void*)v.data, v.step * (size_t)v.rows);
  }
  template <typename Mat>
  void Unregister(Mat& im) {
    UnregisterRange((void*)detail::mview(im).data);
  }
  int InFlight() const { return handle_ ? pm_in_flight(handle_) : 0; }

  // The sparse-init maps Match() starts from (what SparseInit returns, patchmatch_gpu.cu:414-442):
  // left-image and right-image coordinates.  Kept (copied) until replaced; pass empty images to clear.
  void SetSeeds(const Image1f& seed_l, const Image1f& seed_r) { SetSeedViews(detail::cview(seed_l), detail::cview(seed_r)); }

  // patchmatch_gpu.h:110-112 -- GFTT corners + rectified template matching + dilation, on the device.
  Image1f SparseInit(const Image1b& iml, const Image1b& imr, int dilate_factor) {
    Image1f seed(iml.rows > 0 ? iml.rows : 1, iml.cols > 0 ? iml.cols : 1);
    SparseInitViews(detail::cview(iml), detail::cview(imr), dilate_factor, detail::mview(seed));
    return seed;
  }

  // Match() for any image class detail::cview / mview / alloc know: bm::core::Image<T> always, cv::Mat_<T> when the
  // OpenCV types are on (then Image1b / Image1f ARE cv::Mat1b / cv::Mat1f and this is what Match() calls).
  template <typename MatB, typename MatF>
  void MatchMat(const MatB& iml, const MatB& imr, MatF& disp, MatF& dispr) {
    const View1b l = detail::cview(iml), r = detail::cview(imr);
    if (l.empty() || r.empty() || l.rows != r.rows || l.cols != r.cols)
      throw std::invalid_argument("PatchmatchGpu::Match: images empty or of different size");
    detail::alloc(disp, l.rows, l.cols);
    detail::alloc(dispr, l.rows, l.cols);
    MatchViews(l, r, detail::mview(disp), detail::mview(dispr));
  }

  // ---- what libvehicle_pm_gpu.so exports of this class: the same calls on plain views --------------------------------
  void MatchViews(View1b iml, View1b imr, View1f disp, View1f dispr);
  void MatchBatchViews(const std::vector<View1b>& imls, const std::vector<View1b>& imrs, const std::vector<View1f>& disps,
                       const std::vector<View1f>& disprs);
  bool SubmitViews(View1b iml, View1b imr, uint64_t tag);
  bool SubmitBoundViews(View1b iml, View1b imr, View1f disp, View1f dispr, uint64_t tag);
  bool NextCollectSize(int* rows, int* cols) const;  // size of the oldest pair in flight; false: none
  bool CollectViews(View1f disp, View1f dispr, uint64_t* tag);
  void SetSeedViews(ConstView1f seed_l, ConstView1f seed_r);
  void SparseInitViews(View1b iml, View1b imr, int dilate_factor, View1f seed);

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
  core::Image<float> seed_l_, seed_r_;  // tightly packed copies
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

  void SetSeeds(const Image1f& seed_l, const Image1f& seed_r) { SetSeedViews(detail::cview(seed_l), detail::cview(seed_r)); }
  // rounds: boundary exchange rounds per vertical sweep; -1 = bands - 1, always exact without a repeat (pm_tiled_run)
  void Match(const Image1b& iml, const Image1b& imr, Image1f& disp, Image1f& dispr, int rounds = -1) {
    detail::alloc(disp, rows_, cols_);
    detail::alloc(dispr, rows_, cols_);
    MatchViews(detail::cview(iml), detail::cview(imr), detail::mview(disp), detail::mview(dispr), rounds);
  }
  const pm_tiled_info& LastInfo() const { return info_; }
  // neighbouring bands on different devices, and how many of those boundaries have direct peer access (pm_tiled_topology)
  std::pair<int, int> Topology() const {
    int a = 0, b = 0;
    pm_tiled_topology(plan_, &a, &b);
    return {a, b};
  }
  // how a band reads its neighbour's boundary row (pm_tiled_exchange: AUTO, COPY, DIRECT); results do not depend on it
  void SetExchange(int mode);
  // how a vertical sweep crosses the bands (pm_tiled_schedule: SPECULATIVE rounds or PIPELINED in order); same maps
  void SetSchedule(int schedule);

  void SetSeedViews(ConstView1f seed_l, ConstView1f seed_r);
  void MatchViews(View1b iml, View1b imr, View1f disp, View1f dispr, int rounds);

 private:
  int rows_, cols_;
  std::vector<pm_handle*> bands_;
  pm_tiled_plan* plan_ = nullptr;
  pm_tiled_info info_{};
  core::Image<float> seed_l_, seed_r_;
};

}  // namespace pm
}  // namespace bm
