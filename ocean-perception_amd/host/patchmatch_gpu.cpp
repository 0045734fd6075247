// patchmatch_gpu.cpp -- bm::pm::PatchmatchGpu over the C ABI (see patchmatch_gpu.hpp).
#include "patchmatch_gpu.hpp"

#include <cmath>
#include <cstring>

namespace bm {
namespace pm {

pm_params PatchmatchGpu::Params::ToC() const {
  pm_params p;
  pm_params_default(&p, semantics);
  p.cost_alpha = cost_alpha;
  p.patchmatch_iters = patchmatch_iters;
  p.init_dilate_factor = init_dilate_factor;
  p.cost_improve_factor = cost_improve_factor;
  p.engine = engine;
  p.mode = mode;
  p.state_dtype = state_dtype;
  for (int i = 0; i < PM_MAX_ITERS; ++i) {
    p.patch_w[i] = patch_size;
    p.patch_h[i] = patch_size;
  }
  p.bg_patch_w = patch_size;
  p.bg_patch_h = patch_size;
  p.left_right_check = left_right_check ? 1 : 0;
  // like the reference, Match() seeds itself (SparseInit) unless seed maps were supplied through SetSeeds()
  p.sparse_init = 1;
  p.max_features_per_frame = detector_params.max_features_per_frame;
  p.min_distance_btw_features = detector_params.min_distance_btw_tracked_and_detected_features;
  p.gftt_block_size = detector_params.gftt_block_size;
  p.gftt_quality_level = detector_params.gftt_quality_level;
  p.gftt_use_harris = detector_params.gftt_use_harris_corner_detector ? 1 : 0;
  p.gftt_k = detector_params.gftt_k;
  // cv::cornerSubPix on the corners (feature_detector.cpp:110-120) and on the matches (stereo_matcher.cpp:94-103).
  // StereoMatcher::Params::bidirectional is never read by the reference either (stereo_matcher.cpp:11-116).
  p.subpixel_corners = detector_params.subpixel_corners ? 1 : 0;
  p.subpix_winsize = detector_params.subpix_winsize;
  p.subpix_zerozone = detector_params.subpix_zerozone;
  p.subpix_maxiters = detector_params.subpix_maxiters;
  p.subpix_epsilon = detector_params.subpix_epsilon;
  p.subpixel_refinement = matcher_params.subpixel_refinement ? 1 : 0;
  p.templ_cols = matcher_params.templ_cols;
  p.templ_rows = matcher_params.templ_rows;
  p.max_disp = matcher_params.max_disp;
  p.max_matching_cost = matcher_params.max_matching_cost;
  p.stream_priority = stream_priority;
  return p;
}

PatchmatchGpu::PatchmatchGpu(const Params& params) : params_(params) {
  if (params_.max_rows > 0 && params_.max_cols > 0) EnsurePlan(params_.max_rows, params_.max_cols);
}

PatchmatchGpu::~PatchmatchGpu() { pm_destroy(handle_); }

void PatchmatchGpu::Check(int status, const char* what) const {
  if (status == PM_OK) return;
  std::string msg = std::string(what) + ": " + pm_status_string(status);
  if (handle_) msg += std::string(" -- ") + pm_last_error(handle_);
  throw std::runtime_error(msg);
}

// The reference allocates its GpuMats lazily on the first Match and never resizes the noise image
// (patchmatch_gpu.cu:339-344); here a size that exceeds the plan re-plans explicitly.
void PatchmatchGpu::EnsurePlan(int rows, int cols) {
  if (handle_ && rows <= plan_rows_ && cols <= plan_cols_) return;
  if (InFlight() > 0)  // re-planning destroys the handle the submitted pairs live in
    throw std::runtime_error("PatchmatchGpu: image larger than the plan while pairs are in flight; Collect() them first");
  // grow to the envelope of everything seen so far: alternating 720x1280 / 1280x720 inputs re-plan once, not per call
  rows = rows > plan_rows_ ? rows : plan_rows_;
  cols = cols > plan_cols_ ? cols : plan_cols_;
  if (handle_) {
    pm_destroy(handle_);
    handle_ = nullptr;
  }
  const pm_params p = params_.ToC();
  const int rc = pm_create(&p, params_.device, rows, cols, params_.max_batch < 1 ? 1 : params_.max_batch, &handle_);
  if (rc != PM_OK) {
    std::string msg = std::string("pm_create: ") + pm_status_string(rc);
    if (handle_) msg += std::string(" -- ") + pm_last_error(handle_);
    pm_destroy(handle_);
    handle_ = nullptr;
    throw std::runtime_error(msg);
  }
  plan_rows_ = rows;
  plan_cols_ = cols;
  for (const auto& r : registered_)  // registrations belong to the handle: renew them on the new one
    Check(pm_host_register(handle_, r.first, r.second), "pm_host_register");
}

void PatchmatchGpu::RegisterRange(void* ptr, size_t bytes) {
  if (!ptr || bytes == 0) throw std::invalid_argument("PatchmatchGpu::Register: empty image");
  if (handle_) Check(pm_host_register(handle_, ptr, bytes), "pm_host_register");
  registered_.emplace_back(ptr, bytes);
}

void PatchmatchGpu::UnregisterRange(void* ptr) {
  for (size_t i = 0; i < registered_.size(); ++i) {
    if (registered_[i].first != ptr) continue;
    if (handle_) Check(pm_host_unregister(handle_, ptr), "pm_host_unregister");
    registered_.erase(registered_.begin() + (long)i);
    return;
  }
  throw std::invalid_argument("PatchmatchGpu::Unregister: this image was not registered");
}

namespace {
// a tightly packed copy of a float map (seed maps are kept until replaced)
void keep(core::Image<float>& dst, ConstView1f src) {
  if (src.empty()) {
    dst = core::Image<float>();
    return;
  }
  dst.create(src.rows, src.cols);
  for (int r = 0; r < src.rows; ++r)
    std::memcpy(dst.ptr(r), reinterpret_cast<const char*>(src.data) + (size_t)r * src.step, sizeof(float) * (size_t)src.cols);
}
}  // namespace

void PatchmatchGpu::SetSeedViews(ConstView1f seed_l, ConstView1f seed_r) {
  keep(seed_l_, seed_l);
  keep(seed_r_, seed_r);
}

void PatchmatchGpu::MatchViews(View1b iml, View1b imr, View1f disp, View1f dispr) {
  if (iml.empty() || imr.empty() || iml.rows != imr.rows || iml.cols != imr.cols)
    throw std::invalid_argument("PatchmatchGpu::Match: images empty or of different size");
  if (!seed_l_.empty() && (seed_l_.rows != iml.rows || seed_l_.cols != iml.cols))
    throw std::invalid_argument("PatchmatchGpu::Match: left seed map does not match the image size");
  if (!seed_r_.empty() && (seed_r_.rows != iml.rows || seed_r_.cols != iml.cols))
    throw std::invalid_argument("PatchmatchGpu::Match: right seed map does not match the image size");
  if (iml.step != imr.step) throw std::invalid_argument("PatchmatchGpu::Match: the two images have different row steps");
  const bool lr = params_.left_right_check;
  if (!disp.data || disp.rows != iml.rows || disp.cols != iml.cols || disp.step < sizeof(float) * (size_t)iml.cols)
    throw std::invalid_argument("PatchmatchGpu::Match: the left output map does not have the image size");
  if (lr && (!dispr.data || dispr.rows != iml.rows || dispr.cols != iml.cols || dispr.step != disp.step))
    throw std::invalid_argument("PatchmatchGpu::Match: the right output map does not have the left one's size and step");
  EnsurePlan(iml.rows, iml.cols);
  Check(pm_match_u8(handle_, iml.data, imr.data, iml.rows, iml.cols, iml.step, seed_l_.empty() ? nullptr : seed_l_.data(),
                    seed_r_.empty() ? nullptr : seed_r_.data(), 0, disp.data, dispr.data, disp.step),
        "pm_match_u8");
}

void PatchmatchGpu::MatchBatchViews(const std::vector<View1b>& imls, const std::vector<View1b>& imrs,
                                    const std::vector<View1f>& disps, const std::vector<View1f>& disprs) {
  const size_t n = imls.size();
  if (n == 0 || imrs.size() != n || disps.size() != n || disprs.size() != n)
    throw std::invalid_argument("PatchmatchGpu::MatchBatch: no pairs, or left / right / output counts differ");
  if ((int)n > params_.max_batch)
    throw std::invalid_argument("PatchmatchGpu::MatchBatch: more pairs than Params::max_batch");
  if (!seed_l_.empty() || !seed_r_.empty())
    throw std::invalid_argument("PatchmatchGpu::MatchBatch: seed maps set through SetSeeds() apply to single pairs only");
  const int rows = imls[0].rows, cols = imls[0].cols;
  std::vector<const uint8_t*> pl(n), pr(n);
  std::vector<float*> dl(n), dr(n);
  for (size_t i = 0; i < n; ++i) {
    const auto ok8 = [&](const View1b& v) { return !v.empty() && v.rows == rows && v.cols == cols && v.step == (size_t)cols; };
    const auto okf = [&](const View1f& v) { return !v.empty() && v.rows == rows && v.cols == cols && v.step == sizeof(float) * (size_t)cols; };
    if (!ok8(imls[i]) || !ok8(imrs[i]) || !okf(disps[i]) || !okf(disprs[i]))
      throw std::invalid_argument("PatchmatchGpu::MatchBatch: all images of a batch must be continuous and of one size");
    pl[i] = imls[i].data;
    pr[i] = imrs[i].data;
    dl[i] = disps[i].data;
    dr[i] = disprs[i].data;
  }
  EnsurePlan(rows, cols);
  Check(pm_match_batch_u8(handle_, (int)n, pl.data(), pr.data(), rows, cols, nullptr, nullptr, dl.data(), dr.data()),
        "pm_match_batch_u8");
}

bool PatchmatchGpu::SubmitViews(View1b iml, View1b imr, uint64_t tag) {
  if (iml.empty() || imr.empty() || iml.rows != imr.rows || iml.cols != imr.cols || iml.step != imr.step)
    throw std::invalid_argument("PatchmatchGpu::Submit: images empty or of different size / row step");
  if (InFlight() == 0) EnsurePlan(iml.rows, iml.cols);  // re-planning destroys the handle: only when idle
  const bool seeded_l = !seed_l_.empty() && seed_l_.rows == iml.rows && seed_l_.cols == iml.cols;
  const bool seeded_r = !seed_r_.empty() && seed_r_.rows == iml.rows && seed_r_.cols == iml.cols;
  const int rc = pm_submit_u8(handle_, iml.data, imr.data, iml.rows, iml.cols, iml.step,
                              seeded_l ? seed_l_.data() : nullptr, seeded_r ? seed_r_.data() : nullptr, 0, tag);
  if (rc == PM_ERR_BUSY) return false;
  Check(rc, "pm_submit_u8");
  in_flight_sizes_.emplace_back(iml.rows, iml.cols);
  return true;
}

bool PatchmatchGpu::SubmitBoundViews(View1b iml, View1b imr, View1f disp, View1f dispr, uint64_t tag) {
  if (iml.empty() || imr.empty() || iml.rows != imr.rows || iml.cols != imr.cols || iml.step != imr.step)
    throw std::invalid_argument("PatchmatchGpu::Submit: images empty or of different size / row step");
  if (disp.empty() || dispr.empty() || disp.rows != iml.rows || disp.cols != iml.cols || dispr.rows != iml.rows ||
      dispr.cols != iml.cols || disp.step != dispr.step)
    throw std::invalid_argument("PatchmatchGpu::Submit: bound maps must already have the image size");
  if (InFlight() == 0) EnsurePlan(iml.rows, iml.cols);
  const bool seeded_l = !seed_l_.empty() && seed_l_.rows == iml.rows && seed_l_.cols == iml.cols;
  const bool seeded_r = !seed_r_.empty() && seed_r_.rows == iml.rows && seed_r_.cols == iml.cols;
  const int rc = pm_submit_bound_u8(handle_, iml.data, imr.data, iml.rows, iml.cols, iml.step,
                                    seeded_l ? seed_l_.data() : nullptr, seeded_r ? seed_r_.data() : nullptr, 0, disp.data,
                                    dispr.data, disp.step, tag);
  if (rc == PM_ERR_BUSY) return false;
  Check(rc, "pm_submit_bound_u8");
  in_flight_sizes_.emplace_back(iml.rows, iml.cols);
  return true;
}

bool PatchmatchGpu::Collect(uint64_t* tag) {
  if (in_flight_sizes_.empty()) return false;
  Check(pm_collect(handle_, nullptr, nullptr, 0, tag), "pm_collect");
  in_flight_sizes_.erase(in_flight_sizes_.begin());
  return true;
}

bool PatchmatchGpu::NextCollectSize(int* rows, int* cols) const {
  if (in_flight_sizes_.empty()) return false;
  *rows = in_flight_sizes_.front().first;
  *cols = in_flight_sizes_.front().second;
  return true;
}

bool PatchmatchGpu::CollectViews(View1f disp, View1f dispr, uint64_t* tag) {
  if (in_flight_sizes_.empty()) return false;
  const int rows = in_flight_sizes_.front().first, cols = in_flight_sizes_.front().second;
  if (disp.empty() || dispr.empty() || disp.rows != rows || disp.cols != cols || dispr.rows != rows || dispr.cols != cols ||
      disp.step != dispr.step)
    throw std::invalid_argument("PatchmatchGpu::Collect: the maps do not have the size of the pair being collected");
  Check(pm_collect(handle_, disp.data, dispr.data, disp.step, tag), "pm_collect");
  in_flight_sizes_.erase(in_flight_sizes_.begin());
  return true;
}

void PatchmatchGpu::Match(const uint8_t* d_iml, const uint8_t* d_imr, int rows, int cols, const float* d_seed_l,
                          const float* d_seed_r, float* d_disp, float* d_dispr) {
  EnsurePlan(rows, cols);
  Check(pm_match_device(handle_, 1, d_iml, d_imr, rows, cols, d_seed_l, d_seed_r, d_disp, d_dispr),
        "pm_match_device");
  Check(pm_synchronize(handle_), "pm_synchronize");
}

void PatchmatchGpu::Match(const GpuImage1f& iml, const GpuImage1f& imr, const GpuImage1f& Gl, const GpuImage1f& Gr,
                          GpuImage1f& disp, void* stream) {
  const auto same = [&](const GpuImage1f& m) { return m.data && m.rows == iml.rows && m.cols == iml.cols; };
  if (!same(iml) || !same(imr) || !same(Gl) || !same(Gr) || !same(disp))
    throw std::invalid_argument("PatchmatchGpu::Match(GpuImage1f...): images empty or of different size");
  if (imr.step != iml.step || Gl.step != iml.step || Gr.step != iml.step)
    throw std::invalid_argument("PatchmatchGpu::Match(GpuImage1f...): images and gradients must share one row step");
  EnsurePlan(iml.rows, iml.cols);
  Check(pm_match_view_device(handle_, iml.data, imr.data, Gl.data, Gr.data, iml.rows, iml.cols, iml.step, disp.data,
                             disp.step, stream),
        "pm_match_view_device");
}

void PatchmatchGpu::SparseInitViews(View1b iml, View1b imr, int dilate_factor, View1f seed) {
  if (iml.empty() || imr.empty() || iml.rows != imr.rows || iml.cols != imr.cols)
    throw std::invalid_argument("PatchmatchGpu::SparseInit: images empty or of different size");
  if (iml.step != (size_t)iml.cols || imr.step != (size_t)imr.cols)
    throw std::invalid_argument("PatchmatchGpu::SparseInit: images must be continuous (step == cols)");
  if (seed.empty() || seed.rows != iml.rows || seed.cols != iml.cols || seed.step != sizeof(float) * (size_t)iml.cols)
    throw std::invalid_argument("PatchmatchGpu::SparseInit: the seed map must be continuous and of the image size");
  EnsurePlan(iml.rows, iml.cols);
  Check(pm_sparse_init(handle_, iml.data, imr.data, iml.rows, iml.cols, dilate_factor, seed.data), "pm_sparse_init");
}

// ---- TiledPatchmatchGpu ----------------------------------------------------------------------------------------------
TiledPatchmatchGpu::TiledPatchmatchGpu(const PatchmatchGpu::Params& params, int rows, int cols,
                                       const std::vector<int>& devices)
    : rows_(rows), cols_(cols) {
  if (devices.empty()) throw std::runtime_error("TiledPatchmatchGpu: no devices given");
  pm_params p = params.ToC();
  p.sparse_init = 0;  // the device seeder sees whole images only: seed maps come in through SetSeeds()
  const int n = (int)devices.size();
  const int band_rows = pm_tiled_band_rows(&p, rows, n);
  if (band_rows < 0) throw std::runtime_error("TiledPatchmatchGpu: cannot split the image into that many bands");
  auto cleanup = [&]() {
    pm_tiled_destroy(plan_);
    plan_ = nullptr;
    for (pm_handle* h : bands_) pm_destroy(h);
    bands_.clear();
  };
  for (int k = 0; k < n; ++k) {
    pm_handle* h = nullptr;
    const int rc = pm_create(&p, devices[(size_t)k], band_rows, cols, 1, &h);
    if (rc != PM_OK) {
      const std::string msg = std::string("TiledPatchmatchGpu: pm_create: ") + pm_status_string(rc) +
                              (h ? std::string(" -- ") + pm_last_error(h) : std::string());
      pm_destroy(h);
      cleanup();
      throw std::runtime_error(msg);
    }
    bands_.push_back(h);
  }
  const int rc = pm_tiled_create(bands_.data(), n, rows, cols, &plan_);
  if (rc != PM_OK) {
    const std::string msg = std::string("TiledPatchmatchGpu: pm_tiled_create: ") + pm_status_string(rc) + " -- " +
                            pm_tiled_last_error(plan_);
    cleanup();
    throw std::runtime_error(msg);
  }
}

TiledPatchmatchGpu::~TiledPatchmatchGpu() {
  pm_tiled_destroy(plan_);
  for (pm_handle* h : bands_) pm_destroy(h);
}

void TiledPatchmatchGpu::SetSeedViews(ConstView1f seed_l, ConstView1f seed_r) {
  keep(seed_l_, seed_l);
  keep(seed_r_, seed_r);
}

void TiledPatchmatchGpu::SetExchange(int mode) {
  const int rc = pm_tiled_set_exchange(plan_, mode);
  if (rc != PM_OK)
    throw std::runtime_error(std::string("pm_tiled_set_exchange: ") + pm_status_string(rc) + " -- " + pm_tiled_last_error(plan_));
}

void TiledPatchmatchGpu::SetSchedule(int schedule) {
  const int rc = pm_tiled_set_schedule(plan_, schedule);
  if (rc != PM_OK)
    throw std::runtime_error(std::string("pm_tiled_set_schedule: ") + pm_status_string(rc) + " -- " + pm_tiled_last_error(plan_));
}

void TiledPatchmatchGpu::MatchViews(View1b iml, View1b imr, View1f disp, View1f dispr, int rounds) {
  if (iml.rows != rows_ || iml.cols != cols_ || imr.rows != rows_ || imr.cols != cols_ || iml.step != imr.step)
    throw std::runtime_error("TiledPatchmatchGpu::Match: image size differs from the plan");
  if (disp.empty() || dispr.empty() || disp.rows != rows_ || disp.cols != cols_ || dispr.rows != rows_ ||
      dispr.cols != cols_ || disp.step != dispr.step)
    throw std::runtime_error("TiledPatchmatchGpu::Match: output maps differ from the plan's size");
  const int rc = pm_tiled_match_u8(plan_, iml.data, imr.data, iml.step, seed_l_.empty() ? nullptr : seed_l_.data(),
                                   seed_r_.empty() ? nullptr : seed_r_.data(), 0, disp.data, dispr.data, disp.step, rounds,
                                   &info_);
  if (rc != PM_OK)
    throw std::runtime_error(std::string("pm_tiled_match_u8: ") + pm_status_string(rc) + " -- " + pm_tiled_last_error(plan_));
}

}  // namespace pm
}  // namespace bm
