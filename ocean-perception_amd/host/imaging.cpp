// imaging.cpp -- bm::imaging free functions over the C ABI (see imaging.hpp).
#include "imaging.hpp"

#include <mutex>
#include <stdexcept>
#include <string>

namespace bm {
namespace imaging {
namespace {

int g_device = 0;
pm_handle* g_handle = nullptr;
std::mutex g_mutex;

void Check(int status, const char* what) {
  if (status == PM_OK) return;
  std::string msg = std::string(what) + ": " + pm_status_string(status);
  if (g_handle) msg += std::string(" -- ") + pm_last_error(g_handle);
  throw std::runtime_error(msg);
}

pm_handle* Context() {
  if (g_handle) return g_handle;
  pm_params p;
  pm_params_default(&p, PM_SEM_GPU);
  pm_handle* h = nullptr;
  const int rc = pm_create(&p, g_device, 16, 16, 1, &h);
  if (rc != PM_OK) {
    std::string msg = std::string("bm::imaging: pm_create: ") + pm_status_string(rc);
    if (h) msg += std::string(" -- ") + pm_last_error(h);
    pm_destroy(h);
    throw std::runtime_error(msg);
  }
  g_handle = h;
  return g_handle;
}

// A device buffer freed on scope exit.
class DeviceBuffer {
 public:
  DeviceBuffer(pm_handle* h, size_t bytes) : h_(h) { Check(pm_device_malloc(h, bytes, &p_), "pm_device_malloc"); }
  ~DeviceBuffer() { pm_device_free(h_, p_); }
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  template <typename T>
  T* as() { return static_cast<T*>(p_); }
  void Upload(const void* src, size_t bytes) { Check(pm_upload(h_, p_, src, bytes), "pm_upload"); }
  void Download(void* dst, size_t bytes) { Check(pm_download(h_, dst, p_, bytes), "pm_download"); }

 private:
  pm_handle* h_;
  void* p_ = nullptr;
};

template <typename T>
size_t Bytes(const Image<T>& im) { return sizeof(T) * (size_t)im.rows * im.cols; }

void SameSize(int r0, int c0, int r1, int c1, const char* what) {
  if (r0 != r1 || c0 != c1 || r0 <= 0 || c0 <= 0) throw std::invalid_argument(std::string(what) + ": image sizes differ or are empty");
}

}  // namespace

void SetDevice(int device) {
  std::lock_guard<std::mutex> lock(g_mutex);
  if (g_handle && device != g_device) {
    pm_destroy(g_handle);
    g_handle = nullptr;
  }
  g_device = device;
}

Image3f CastImage3bTo3f(const Image3b& im) {
  Image3f out(im.rows, im.cols);
  const float s = (float)(1.0 / 255.0);  // kUint8ToFloat, image_util.cpp:10
  for (int y = 0; y < im.rows; ++y)
    for (int x = 0; x < im.cols; ++x)
      for (int c = 0; c < 3; ++c) out.at(y, x).v[c] = (float)im.at(y, x).v[c] * s;
  return out;
}

Image1f ComputeIntensity(const Image3f& bgr) {
  std::lock_guard<std::mutex> lock(g_mutex);
  pm_handle* h = Context();
  DeviceBuffer d_in(h, Bytes(bgr)), d_out(h, sizeof(float) * (size_t)bgr.rows * bgr.cols);
  d_in.Upload(bgr.data(), Bytes(bgr));
  Check(pm_compute_intensity(h, d_in.as<float>(), bgr.rows, bgr.cols, d_out.as<float>()), "pm_compute_intensity");
  Image1f out(bgr.rows, bgr.cols);
  d_out.Download(out.data(), Bytes(out));
  return out;
}

Image3f EstimateIlluminantGaussian(const Image3f& bgr, int ksizeX, int ksizeY, double sigmaX, double sigmaY) {
  if (ksizeX != ksizeY || sigmaX != sigmaY)
    throw std::invalid_argument("EstimateIlluminantGaussian: only square kernels (the reference's only use)");
  std::lock_guard<std::mutex> lock(g_mutex);
  pm_handle* h = Context();
  DeviceBuffer d_in(h, Bytes(bgr)), d_out(h, Bytes(bgr));
  d_in.Upload(bgr.data(), Bytes(bgr));
  Check(pm_gaussian_blur(h, d_in.as<float>(), bgr.rows, bgr.cols, 3, ksizeX, sigmaX, d_out.as<float>()),
        "pm_gaussian_blur");
  Image3f out(bgr.rows, bgr.cols);
  d_out.Download(out.data(), Bytes(out));
  // Akkaynak et al. multiply by a factor of 2 to get the illuminant map (illuminant.cpp:19-20); exact in float
  for (int y = 0; y < out.rows; ++y)
    for (int x = 0; x < out.cols; ++x)
      for (int c = 0; c < 3; ++c) out.at(y, x).v[c] *= 2.0f;
  return out;
}

Image3f Normalize(const Image3f& bgr) {
  std::lock_guard<std::mutex> lock(g_mutex);
  pm_handle* h = Context();
  DeviceBuffer d_in(h, Bytes(bgr)), d_out(h, Bytes(bgr));
  d_in.Upload(bgr.data(), Bytes(bgr));
  Check(pm_normalize(h, d_in.as<float>(), bgr.rows, bgr.cols, d_out.as<float>()), "pm_normalize");
  Image3f out(bgr.rows, bgr.cols);
  d_out.Download(out.data(), Bytes(out));
  return out;
}

Image3f NormalizeColorIlluminant(const Image3f bgr) {
  std::lock_guard<std::mutex> lock(g_mutex);
  pm_handle* h = Context();
  DeviceBuffer d_in(h, Bytes(bgr)), d_out(h, Bytes(bgr));
  d_in.Upload(bgr.data(), Bytes(bgr));
  Check(pm_normalize_color_illuminant(h, d_in.as<float>(), bgr.rows, bgr.cols, d_out.as<float>()),
        "pm_normalize_color_illuminant");
  Image3f out(bgr.rows, bgr.cols);
  d_out.Download(out.data(), Bytes(out));
  return out;
}

float FindDarkFast(const Image1f& intensity, const Image1f& range, float percentile, Image1b& mask) {
  SameSize(intensity.rows, intensity.cols, range.rows, range.cols, "FindDarkFast");
  std::lock_guard<std::mutex> lock(g_mutex);
  pm_handle* h = Context();
  DeviceBuffer d_i(h, Bytes(intensity)), d_r(h, Bytes(range)), d_m(h, (size_t)intensity.rows * intensity.cols);
  d_i.Upload(intensity.data(), Bytes(intensity));
  d_r.Upload(range.data(), Bytes(range));
  float thr = 0.f;
  Check(pm_find_dark(h, d_i.as<float>(), d_r.as<float>(), intensity.rows, intensity.cols, percentile,
                     d_m.as<uint8_t>(), &thr),
        "pm_find_dark");
  if (mask.rows != intensity.rows || mask.cols != intensity.cols) mask.create(intensity.rows, intensity.cols);
  d_m.Download(mask.data(), Bytes(mask));
  return thr;
}

Image3f RemoveBackscatter(const Image3f& bgr, const Image1f& range, const Vector3f& B, const Vector3f& beta_B) {
  SameSize(bgr.rows, bgr.cols, range.rows, range.cols, "RemoveBackscatter");
  std::lock_guard<std::mutex> lock(g_mutex);
  pm_handle* h = Context();
  DeviceBuffer d_in(h, Bytes(bgr)), d_r(h, Bytes(range)), d_out(h, Bytes(bgr));
  d_in.Upload(bgr.data(), Bytes(bgr));
  d_r.Upload(range.data(), Bytes(range));
  Check(pm_remove_backscatter(h, d_in.as<float>(), d_r.as<float>(), bgr.rows, bgr.cols, B.data(), beta_B.data(),
                              d_out.as<float>()),
        "pm_remove_backscatter");
  Image3f out(bgr.rows, bgr.cols);
  d_out.Download(out.data(), Bytes(out));
  return out;
}

Image3f CorrectAttenuation(const Image3f& bgr, const Image1f& range, const Vector12f& X) {
  SameSize(bgr.rows, bgr.cols, range.rows, range.cols, "CorrectAttenuation");
  std::lock_guard<std::mutex> lock(g_mutex);
  pm_handle* h = Context();
  DeviceBuffer d_in(h, Bytes(bgr)), d_r(h, Bytes(range)), d_out(h, Bytes(bgr));
  d_in.Upload(bgr.data(), Bytes(bgr));
  d_r.Upload(range.data(), Bytes(range));
  Check(pm_correct_attenuation(h, d_in.as<float>(), d_r.as<float>(), bgr.rows, bgr.cols, X.data(), d_out.as<float>()),
        "pm_correct_attenuation");
  Image3f out(bgr.rows, bgr.cols);
  d_out.Download(out.data(), Bytes(out));
  return out;
}

Image1b StereoReady(const Image3b& bgr) {
  std::lock_guard<std::mutex> lock(g_mutex);
  pm_handle* h = Context();
  DeviceBuffer d_in(h, Bytes(bgr)), d_g(h, (size_t)bgr.rows * bgr.cols);
  d_in.Upload(bgr.data(), Bytes(bgr));
  Check(pm_stereo_ready(h, d_in.as<uint8_t>(), bgr.rows, bgr.cols, nullptr, d_g.as<uint8_t>()), "pm_stereo_ready");
  Image1b out(bgr.rows, bgr.cols);
  d_g.Download(out.data(), Bytes(out));
  return out;
}

Image1f DispToDepth(const Image1f& disp, double fx, double baseline) {
  std::lock_guard<std::mutex> lock(g_mutex);
  pm_handle* h = Context();
  DeviceBuffer d_in(h, Bytes(disp)), d_out(h, Bytes(disp));
  d_in.Upload(disp.data(), Bytes(disp));
  Check(pm_disp_to_range(h, d_in.as<float>(), disp.rows, disp.cols, fx, baseline, d_out.as<float>()),
        "pm_disp_to_range");
  Image1f out(disp.rows, disp.cols);
  d_out.Download(out.data(), Bytes(out));
  return out;
}

}  // namespace imaging
}  // namespace bm
