// pm_imaging.hip -- C-ABI implementation of include/pm/imaging.h: the rows either side of the stereo hot path
//  and the device-buffer helpers.  A separate translation unit: it reaches the handle only
// through pm_internal.hpp (device, stream, error text, one opaque state slot).
#include "pm/imaging.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <new>
#include <vector>

#include "pm_enhance.hpp"
#include "pm_imaging.hpp"
#include "pm_internal.hpp"
#include "pm_tune.hpp"

using namespace pm;

namespace {

// Device state of the imaging entry points, created on first use and released by pm_destroy.
struct ImagingState {
  unsigned* img_scalars = nullptr;  // device: [0] max range / min disparity bits, [1] dark-pixel count, [2..3] V min / max
  // stereo-ready enhancement: row-pass output, bgr / illuminant, Gaussian taps
  float* enh_tmp = nullptr;
  float* enh_q = nullptr;
  float* enh_taps = nullptr;
  size_t enh_values = 0;  // floats allocated in enh_tmp / enh_q
  // pm_match_bgr_device: blurred illuminants of n pairs (left, right) and their value min / max words
  float* bgr_blur = nullptr;
  unsigned* bgr_mm = nullptr;
  size_t bgr_values = 0;  // floats allocated in bgr_blur
  int bgr_pairs = 0;      // pairs bgr_mm holds
  int enh_taps_cap = 0, enh_ksize = 0;
  double enh_sigma = 0;
};

#define PM_HIP(h, call)                                                                                     \
  do {                                                                                                      \
    hipError_t e_ = (call);                                                                                 \
    if (e_ != hipSuccess) {                                                                                 \
      pm_internal::set_error((h), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return PM_ERR_HIP;                                                                                    \
    }                                                                                                       \
  } while (0)

#define set_err pm_internal::set_error

int launch_check(pm_handle* h, const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_err(h, "launch of %s failed: %s", what, hipGetErrorString(e));
    return PM_ERR_HIP;
  }
  return PM_OK;
}

ImagingState* state_of(pm_handle* h) {
  void** slot = pm_internal::imaging_slot(h);
  if (!*slot) *slot = new (std::nothrow) ImagingState();
  return static_cast<ImagingState*>(*slot);
}

}  // namespace

void pm_internal::release_imaging(pm_handle* h) {
  void** slot = pm_internal::imaging_slot(h);
  ImagingState* st = static_cast<ImagingState*>(*slot);
  if (!st) return;
  void* dev[] = {st->img_scalars, st->enh_tmp, st->enh_q, st->enh_taps, st->bgr_blur, st->bgr_mm};
  for (void* p : dev)
    if (p) (void)hipFree(p);
  delete st;
  *slot = nullptr;
}

// ---- pm/imaging.h: disparity -> range -> range-dependent correction -----------------------------
namespace {

int imaging_begin(pm_handle* h, const char* what, const void* a, const void* b, int rows, int cols) {
  if (!h) return PM_ERR_INVALID_ARG;
  if (!a || !b || rows <= 0 || cols <= 0) {
    set_err(h, "%s: null pointer or empty image", what);
    return PM_ERR_INVALID_ARG;
  }
  PM_HIP(h, hipSetDevice(pm_internal::device(h)));
  if (!state_of(h)) {
    set_err(h, "%s: out of host memory", what);
    return PM_ERR_NOMEM;
  }
  if (!state_of(h)->img_scalars) {
    PM_HIP(h, hipMalloc((void**)&state_of(h)->img_scalars, sizeof(unsigned) * 4));
    PM_HIP(h, hipMemsetAsync(state_of(h)->img_scalars, 0, sizeof(unsigned) * 4, pm_internal::stream(h)));
  }
  return PM_OK;
}

inline dim3 stream_grid(size_t n_items) {  // grid-stride: enough blocks to fill 256 CUs a few times over
  size_t b = (n_items + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}

inline dim3 reduce_grid(size_t n_items) {  // reductions: one atomic per block, so no more blocks than fill the chip
  size_t b = (n_items + 255) / 256;
  if (b > 256 * 8) b = 256 * 8;
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

BackscatterParams backscatter_params(const float B[3], const float beta_B[3]) {
  BackscatterParams bp{};
  for (int c = 0; c < 3; ++c) {
    bp.B[c] = B ? B[c] : 0.f;
    bp.beta_B[c] = beta_B ? beta_B[c] : 0.f;
  }
  return bp;
}
AttenuationParams attenuation_params(const float X[12]) {
  AttenuationParams ap{};
  for (int c = 0; c < 3; ++c) {
    ap.a[c] = X ? X[c] : 0.f;
    ap.b[c] = X ? X[3 + c] : 0.f;
    ap.c[c] = X ? X[6 + c] : 0.f;
    ap.d[c] = X ? X[9 + c] : 0.f;
  }
  return ap;
}

}  // namespace

int pm_disp_to_range(pm_handle* h, const float* d_disp, int rows, int cols, double fx, double baseline,
                     float* d_range) {
  if (int rc = imaging_begin(h, "pm_disp_to_range", d_disp, d_range, rows, cols)) return rc;
  const size_t n = (size_t)rows * cols;
  hipLaunchKernelGGL(k_disp_to_range, stream_grid(n), dim3(256), 0, pm_internal::stream(h), d_disp, n, fx * baseline, d_range,
                     (unsigned*)nullptr);
  return launch_check(h, "disp_to_range");
}

int pm_remove_backscatter(pm_handle* h, const float* d_bgr, const float* d_range, int rows, int cols,
                          const float B[3], const float beta_B[3], float* d_out) {
  if (int rc = imaging_begin(h, "pm_remove_backscatter", d_bgr, d_range, rows, cols)) return rc;
  if (!B || !beta_B || !d_out) {
    set_err(h, "pm_remove_backscatter: null parameter");
    return PM_ERR_INVALID_ARG;
  }
  const size_t n = (size_t)rows * cols;
  const int vec = aligned16(d_bgr) && aligned16(d_range) && aligned16(d_out);
  hipLaunchKernelGGL((k_range_enhance<1>), stream_grid(n / 4 + 1), dim3(256), 0, pm_internal::stream(h), d_bgr, d_range, n, 0.0,
                     backscatter_params(B, beta_B), attenuation_params(nullptr), (const unsigned*)state_of(h)->img_scalars,
                     (float*)nullptr, d_out, vec);
  return launch_check(h, "remove_backscatter");
}

int pm_correct_attenuation(pm_handle* h, const float* d_bgr, const float* d_range, int rows, int cols,
                           const float X[12], float* d_out) {
  if (int rc = imaging_begin(h, "pm_correct_attenuation", d_bgr, d_range, rows, cols)) return rc;
  if (!X || !d_out) {
    set_err(h, "pm_correct_attenuation: null parameter");
    return PM_ERR_INVALID_ARG;
  }
  const size_t n = (size_t)rows * cols;
  PM_HIP(h, hipMemsetAsync(state_of(h)->img_scalars, 0, sizeof(unsigned), pm_internal::stream(h)));
  hipLaunchKernelGGL(k_range_max, reduce_grid(n), dim3(256), 0, pm_internal::stream(h), d_range, n, state_of(h)->img_scalars);
  const int vec = aligned16(d_bgr) && aligned16(d_range) && aligned16(d_out);
  hipLaunchKernelGGL((k_range_enhance<2>), stream_grid(n / 4 + 1), dim3(256), 0, pm_internal::stream(h), d_bgr, d_range, n, 0.0,
                     backscatter_params(nullptr, nullptr), attenuation_params(X), (const unsigned*)state_of(h)->img_scalars,
                     (float*)nullptr, d_out, vec);
  return launch_check(h, "correct_attenuation");
}

int pm_range_enhance(pm_handle* h, const float* d_bgr, const float* d_disp, int rows, int cols, double fx,
                     double baseline, const float B[3], const float beta_B[3], const float X[12],
                     float* d_range_out, float* d_out) {
  if (int rc = imaging_begin(h, "pm_range_enhance", d_bgr, d_disp, rows, cols)) return rc;
  if (!B || !beta_B || !X || !d_out) {
    set_err(h, "pm_range_enhance: null parameter");
    return PM_ERR_INVALID_ARG;
  }
  const size_t n = (size_t)rows * cols;
  // pass 1: the largest range (CorrectAttenuation gives it to pixels without range); reads the disparity only
  PM_HIP(h, hipMemsetD32Async((hipDeviceptr_t)state_of(h)->img_scalars, 0x7f800000u, 1, pm_internal::stream(h)));
  hipLaunchKernelGGL(k_disp_min_positive, reduce_grid(n / 4 + 1), dim3(256), 0, pm_internal::stream(h), d_disp, n, state_of(h)->img_scalars,
                     aligned16(d_disp) ? 1 : 0);
  const int vec = aligned16(d_bgr) && aligned16(d_disp) && aligned16(d_out) && (!d_range_out || aligned16(d_range_out));
  hipLaunchKernelGGL((k_range_enhance<7>), stream_grid(n / 4 + 1), dim3(256), 0, pm_internal::stream(h), d_bgr, d_disp, n,
                     fx * baseline, backscatter_params(B, beta_B), attenuation_params(X),
                     (const unsigned*)state_of(h)->img_scalars, d_range_out, d_out, vec);
  return launch_check(h, "range_enhance");
}

int pm_compute_intensity(pm_handle* h, const float* d_bgr, int rows, int cols, float* d_gray) {
  if (int rc = imaging_begin(h, "pm_compute_intensity", d_bgr, d_gray, rows, cols)) return rc;
  const size_t n = (size_t)rows * cols;
  hipLaunchKernelGGL(k_intensity, stream_grid(n), dim3(256), 0, pm_internal::stream(h), d_bgr, n, d_gray);
  return launch_check(h, "intensity");
}

int pm_find_dark(pm_handle* h, const float* d_intensity, const float* d_range, int rows, int cols, float percentile,
                 uint8_t* d_mask, float* threshold) {
  if (int rc = imaging_begin(h, "pm_find_dark", d_intensity, d_range, rows, cols)) return rc;
  if (!d_mask || !threshold) {
    set_err(h, "pm_find_dark: null output");
    return PM_ERR_INVALID_ARG;
  }
  const size_t n = (size_t)rows * cols;
  // backscatter.cpp:41-78
  const float N = (float)(rows * cols);
  const int n_desired = (int)(percentile * N);
  auto count_at = [&](float thr, unsigned* out) -> int {
    PM_HIP(h, hipMemsetAsync(state_of(h)->img_scalars + 1, 0, sizeof(unsigned), pm_internal::stream(h)));
    hipLaunchKernelGGL(k_dark_count, reduce_grid(n), dim3(256), 0, pm_internal::stream(h), d_intensity, d_range, n, thr, d_mask,
                       state_of(h)->img_scalars + 1);
    PM_HIP(h, hipMemcpyAsync(out, state_of(h)->img_scalars + 1, sizeof(unsigned), hipMemcpyDeviceToHost, pm_internal::stream(h)));
    PM_HIP(h, hipStreamSynchronize(pm_internal::stream(h)));
    return PM_OK;
  };
  float low = 0.f, high = 0.5f;
  const float first = (float)(1.5 * percentile);
  unsigned n_dark = 0;
  if (int rc = count_at(first, &n_dark)) return rc;
  if ((int)n_dark < n_desired) {
    low = first;
  } else if ((int)n_dark > n_desired) {
    high = first;
  } else {
    *threshold = first;
    return PM_OK;
  }
  for (int iter = 0; iter < 8; ++iter) {
    const float thr = (high + low) / 2.0f;
    if (int rc = count_at(thr, &n_dark)) return rc;
    if ((int)n_dark < n_desired) {
      low = thr;
    } else if ((int)n_dark > n_desired) {
      high = thr;
    } else {
      *threshold = thr;
      return PM_OK;
    }
  }
  *threshold = (high + low) / 2.0f;
  return PM_OK;
}

// ---- stereo-ready enhancement ---------------------------------------------------------------------
namespace {

// cv::getGaussianKernel(n, sigma, CV_32F), uploaded once per (n, sigma)
int ensure_taps(pm_handle* h, int ksize, double sigma) {
  if (state_of(h)->enh_ksize == ksize && state_of(h)->enh_sigma == sigma && state_of(h)->enh_taps) return PM_OK;
  if (ksize > state_of(h)->enh_taps_cap) {
    PM_HIP(h, hipStreamSynchronize(pm_internal::stream(h)));
    if (state_of(h)->enh_taps) PM_HIP(h, hipFree(state_of(h)->enh_taps));
    state_of(h)->enh_taps = nullptr;
    PM_HIP(h, hipMalloc((void**)&state_of(h)->enh_taps, sizeof(float) * (size_t)ksize));
    state_of(h)->enh_taps_cap = ksize;
  }
  std::vector<float> k((size_t)ksize);
  const double scale2x = -0.5 / (sigma * sigma);
  double sum = 0;
  for (int i = 0; i < ksize; ++i) {
    const double x = i - (ksize - 1) * 0.5;
    k[(size_t)i] = (float)std::exp(scale2x * x * x);
    sum += k[(size_t)i];
  }
  sum = 1. / sum;
  for (int i = 0; i < ksize; ++i) k[(size_t)i] = (float)(k[(size_t)i] * sum);
  PM_HIP(h, hipStreamSynchronize(pm_internal::stream(h)));  // the previous taps may still be in use
  PM_HIP(h, hipMemcpy(state_of(h)->enh_taps, k.data(), sizeof(float) * (size_t)ksize, hipMemcpyHostToDevice));
  state_of(h)->enh_ksize = ksize;
  state_of(h)->enh_sigma = sigma;
  return PM_OK;
}

int ensure_enh_scratch(pm_handle* h, size_t values) {
  if (values <= state_of(h)->enh_values) return PM_OK;
  PM_HIP(h, hipStreamSynchronize(pm_internal::stream(h)));
  if (state_of(h)->enh_tmp) PM_HIP(h, hipFree(state_of(h)->enh_tmp));
  if (state_of(h)->enh_q) PM_HIP(h, hipFree(state_of(h)->enh_q));
  state_of(h)->enh_tmp = state_of(h)->enh_q = nullptr;
  state_of(h)->enh_values = 0;
  PM_HIP(h, hipMalloc((void**)&state_of(h)->enh_tmp, sizeof(float) * values));
  PM_HIP(h, hipMalloc((void**)&state_of(h)->enh_q, sizeof(float) * values));
  state_of(h)->enh_values = values;
  return PM_OK;
}

// separable Gaussian, replicate border, of `count` <= kBlurBatch images of one size in one launch per pass; divide:
// dst = orig / (2 * blur) (the illuminant normalisation)
template <bool SRC_U8>
int run_gaussian_batch(pm_handle* h, const void* const* d_src, float* const* d_dst, int count, int rows, int cols, int ch,
                       int ksize, double sigma, bool divide) {
  if (ksize < 1 || (ksize % 2) == 0 || !(sigma > 0)) {
    set_err(h, "gaussian: ksize %d must be odd and sigma %g positive", ksize, sigma);
    return PM_ERR_INVALID_ARG;
  }
  if (count < 1 || count > kBlurBatch) {
    set_err(h, "gaussian: %d images in one batch (1 .. %d)", count, kBlurBatch);
    return PM_ERR_INVALID_ARG;
  }
  const size_t values = (size_t)rows * cols * ch;
  if (int rc = ensure_enh_scratch(h, values * (size_t)count)) return rc;
  if (int rc = ensure_taps(h, ksize, sigma)) return rc;
  const size_t row_lds = sizeof(float) * ((size_t)(blur_skew(kBlurRowPx + ksize - 1) + 1) * ch + ksize);
  if (row_lds > 64 * 1024) {
    set_err(h, "gaussian: kernel of %d taps x %d channels exceeds the row tile", ksize, ch);
    return PM_ERR_SIZE;
  }
  BlurBatch bb{};
  for (int i = 0; i < count; ++i) {
    bb.src[i] = d_src[i];
    bb.dst[i] = d_dst[i];
  }
  bb.tmp = state_of(h)->enh_tmp;
  bb.tmp_stride = values;
  const dim3 rgrid((unsigned)((cols + kBlurRowPx - 1) / kBlurRowPx), (unsigned)rows, (unsigned)count);
  const float* taps = state_of(h)->enh_taps;
  hipStream_t stream = pm_internal::stream(h);
  switch (ch) {
    case 1: hipLaunchKernelGGL((k_blur_rows<SRC_U8, 1>), rgrid, dim3(kBlurRowThreads), row_lds, stream, bb, rows, cols, ksize, taps); break;
    case 2: hipLaunchKernelGGL((k_blur_rows<SRC_U8, 2>), rgrid, dim3(kBlurRowThreads), row_lds, stream, bb, rows, cols, ksize, taps); break;
    case 3: hipLaunchKernelGGL((k_blur_rows<SRC_U8, 3>), rgrid, dim3(kBlurRowThreads), row_lds, stream, bb, rows, cols, ksize, taps); break;
    default: hipLaunchKernelGGL((k_blur_rows<SRC_U8, 4>), rgrid, dim3(kBlurRowThreads), row_lds, stream, bb, rows, cols, ksize, taps); break;
  }
  // column tile: W columns x T rows of outputs with T = (threads / (W / 2)) * 4 -- one group of four rows per thread --
  // and ((T + 2c + 3) x (W + 2) + c + 1) floats of LDS; the widest tile that fits 48 KB (wider = better coalesced fill)
  const int c = ksize / 2;
  int W = 0, T = 0, NT = 256;
  {
    static const int forced[3] = {[] {
      const char* e = pm::tune_env("PM_BLUR_COL");  // "W,T,threads" (experiments)
      int w = 0, t = 0, n = 0;
      if (e && sscanf(e, "%d,%d,%d", &w, &t, &n) == 3) return w;
      return 0;
    }(), [] {
      const char* e = pm::tune_env("PM_BLUR_COL");
      int w = 0, t = 0, n = 0;
      if (e && sscanf(e, "%d,%d,%d", &w, &t, &n) == 3) return t;
      return 0;
    }(), [] {
      const char* e = pm::tune_env("PM_BLUR_COL");
      int w = 0, t = 0, n = 0;
      if (e && sscanf(e, "%d,%d,%d", &w, &t, &n) == 3) return n;
      return 0;
    }()};
    auto lds_of = [&](int w, int t) { return sizeof(float) * ((size_t)(t + 2 * c + 3) * (w + 2) + c + 1); };
    if (forced[0]) {
      W = forced[0];
      T = forced[1];
      NT = forced[2];
    } else {
      for (int w : {32, 16, 8}) {
        const int t = (256 / (w / 2)) * 4;
        if (lds_of(w, t) <= 48 * 1024) {
          W = w;
          T = t;
          break;
        }
      }
    }
    if (!W || lds_of(W, T) > 64 * 1024) {
      set_err(h, "gaussian: kernel of %d taps exceeds the column tile", ksize);
      return PM_ERR_SIZE;
    }
  }
  const int width = cols * ch;
  const size_t col_lds = sizeof(float) * ((size_t)(T + 2 * c + 3) * (W + 2) + c + 1);
  const dim3 cgrid((unsigned)((width + W - 1) / W), (unsigned)((rows + T - 1) / T), (unsigned)count);
  if (divide)
    hipLaunchKernelGGL((k_blur_cols<true, SRC_U8>), cgrid, dim3(NT), col_lds, stream, bb, rows, width, ksize, taps, W, T);
  else
    hipLaunchKernelGGL((k_blur_cols<false, SRC_U8>), cgrid, dim3(NT), col_lds, stream, bb, rows, width, ksize, taps, W, T);
  return launch_check(h, "gaussian");
}
template <bool SRC_U8>
int run_gaussian(pm_handle* h, const void* d_src, int rows, int cols, int ch, int ksize, double sigma, bool divide,
                 float* d_dst) {
  return run_gaussian_batch<SRC_U8>(h, &d_src, &d_dst, 1, rows, cols, ch, ksize, sigma, divide);
}

// imaging::Normalize on d_q -> J and / or gray8
int run_normalize(pm_handle* h, const float* d_q, int rows, int cols, float* d_J, uint8_t* d_gray8) {
  if (rows < 8 || cols < 8) {
    set_err(h, "normalize: the image must be at least 8x8 (its 1/8 resize would be empty)");
    return PM_ERR_INVALID_ARG;
  }
  const unsigned init[2] = {0x7f7fffffu, 0u};
  PM_HIP(h, hipMemcpyAsync(state_of(h)->img_scalars + 2, init, sizeof(init), hipMemcpyHostToDevice, pm_internal::stream(h)));
  const size_t small = (size_t)(rows / 8) * (cols / 8);
  hipLaunchKernelGGL(k_value_minmax, reduce_grid(small), dim3(256), 0, pm_internal::stream(h), d_q, rows, cols, state_of(h)->img_scalars + 2);
  const size_t n = (size_t)rows * cols;
  hipLaunchKernelGGL(k_normalize_gray, stream_grid(n), dim3(256), 0, pm_internal::stream(h), d_q, n,
                     (const unsigned*)(state_of(h)->img_scalars + 2), d_J, d_gray8);
  return launch_check(h, "normalize");
}

}  // namespace

int pm_gaussian_blur(pm_handle* h, const float* d_src, int rows, int cols, int channels, int ksize, double sigma,
                     float* d_dst) {
  if (int rc = imaging_begin(h, "pm_gaussian_blur", d_src, d_dst, rows, cols)) return rc;
  if (channels < 1 || channels > 4) {
    set_err(h, "pm_gaussian_blur: %d channels", channels);
    return PM_ERR_INVALID_ARG;
  }
  return run_gaussian<false>(h, d_src, rows, cols, channels, ksize, sigma, false, d_dst);
}

int pm_normalize(pm_handle* h, const float* d_bgr, int rows, int cols, float* d_out) {
  if (int rc = imaging_begin(h, "pm_normalize", d_bgr, d_out, rows, cols)) return rc;
  return run_normalize(h, d_bgr, rows, cols, d_out, nullptr);
}

int pm_stereo_ready(pm_handle* h, const uint8_t* d_bgr8, int rows, int cols, float* d_J, uint8_t* d_gray8) {
  if (int rc = imaging_begin(h, "pm_stereo_ready", d_bgr8, d_bgr8, rows, cols)) return rc;
  if (!d_J && !d_gray8) {
    set_err(h, "pm_stereo_ready: no output requested");
    return PM_ERR_INVALID_ARG;
  }
  // NormalizeColorIlluminant (normalization.cpp:178-185): ksize = NextOddInt(cols / 3), sigma = (float)ksize / 4
  const int third = cols / 3;
  const int ksize = third + (1 - third % 2);
  const double sigma = (double)((float)ksize / 4.0f);
  if (int rc = ensure_enh_scratch(h, (size_t)rows * cols * 3)) return rc;
  if (int rc = run_gaussian<true>(h, d_bgr8, rows, cols, 3, ksize, sigma, true, state_of(h)->enh_q)) return rc;
  // enhance_test.cpp:69 applies Normalize to NormalizeColorIlluminant's result, which already ends with a
  // Normalize (normalization.cpp:184): two value stretches.  The row-pass scratch is free again: it takes the first.
  if (int rc = run_normalize(h, state_of(h)->enh_q, rows, cols, state_of(h)->enh_tmp, nullptr)) return rc;
  return run_normalize(h, state_of(h)->enh_tmp, rows, cols, d_J, d_gray8);
}

// Match() on BGR inputs with the stereo-ready enhancement folded into the load path (BASELINE configs[4]): per image the
// two Gaussian passes (illuminant estimate) and the two tiny min / max passes of the value stretches; everything per
// pixel happens inside k_prep_bgr.  Results equal pm_stereo_ready x 2 followed by pm_match_device bit for bit.
int pm_match_bgr_device(pm_handle* h, int n, const uint8_t* d_left_bgr8, const uint8_t* d_right_bgr8, int rows, int cols,
                        const float* d_seed_l, const float* d_seed_r, float* d_disp_l, float* d_disp_r) {
  if (int rc = imaging_begin(h, "pm_match_bgr_device", d_left_bgr8, d_right_bgr8, rows, cols)) return rc;
  if (n < 1 || rows < 8 || cols < 8) {
    set_err(h, "pm_match_bgr_device: n >= 1 pairs of at least 8x8 pixels");
    return PM_ERR_INVALID_ARG;
  }
  ImagingState* st = state_of(h);
  const size_t ipx = (size_t)rows * cols, values = ipx * 3;
  if (st->bgr_values < values * 2 * (size_t)n) {
    if (st->bgr_blur) PM_HIP(h, hipFree(st->bgr_blur));
    st->bgr_blur = nullptr;
    st->bgr_values = 0;
    PM_HIP(h, hipMalloc((void**)&st->bgr_blur, sizeof(float) * values * 2 * (size_t)n));
    st->bgr_values = values * 2 * (size_t)n;
  }
  if (st->bgr_pairs < n) {
    if (st->bgr_mm) PM_HIP(h, hipFree(st->bgr_mm));
    st->bgr_mm = nullptr;
    st->bgr_pairs = 0;
    PM_HIP(h, hipMalloc((void**)&st->bgr_mm, sizeof(unsigned) * 8 * (size_t)n));
    st->bgr_pairs = n;
  }
  // NormalizeColorIlluminant (normalization.cpp:178-185): ksize = NextOddInt(cols / 3), sigma = (float)ksize / 4
  const int third = cols / 3;
  const int ksize = third + (1 - third % 2);
  const double sigma = (double)((float)ksize / 4.0f);
  std::vector<unsigned> init((size_t)n * 8);
  for (size_t i = 0; i < init.size(); i += 2) {
    init[i] = 0x7f7fffffu;  // min
    init[i + 1] = 0u;       // max
  }
  PM_HIP(h, hipMemcpyAsync(st->bgr_mm, init.data(), sizeof(unsigned) * init.size(), hipMemcpyHostToDevice, pm_internal::stream(h)));
  float* blur_l = st->bgr_blur;
  float* blur_r = st->bgr_blur + values * (size_t)n;
  const size_t small = (size_t)(rows / 8) * (cols / 8);
  // image z = b * 2 + i (pair b, left / right); up to kBlurBatch images per launch of each pass
  for (int z0 = 0; z0 < 2 * n; z0 += kBlurBatch) {
    const int count = 2 * n - z0 < kBlurBatch ? 2 * n - z0 : kBlurBatch;
    const void* srcs[kBlurBatch];
    float* dsts[kBlurBatch];
    BlurBatch bb{};
    for (int k = 0; k < count; ++k) {
      const int b = (z0 + k) >> 1, i = (z0 + k) & 1;
      srcs[k] = (i == 0 ? d_left_bgr8 : d_right_bgr8) + (size_t)b * values;
      dsts[k] = (i == 0 ? blur_l : blur_r) + (size_t)b * values;
      bb.src[k] = srcs[k];
      bb.dst[k] = dsts[k];
    }
    if (int rc = run_gaussian_batch<true>(h, srcs, dsts, count, rows, cols, 3, ksize, sigma, false)) return rc;
    dim3 mgrid = reduce_grid(small);
    mgrid.y = (unsigned)count;
    unsigned* mm = st->bgr_mm + (size_t)z0 * 4;
    hipLaunchKernelGGL((k_value_minmax_fused<1>), mgrid, dim3(256), 0, pm_internal::stream(h), bb, rows, cols, mm);
    hipLaunchKernelGGL((k_value_minmax_fused<2>), mgrid, dim3(256), 0, pm_internal::stream(h), bb, rows, cols, mm);
  }
  if (int rc = launch_check(h, "value min / max")) return rc;
  BgrSource src;
  src.left = d_left_bgr8;
  src.right = d_right_bgr8;
  src.blur_l = blur_l;
  src.blur_r = blur_r;
  src.mm = st->bgr_mm;
  pm_internal::set_bgr_source(h, &src);
  // the image pointers below only have to be non-null: the prep stage reads `src`
  const int rc = pm_match_device(h, n, d_left_bgr8, d_right_bgr8, rows, cols, d_seed_l, d_seed_r, d_disp_l, d_disp_r);
  pm_internal::set_bgr_source(h, nullptr);
  return rc;
}

int pm_normalize_color_illuminant(pm_handle* h, const float* d_bgr, int rows, int cols, float* d_out) {
  if (int rc = imaging_begin(h, "pm_normalize_color_illuminant", d_bgr, d_out, rows, cols)) return rc;
  const int third = cols / 3;
  const int ksize = third + (1 - third % 2);
  const double sigma = (double)((float)ksize / 4.0f);
  if (int rc = ensure_enh_scratch(h, (size_t)rows * cols * 3)) return rc;
  if (int rc = run_gaussian<false>(h, d_bgr, rows, cols, 3, ksize, sigma, true, state_of(h)->enh_q)) return rc;
  return run_normalize(h, state_of(h)->enh_q, rows, cols, d_out, nullptr);
}

int pm_device_malloc(pm_handle* h, size_t bytes, void** d_ptr) {
  if (!h || !d_ptr) return PM_ERR_INVALID_ARG;
  PM_HIP(h, hipSetDevice(pm_internal::device(h)));
  *d_ptr = nullptr;
  if (hipMalloc(d_ptr, bytes ? bytes : 1) != hipSuccess) {
    set_err(h, "pm_device_malloc: %zu bytes", bytes);
    return PM_ERR_NOMEM;
  }
  return PM_OK;
}

int pm_device_free(pm_handle* h, void* d_ptr) {
  if (!h) return PM_ERR_INVALID_ARG;
  if (!d_ptr) return PM_OK;
  PM_HIP(h, hipSetDevice(pm_internal::device(h)));
  PM_HIP(h, hipStreamSynchronize(pm_internal::stream(h)));
  PM_HIP(h, hipFree(d_ptr));
  return PM_OK;
}

int pm_upload(pm_handle* h, void* d_dst, const void* src, size_t bytes) {
  if (!h || !d_dst || !src) return PM_ERR_INVALID_ARG;
  PM_HIP(h, hipSetDevice(pm_internal::device(h)));
  PM_HIP(h, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, pm_internal::stream(h)));
  PM_HIP(h, hipStreamSynchronize(pm_internal::stream(h)));  // pageable source: the caller may reuse it right away
  return PM_OK;
}

int pm_download(pm_handle* h, void* dst, const void* d_src, size_t bytes) {
  if (!h || !dst || !d_src) return PM_ERR_INVALID_ARG;
  PM_HIP(h, hipSetDevice(pm_internal::device(h)));
  PM_HIP(h, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, pm_internal::stream(h)));
  PM_HIP(h, hipStreamSynchronize(pm_internal::stream(h)));
  return PM_OK;
}
