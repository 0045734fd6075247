// pm_seed.hip -- translation unit of the device seeder (pm_seed.hpp: kernels and their launch sequence).
#include <cmath>
#include <vector>

#include "pm_seed.hpp"

namespace pm {

hipError_t seed_scratch_alloc(SeedScratch& sc, size_t plane_elems, hipStream_t stream) {
  hipError_t e;
  sc.cap = (int)(plane_elems + 64);
  if ((e = hipMalloc((void**)&sc.eig, sizeof(float) * plane_elems)) != hipSuccess) return e;
  if ((e = hipMalloc((void**)&sc.keys, sizeof(unsigned long long) * sc.cap)) != hipSuccess) return e;
  if ((e = hipMalloc((void**)&sc.keys_sorted, sizeof(unsigned long long) * sc.cap)) != hipSuccess) return e;
  if ((e = hipMalloc((void**)&sc.counters, sizeof(unsigned) * kSeedCounters)) != hipSuccess) return e;
  if ((e = hipMalloc((void**)&sc.kp_xy, sizeof(int) * 2 * kSeedMaxFeatures)) != hipSuccess) return e;
  if ((e = hipMalloc((void**)&sc.kp_d, sizeof(float) * kSeedMaxFeatures)) != hipSuccess) return e;
  if ((e = hipMalloc((void**)&sc.kp_f, sizeof(float) * 2 * kSeedMaxFeatures)) != hipSuccess) return e;
  sc.sp_buf = nullptr;   // allocated with the masks, only for handles that ask for cornerSubPix (seed_subpix_prepare)
  sc.sp_mask = nullptr;
  sc.sp_mask_win = sc.sp_mask_zero = 0;
  sc.sort_tmp = nullptr;
  sc.sort_tmp_bytes = 0;
  sc.counters_clean = false;
  if ((e = hipcub::DeviceRadixSort::SortKeysDescending(nullptr, sc.sort_tmp_bytes, sc.keys, sc.keys_sorted, sc.cap, 0, 64,
                                                       stream)) != hipSuccess)
    return e;
  return hipMalloc(&sc.sort_tmp, sc.sort_tmp_bytes);
}

// cornerSubPix needs its two window masks (computed on the HOST: std::exp of the C library, the function the oracle and
// OpenCV call -- the device's expf may differ in the last place) and a neighbourhood buffer per corner.
hipError_t seed_subpix_prepare(SeedScratch& sc, const SeedParams& sp, hipStream_t stream) {
  if (!sp.subpixel_corners && !sp.subpixel_refinement) return hipSuccess;
  hipError_t e;
  if (!sc.sp_buf) {
    const size_t side = 2 * kSubpixMaxWin + 3;
    if ((e = hipMalloc((void**)&sc.sp_buf, sizeof(float) * side * side * kSeedMaxFeatures)) != hipSuccess) return e;
    if ((e = hipMalloc((void**)&sc.sp_mask, sizeof(float) * 2 * kSubpixMaskStride)) != hipSuccess) return e;
    sc.sp_mask_win = 0;
  }
  if (sc.sp_mask_win == sp.subpix_winsize && sc.sp_mask_zero == sp.subpix_zerozone) return hipSuccess;
  std::vector<float> m(2 * kSubpixMaskStride, 0.f);
  auto fill = [&](float* mask, int win, int zero_zone) {
    const int ww = 2 * win + 1;
    for (int i = 0; i < ww; ++i) {
      const float y = (float)(i - win) / (float)win;
      const float vy = std::exp(-y * y);
      for (int j = 0; j < ww; ++j) {
        const float x = (float)(j - win) / (float)win;
        mask[i * ww + j] = (float)(vy * std::exp(-x * x));
      }
    }
    if (zero_zone >= 0 && zero_zone * 2 + 1 < ww)
      for (int i = win - zero_zone; i <= win + zero_zone; ++i)
        for (int j = win - zero_zone; j <= win + zero_zone; ++j) mask[i * ww + j] = 0.f;
  };
  fill(m.data(), sp.subpix_winsize, sp.subpix_zerozone);
  fill(m.data() + kSubpixMaskStride, kSubpixMatchWin, -1);
  if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
  if ((e = hipMemcpy(sc.sp_mask, m.data(), sizeof(float) * m.size(), hipMemcpyHostToDevice)) != hipSuccess) return e;
  sc.sp_mask_win = sp.subpix_winsize;
  sc.sp_mask_zero = sp.subpix_zerozone;
  return hipSuccess;
}

void seed_scratch_free(SeedScratch& sc) {
  void* dev[] = {sc.eig, sc.keys, sc.keys_sorted, sc.counters, sc.kp_xy, sc.kp_d, sc.kp_f, sc.sp_buf, sc.sp_mask, sc.sort_tmp};
  for (void* p : dev)
    if (p) (void)hipFree(p);
  sc = SeedScratch{};
}

}  // namespace pm
