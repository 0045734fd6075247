// pm_seed.hip -- translation unit of the device seeder (pm_seed.hpp: kernels and their launch sequence).
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
  sc.sort_tmp = nullptr;
  sc.sort_tmp_bytes = 0;
  if ((e = hipcub::DeviceRadixSort::SortKeysDescending(nullptr, sc.sort_tmp_bytes, sc.keys, sc.keys_sorted, sc.cap, 0, 64,
                                                       stream)) != hipSuccess)
    return e;
  return hipMalloc(&sc.sort_tmp, sc.sort_tmp_bytes);
}

void seed_scratch_free(SeedScratch& sc) {
  void* dev[] = {sc.eig, sc.keys, sc.keys_sorted, sc.counters, sc.kp_xy, sc.kp_d, sc.sort_tmp};
  for (void* p : dev)
    if (p) (void)hipFree(p);
  sc = SeedScratch{};
}

}  // namespace pm
