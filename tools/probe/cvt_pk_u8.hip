// Probe: does v_cvt_pk_u8_f32 equal saturate_cast<uchar>(float) = clamp(rint(x), 0, 255) (ties to even)?
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float* in, unsigned* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0, 0u);
}
int main() {
  std::vector<float> h;
  for (int i = -8; i < 260 * 16; ++i) h.push_back(i / 16.0f);           // all multiples of 1/16 incl. every .5 tie
  for (int i = 0; i < 256; ++i) { h.push_back(std::nextafterf(i + 0.5f, 0.f)); h.push_back(std::nextafterf(i + 0.5f, 1e9f)); }
  h.push_back(-1e30f); h.push_back(1e30f); h.push_back(-0.f); h.push_back(255.5f); h.push_back(254.5f);
  int n = (int)h.size();
  float* d_in; unsigned* d_out;
  hipMalloc(&d_in, n * 4); hipMalloc(&d_out, n * 4);
  hipMemcpy(d_in, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, d_in, d_out, n);
  std::vector<unsigned> o(n);
  hipMemcpy(o.data(), d_out, n * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i) {
    float r = nearbyintf(h[i]);
    int want = r < 0 ? 0 : (r > 255 ? 255 : (int)r);
    if ((int)(o[i] & 0xff) != want) { if (bad < 10) printf("x=%.9g got %u want %d\n", h[i], o[i] & 0xff, want); ++bad; }
  }
  printf("cvt_pk_u8_f32: %d of %d differ from clamp(rint(x),0,255)\n", bad, n);
  return 0;
}
