// pm_imaging.hpp -- per-pixel range-dependent post-processing (include/pm/imaging.h).
// All kernels are streaming passes: one read of each input, one write of each output, 16-byte vector
// accesses where the layout allows (interleaved BGR float = 12 B per pixel, so a lane takes 4 pixels =
// three float4).  Bound: HBM.  Float arithmetic in the reference's operation order, no FMA.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace pm {

struct BackscatterParams {  // imaging::RemoveBackscatter (backscatter.cpp:277-308)
  float B[3], beta_B[3];
};
struct AttenuationParams {  // imaging::CorrectAttenuation (attenuation.cpp:269-299): X = a, b, c, d per channel
  float a[3], b[3], c[3], d[3];
};

// StereoCamera::DispToDepth (stereo_camera.cpp:49-53): fx * Baseline() / disp in double.
__device__ __forceinline__ float disp_to_range(float disp, double fxb) {
  return disp > 0.f ? (float)(fxb / (double)disp) : 0.f;
}

// One channel of RemoveBackscatter.  The reference builds it from cv::Mat expressions, which evaluate as
//   z   = range + (range > 1e-3 ? 0 : 20)                  cv::threshold BINARY_INV, backscatter.cpp:286-288
//   e   = exp(z * (-beta))                                  scaled copy, cv::exp                 :290-293
//   bs  = e * (-B) + B                                      B * (1 - e) folded into one scale + shift :295-297
//   out = max(I - bs, 0)                                    :299-305
__device__ __forceinline__ float remove_backscatter_1(float I, float z, float B, float beta) {
  const float e = expf(z * (-beta));
  const float bs = e * (-B) + B;
  const float o = I - bs;
  return o > 0.f ? o : 0.f;
}
__device__ __forceinline__ float backscatter_range(float range) { return range > 1e-3f ? range : range + 20.0f; }

// One channel of CorrectAttenuation (attenuation.cpp:285-298):
//   beta_cz = z * (a * exp(z * b) + c * exp(z * d));  out = I * exp(beta_cz)
__device__ __forceinline__ float correct_attenuation_1(float I, float z, float a, float b, float c, float d) {
  const float e1 = expf(z * b), e2 = expf(z * d);
  const float w = e1 * a + e2 * c;  // cv::addWeighted form of a * M1 + c * M2
  const float bz = z * w;
  return I * expf(bz);
}
// SetMaxRangeWhereZero (attenuation.cpp:255-266): range + (range > 0 ? 0 : rmax)
__device__ __forceinline__ float attenuation_range(float range, float rmax) { return range > 0.f ? range : range + rmax; }

// Block reductions: one same-address atomic per BLOCK (per-wave atomics on one address serialise: the first
// version of the fused pass spent 190 of its 250 us there), skipped when the value cannot change the result.
__device__ __forceinline__ float block_max(float m) {
  __shared__ float s[4];
#pragma unroll
  for (int ofs = 32; ofs > 0; ofs >>= 1) m = fmaxf(m, __shfl_xor(m, ofs, 64));
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = m;
  __syncthreads();
  return fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
}
__device__ __forceinline__ float block_min(float m) {
  __shared__ float s[4];
#pragma unroll
  for (int ofs = 32; ofs > 0; ofs >>= 1) m = fminf(m, __shfl_xor(m, ofs, 64));
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = m;
  __syncthreads();
  return fminf(fminf(s[0], s[1]), fminf(s[2], s[3]));
}
__device__ __forceinline__ unsigned block_sum(unsigned c) {
  __shared__ unsigned s[4];
#pragma unroll
  for (int ofs = 32; ofs > 0; ofs >>= 1) c += __shfl_xor(c, ofs, 64);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = c;
  __syncthreads();
  return s[0] + s[1] + s[2] + s[3];
}

// ---- max of a non-negative float map (bit pattern order == value order) -> *out (zeroed by the caller)
__global__ void __launch_bounds__(256) k_range_max(const float* __restrict__ range, size_t n, unsigned* out) {
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float r = range[i];
    m = r > m ? r : m;
  }
  m = block_max(m);
  if (threadIdx.x == 0 && m > __uint_as_float(*(volatile unsigned*)out)) atomicMax(out, __float_as_uint(m));
}

// disparity -> range, optionally with the running max for CorrectAttenuation
__global__ void __launch_bounds__(256) k_disp_to_range(const float* __restrict__ disp, size_t n, double fxb,
                                                       float* __restrict__ range, unsigned* max_out) {
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float r = disp_to_range(disp[i], fxb);
    range[i] = r;
    m = r > m ? r : m;
  }
  if (max_out) {
    m = block_max(m);
    if (threadIdx.x == 0 && m > __uint_as_float(*(volatile unsigned*)max_out)) atomicMax(max_out, __float_as_uint(m));
  }
}

// Smallest positive disparity of a map (pass 1 of the fused path: reads 4 B per pixel, no division).
// fx*b/d is decreasing in d and rounding is monotone, so max(range) = (float)(fx*b / min positive disparity)
// exactly.  *out starts as +inf (0x7f800000); positive floats order like their bit patterns.
__global__ void __launch_bounds__(256) k_disp_min_positive(const float* __restrict__ disp, size_t n, unsigned* out,
                                                           int vec_ok) {
  float m = __uint_as_float(0x7f800000u);
  auto take = [&](float d) { m = (d > 0.f && d < m) ? d : m; };
  const size_t n4 = vec_ok ? n / 4 : 0;
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (size_t)gridDim.x * blockDim.x) {
    const float4 v = ((const float4*)disp)[q];
    take(v.x);
    take(v.y);
    take(v.z);
    take(v.w);
  }
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    take(disp[i]);
  m = block_min(m);
  if (threadIdx.x == 0 && m < __uint_as_float(*(volatile unsigned*)out)) atomicMin(out, __float_as_uint(m));
}

// MODE bit 0: RemoveBackscatter, bit 1: CorrectAttenuation (on the result of bit 0 if both), bit 2: the range
// comes from a disparity map (DispToDepth) and is optionally written out.  One pixel per lane iteration; the
// three channels of a pixel are 12 contiguous bytes, four pixels per lane make three aligned float4.
template <int MODE>
__global__ void __launch_bounds__(256) k_range_enhance(const float* __restrict__ bgr, const float* __restrict__ rng_or_disp,
                                                       size_t n_px, double fxb, BackscatterParams bp,
                                                       AttenuationParams ap, const unsigned* __restrict__ rmax_bits,
                                                       float* __restrict__ range_out, float* __restrict__ out,
                                                       int vec_ok) {
  // MODE & 4: the scalar is the smallest positive disparity (k_disp_min_positive), else the largest range
  float rmax = 0.f;
  if (MODE & 2) {
    const unsigned sb = *rmax_bits;
    rmax = (MODE & 4) ? (sb == 0x7f800000u ? 0.f : disp_to_range(__uint_as_float(sb), fxb)) : __uint_as_float(sb);
  }
  const size_t n4 = vec_ok ? n_px / 4 : 0;  // vec_ok: every pointer is 16-byte aligned
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (size_t)gridDim.x * blockDim.x) {
    const float4 r4 = ((const float4*)rng_or_disp)[q];
    const float4* src = (const float4*)bgr + q * 3;
    const float4 v0 = src[0], v1 = src[1], v2 = src[2];
    float px[12] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w};
    float rr[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float range = (MODE & 4) ? disp_to_range(rr[k], fxb) : rr[k];
      rr[k] = range;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float I = px[3 * k + c];
        if (MODE & 1) I = remove_backscatter_1(I, backscatter_range(range), bp.B[c], bp.beta_B[c]);
        if (MODE & 2) I = correct_attenuation_1(I, attenuation_range(range, rmax), ap.a[c], ap.b[c], ap.c[c], ap.d[c]);
        px[3 * k + c] = I;
      }
    }
    float4* dst = (float4*)out + q * 3;
    dst[0] = make_float4(px[0], px[1], px[2], px[3]);
    dst[1] = make_float4(px[4], px[5], px[6], px[7]);
    dst[2] = make_float4(px[8], px[9], px[10], px[11]);
    if ((MODE & 4) && range_out) ((float4*)range_out)[q] = make_float4(rr[0], rr[1], rr[2], rr[3]);
  }
  // tail (n_px not a multiple of 4): scalar
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_px; i += (size_t)gridDim.x * blockDim.x) {
    const float range = (MODE & 4) ? disp_to_range(rng_or_disp[i], fxb) : rng_or_disp[i];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float I = bgr[i * 3 + c];
      if (MODE & 1) I = remove_backscatter_1(I, backscatter_range(range), bp.B[c], bp.beta_B[c]);
      if (MODE & 2) I = correct_attenuation_1(I, attenuation_range(range, rmax), ap.a[c], ap.b[c], ap.c[c], ap.d[c]);
      out[i * 3 + c] = I;
    }
    if ((MODE & 4) && range_out) range_out[i] = range;
  }
}

// ComputeIntensity: cv::cvtColor BGR2GRAY on floats = b * 0.114f + g * 0.587f + r * 0.299f
__global__ void __launch_bounds__(256) k_intensity(const float* __restrict__ bgr, size_t n_px, float* __restrict__ gray) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_px; i += (size_t)gridDim.x * blockDim.x) {
    const float b = bgr[i * 3], g = bgr[i * 3 + 1], r = bgr[i * 3 + 2];
    float s = b * 0.114f;
    s = s + g * 0.587f;
    s = s + r * 0.299f;
    gray[i] = s;
  }
}

// One counting step of FindDarkFast: mask = (intensity <= thr) & (range > 0.1); *count += popcount
__global__ void __launch_bounds__(256) k_dark_count(const float* __restrict__ intensity, const float* __restrict__ range,
                                                    size_t n, float thr, uint8_t* __restrict__ mask,
                                                    unsigned* count) {
  unsigned c = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const bool dark = (intensity[i] <= thr) && (range[i] > 0.1f);
    mask[i] = dark ? 255 : 0;
    c += dark ? 1u : 0u;
  }
  c = block_sum(c);
  if (threadIdx.x == 0 && c) atomicAdd(count, c);
}

}  // namespace pm
