// Microbenchmark: what does the vector memory pipeline charge for a wave64 load of 16-byte (or 12-byte) records at
// lane stride = record size, with the first lane on / off a 64-byte boundary?  tools/probe/tcp_align (round 3).
// Each wavefront streams through an L2-resident buffer (no reuse inside a wavefront); all CUs busy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct R12 { unsigned a, b, c; };

template <int REC, int SHIFT, int GROUPS>
__global__ void k(const char* buf, size_t bytes, unsigned* out, int iters) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  // GROUPS contiguous runs per wavefront (16-lane groups at unrelated places when GROUPS = 4)
  const int gl = lane % (64 / GROUPS), grp = lane / (64 / GROUPS);
  unsigned acc = 0;
  size_t pos = ((size_t)wave * 7919u * 4096u) % (bytes - 65536);
  for (int i = 0; i < iters; ++i) {
    const size_t base = ((pos + (size_t)grp * 16384u) & ~(size_t)63) + (size_t)SHIFT * REC;
    const char* p = buf + base + (size_t)gl * REC;
    if (REC == 16) {
      const u32x4 t = *(const u32x4*)p;
      acc += t.x ^ t.w;
    } else {
      const R12 t = *(const R12*)p;
      acc += t.a ^ t.c;
    }
    pos = (pos + 4096u * 13u) % (bytes - 65536);
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int REC, int SHIFT, int GROUPS>
float run(const char* d, size_t bytes, unsigned* out, int iters) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  k<REC, SHIFT, GROUPS><<<256 * 8, 256>>>(d, bytes, out, 16);
  hipEventRecord(a);
  k<REC, SHIFT, GROUPS><<<256 * 8, 256>>>(d, bytes, out, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double loads = 256.0 * 8 * 4 * iters;  // wave-level load instructions
  printf("rec %2d B, first lane %d records off a 64-byte line, %d run(s) per wave: %.3f ms, %.1f ns per wave-load per CU-slot, %.1f GB/s\n",
         REC, SHIFT, GROUPS, ms, ms * 1e6 / loads * 256, loads * 64 * REC / ms / 1e6);
  return ms;
}

int main() {
  const size_t bytes = 24u << 20;  // sits in L2 + MALL
  char* d;
  unsigned* out;
  hipMalloc(&d, bytes);
  hipMalloc(&out, 64);
  hipMemset(d, 1, bytes);
  const int it = 4000;
  run<16, 0, 1>(d, bytes, out, it);
  run<16, 1, 1>(d, bytes, out, it);
  run<16, 2, 1>(d, bytes, out, it);
  run<16, 0, 4>(d, bytes, out, it);
  run<16, 1, 4>(d, bytes, out, it);
  run<16, 3, 4>(d, bytes, out, it);
  run<12, 0, 1>(d, bytes, out, it);
  run<12, 1, 1>(d, bytes, out, it);
  run<12, 0, 4>(d, bytes, out, it);
  run<12, 3, 4>(d, bytes, out, it);
  return 0;
}
