// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths THIS engine uses
// (/opt/skills/guides/MI355X_MICROARCH.md, HBM section: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced
// streaming read (16 B/lane) ... other access widths are uncalibrated: calibrate on a known byte count in your own access
// pattern").  Every kernel below moves a KNOWN number of bytes of a 512 MiB buffer (beyond L2 + Infinity Cache) exactly
// once; tools/fetch_calib.sh runs it under --pmc FETCH_SIZE and --pmc WRITE_SIZE and divides.
//   rd<2> / rd<4> / rd<8> / rd<16>   coalesced streaming reads of 2 / 4 / 8 / 16 bytes per lane
//   rd_rec16                         16-byte records in runs of 16 lanes at unrelated places (a sweep step's gather)
//   wr<4> / wr<16>                   coalesced streaming writes
//   wr_strided4                      one dword per lane, every fourth word of a stretch (a row chain's state write-back)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int W> struct T;
template <> struct T<2> { typedef unsigned short t; };
template <> struct T<4> { typedef unsigned t; };
template <> struct T<8> { typedef u32x2 t; };
template <> struct T<16> { typedef u32x4 t; };
__device__ inline unsigned fold(unsigned short v) { return v; }
__device__ inline unsigned fold(unsigned v) { return v; }
__device__ inline unsigned fold(u32x2 v) { return v.x ^ v.y; }
__device__ inline unsigned fold(u32x4 v) { return v.x ^ v.w; }

template <int W>
__global__ void rd(const char* buf, size_t n_elems, unsigned* out) {
  typedef typename T<W>::t E;
  const E* p = (const E*)buf;
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_elems; i += (size_t)gridDim.x * blockDim.x) acc ^= fold(p[i]);
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void rd_rec16(const char* buf, size_t n_recs, unsigned* out) {
  // wave w, iteration i: four runs of 16 consecutive records, the runs 1 MiB apart; every record read exactly once
  const u32x4* p = (const u32x4*)buf;
  const size_t quarter = n_recs / 4;
  const int lane = threadIdx.x & 63, run = lane >> 4, gl = lane & 15;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = ((size_t)gridDim.x * blockDim.x) >> 6;
  unsigned acc = 0;
  for (size_t i = wave; i * 16 + 15 < quarter; i += waves) acc ^= fold(p[(size_t)run * quarter + i * 16 + gl]);
  if (acc == 0x12345678u) out[0] = acc;
}
template <int W>
__global__ void wr(char* buf, size_t n_elems) {
  typedef typename T<W>::t E;
  E* p = (E*)buf;
  E v{};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_elems; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void wr_strided4(char* buf, size_t n_words) {  // every fourth dword
  unsigned* p = (unsigned*)buf;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i * 4 < n_words; i += (size_t)gridDim.x * blockDim.x) p[i * 4] = 0u;
}

int main() {
  const size_t bytes = (size_t)512 << 20;
  char* d;
  unsigned* out;
  if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
  (void)hipMemset(d, 1, bytes);
  (void)hipDeviceSynchronize();
  const dim3 grid(256 * 16), block(256);
  rd<4><<<grid, block>>>(d, bytes / 4, out);  // (a warm-up dispatch; rd<4> appears twice and is averaged per dispatch)
  (void)hipDeviceSynchronize();
  rd<2><<<grid, block>>>(d, bytes / 2, out);
  rd<4><<<grid, block>>>(d, bytes / 4, out);
  rd<8><<<grid, block>>>(d, bytes / 8, out);
  rd<16><<<grid, block>>>(d, bytes / 16, out);
  rd_rec16<<<grid, block>>>(d, bytes / 16, out);
  wr<4><<<grid, block>>>(d, bytes / 4);
  wr<16><<<grid, block>>>(d, bytes / 16);
  wr_strided4<<<grid, block>>>(d, bytes / 4);
  (void)hipDeviceSynchronize();
  printf("bytes moved per kernel: rd* %zu, rd_rec16 %zu, wr<4>/wr<16> %zu, wr_strided4 %zu (dwords written; lines touched %zu)\n",
         bytes, (bytes / 16 / 4 / 16) * 16 * 4 * 16, bytes, bytes / 4, bytes);
  return 0;
}
