// pm_sweep_defs.hpp -- types and helpers shared by the sweep kernels (pm_sweeps.hip) and their caller (pm_engine.hip).
#pragma once

#include "pm_device.hpp"

namespace pm {

constexpr int kWave = 64;  // CDNA4 wavefront

// The chain engines keep a whole chain (4 floats per position) in LDS.  Up to 64 KB of dynamic LDS is available
// by default; beyond that the kernel needs its limit raised (160 KB per CU on gfx950: chains of up to ~10 000
// positions).  run_sweep falls back to the serial engine for longer chains.
constexpr size_t kChainLdsMax = 160 * 1024 - 1024;
inline size_t chain_lds_bytes(int n, int extra_words, int planes = 4) {
  const int n1 = (n + 1 + 3) & ~3;
  return sizeof(float) * ((size_t)planes * (size_t)n1 + (size_t)extra_words);
}
template <typename K>
inline void allow_big_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

struct SweepGeom {
  int axis;           // 0 = along a row, 1 = along a column
  int dir;            // +1 / -1
  int c_lo, c_hi;     // chains (inclusive)
  int s_first, s_last;  // first and last visited position along the chain (inclusive)
};

}  // namespace pm
