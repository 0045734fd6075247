// pm_run.hpp -- the run step: directional sweeps that advance a whole adoption run per step.  This header
// holds the idea, the exactness notes and the helpers; the engine built on it is pm_run2.hpp
// (PM_ENGINE_RUNBLK2; its one-segment-per-wavefront predecessors PM_ENGINE_RUN / _RUNBLK were retired in round 2).
//
// Measured on the benchmark workload (8 iterations, 11x11): in the first pass of an iteration
// 85-97 % of the pixels adopt their predecessor's value and the adopted value travels in runs of
// 7-50 pixels; in the reverse passes 60-90 % of the steps offer a candidate equal to the pixel's own
// value.  A sweep is therefore mostly "one value walking along the chain until some pixel rejects
// it" (patchmatch.cpp:158-196 applied pixel after pixel, :264-310).
//
// A wavefront tests ONE candidate value v at up to 64-win consecutive positions per step:
//   * lane l computes the window LINE sum of image column X0+l (row sweep) or image row Y0+l (column
//     sweep, on the transposed planes): the sum over the window's other dimension of the colour and
//     saturated-gradient absolute differences.  Lanes are always in increasing coordinate order, so
//     every load is a coalesced row read, and for a row sweep the second bilinear tap of a lane is
//     the first tap of its right neighbour (DPP wave_shl:1, no second load);
//   * the window sums of a position are `win` adjacent lines: a sliding sum across lanes.  Both sums
//     are integers (< 2^16 each, packed into one register), so regrouping them is exact;
//   * every lane turns its window sums into the cost functor's value and compares with the stored
//     cost of its pixel; a ballot finds the first position (in sweep order) that does not continue
//     the run.  Positions before it adopt v (or already hold it), that position keeps its own value,
//     which becomes the next candidate.
// The disparity / cost values of the chain live in LDS for the whole kernel (loaded and stored once),
// so a step touches global memory only for the image lines.  The result is bit-identical to the
// sequential loop.
//
// Exactness notes.  (1) cv::getRectSubPix derives the bilinear weight from fl(fl(x - d) - (pw-1)/2);
// for a fixed d that fraction is the same for all x with x - d in one binade and changes when x - d
// crosses a power of two, so a step only decides positions whose (fraction, integer offset) equal
// the reference position's; the rest wait for the next step.  (2) a candidate with x - d < pw/2 is not
// considered (patchmatch.cpp:186): such a position ends the run without an evaluation.
#pragma once

#include "pm_kernels.hpp"

namespace pm {

constexpr int kMaxSegWaves = 16;


// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an XCD and its 4 MiB L2).
// Adjacent chains read almost the same image rows, so chain k of the sweep goes to the block whose
// XCD owns the band around k: block b -> chain (b % 8) * band + b / 8 (bijective for any count).
// Speed only: any placement gives the same result.
__device__ __forceinline__ int xcd_band_index(int b, int nb) {
  const int xcd = b & 7, j = b >> 3;
  const int q = nb >> 3, r = nb & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

// lane l <- lane l+1 across the whole wavefront (DPP wave_shl:1, gfx9 incl. gfx950); lane 63 has no
// source and receives 0.
__device__ __forceinline__ int wave_shl1(int v) {
  // bound_ctrl: the lane without a source reads 0 and the old value is dead -- no zero-initialising v_mov per use
  return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xF, 0xF, true);
}
__device__ __forceinline__ float wave_shl1f(float v) {
  return __builtin_bit_cast(float, wave_shl1(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
// Segments + fix-up (pm_run2.hpp::k_runblk2): one WORKGROUP per chain, the chain cut into segments, fix-up
// iterated to a fixpoint.  Round 1 sweeps every segment speculatively, starting from the OLD value of the pixel
// before it (exact for the first segment, whose predecessor is never swept).  In each later round a segment whose
// predecessor ended on a different value than the one it started from re-runs from its start with that value
// until its state merges with the trajectory it had stored (or the segment ends, which may change ITS last value
// and trigger its successor in the next round).  Segment k is final after round k+1, so at most S rounds happen
// and the fixpoint is the unique solution of the recurrence = the sequential sweep; in practice a value crosses
// one or two boundaries and 2-3 rounds suffice.

}  // namespace pm
