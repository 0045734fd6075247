// pm_run3.hpp -- PM_ENGINE_RUNBLK2 for PM_SEM_CPU: the run step with its decisions on the scalar unit.
//
// The run step itself (one candidate value tested at up to GS - win consecutive positions of a chain segment, lane =
// window LINE, window sums as a sliding sum across lanes, segments + fix-up rounds) is described in pm_run2.hpp.
// Round-2 counters showed the sweeps bound by vector-instruction issue, with ~70 of a step's 168 vector instructions
// spent on DECISIONS, not on taps: every ballot was shifted into the lane's group by a 64-bit vector shift, turned
// into "first position" indices by vector ffs / min / select chains, and broadcast back by ds_bpermute.  Here
//   * every predicate is ONE v_cmp into a 64-bit scalar mask and all logic on the masks runs on the scalar unit, for
//     the two or four groups of the wavefront at once (fields of 32 / 16 bits, SWAR):  a sentinel bit behind the
//     last position of every field makes "x - 1 per field" borrow-free, so `x & ~(x - F)` is the first stopping
//     position of every group and `that - F0` the positions the run passed;
//   * lanes always sit in SWEEP order: a backward sweep mirrors the lane -> line mapping (the loads stay coalesced,
//     the neighbour tap comes from lane - 1 instead of lane + 1), so no mask is ever bit-reversed;
//   * "same bilinear parameters as the step's first position" is an exponent compare: within one binade of
//     fl(x - d) the float grid is invariant under integer shifts of x, so fl((x + k) - d) = fl(x - d) + k exactly
//     and fraction and column offset agree; positions in another binade wait for the next step (as before);
//   * the candidate of a group lives in an LDS slot: the rejecting lane stores its value, everybody reads it back
//     (no ds_bpermute address arithmetic); a lane's chain index advances by a popcount of its group's field;
//   * chain state is one float4 per position {old value, old cost, new value, new cost}: one ds_read_b64 and one
//     ds_write_b64 per step at the same address, no conditional read;
//   * three window lines per load: 16-byte records {three gradients, three colour bytes} from LINE-indexed planes
//     (PlaneSet::rpg / cpg, pm_kernels.hpp::triples_block); the line offsets are wave-uniform base pointers (scalar registers), the
//     lane's column is ONE vector offset shared by all loads of a step.  Round-3 counters (profiles/r03b_*): with the
//     chip full of sweeps the vector units issue 31-35 % of the time while the texture address / data units are
//     73-92 % busy -- the memory pipeline works per 4 lanes and cache line (23 accesses per wave-load of 12-byte
//     records), so the number and width of the loads is what a step costs;
//   * 16-lane groups add up their window lines with row_shl DPP in log steps (5 adds for 11 lines instead of 10).
// Results are bit-identical to the serial engine, the wave engine and the oracle (tests/test_gpu_parity.py,
// tools/fuzz_engines.py).
#pragma once
#include <vector>

#include "pm_run2.hpp"
#include "pm_tune.hpp"

namespace pm {

template <int GS> struct Fields;
template <> struct Fields<16> { static constexpr unsigned long long lsb = 0x0001000100010001ull; };
template <> struct Fields<32> { static constexpr unsigned long long lsb = 0x0000000100000001ull; };

__device__ __forceinline__ unsigned long long mask_of(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// lane predicate from a wave-uniform mask: the mask is used as it is (exec / vcc), no vector instruction
__device__ __forceinline__ bool in_mask(unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

// lane l <- lane l + N of its 16-lane row (DPP row_shl:N); lanes without a source read 0
template <int N>
__device__ __forceinline__ int row_shl(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x100 + N, 0xF, 0xF, true);
}
// the lane holding the next line in increasing image coordinate: lane + 1 in a forward sweep, lane - 1 in a backward
// one (wave_shl:1 / wave_shr:1; the lane without a source reads 0: it is the group's spare line)
template <int DIR>
__device__ __forceinline__ int next_line_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, DIR > 0 ? 0x130 : 0x138, 0xF, 0xF, true);
}
template <int DIR>
__device__ __forceinline__ float next_line_f(float v) {
  return __builtin_bit_cast(float, next_line_i<DIR>(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ int clamp_med3(int x, int hi) {  // min(max(x, 0), hi) as one v_med3_i32
  int r;
  asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "s"(hi));
  return r;
}

// sum of the `win` lines l .. l + win - 1 (packed colour | gradient << 16 sums) in every lane l that is a position
template <int GS, int WIN>
__device__ __forceinline__ int window_sum(int line, int win) {
  if constexpr (GS == 16 && WIN >= 3 && WIN <= 11) {
    const int s2 = line + row_shl<1>(line);
    if constexpr (WIN == 3) return s2 + row_shl<2>(line);
    const int s4 = s2 + row_shl<2>(s2);
    if constexpr (WIN == 5) return s4 + row_shl<4>(line);
    if constexpr (WIN == 7) return (s4 + row_shl<4>(s2)) + row_shl<6>(line);
    const int s8 = s4 + row_shl<4>(s4);
    if constexpr (WIN == 9) return s8 + row_shl<8>(line);
    return (s8 + row_shl<8>(s2)) + row_shl<10>(line);
  } else {
    int w = line;
    if constexpr (WIN > 0) {
#pragma unroll
      for (int t = 1; t < WIN; ++t) w = line + wave_shl1(w);
    } else {
      for (int t = 1; t < win; ++t) w = line + wave_shl1(w);
    }
    return w;
  }
}

// cpu_cost_from_sums (pm_device.hpp) on both sums at once with packed-f32 instructions: the same operations in the
// same order per component (mean_from_sum: p = s * hi, e1 = fma(s, hi, -p), e2 = s * lo, mean = p + (e1 + e2)).
__device__ __forceinline__ float cost_from_packed_sums(int wsum, const CostParams& cp) {
  const f32x2 sf = {(float)(wsum & 0xffff), (float)((unsigned)wsum >> 16)};
  const f32x2 hi = {cp.inv_n_hi, cp.inv_n_hi}, lo = {cp.inv_n_lo, cp.inv_n_lo};
  const f32x2 p = sf * hi;
  const f32x2 e1 = __builtin_elementwise_fma(sf, hi, -p);
  const f32x2 e2 = sf * lo;
  const f32x2 m = p + (e1 + e2);
  const f32x2 e = {fminf(m.x, cp.tau_color), fminf(m.y, cp.tau_grad)};
  const f32x2 wgt = {cp.alpha, cp.one_minus_alpha};
  const f32x2 t = wgt * e;
  return t.x + t.y;
}

// LDS bytes of the column sweeps' staged reference lines (LREF): kLref4Stride dwords per image row.
template <int AXIS>
inline size_t run3_lref_bytes(const PlaneSet& ps) {
  return sizeof(unsigned) * (size_t)kLref4Stride * (AXIS == 1 ? ps.rows : ps.cols);
}
// PM_RUN2_LREF: bit 0 = row sweeps, bit 1 = column sweeps (default 2); PM_RUN2_LREF_KB: LDS budget per workgroup (A/B knobs)
inline bool run3_lref_enabled(int axis) {
  static const int v = [] {
    const char* e = pm::tune_env("PM_RUN2_LREF");
    return e ? atoi(e) : 2;
  }();
  return (v >> axis) & 1;
}
inline size_t run3_lref_limit() {
  static const size_t v = [] {
    const char* e = pm::tune_env("PM_RUN2_LREF_KB");
    return (size_t)(e ? atoi(e) : 40) * 1024;
  }();
  return v;
}

// The LDS word that carries a group's candidate from the rejecting lane to the others: volatile keeps the accesses in
// program order, the explicit address space keeps them ds_read / ds_write (a volatile generic pointer is a flat access).
typedef __attribute__((address_space(3))) volatile float* LdsSlot;

// per-lane constants of the kernel
struct Run3Lane {
  int gl, gbase;
  int mpos;          // position of the lane inside a step (lane - POS0); < 0 or >= nd: a line-only lane
  float dmposf;      // (float)(DIR * mpos)
  int rofs;          // row sweeps: target column of the lane's line minus the first target column of the step
  int lim;           // segment end for position lanes of an active segment, INT_MIN otherwise
};

// Wave-uniform base pointers of the window's line pairs (and reference quads), fixed for the whole kernel: with them
// in scalar registers a load is global_load ... v_off, s[base] and ONE vector offset serves all loads of a step.
typedef __attribute__((address_space(1))) const char* GlobalPtr;  // explicit: a pinned generic pointer loads as flat_load
struct Run3Bases {
  GlobalPtr line[4];
  GlobalPtr quad[3];
};
// One record of the line-triple planes: the gradients of three consecutive lines and their colour bytes (pm_kernels.hpp::triples_block).
struct alignas(16) TripleRec {
  float g0, g1, g2;
  unsigned c;
};
struct alignas(8) QuadRec {
  unsigned c, g;
};
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned pm_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ TripleRec ld_rec_g(GlobalPtr base, unsigned byte_off) {
  // one global_load_dwordx4 (a member-wise struct load is split into dwordx2 + dword when a line is unused: two
  // trips through the memory pipeline).  The elements go through scalar temporaries: hipcc 7.2 evaluates
  // __builtin_bit_cast(float, t.y) on a vector element as element 0.
  const u32x4 t = *(const __attribute__((address_space(1))) u32x4*)(base + (size_t)byte_off);
  const unsigned t0 = t.x, t1 = t.y, t2 = t.z, t3 = t.w;
  TripleRec r;
  r.g0 = __builtin_bit_cast(float, t0);
  r.g1 = __builtin_bit_cast(float, t1);
  r.g2 = __builtin_bit_cast(float, t2);
  r.c = t3;
  return r;
}
__device__ __forceinline__ QuadRec ld_quad_g(GlobalPtr base, unsigned byte_off) {
  const __attribute__((address_space(1))) QuadRec* q =
      (const __attribute__((address_space(1))) QuadRec*)(base + (size_t)byte_off);
  QuadRec r;
  r.c = q->c;
  r.g = q->g;
  return r;
}
template <int AXIS, int TP>
__device__ __forceinline__ Run3Bases run3_bases(const View& v, const PlaneSet& ps, int chain) {
  Run3Bases b{};
  if constexpr (TP > 0) {
    if constexpr (AXIS == 0) {
      const int y0 = chain - TP / 2;  // first window row
#pragma unroll
      for (int m = 0; m < (TP + 2) / 3; ++m)
        b.line[m] = (GlobalPtr)v.rpg + ((size_t)(y0 + 3 * m) * (size_t)ps.pitch) * 16u;
#pragma unroll
      for (int q = 0; q < (TP + 3) / 4; ++q)
        b.quad[q] = (GlobalPtr)v.rqk + ((size_t)(y0 + 4 * q) * (size_t)ps.pitch) * 8u;
    } else {
#pragma unroll
      for (int m = 0; m < (TP + 3) / 3; ++m) b.line[m] = (GlobalPtr)v.cpg + ((size_t)(3 * m) * (size_t)ps.pitch_t) * 16u;
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) asm volatile("" : "+s"(b.line[m]));  // pinned: not re-derived per step in vector registers
#pragma unroll
    for (int q = 0; q < 3; ++q) asm volatile("" : "+s"(b.quad[q]));
  }
  return b;
}

// TP = square window 3 .. 11, or 0: any window from cp (GS = 32).  One step of every group of the wavefront.
// inr_m: lanes whose position exists (inside their segment, group still running).
// WIDE (fix-up rounds only): ALL groups of the wavefront work on ONE run -- group j tests the candidate at the positions
// behind group j - 1's (the caller set ipm that way and hands every lane the same candidate slot); what a group found
// counts only if every group before it passed all of its positions without merging (see the decisions below).
template <int GS, int AXIS, int TP, int DIR, bool LREF, bool FIX, bool WIDE = false>
__device__ __forceinline__ void run3_step(const View& v, const PlaneSet& ps, const CostParams& cp, const SweepGeom& g,
                                          int chain, const Run3Lane& k, const Run3Bases& bases,
                                          unsigned long long inr_m, int& ipm, float4* st4, LdsSlot cand_slot,
                                          unsigned long long& merged_fill) {
  const int pitch = ps.pitch, cols = ps.cols, rows = ps.rows;
  const int pw = TP > 0 ? TP : cp.pw, ph = TP > 0 ? TP : cp.ph;
  const int half_w = pw / 2, half_h = ph / 2;
  const int win = AXIS == 0 ? pw : ph;
  const int half = win / 2;
  const int nd = AXIS == 0 ? GS - pw : GS - ph + 1;
  constexpr int POS0 = (AXIS == 0 && DIR < 0) ? 1 : 0;
  constexpr unsigned long long F = Fields<GS>::lsb;
  constexpr unsigned long long F0 = F << POS0;
  const unsigned long long S = F << (POS0 + nd);
  const unsigned long long POSM = S - F0;
  const float shift = (float)(pw - 1) * 0.5f;

  // ---- state of the lane's position and the group's candidate -------------------------------------------------------
  // (every lane reads: lanes beyond their segment see some other position of the chain, masked off by inr_m)
  float4* const my = st4 + (ipm + 1);
  const float2 dc = *(const float2*)my;
  const float d0 = dc.x, c0 = dc.y;
  const float cand = *cand_slot;
  const unsigned long long neutral_m = mask_of(d0 == cand) & inr_m;
  const unsigned long long need_m = inr_m & ~neutral_m;

  const int pos = g.s_first + DIR * ipm;  // image column (row sweep) / row (column sweep) of the lane's position
  float cx, cx0;                          // x - d of the lane's position and of the step's first position
  if (AXIS == 0) {
    const float posf = (float)pos;
    cx = posf - cand;
    cx0 = (posf - k.dmposf) - cand;
  } else {
    cx = cx0 = (float)chain - cand;
  }
  const unsigned long long valid_m = mask_of(cx >= (float)half_w);  // patchmatch.cpp:186
  // same sign and exponent as the first position's x - d: same bilinear fraction and column offset (header)
  const unsigned long long same_m =
      AXIS == 0 ? mask_of((__builtin_bit_cast(unsigned, cx) ^ __builtin_bit_cast(unsigned, cx0)) < 0x00800000u) : ~0ull;
  const float t0 = cx0 - shift;
  const float fl0 = floorf(t0);
  const float a_r = t0 - fl0;
  const int ipx_r = (int)fl0;  // first target column of the first position's window

  float cost = 0.f;
  if ((need_m & valid_m) != 0ull) {  // some lane may adopt: the wavefront evaluates (groups without need compute along)
    const float ia_r = 1.f - a_r;
    unsigned sc = 0, sg = 0;
    if constexpr (TP > 0) {
      // colour weights cvRound(w * 2^16), clamped to 16 bits (pm_device.hpp::cpu_color_weights): w * 2^16 is exact, so
      // adding 2^23 rounds it to the nearest-even integer, which then sits in the low mantissa bits
      const float m_a = __builtin_fmaf(a_r, 65536.f, 8388608.f), m_ia = __builtin_fmaf(ia_r, 65536.f, 8388608.f);
      const float cap = __builtin_bit_cast(float, 0x4b00ffffu);  // 2^23 + 65535
      const unsigned cw = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, fminf(m_a, cap)),
                                                __builtin_bit_cast(unsigned, fminf(m_ia, cap)), 0x05040100u);
      const f32x2 w2 = {ia_r, a_r};
      if constexpr (AXIS == 0) {
        // lane = image column X (its own position's column -+ half), window rows in the lane, three per load
        constexpr int NT = (TP + 2) / 3, NQR = (TP + 3) / 4;
        const int X = clamp_med3(pos - DIR * half, cols - 1);
        const int R0 = clamp_med3(ipx_r + k.rofs, cols - 1);
        const unsigned rv = (unsigned)R0 << 4;  // the lane's byte offset inside every line
        unsigned tcol[4 * NQR + 4];  // colour lerp sums r0 * a11 + r1 * a12 + 2^15: the sample is byte 2
        float gv[3 * NT + 1];
        TripleRec recs[NT];
#pragma unroll
        for (int m = 0; m < NT; ++m) recs[m] = ld_rec_g(bases.line[m], rv);
#pragma unroll
        for (int m = 0; m < NT; ++m) {
          const TripleRec rec = recs[m];
          gv[3 * m] = rec.g0;
          gv[3 * m + 1] = rec.g1;
          gv[3 * m + 2] = rec.g2;
          // the second bilinear tap of the three rows is the next line's first one: one DPP move per record, then per
          // row (own byte | neighbour's byte << 16) by one v_perm and the lerp sum by one v_dot2_u32_u16
          const unsigned cn = (unsigned)next_line_i<DIR>((int)rec.c);
          tcol[3 * m] = cpu_color_sum_pk(__builtin_amdgcn_perm(cn, rec.c, 0x0c040c00u), cw);
          tcol[3 * m + 1] = cpu_color_sum_pk(__builtin_amdgcn_perm(cn, rec.c, 0x0c050c01u), cw);
          tcol[3 * m + 2] = cpu_color_sum_pk(__builtin_amdgcn_perm(cn, rec.c, 0x0c060c02u), cw);
        }
#pragma unroll
        for (int t = 3 * NT; t < 4 * NQR + 4; ++t) tcol[t] = 0u;
        gv[3 * NT] = 0.f;
        float sgr[4 * NQR + 4];  // gradient lerp sums g0 * (1 - a) + g1 * a, g1 = the next line's g0
#pragma unroll
        for (int t = 0; t < TP; ++t) {  // one packed multiply per sample: the sample splat against (1 - a, a)
          const f32x2 p = f32x2{gv[t], gv[t]} * w2;
          sgr[t] = p.x + next_line_f<DIR>(p.y);
        }
        unsigned rq_c[NQR], rq_g[NQR];  // reference bytes of four window rows per dword
        if constexpr (LREF) {  // staged by the kernel: [image column][kLref4Stride] dwords
          unsigned x8 = (unsigned)X << 3;
          asm volatile("" : "+v"(x8));
          unsigned x7 = x8 - (unsigned)X;
          asm volatile("" : "+v"(x7));
          const unsigned* rr = v.lds_ref4 + x7;
#pragma unroll
          for (int q = 0; q < NQR; ++q) {
            const int rem = TP - 4 * q;
            const unsigned mask = rem >= 4 ? 0xffffffffu : ((1u << (8 * rem)) - 1u);
            rq_c[q] = rr[q] & mask;
            rq_g[q] = rr[NQR + q] & mask;
          }
        } else {
          const unsigned qv = (unsigned)X << 3;
#pragma unroll
          for (int q = 0; q < NQR; ++q) {
            const QuadRec rr = ld_quad_g(bases.quad[q], qv);
            const int rem = TP - 4 * q;
            const unsigned mask = rem >= 4 ? 0xffffffffu : ((1u << (8 * rem)) - 1u);
            rq_c[q] = rr.c & mask;
            rq_g[q] = rr.g & mask;
          }
        }
#pragma unroll
        for (int q = 0; 4 * q < TP; ++q) {  // four rows per v_sad_u8
          const int r0 = 4 * q;
          const bool h1 = r0 + 1 < TP, h2 = r0 + 2 < TP, h3 = r0 + 3 < TP;
          const unsigned u = __builtin_amdgcn_perm(tcol[r0 + 1], tcol[r0], h1 ? 0x0c0c0602u : 0x0c0c0c02u);
          unsigned s4 = u;
          if (h2) {
            const unsigned w = __builtin_amdgcn_perm(tcol[r0 + 3], tcol[r0 + 2], h3 ? 0x0c0c0602u : 0x0c0c0c02u);
            s4 = (w << 16) | u;
          }
          unsigned g4 = __builtin_amdgcn_cvt_pk_u8_f32(sgr[r0], 0, 0u);
          if (h1) g4 = __builtin_amdgcn_cvt_pk_u8_f32(sgr[r0 + 1], 1, g4);
          if (h2) g4 = __builtin_amdgcn_cvt_pk_u8_f32(sgr[r0 + 2], 2, g4);
          if (h3) g4 = __builtin_amdgcn_cvt_pk_u8_f32(sgr[r0 + 3], 3, g4);
          sc = __builtin_amdgcn_sad_u8(rq_c[q], s4, sc);
          sg = __builtin_amdgcn_sad_u8(rq_g[q], g4, sg);
        }
      } else {
        // lane = image row Y (transposed planes), window columns in the lane: samples 0 .. TP of the lane's row are
        // TP + 1 consecutive image columns from ipx_r: whole triples of the line-indexed plane
        constexpr int NTC = (TP + 3) / 3, NQ = (TP + 3) / 4;
        const int pt = ps.pitch_t;
        const int Y = clamp_med3(pos - DIR * half, rows - 1);
        const int ipx_c = clamp_med3(ipx_r, cols - 1);  // (a group that cannot adopt may carry any ipx_r)
        // element (line ipx_c, row Y); both factors < 2^16: the 24-bit multiply-add is exact and full rate
        const unsigned cv = (__umul24((unsigned)ipx_c, (unsigned)pt) + (unsigned)Y) << 4;
        unsigned prv[NTC + 1];
        float gv[3 * NTC + 1];
#pragma unroll
        for (int m = 0; m < NTC; ++m) {
          const TripleRec rec = ld_rec_g(bases.line[m], cv);
          prv[m] = rec.c;
          gv[3 * m] = rec.g0;
          gv[3 * m + 1] = rec.g1;
          gv[3 * m + 2] = rec.g2;
        }
        prv[NTC] = 0u;
        gv[3 * NTC] = 0.f;
        unsigned rc4[NQ], rg4[NQ];  // reference bytes of the lane's row, four window columns per dword
        if constexpr (LREF) {
          const unsigned* rr = v.lds_ref4 + __umul24((unsigned)Y, (unsigned)kLref4Stride);  // (24-bit multiply: full rate)
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            rc4[q] = rr[q];
            rg4[q] = rr[NQ + q];
          }
        } else {
          const int lorg = (chain - half_w) * pt;
          unsigned pkv[4 * NQ];
#pragma unroll
          for (int t = 0; t < 4 * NQ; ++t)
            pkv[t] = t < TP ? (unsigned)ld_u16(v.trefpk, (unsigned)((Y + lorg + t * pt) * 2)) : 0u;
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            const unsigned lo = __builtin_amdgcn_perm(pkv[4 * q + 1], pkv[4 * q], 0x05010400u);      // c0 c1 g0 g1
            const unsigned hi = __builtin_amdgcn_perm(pkv[4 * q + 3], pkv[4 * q + 2], 0x05010400u);  // c2 c3 g2 g3
            rc4[q] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
            rg4[q] = __builtin_amdgcn_perm(hi, lo, 0x07060302u);
          }
        }
        unsigned tcol[4 * NQ];
#pragma unroll
        for (int t = 0; t < 4 * NQ; ++t) {
          if (t < TP) {
            // samples t, t + 1 as halfwords: bytes t % 3 of record t / 3 and (t + 1) % 3 of record (t + 1) / 3
            const unsigned sel = 0x0c000c00u | (unsigned)(t % 3) |
                                 ((unsigned)(((t + 1) / 3 != t / 3 ? 4 : 0) + (t + 1) % 3) << 16);
            const unsigned r01 = __builtin_amdgcn_perm(prv[(t + 1) / 3], prv[t / 3], sel);
            tcol[t] = cpu_color_sum_pk(r01, cw);
          } else {
            tcol[t] = 0u;
          }
        }
        // samples 0 .. TP, one packed multiply each: the sample splat against (1 - a, a) takes its register wherever the
        // load left it (pairs of samples would have to be moved into aligned register pairs first)
        f32x2 pw2[TP + 1];
#pragma unroll
        for (int t = 0; t <= TP; ++t) pw2[t] = f32x2{gv[t], gv[t]} * w2;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {  // four window columns per v_sad_u8
          const int t0q = 4 * q;
          const bool h1 = t0q + 1 < TP, h2 = t0q + 2 < TP, h3 = t0q + 3 < TP;
          const unsigned u = __builtin_amdgcn_perm(tcol[t0q + 1], tcol[t0q], h1 ? 0x0c0c0602u : 0x0c0c0c02u);
          unsigned s4 = u;
          if (h2) {
            const unsigned w = __builtin_amdgcn_perm(tcol[t0q + 3], tcol[t0q + 2], h3 ? 0x0c0c0602u : 0x0c0c0c02u);
            s4 = (w << 16) | u;
          }
          unsigned g4 = 0u;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int t = t0q + j;
            if (t < TP) g4 = __builtin_amdgcn_cvt_pk_u8_f32(pw2[t].x + pw2[t + 1].y, j, g4);
          }
          sc = __builtin_amdgcn_sad_u8(rc4[q], s4, sc);
          sg = __builtin_amdgcn_sad_u8(rg4[q], g4, sg);
        }
      }
    } else {
      // any window: one load per tap and plane (the form of pm_device.hpp::cpu_cost_lane, one line per lane)
      CpuLerp l;
      l.a = a_r;
      l.ia = ia_r;
      l.a11 = __float2int_rn(ia_r * 65536.f);
      l.a12 = __float2int_rn(a_r * 65536.f);
      l.ipx = 0;
      if constexpr (AXIS == 0) {
        const int X = clamp_med3(pos - DIR * half, cols - 1);
        const int R0 = clamp_med3(ipx_r + k.rofs, cols - 1);
        const int org = (chain - half_h) * pitch;
        for (int t = 0; t < ph; ++t) {
          const int so = org + t * pitch;
          const int l8 = ld_u8(v.ref8, (unsigned)(X + so));
          const int lg = ld_u8(v.refg8, (unsigned)(X + so));
          const int r0 = ld_u8(v.tgt8, (unsigned)(R0 + so));
          const float g0 = ld_f32(v.tgtg, (unsigned)(R0 + so) * 4u);
          sc = cpu_acc_color(sc, l8, r0, next_line_i<DIR>(r0), l);
          sg = cpu_acc_grad(sg, lg, g0, next_line_f<DIR>(g0), l);
        }
      } else {
        const int pt = ps.pitch_t;
        const int Y = clamp_med3(pos - DIR * half, rows - 1);
        const int ipx_c = clamp_med3(ipx_r, cols - 1);
        const int vb = ipx_c * pt + Y;
        const int lorg = (chain - half_w) * pt;
        int r0 = ld_u8(v.ttgt8, (unsigned)vb);
        float g0 = ld_f32(v.ttgtg, (unsigned)vb * 4u);
        for (int t = 0; t < pw; ++t) {
          const int lso = lorg + t * pt, rso = (t + 1) * pt;
          const int l8 = ld_u8(v.tref8, (unsigned)(Y + lso));
          const int lg = ld_u8(v.trefg8, (unsigned)(Y + lso));
          const int r1 = ld_u8(v.ttgt8, (unsigned)(vb + rso));
          const float g1 = ld_f32(v.ttgtg, (unsigned)(vb + rso) * 4u);
          sc = cpu_acc_color(sc, l8, r0, r1, l);
          sg = cpu_acc_grad(sg, lg, g0, g1, l);
          r0 = r1;
          g0 = g1;
        }
      }
    }
    const int wsum = window_sum<GS, TP>((int)(sc | (sg << 16)), win);
    cost = cost_from_packed_sums(wsum, cp);
  }

  // ---- decisions: scalar unit, all groups at once -----------------------------------------------------------------
  // The run passes a position iff it ends up holding `cand`: already equal (neutral) or adopted (needs an allowed
  // candidate, the step's bilinear parameters and a strictly smaller cost).  The first position that does not pass
  // is decided in this step if its candidate is not allowed at all or was evaluated with its own parameters; then
  // it keeps its value, which becomes the next candidate.  Otherwise the next step starts there.
  const unsigned long long lt_m = mask_of(cost < c0);
  const unsigned long long adopt_m = need_m & valid_m & same_m & lt_m;
  const unsigned long long stop_m = (POSM & ~(neutral_m | adopt_m)) | S;
  const unsigned long long q_m = stop_m & ~(stop_m - F);   // one bit per group: its first stop (or the sentinel)
  const unsigned long long passed_m = q_m - F0;            // positions before it
  unsigned long long real_m = q_m & inr_m & (~valid_m | same_m);  // the stop is a decided position
  unsigned long long done_m = passed_m | real_m;           // positions resolved by this step
  const float dval = in_mask(real_m) ? d0 : cand;
  const float cval = in_mask(adopt_m & passed_m) ? cost : c0;
  unsigned long long write_m = done_m;
  unsigned long long m_m = S;  // first position of every field where a re-run meets its stored trajectory (S: none)
  if constexpr (FIX) {
    // a re-run merges with the stored trajectory at the first position where both hold the same value
    const float spec = my->z;
    const unsigned long long eq_m = (mask_of(dval == spec) & done_m) | S;
    m_m = eq_m & ~(eq_m - F);
    write_m = done_m & (m_m - F0);
  }
  const unsigned long long merged1 = m_m & ~S;  // at most one bit per field, below the field's top bit
  if constexpr (WIDE) {
    // The fields are consecutive stretches of ONE run: field j + 1 was offered the candidate on the assumption that
    // field j hands it on, i.e. that all of field j's positions passed (first stop = sentinel) and none merged (first
    // merge = sentinel).  Fields behind the first one that did not are void: nothing written, nothing counted.
    const unsigned long long clean = q_m & m_m & S;
    constexpr unsigned long long fld = GS == 32 ? 0xffffffffull : 0xffffull;
    unsigned long long vfill = fld;
    bool ok = true;
#pragma unroll
    for (int j = 0; j + 1 < kWave / GS; ++j) {
      ok = ok && ((clean >> (j * GS + POS0 + nd)) & 1ull) != 0ull;
      if (ok) vfill |= fld << ((j + 1) * GS);
    }
    real_m &= vfill;
    done_m &= vfill;
    write_m &= vfill;
    if (FIX && (merged1 & vfill) != 0ull) merged_fill = ~0ull;  // the run has met its stored trajectory: the re-run is over
  } else if constexpr (FIX) {
    // groups that merged stop: fill their fields
    constexpr unsigned long long H = F << (GS - 1);
    const unsigned long long nz = F & ~((H - merged1) >> (GS - 1));
    merged_fill |= (nz << GS) - nz;
  }
  if (in_mask(write_m)) *(float2*)&my->z = make_float2(dval, cval);
  if (in_mask(real_m)) *cand_slot = d0;
  if constexpr (WIDE) {
    ipm += __builtin_popcountll(done_m);  // the resolved positions are one stretch from the run's first position on
  } else {
    const unsigned field = GS == 16 ? ((unsigned)(done_m >> k.gbase) & 0xffffu)
                                    : (k.gbase ? (unsigned)(done_m >> 32) : (unsigned)done_m);
    ipm += __builtin_popcount(field);
  }
}

// One workgroup per chain; a wavefront carries 64 / GS segments.  grid = (chains, 1, slots), block = 64 * nw,
// dynamic LDS = run3_lds_bytes().
template <int GS, int AXIS, int TP, int DIR, bool LREF>
__global__ void __launch_bounds__(64 * kMaxSegWaves) k_runblk3(PlaneSet ps, CostParams cp, SweepGeom g, int seg_len
#ifdef PM_RUN3_STATS
                                                               , unsigned* chain_log  // [workgroup][8], this launch's
#endif
) {
  extern __shared__ float lds[];
#ifdef PM_RUN3_STATS
  const unsigned t_wall0 = (unsigned)wall_clock64();  // 100 MHz, shared by all CUs: launch time = max end - min start
#endif
  const int n = (g.s_last - g.s_first) * DIR + 1;
  const int n1 = n + 1;
  constexpr int kPerWave = kWave / GS;
  const int nw = blockDim.x >> 6;
  const int nseg = kPerWave * nw;
  // [k] = {disparity, cost, new disparity, new cost} of chain position k - 1; [0] = the pixel before the chain
  float4* st4 = (float4*)lds;
  float* s_last = lds + 4 * n1;     // [nseg + 1] last value of every segment
  float* s_cand = s_last + nseg + 1;  // [nseg] current candidate of every segment
  int* s_changed = (int*)(s_cand + nseg);  // [2]

  const int chain = g.c_lo + xcd_band_index(blockIdx.x, gridDim.x);
  if (!chain_active(ps, blockIdx.z, chain)) return;  // uniform for the workgroup, before any barrier
  View v = make_view(ps, blockIdx.z);
#ifdef PM_TUNING
  const int dbg = seg_len >> 24;  // timing experiments (PM_RUN3_DBG): bit 0 no steps, bit 1 no reference staging
  seg_len &= 0xffffff;
  if (dbg & 2) {
    v.lds_ref4 = (unsigned*)(lds + 4 * n1 + 2 * nseg + 3);
  } else
#endif
  if constexpr (LREF && AXIS == 0) {
    // row sweeps: the reference quads of the chain's window rows (quads_block lines y0, y0 + 4, y0 + 8), per image column
    // NQ colour dwords and NQ gradient dwords at stride kLref4Stride
    static_assert(TP > 0, "staged reference lines need a fixed window");
    constexpr int NQ = (TP + 3) / 4;
    static_assert(2 * NQ <= kLref4Stride, "reference column does not fit its LDS stride");
    unsigned* sref4 = (unsigned*)(s_changed + 2);
    const int y0 = chain - TP / 2;
    for (int e = threadIdx.x; e < NQ * ps.cols; e += blockDim.x) {
      const int q = e / ps.cols, x = e - q * ps.cols;
      const uint32_t* src = v.rqk + ((size_t)(y0 + 4 * q) * ps.pitch + x) * 2;
      sref4[x * kLref4Stride + q] = src[0];
      sref4[x * kLref4Stride + NQ + q] = src[1];
    }
    v.lds_ref4 = sref4;
  } else if constexpr (LREF) {
    static_assert(TP > 0, "staged reference lines need a fixed window");
    // per image row the TP packed reference values of window columns chain - TP/2 .. + TP/2 as bytes, four columns
    // per dword: NQ colour dwords, NQ gradient dwords, row stride kLref4Stride dwords (odd: no bank conflicts)
    constexpr int NQ = (TP + 3) / 4;
    static_assert(2 * NQ <= kLref4Stride, "reference row does not fit its LDS stride");
    const int len = ps.rows;
    unsigned* sref4 = (unsigned*)(s_changed + 2);
    const uint16_t* src = v.trefpk + (size_t)(chain - TP / 2) * ps.pitch_t;
    // eight rows per task: one 16-byte load per window column of the dword (4 * NQ loads in flight per thread), the
    // bytes of a row gathered by four v_perm_b32
    static_assert(NQ <= 3, "task index -> dword by two comparisons");
    const bool wide = (reinterpret_cast<uintptr_t>(src) & 15u) == 0 && (ps.pitch_t & 7) == 0;  // uniform
    const int n8 = wide ? len >> 3 : 0;
    for (int e = threadIdx.x; e < NQ * n8; e += blockDim.x) {
      const int q = (e >= n8 ? 1 : 0) + (e >= 2 * n8 ? 1 : 0), r8 = e - q * n8;
      u32x4 col[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int t = 4 * q + j;  // a column beyond the window stages zeros (the step compares whole dwords)
        col[j] = u32x4{0u, 0u, 0u, 0u};
        if (t < TP) col[j] = *reinterpret_cast<const u32x4*>(src + (size_t)t * ps.pitch_t + 8 * r8);
      }
      unsigned* dst = sref4 + (8 * r8) * kLref4Stride + q;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        // halfword i of every column = {colour, gradient} of row 8 * r8 + i
        const unsigned sel = (i & 1) ? 0x07030602u : 0x05010400u;
        const unsigned lo = __builtin_amdgcn_perm(col[1][i >> 1], col[0][i >> 1], sel);  // c0 c1 g0 g1
        const unsigned hi = __builtin_amdgcn_perm(col[3][i >> 1], col[2][i >> 1], sel);  // c2 c3 g2 g3
        dst[i * kLref4Stride] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
        dst[i * kLref4Stride + NQ] = __builtin_amdgcn_perm(hi, lo, 0x07060302u);
      }
    }
    for (int e = threadIdx.x; e < NQ * (len - 8 * n8); e += blockDim.x) {  // the rows left over (all, if not `wide`)
      const int nl = len - 8 * n8;
      const int q = e / nl, row = 8 * n8 + (e - q * nl);
      unsigned cw = 0u, gw = 0u;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int t = 4 * q + j;
        if (t < TP) {
          const unsigned pk = src[(size_t)t * ps.pitch_t + row];
          cw |= (pk & 0xffu) << (8 * j);
          gw |= (pk >> 8) << (8 * j);
        }
      }
      sref4[row * kLref4Stride + q] = cw;
      sref4[row * kLref4Stride + NQ + q] = gw;
    }
    v.lds_ref4 = sref4;
  }
  const Run3Bases bases = run3_bases<AXIS, TP>(v, ps, chain);
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int pw = TP > 0 ? TP : cp.pw, ph = TP > 0 ? TP : cp.ph;
  const int nd = AXIS == 0 ? GS - pw : GS - ph + 1;
  constexpr int POS0 = (AXIS == 0 && DIR < 0) ? 1 : 0;
  Run3Lane k;
  k.gl = lane & (GS - 1);
  k.gbase = lane & ~(GS - 1);
  k.mpos = k.gl - POS0;
  k.dmposf = (float)(DIR * k.mpos);
  k.rofs = DIR > 0 ? k.gl : pw - k.gl;
  const int sidx = kPerWave * w + lane / GS;

  {
    // the chain's state into LDS, four positions per thread in flight (the latency of one load, not of four in a row;
    // four consecutive positions of a column chain share a 16-byte piece of the state planes, pm_device.hpp::state_at)
    constexpr int U = 4;
    const int bd = blockDim.x;
    for (int j0 = threadIdx.x; j0 <= n; j0 += U * bd) {
      float dd[U], cc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = j0 + u * bd;
        dd[u] = cc[u] = 0.f;
        if (j <= n) {
          const size_t o = chain_at(AXIS, chain, g.s_first + DIR * (j - 1), ps.pitch);
          dd[u] = v.disp[o];
          if (j > 0) cc[u] = v.cost[o];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = j0 + u * bd;
        if (j <= n) st4[j] = make_float4(dd[u], cc[u], dd[u], cc[u]);
      }
    }
  }
  __syncthreads();

  const int i0 = sidx * seg_len;
  const int i1 = min(n, i0 + seg_len);
  const bool active = i0 < n;
  k.lim = (active && k.mpos >= 0 && k.mpos < nd) ? i1 : (int)0x80000000;
#ifdef PM_TUNING
  if (dbg & 1) k.lim = (int)0x80000000;
#endif
  LdsSlot cand_slot = (LdsSlot)(s_cand + sidx);
  unsigned n_steps = 0, n_fix = 0, n_rounds = 0;
  unsigned long long no_merge = 0ull;

  // ---- round 1: every segment speculatively from the OLD value of the pixel before it -----------------------------
  float in_used = active ? st4[i0].x : 0.f;
  if (k.gl == 0) *cand_slot = in_used;
  int ipm = i0 + k.mpos;
#ifdef PM_RUN3_STATS
  const long long t_start = clock64();
  unsigned n_gsteps = 0;
#endif
  // (Round 1 does not go wide when a wavefront's last group is alone: built and measured, 450 -> 445 pairs/s, and 438 when
  // the helpers join only behind a step that passed all of a group's positions.  The chip is full of round-1 steps then:
  // the helpers' loads compete with them for the memory pipeline, short runs void most of what the helpers evaluate, and
  // a second copy of the step in this loop is 2000 more instructions in every wavefront's path.  In the fix-up rounds
  // below few groups are at work and the same step is free.)
  for (;;) {
    const unsigned long long inr_m = mask_of(ipm < k.lim);
    if (inr_m == 0ull) break;
#ifdef PM_RUN3_STATS
    n_gsteps += __builtin_popcountll(inr_m & (Fields<GS>::lsb << POS0));
#endif
    run3_step<GS, AXIS, TP, DIR, LREF, false>(v, ps, cp, g, chain, k, bases, inr_m, ipm, st4, cand_slot, no_merge);
    ++n_steps;
  }
#ifdef PM_RUN3_STATS
  const long long t_r1 = clock64();
#endif
  float lastv = *cand_slot;
  if (active && k.gl == 0) s_last[sidx + 1] = lastv;
  if (threadIdx.x == 0) s_last[0] = in_used;

  // ---- fix-up rounds: a segment whose predecessor ended on another value re-runs until it merges ------------------
  for (int round = 1; round < nseg; ++round) {
    if (threadIdx.x == 0) s_changed[round & 1] = 0;
    __syncthreads();
    const float in = (active && sidx > 0) ? s_last[sidx] : in_used;
    const bool redo = active && sidx > 0 && (in != in_used);
    const unsigned long long redo_m = mask_of(redo);
    bool new_last = false;
    if (redo_m != 0ull) {
      if (redo) {
        in_used = in;
        ipm = i0 + k.mpos;
        if (k.gl == 0) *cand_slot = in;
      }
      unsigned long long merged_fill = 0ull;
      // A re-run that is ALONE in its wavefront -- the other groups have merged, or had nothing to re-run -- goes WIDE:
      // the wavefront's groups line up behind each other on its segment and test its candidate at kPerWave x the positions
      // per step (run3_step<WIDE>).  Long re-runs are what the fix-up rounds of the forward sweeps consist of: a value
      // that runs through whole segments, one round per segment, nd positions per step (profiles/r06_chain_balance.txt).
      const int lim_own = k.lim;
      bool wide = false;       // uniform in the wavefront
      LdsSlot wide_slot = cand_slot;
      unsigned long long merged_before = 0ull, own_fill = 0ull;  // who had merged when the last group went wide; its field
      for (;;) {
        unsigned long long inr_m = mask_of(ipm < k.lim) & ~merged_fill;
        if (!wide) {
          inr_m &= redo_m;
          if (inr_m == 0ull) break;
#ifndef PM_RUN3_NO_WIDE  // (A/B switch of tools/build_variant.sh)
          if constexpr (kPerWave > 1) {
            unsigned act = 0u;  // groups of the wavefront with a position left
#pragma unroll
            for (int j = 0; j < kPerWave; ++j)
              act |= ((inr_m >> (j * GS)) & (GS == 32 ? 0xffffffffull : 0xffffull)) != 0ull ? 1u << j : 0u;
            if (__builtin_popcount(act) == 1) {
              const int a = __builtin_ctz(act);
              const int src = a * GS + POS0;  // the lane of the running group that holds its first unresolved position
              const int wp = __builtin_amdgcn_readlane(ipm, src);
              const int wlim = __builtin_amdgcn_readlane(k.lim, src);
              wide_slot = (LdsSlot)(s_cand + kPerWave * w + a);
              wide = true;
              merged_before = merged_fill;  // (their fields take part in the wide steps: the mask starts afresh)
              merged_fill = 0ull;
              own_fill = (GS == 32 ? 0xffffffffull : 0xffffull) << (a * GS);
              ipm = wp + (lane / GS) * nd + k.mpos;
              k.lim = (k.mpos >= 0 && k.mpos < nd) ? wlim : (int)0x80000000;
              inr_m = mask_of(ipm < k.lim);
            }
          }
#endif
        } else if (inr_m == 0ull) {
          break;
        }
        if (wide)
          run3_step<GS, AXIS, TP, DIR, LREF, true, true>(v, ps, cp, g, chain, k, bases, inr_m, ipm, st4, wide_slot, merged_fill);
        else
          run3_step<GS, AXIS, TP, DIR, LREF, true>(v, ps, cp, g, chain, k, bases, inr_m, ipm, st4, cand_slot, merged_fill);
        ++n_fix;
      }
      k.lim = lim_own;
      if (wide) merged_fill = merged_before | (merged_fill != 0ull ? own_fill : 0ull);
      const float c2 = *cand_slot;
      if (redo && !in_mask(merged_fill) && c2 != lastv) {
        lastv = c2;
        new_last = true;
      }
    }
    __syncthreads();
    if (new_last && k.gl == 0) {
      s_last[sidx + 1] = lastv;
      s_changed[round & 1] = 1;
    }
    __syncthreads();
    ++n_rounds;
    if (!s_changed[round & 1]) break;
  }
  __syncthreads();
#ifdef PM_RUN3_STATS
  if (ps.counters) {  // load balance of a launch: what the slowest wavefront of a workgroup does vs the average one
    const long long t_end = clock64();
    __shared__ unsigned s_max_steps, s_max_fix;
    if (threadIdx.x == 0) s_max_steps = s_max_fix = 0;
    __syncthreads();
    if (lane == 0) {
      atomicMax(&s_max_steps, n_steps);
      atomicMax(&s_max_fix, n_fix);
      atomicAdd(&ps.counters[10], (unsigned long long)n_gsteps);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      atomicAdd(&ps.counters[8], (unsigned long long)s_max_steps);
      atomicAdd(&ps.counters[9], (unsigned long long)s_max_fix);
      atomicAdd(&ps.counters[11], 1ull);
      atomicAdd(&ps.counters[12], (unsigned long long)(t_r1 - t_start));
      atomicAdd(&ps.counters[13], (unsigned long long)(t_end - t_r1));
      atomicAdd(&ps.counters[14], (unsigned long long)nseg);
    }
  }
  if (chain_log && blockIdx.z == 0) {  // per chain: what its slowest wavefront did, and when (tools/chain_tail.py)
    __shared__ unsigned s_ms, s_mf, s_ss, s_sf, s_ws, s_wf;
    if (threadIdx.x == 0) s_ms = s_mf = s_ss = s_sf = s_ws = s_wf = 0;
    __syncthreads();
    if (lane == 0) {
      atomicMax(&s_ms, n_steps);
      atomicMax(&s_mf, n_fix);
      atomicAdd(&s_ss, n_steps);  // all wavefronts' steps: balance inside the chain (tools/chain_tail.py)
      atomicAdd(&s_sf, n_fix);
      if (w < 4) {  // per wavefront, 8 bits each (saturating): WHICH part of the chain is the slow one
        atomicOr(&s_ws, (n_steps > 255u ? 255u : n_steps) << (8 * w));
        atomicOr(&s_wf, (n_fix > 255u ? 255u : n_fix) << (8 * w));
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned* r = chain_log + 8 * (size_t)blockIdx.x;
      r[0] = (unsigned)chain;
      r[1] = s_ms;
      r[2] = s_mf;
      r[4] = t_wall0;
      r[5] = (unsigned)wall_clock64();
      r[3] = n_rounds | (s_ss << 8);  // rounds (8 bits) | steps of all wavefronts, round 1
      r[6] = s_ws;
      r[7] = s_wf;
      (void)s_sf;
    }
  }
#endif
  if (ps.counters && lane == 0) {
    const int base = AXIS * 4;
    atomicAdd(&ps.counters[base + 0], (unsigned long long)n_steps);
    atomicAdd(&ps.counters[base + 1], (unsigned long long)n_fix);
    if (w == 0) atomicAdd(&ps.counters[base + 2], (unsigned long long)n_rounds);
    if (w == 0) atomicAdd(&ps.counters[base + 3], (unsigned long long)n);
  }

  for (int j = threadIdx.x + 1; j <= n; j += blockDim.x) {
    const float4 t = st4[j];
    if (t.z != t.x) {
      const size_t o = chain_at(AXIS, chain, g.s_first + DIR * (j - 1), ps.pitch);
      v.disp[o] = t.z;
      v.cost[o] = t.w;
    }
  }
}


inline int run3_dbg() {
  static const int v = [] {
    const char* e = pm::tune_env("PM_RUN3_DBG");
    return e ? atoi(e) : 0;
  }();
  return v;
}
#ifdef PM_RUN3_STATS
// ---- per-chain log of the stats build (make tuning TUNE_DEFS=-DPM_RUN3_STATS; tools/chain_tail.py) -----------------------
// Every k_runblk3 launch gets a slot of [workgroups][8] words in one device buffer and a host record of what it was;
// pm_run3_stats_dump (pm_sweeps.hip, exported by this build only) writes both to a file.
struct Run3StatsRec {
  unsigned long long stream;
  int axis, dir, gs, n, chains, waves;
};
struct Run3Stats {
  static constexpr int kMaxLaunch = 4096, kMaxChains = 4096;
  unsigned* d_log = nullptr;
  std::vector<Run3StatsRec> recs;
  bool on = false;
};
inline Run3Stats& run3_stats() {
  static Run3Stats s;
  return s;
}
inline unsigned* run3_stats_slot(hipStream_t stream, int axis, int dir, int gs, int n, int chains, int waves) {
  Run3Stats& s = run3_stats();
  if (!s.on || chains > Run3Stats::kMaxChains || (int)s.recs.size() >= Run3Stats::kMaxLaunch) return nullptr;
  if (!s.d_log) {
    if (hipMalloc((void**)&s.d_log, sizeof(unsigned) * 8 * (size_t)Run3Stats::kMaxChains * Run3Stats::kMaxLaunch) != hipSuccess)
      return nullptr;
  }
  s.recs.push_back({(unsigned long long)(uintptr_t)stream, axis, dir, gs, n, chains, waves});
  return s.d_log + 8 * (size_t)Run3Stats::kMaxChains * (s.recs.size() - 1);
}
#endif
inline size_t run3_lds_bytes(int n, int nseg) { return sizeof(float) * (4 * (size_t)(n + 1) + 2 * (size_t)nseg + 3); }

template <int GS, int AXIS, int TP, int DIR, bool LREF>
inline void launch_run3_l(const PlaneSet& ps, const CostParams& cp, const SweepGeom& g, int slots, int waves,
                          hipStream_t stream) {
  const int chains = g.c_hi - g.c_lo + 1;
  const int n = (g.s_last - g.s_first) * g.dir + 1;
  const int nwv = waves < 1 ? 1 : (waves > kMaxSegWaves ? kMaxSegWaves : waves);
  const int nseg = (kWave / GS) * nwv;
  int len = (n + nseg - 1) / nseg;
  if (len < 8) len = 8;
  size_t lds_bytes = run3_lds_bytes(n, nseg);
  if (LREF) lds_bytes += run3_lref_bytes<AXIS>(ps);
  {  // tuning build: PM_RUN3_LDS_EXTRA_KB pads the allocation (how sensitive is the step to workgroups per CU?)
    static const int extra = [] {
      const char* e = pm::tune_env("PM_RUN3_LDS_EXTRA_KB");
      return e ? atoi(e) : 0;
    }();
    lds_bytes += (size_t)extra * 1024;
  }
  allow_big_lds(k_runblk3<GS, AXIS, TP, DIR, LREF>, lds_bytes);
#ifdef PM_RUN3_STATS
  hipLaunchKernelGGL((k_runblk3<GS, AXIS, TP, DIR, LREF>), dim3((unsigned)chains, 1, (unsigned)slots),
                     dim3(kWave * nwv), lds_bytes, stream, ps, cp, g, len | (run3_dbg() << 24),
                     run3_stats_slot(stream, AXIS, DIR, GS, n, chains, nwv));
#else
  hipLaunchKernelGGL((k_runblk3<GS, AXIS, TP, DIR, LREF>), dim3((unsigned)chains, 1, (unsigned)slots),
                     dim3(kWave * nwv), lds_bytes, stream, ps, cp, g, len | (run3_dbg() << 24));
#endif
}
template <int GS, int AXIS, int TP>
inline void launch_run3_d(const PlaneSet& ps, const CostParams& cp, const SweepGeom& g, int slots, int waves,
                          hipStream_t stream) {
  // column sweeps of the benchmark window stage their reference lines in LDS while that leaves room for at least
  // four workgroups per CU (PM_RUN2_LREF / PM_RUN2_LREF_KB: A/B knobs)
  if constexpr (TP == 11) {
    const int n = (g.s_last - g.s_first) * g.dir + 1;
    const size_t total = run3_lds_bytes(n, 64) + run3_lref_bytes<AXIS>(ps);
    if (run3_lref_enabled(AXIS) && total <= run3_lref_limit()) {
      if (g.dir > 0) launch_run3_l<GS, AXIS, TP, 1, true>(ps, cp, g, slots, waves, stream);
      else launch_run3_l<GS, AXIS, TP, -1, true>(ps, cp, g, slots, waves, stream);
      return;
    }
  }
  if (g.dir > 0) launch_run3_l<GS, AXIS, TP, 1, false>(ps, cp, g, slots, waves, stream);
  else launch_run3_l<GS, AXIS, TP, -1, false>(ps, cp, g, slots, waves, stream);
}

// group = lanes per chain segment (32 or 16); windows of 3 and 5 always take 16, windows the fixed-size kernels do
// not cover (not square, or wider than 11) take the general kernel with 32.
template <int AXIS>
inline void launch_run3_axis(const PlaneSet& ps, const CostParams& cp, const SweepGeom& g, int slots, int waves,
                             int group, hipStream_t stream) {
  const int sq = (cp.pw == cp.ph && cp.pw <= 11) ? cp.pw : 0;
  const bool g16 = group <= 16;
  switch (sq) {
    case 3: launch_run3_d<16, AXIS, 3>(ps, cp, g, slots, waves, stream); break;
    case 5: launch_run3_d<16, AXIS, 5>(ps, cp, g, slots, waves, stream); break;
    case 7:
      if (g16) launch_run3_d<16, AXIS, 7>(ps, cp, g, slots, waves, stream);
      else launch_run3_d<32, AXIS, 7>(ps, cp, g, slots, waves, stream);
      break;
    case 9:
      if (g16) launch_run3_d<16, AXIS, 9>(ps, cp, g, slots, waves, stream);
      else launch_run3_d<32, AXIS, 9>(ps, cp, g, slots, waves, stream);
      break;
    case 11:
      if (g16) launch_run3_d<16, AXIS, 11>(ps, cp, g, slots, waves, stream);
      else launch_run3_d<32, AXIS, 11>(ps, cp, g, slots, waves, stream);
      break;
    default: launch_run3_d<32, AXIS, 0>(ps, cp, g, slots, waves, stream); break;
  }
}
inline void launch_sweep_run3(const PlaneSet& ps, const CostParams& cp, const SweepGeom& g, int slots, int waves,
                              int group, hipStream_t stream) {
  if (g.axis == 0) launch_run3_axis<0>(ps, cp, g, slots, waves, group, stream);
  else launch_run3_axis<1>(ps, cp, g, slots, waves, group, stream);
}

}  // namespace pm
