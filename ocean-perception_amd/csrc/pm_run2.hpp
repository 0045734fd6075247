// pm_run2.hpp -- PM_ENGINE_RUNBLK2: the run engine with TWO chain segments per wavefront.
//
// PMC profiles show the run-step kernels are VALU-issue bound (SQ_ACTIVE_INST_VALU ~ 86 % of SIMD
// time) while a 64-lane step consumes on average only ~8 positions, i.e. ~18 of the 64 window lines
// it computes.  Here a wavefront carries two independent segments of the same chain, one per 32-lane
// half ("group"): each instruction of a step now serves two steps, and a 32-line strip wastes far less
// (win = 11: up to 21 positions per step and group).  Everything that was wave-uniform in
// pm_run.hpp::run_step (position index, candidate, reference lane, bilinear parameters, outcome) is
// group-uniform here and lives in vector registers; cross-lane traffic stays inside a group:
// ballots are split into their 32-bit halves, broadcasts use ds_bpermute.  The DPP neighbour tap and
// the sliding window sum may cross from lane 31 into lane 32, which only touches lines no position of
// group 0 uses (its last lane is the spare one).
// The run step itself, its exactness arguments and the fix-up scheme are described in pm_run.hpp; results are
// bit-identical to the serial and wave engines and the oracle.
#pragma once

#include "pm_run.hpp"

namespace pm {

constexpr int kGroup = 32;   // default group width; 16 (four segments per wavefront) is the other choice
constexpr int kLref4Stride = 7;  // dwords per image row of the column sweeps' staged reference bytes (odd)

// Occupancy experiment knob: -DPM_RUNBLK2_MIN_WAVES=8 caps the kernel at 64 VGPRs (8 waves per SIMD).
#ifdef PM_RUNBLK2_MIN_WAVES
#define PM_RUNBLK2_BOUNDS __launch_bounds__(64 * kMaxSegWaves, PM_RUNBLK2_MIN_WAVES)
#else
#define PM_RUNBLK2_BOUNDS __launch_bounds__(64 * kMaxSegWaves)
#endif

struct RunStep2 {
  // group-uniform
  int advance;   // positions resolved (0 if the group is idle)
  int rej_pos;   // -1: none
  float rej_d0;  // value the position rej_pos holds after the step = candidate of the next step
  // per lane
  int mpos;
  bool adopt;
  float d0, c0, cost;
#ifdef PM_RUN2_TIMING
  long long t[5];  // s_memtime at: entry, after the need ballot, after the line sums, after the cost, at return
#endif
#ifdef PM_RUN2_STATS
  bool evald, g_need, pred_ok;  // pred_ok: the step ended in a reject at its first evaluated position
#endif
};
#ifdef PM_RUN2_TIMING
#define PM_T(k) st.t[k] = clock64()
#else
#define PM_T(k)
#endif

template <int GS, int AXIS, int TPW, int TPH>
__device__ __forceinline__ int run2_nd(const CostParams& cp) {
  return AXIS == 0 ? GS - (TPW > 0 ? TPW : cp.pw) : GS - (TPH > 0 ? TPH : cp.ph) + 1;
}

// Ballot of this lane's group (GS = 32, 16 or 8 lanes), in the low GS bits.
template <int GS>
__device__ __forceinline__ unsigned gballot(bool p, int gbase) {
  const unsigned long long b = __builtin_amdgcn_ballot_w64(p);  // (the int form costs a v_cndmask + v_cmp per ballot)
  if (GS == 32) return gbase ? (unsigned)(b >> 32) : (unsigned)b;
  return (unsigned)(b >> gbase) & ((1u << GS) - 1u);
}

// Buffer descriptors of one view's planes: MUBUF addressing = descriptor base + SGPR offset + VGPR
// offset + immediate, so a window row costs no VALU address arithmetic -- the row offset rides in the
// scalar operand, the lane's column in the vector operand (which is the same for every row).
struct RowBufs {
  __amdgpu_buffer_rsrc_t ref8, refg8, tgt8, tgtg;
};
// Measured (profiles/r01f_ab_loads.txt): the MUBUF form was SLOWER in the lockstep configuration (7.26 vs 5.80 ms
// per frame) although it removes ~60 VALU address instructions per step, and is on par under per-view streams
// (4.08 vs 4.12 ms).  Default: plain global loads with the same scalar-row + vector-column addressing;
// -DPM_RUN2_GLOBAL_LOADS=0 selects the MUBUF form.
// 1: the reference pixel's colour and gradient bytes come from the packed u16 plane with one load
#ifndef PM_RUN2_REF_PK16
#define PM_RUN2_REF_PK16 1
#endif
#ifndef PM_RUN2_PK_GRAD
#define PM_RUN2_PK_GRAD 1
#endif
#ifndef PM_RUN2_GLOBAL_LOADS
#define PM_RUN2_GLOBAL_LOADS 1
#endif
// 1: two window lines per load from the pair planes (PlaneSet::rpg ...), colour bytes and gradients of the target in
// one 12-byte record: 12 instead of 33 loads per row-sweep step, 6 instead of 24 per column-sweep step.  The sweeps are bound by the number of memory instructions: one
// extra byte load per window line costs 28 % of the frame (A/B in DESIGN.md).
#ifndef PM_RUN2_PAIRS
#define PM_RUN2_PAIRS 1
#endif
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
// 32-bit byte offsets (a view's pair planes are far below 4 GiB): (wave-uniform base) + (VGPR offset) is the
// global_load ... v_off, s[base:base+1] form -- one v_add_lshl_u32 per load instead of a 64-bit address pair
__device__ __forceinline__ unsigned ld_u32(const uint32_t* base, unsigned elem) {
  return *(const uint32_t*)((const char*)base + (size_t)(elem << 2));
}
// one target record: .x / .y = the gradients of the pair's two lines, .c = their colour bytes (line 0 | line 1 << 8)
struct PairRec {
  float x, y;
  unsigned c;
};
// `byte_off` = 12 * element: the callers form it as 12 * (first element) -- one shift-add and one shift, no 32-bit
// multiply (quarter rate) -- plus a wave-uniform 12 * pitch per line pair
__device__ __forceinline__ PairRec ld_rec(const float* base, unsigned byte_off) {
  return *(const PairRec*)((const char*)base + (size_t)byte_off);
}
__device__ __forceinline__ unsigned rec_offset(unsigned elem) {
  unsigned t = (elem << 1) + elem;
  asm volatile("" : "+v"(t));  // opaque: otherwise the optimiser folds this back into a v_mul_lo_u32 by 12 per load
  return t << 2;
}
__device__ __forceinline__ int win_ld8(__amdgpu_buffer_rsrc_t rs, const uint8_t* base, int voff, int soff) {
#if PM_RUN2_GLOBAL_LOADS
  return ld_u8(base, (unsigned)(voff + soff));
#else
  return __builtin_amdgcn_raw_buffer_load_b8(rs, voff, soff, 0);
#endif
}
// voff4 / soff4 are byte offsets
__device__ __forceinline__ float win_ldf(__amdgpu_buffer_rsrc_t rs, const float* base, int voff4, int soff4) {
#if PM_RUN2_GLOBAL_LOADS
  return ld_f32(base, (unsigned)(voff4 + soff4));
#else
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff4, soff4, 0));
#endif
}
// The same for the transposed planes (column sweeps).
__device__ __forceinline__ RowBufs make_col_bufs(const View& v, const PlaneSet& ps) {
  const int bytes = (int)ps.plane_t;
  RowBufs r;
  r.ref8 = __builtin_amdgcn_make_buffer_rsrc((void*)v.tref8, 0, bytes, 0x00020000);
  r.refg8 = __builtin_amdgcn_make_buffer_rsrc((void*)v.trefg8, 0, bytes, 0x00020000);
  r.tgt8 = __builtin_amdgcn_make_buffer_rsrc((void*)v.ttgt8, 0, bytes, 0x00020000);
  r.tgtg = __builtin_amdgcn_make_buffer_rsrc((void*)v.ttgtg, 0, bytes * 4, 0x00020000);
  return r;
}
__device__ __forceinline__ RowBufs make_row_bufs(const View& v, const PlaneSet& ps) {
  const int bytes = (int)ps.plane;
  RowBufs r;
  r.ref8 = __builtin_amdgcn_make_buffer_rsrc((void*)v.ref8, 0, bytes, 0x00020000);
  r.refg8 = __builtin_amdgcn_make_buffer_rsrc((void*)v.refg8, 0, bytes, 0x00020000);
  r.tgt8 = __builtin_amdgcn_make_buffer_rsrc((void*)v.tgt8, 0, bytes, 0x00020000);
  r.tgtg = __builtin_amdgcn_make_buffer_rsrc((void*)v.tgtg, 0, bytes * 4, 0x00020000);
  return r;
}

// DIR = +1 / -1 fixes the sweep direction at compile time (no multiplications by the direction, no selects between
// the "first set bit" forms); 0 = read it from the geometry.
template <int GS, int AXIS, int TPW, int TPH, int DIR, bool LREF>
__device__ __forceinline__ RunStep2 run_step2(const View& v, const PlaneSet& ps, const CostParams& cp,
                                              const SweepGeom& g, int chain, bool act, int i, int n_end, float cand,
                                              const float* din, const float* cin) {
  const int lane = threadIdx.x & (kWave - 1);
  const int gl = lane & (GS - 1);
  const int gbase = lane & ~(GS - 1);
  const int pitch = ps.pitch, cols = ps.cols, rows = ps.rows;
  const int pw = TPW > 0 ? TPW : cp.pw, ph = TPH > 0 ? TPH : cp.ph;
  const int half_w = pw / 2, half_h = ph / 2;
  const int win = AXIS == 0 ? pw : ph;
  const int half = win / 2;
  const int nd = run2_nd<GS, AXIS, TPW, TPH>(cp);
  const int dir = DIR != 0 ? DIR : g.dir;
  const float shift = (float)(pw - 1) * 0.5f;
  const unsigned lanes_nd = (1u << nd) - 1u;

  RunStep2 st;
  PM_T(0);
  st.mpos = dir > 0 ? gl : nd - 1 - gl;
  const bool inr = act && (gl < nd) && (i + st.mpos < n_end);
  st.d0 = inr ? din[i + st.mpos + 1] : 0.f;
  st.c0 = inr ? cin[i + st.mpos + 1] : 0.f;
  const bool neutral = inr && (st.d0 == cand);
  auto first_pos = [&](unsigned m) -> int {  // m != 0
    return dir > 0 ? __ffs((int)m) - 1 : nd - 1 - (31 - __clz((int)m));
  };
  auto glane_of = [&](int m) -> int { return dir > 0 ? m : nd - 1 - m; };

  const unsigned need = gballot<GS>(inr && !neutral, gbase);
  const bool has_need = need != 0u;
  const int r = has_need ? first_pos(need) : 0;
  const int r_gl = glane_of(r);
  PM_T(1);

  const int pos = g.s_first + (i + st.mpos) * dir;
  const int px = AXIS == 0 ? pos : chain;
  float cx = (float)px - cand;
  const bool valid = cx >= (float)half_w;
  cx = cx - shift;
  const float fl = floorf(cx);
  const int ipx = (int)fl;
  const float a = cx - fl;
  const int delta = (px - half_w) - ipx;

  const unsigned valid_m = gballot<GS>(valid, gbase);
  const bool valid_r = has_need && ((valid_m >> r_gl) & 1u);
  // Reference bilinear parameters = those of the step's FIRST position, computed from (i, cand) alone with the
  // same float operations -- group-uniform without any cross-lane traffic, so the window loads do not wait for
  // a ds_bpermute round trip.  A position whose own (a, delta) differ (x - d crossed a binade since position
  // 0) is not decided in this step; if that is r itself the step just advances to r and the next one starts
  // there (then r IS the first position).  Column sweeps: x is the chain, all positions agree by construction.
  const int px0 = AXIS == 0 ? g.s_first + i * dir : chain;
  float cx0 = (float)px0 - cand;
  cx0 = cx0 - shift;
  const float fl0 = floorf(cx0);
  const float a_r = cx0 - fl0;
  const int delta_r = (px0 - half_w) - (int)fl0;
  const bool same = valid && (a == a_r) && (delta == delta_r);

#ifdef PM_RUN2_STATS
  st.evald = __any(valid_r);
  st.g_need = valid_r;
#endif
  st.cost = 0.f;
  if (__builtin_amdgcn_ballot_w64(valid_r) != 0ull) {  // at least one group evaluates; the other computes along and ignores the result
    const float ia_r = 1.f - a_r;
    CpuLerp l;
    l.a = a_r;
    l.ia = ia_r;
    l.a11 = __float2int_rn(ia_r * 65536.f);
    l.a12 = __float2int_rn(a_r * 65536.f);
    l.ipx = 0;
    const int c_i = g.s_first + i * dir;
    const int c_base = dir > 0 ? c_i - half : c_i - half - (nd - 1);
    unsigned sc = 0, sg = 0;
    if (AXIS == 0) {
      const int X = min(max(c_base + gl, 0), cols - 1);
      const int R0 = min(max(c_base + gl - delta_r, 0), cols - 1);
      const int R0x4 = R0 * 4;
      const RowBufs rb = make_row_bufs(v, ps);
      const int org = (chain - half_h) * pitch;  // wave-uniform: scalar offsets below
      if constexpr (TPH > 0 && PM_RUN2_PK_GRAD && PM_RUN2_PAIRS) {
        // Two window lines per load: pair m of the alignment of this chain holds image rows y0 + 2m, y0 + 2m + 1
        // (y0 = first window row).  The colour bytes stay packed -- the multiplies select their byte (SDWA) and ONE
        // DPP move brings the neighbour lane's pair, i.e. the second bilinear tap of both rows.
        constexpr int NPR = (TPH + 1) / 2;
        const int y0 = chain - half_h;                                   // wave-uniform
        const unsigned eo = (unsigned)(y0 & 1) * v.rp_stride + (unsigned)(y0 >> 1) * (unsigned)pitch;
        const unsigned cw = cpu_color_weights(l);
        const unsigned rb0 = rec_offset(eo + (unsigned)R0), rpitch12 = (unsigned)pitch * 12u;  // rpitch12: scalar
        unsigned ppv[NPR + 1];     // reference pairs
        unsigned tcol[2 * NPR + 2];  // colour lerp sums r0 * a11 + r1 * a12 + 2^15: the sample is byte 2 (< 2^24)
        float gv[2 * NPR + 1];
#pragma unroll
        for (int m = 0; m < NPR; ++m) {
          const unsigned em = eo + (unsigned)(m * pitch);
          unsigned pp = 0u;  // LREF only: colour | gradient << 8 of row 2m in the low half, of row 2m + 1 in the high half
          if constexpr (LREF) {
            pp = (unsigned)v.lds_ref[(2 * m) * v.lds_ref_pitch + X];
            if (2 * m + 1 < TPH) pp |= (unsigned)v.lds_ref[(2 * m + 1) * v.lds_ref_pitch + X] << 16;
          }
          const PairRec pg = ld_rec(v.rpg, rb0 + (unsigned)m * rpitch12);
          const unsigned pr = pg.c;
          const unsigned prn = (unsigned)wave_shl1((int)pr);
          gv[2 * m] = pg.x;
          gv[2 * m + 1] = pg.y;
          ppv[m] = pp;
          // (own byte | neighbour's byte << 16) of row 2m, of row 2m + 1: one v_perm each, then one v_dot2_u32_u16
          tcol[2 * m] = cpu_color_sum_pk(__builtin_amdgcn_perm(prn, pr, 0x0c040c00u), cw);
          tcol[2 * m + 1] = cpu_color_sum_pk(__builtin_amdgcn_perm(prn, pr, 0x0c050c01u), cw);
        }
        ppv[NPR] = 0u;
        tcol[2 * NPR] = tcol[2 * NPR + 1] = 0u;
        gv[2 * NPR] = 0.f;
        // gradient lerp sums, one per row (packed-f32 products, the neighbour's product arrives by DPP)
        float sgr[2 * NPR];
        const f32x2 ia2 = {l.ia, l.ia}, a2 = {l.a, l.a};
#pragma unroll
        for (int t = 0; t < TPH; t += 2) {
          const f32x2 gg = {gv[t], gv[t + 1]};
          const f32x2 pa = gg * ia2, pb = gg * a2;
          sgr[t] = pa.x + wave_shl1f(pb.x);
          if (t + 1 < TPH) sgr[t + 1] = pa.y + wave_shl1f(pb.y);
        }
        // FOUR rows per v_sad_u8: the four colour samples (byte 2 of their sums) and the four saturated gradient
        // samples (v_cvt_pk_u8_f32 drops each into its byte) are gathered into one dword each and meet the four
        // reference bytes gathered from two reference pairs -- instead of a shift, a byte extract and a v_sad_u8
        // per row and channel.  Rows the window does not have select the constant 0 on both sides.
        // reference bytes of four rows per dword: one 8-byte load per quad from the quad plane of this chain's
        // alignment (rows the window does not have are masked), or gathered from the staged lines (LREF)
        constexpr int NQR = (TPH + 3) / 4;
        unsigned rq_c[NQR], rq_g[NQR];
        if constexpr (!LREF) {
          const unsigned eq = (unsigned)(y0 & 3) * v.rq_stride + (unsigned)(y0 >> 2) * (unsigned)pitch + (unsigned)X;
#pragma unroll
          for (int q = 0; q < NQR; ++q) {
            const uint2 rr = *(const uint2*)((const char*)v.rqk + (size_t)((eq + (unsigned)(q * pitch)) << 3));
            const int rem = TPH - 4 * q;
            const unsigned mask = rem >= 4 ? 0xffffffffu : ((1u << (8 * rem)) - 1u);
            rq_c[q] = rr.x & mask;
            rq_g[q] = rr.y & mask;
          }
        }
#pragma unroll
        for (int q = 0; 4 * q < TPH; ++q) {
          const int r0 = 4 * q;
          const bool h1 = r0 + 1 < TPH, h2 = r0 + 2 < TPH, h3 = r0 + 3 < TPH;
          // samples: bytes 0..3 = rows r0..r0+3
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
          // references: pair 2q = rows r0, r0 + 1 (bytes c, g, c, g), pair 2q + 1 = rows r0 + 2, r0 + 3
          const unsigned selc = (h3 ? 0x06000000u : 0x0c000000u) | (h2 ? 0x00040000u : 0x000c0000u) |
                                (h1 ? 0x00000200u : 0x00000c00u) | 0x00u;
          const unsigned selg = (h3 ? 0x07000000u : 0x0c000000u) | (h2 ? 0x00050000u : 0x000c0000u) |
                                (h1 ? 0x00000300u : 0x00000c00u) | 0x01u;
          const unsigned rc4 = LREF ? __builtin_amdgcn_perm(ppv[2 * q + 1], ppv[2 * q], selc) : rq_c[q];
          const unsigned rg4 = LREF ? __builtin_amdgcn_perm(ppv[2 * q + 1], ppv[2 * q], selg) : rq_g[q];
          sc = __builtin_amdgcn_sad_u8(rc4, s4, sc);
          sg = __builtin_amdgcn_sad_u8(rg4, g4, sg);
        }
      } else if constexpr (TPH > 0 && PM_RUN2_PK_GRAD) {
        // gradient lerp g0 * (1 - a) + g1 * a with g1 = the neighbour lane's g0: both products of a lane's
        // own sample for two rows per packed-f32 multiply, the neighbour's product arrives by DPP inside
        // the add -- every product and sum is the same single IEEE operation as in cpu_acc_grad
        int lgv[TPH];
        float gv[TPH + 1];
#pragma unroll
        for (int t = 0; t < TPH; ++t) {
          const int so = org + t * pitch;
#if PM_RUN2_REF_PK16
          const int pk = LREF ? (int)v.lds_ref[t * v.lds_ref_pitch + X] : ld_u16(v.refpk, (unsigned)((X + so) * 2));
          const int l8 = pk & 0xff;
          lgv[t] = pk >> 8;
#else
          const int l8 = win_ld8(rb.ref8, v.ref8, X, so);
          lgv[t] = win_ld8(rb.refg8, v.refg8, X, so);
#endif
          const int r0 = win_ld8(rb.tgt8, v.tgt8, R0, so);
          gv[t] = win_ldf(rb.tgtg, v.tgtg, R0x4, so * 4);
          sc = cpu_acc_color(sc, l8, r0, wave_shl1(r0), l);
        }
        gv[TPH] = 0.f;
        const f32x2 ia2 = {l.ia, l.ia}, a2 = {l.a, l.a};
#pragma unroll
        for (int t = 0; t < TPH; t += 2) {
          const f32x2 gg = {gv[t], gv[t + 1]};
          const f32x2 pa = gg * ia2, pb = gg * a2;
          sg = cpu_acc_grad_sum(sg, lgv[t], pa.x + wave_shl1f(pb.x));
          if (t + 1 < TPH) sg = cpu_acc_grad_sum(sg, lgv[t + 1], pa.y + wave_shl1f(pb.y));
        }
      } else {
#pragma unroll
        for (int t = 0; t < ph; ++t) {
          const int so = org + t * pitch;
          const int l8 = win_ld8(rb.ref8, v.ref8, X, so);
          const int lg = win_ld8(rb.refg8, v.refg8, X, so);
          const int r0 = win_ld8(rb.tgt8, v.tgt8, R0, so);
          const float g0 = win_ldf(rb.tgtg, v.tgtg, R0x4, so * 4);
          const int r1 = wave_shl1(r0);
          const float g1 = wave_shl1f(g0);
          sc = cpu_acc_color(sc, l8, r0, r1, l);
          sg = cpu_acc_grad(sg, lg, g0, g1, l);
        }
      }
    } else {
      const int pt = ps.pitch_t;
      const int Y = min(max(c_base + gl, 0), rows - 1);
      // a group that does not evaluate may carry a meaningless delta_r: keep its addresses in range.
      // Window columns beyond cols - 1 read the replicated pad rows of the transposed planes (kTransPad).
      const int ipx_r = min(max((chain - half_w) - delta_r, 0), cols - 1);
      const int vb = ipx_r * pt + Y;  // group-uniform column, lane's row
      const int vb4 = vb * 4;
      const RowBufs cb = make_col_bufs(v, ps);
      const int lorg = (chain - half_w) * pt;  // wave-uniform: scalar offsets below
      if constexpr (TPW > 0 && PM_RUN2_PK_GRAD && PM_RUN2_PAIRS) {
        // samples 0 .. TPW of the lane's row = TPW + 1 consecutive image columns from ipx_r: whole pairs of the
        // alignment ipx_r & 1 (group-uniform, may differ between the groups of a wavefront)
        constexpr int NPC = (TPW + 2) / 2;
        // (both factors < 2^16: the 24-bit multiply is exact and full rate)
        const unsigned e0 = ((ipx_r & 1) ? v.cp_stride : 0u) + __umul24((unsigned)(ipx_r >> 1), (unsigned)pt) + (unsigned)Y;
        const unsigned cb0 = rec_offset(e0), cpitch12 = (unsigned)pt * 12u;  // cpitch12: scalar
        unsigned prv[NPC];
        float gv[2 * NPC + 1];
#pragma unroll
        for (int m = 0; m < NPC; ++m) {
          const unsigned em = e0 + (unsigned)(m * pt);
          const PairRec pg = ld_rec(v.cpg, cb0 + (unsigned)m * cpitch12);
          prv[m] = pg.c;
          gv[2 * m] = pg.x;
          gv[2 * m + 1] = pg.y;
        }
        gv[2 * NPC] = 0.f;
        // reference bytes of the lane's row, four window columns per dword: colour dwords, then gradient dwords.
        // LREF: staged in exactly that form (k_runblk2), six dwords per row; else gathered from TPW packed loads.
        constexpr int NQ = (TPW + 3) / 4;
        unsigned rc4[NQ], rg4[NQ];
        if constexpr (LREF) {
          unsigned y8 = (unsigned)Y << 3;     // Y * kLref4Stride (7) as a shift and a subtraction; opaque, or the
          asm volatile("" : "+v"(y8));       // optimiser turns it back into a (quarter-rate, 64-bit) multiply-add
          unsigned y7 = y8 - (unsigned)Y;
          asm volatile("" : "+v"(y7));
          const unsigned* rr = v.lds_ref4 + y7;
          static_assert(kLref4Stride == 7, "row stride of the staged reference bytes");
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            rc4[q] = rr[q];
            rg4[q] = rr[NQ + q];
          }
        } else {
          unsigned pkv[4 * NQ];
#pragma unroll
          for (int t = 0; t < 4 * NQ; ++t)
            pkv[t] = t < TPW ? (unsigned)ld_u16(v.trefpk, (unsigned)((Y + lorg + t * pt) * 2)) : 0u;
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            const unsigned lo = __builtin_amdgcn_perm(pkv[4 * q + 1], pkv[4 * q], 0x05010400u);      // c0 c1 g0 g1
            const unsigned hi = __builtin_amdgcn_perm(pkv[4 * q + 3], pkv[4 * q + 2], 0x05010400u);  // c2 c3 g2 g3
            rc4[q] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
            rg4[q] = __builtin_amdgcn_perm(hi, lo, 0x07060302u);
          }
        }
        const unsigned cw = cpu_color_weights(l);
        unsigned tcol[4 * NQ];
#pragma unroll
        for (int t = 0; t < 4 * NQ; ++t) {
          if (t < TPW) {
            // samples t, t + 1 as halfwords: both in pair t / 2 (t even) or one in each of two pairs (t odd)
            const unsigned r01 = (t % 2 == 0) ? __builtin_amdgcn_perm(0u, prv[t / 2], 0x0c010c00u)
                                              : __builtin_amdgcn_perm(prv[t / 2 + 1], prv[t / 2], 0x0c040c01u);
            tcol[t] = cpu_color_sum_pk(r01, cw);
          } else {
            tcol[t] = 0u;
          }
        }
        const f32x2 ia2 = {l.ia, l.ia}, a2 = {l.a, l.a};
        f32x2 pa[(TPW + 2) / 2], pb[(TPW + 2) / 2];
#pragma unroll
        for (int k = 0; k < (TPW + 2) / 2; ++k) {
          const f32x2 gg = {gv[2 * k], gv[2 * k + 1]};
          pa[k] = gg * ia2;
          pb[k] = gg * a2;
        }
        // four window columns per v_sad_u8 (see the row sweeps above)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int t0 = 4 * q;
          const bool h1 = t0 + 1 < TPW, h2 = t0 + 2 < TPW, h3 = t0 + 3 < TPW;
          const unsigned u = __builtin_amdgcn_perm(tcol[t0 + 1], tcol[t0], h1 ? 0x0c0c0602u : 0x0c0c0c02u);
          unsigned s4 = u;
          if (h2) {
            const unsigned w = __builtin_amdgcn_perm(tcol[t0 + 3], tcol[t0 + 2], h3 ? 0x0c0c0602u : 0x0c0c0c02u);
            s4 = (w << 16) | u;
          }
          unsigned g4 = 0u;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int t = t0 + k;
            if (t < TPW) g4 = __builtin_amdgcn_cvt_pk_u8_f32(pa[t / 2][t % 2] + pb[(t + 1) / 2][(t + 1) % 2], k, g4);
          }
          sc = __builtin_amdgcn_sad_u8(rc4[q], s4, sc);
          sg = __builtin_amdgcn_sad_u8(rg4[q], g4, sg);
        }
      } else if constexpr (TPW > 0 && PM_RUN2_PK_GRAD) {
        int r0 = win_ld8(cb.tgt8, v.ttgt8, vb, 0);
        // samples g[0 .. PW] of the lane's row; both products per sample with packed-f32 multiplies
        int lgv[TPW];
        float gv[TPW + 2];
        gv[0] = win_ldf(cb.tgtg, v.ttgtg, vb4, 0);
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
          const int lso = lorg + t * pt;
          const int rso = (t + 1) * pt;
#if PM_RUN2_REF_PK16
          // the reference line of window column t: from the chain's LDS copy when the kernel staged it (LREF),
          // else one u16 load from the transposed packed plane
          const int pk = LREF ? (int)v.lds_ref[t * v.lds_ref_pitch + Y] : ld_u16(v.trefpk, (unsigned)((Y + lso) * 2));
          const int l8 = pk & 0xff;
          lgv[t] = pk >> 8;
#else
          const int l8 = win_ld8(cb.ref8, v.tref8, Y, lso);
          lgv[t] = win_ld8(cb.refg8, v.trefg8, Y, lso);
#endif
          const int r1 = win_ld8(cb.tgt8, v.ttgt8, vb, rso);
          gv[t + 1] = win_ldf(cb.tgtg, v.ttgtg, vb4, rso * 4);
          sc = cpu_acc_color(sc, l8, r0, r1, l);
          r0 = r1;
        }
        gv[TPW + 1] = 0.f;
        const f32x2 ia2 = {l.ia, l.ia}, a2 = {l.a, l.a};
        f32x2 pa[(TPW + 2) / 2], pb[(TPW + 2) / 2];
#pragma unroll
        for (int k = 0; k < (TPW + 2) / 2; ++k) {
          const f32x2 gg = {gv[2 * k], gv[2 * k + 1]};
          pa[k] = gg * ia2;
          pb[k] = gg * a2;
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t)
          sg = cpu_acc_grad_sum(sg, lgv[t], pa[t / 2][t % 2] + pb[(t + 1) / 2][(t + 1) % 2]);
      } else {
        int r0 = win_ld8(cb.tgt8, v.ttgt8, vb, 0);
        float g0 = win_ldf(cb.tgtg, v.ttgtg, vb4, 0);
#pragma unroll
        for (int t = 0; t < pw; ++t) {
          const int lso = lorg + t * pt;
          const int rso = (t + 1) * pt;
          const int l8 = win_ld8(cb.ref8, v.tref8, Y, lso);
          const int lg = win_ld8(cb.refg8, v.trefg8, Y, lso);
          const int r1 = win_ld8(cb.tgt8, v.ttgt8, vb, rso);
          const float g1 = win_ldf(cb.tgtg, v.ttgtg, vb4, rso * 4);
          sc = cpu_acc_color(sc, l8, r0, r1, l);
          sg = cpu_acc_grad(sg, lg, g0, g1, l);
          r0 = r1;
          g0 = g1;
        }
      }
    }
    const int line = (int)(sc | (sg << 16));
    PM_T(2);
    int wsum = line;
#pragma unroll
    for (int t = 1; t < win; ++t) wsum = line + wave_shl1(wsum);
    st.cost = cpu_cost_from_sums(wsum & 0xffff, (int)((unsigned)wsum >> 16), cp);
  }
#ifdef PM_RUN2_TIMING
  else st.t[2] = clock64();
#endif
  PM_T(3);

  // The run passes a position iff it ends up holding `cand`: already equal, or adopted.  Positions before
  // r are neutral by the definition of r, so the first position that does not pass is >= r; it is decided
  // in this step (q_real) if its candidate is not allowed at all (!valid) or was evaluated with its own
  // bilinear parameters (same); otherwise the next step starts there.  With no position in need every
  // position in reach passes and q is the end of reach.
  const bool adopt = valid_r && inr && !neutral && same && (st.cost < st.c0);
  const bool cont = neutral || adopt;
  const unsigned stop = gballot<GS>(!cont, gbase) & lanes_nd;
  const int q = stop ? first_pos(stop) : nd;
  const int q_gl = glane_of(min(q, nd - 1));
  const unsigned decided_m = gballot<GS>(inr && (same || !valid), gbase);
  const bool q_real = (q < nd) && ((decided_m >> q_gl) & 1u);
  const int advance = q_real ? q + 1 : q;
  const int rej_pos = q_real ? q : -1;
  const int src_gl = glane_of(max(rej_pos, 0));
  st.rej_d0 = __shfl(st.d0, gbase + src_gl, kWave);
  st.rej_pos = act ? rej_pos : -1;
  st.advance = act ? advance : 0;
  st.adopt = adopt && st.mpos < q;
#ifdef PM_RUN2_STATS
  st.pred_ok = act && has_need && q_real && q == r;
#endif
  PM_T(4);
  return st;
}

// One workgroup per chain; wavefront w carries segments 2w (lanes 0-31) and 2w+1 (lanes 32-63).
// Rounds and fix-up as described in pm_run.hpp, per group.
// grid = (chains, 1, slots), block = 64 * nw, dynamic LDS = 4 * (n + 1) floats + 2 * kMaxSegWaves + 3 words.
// SEM = 0: PM_SEM_CPU (run_step2 above); SEM = 1: PM_SEM_GPU (run_step2_gpu, pm_run_gpu.hpp).
template <int SEM, int GS, int AXIS, int TPW, int TPH, int DIR, bool LREF>
__device__ __forceinline__ RunStep2 run_step2_any(const View& v, const PlaneSet& ps, const CostParams& cp,
                                                  const SweepGeom& g, int chain, bool act, int i, int n_end,
                                                  float cand, const float* din, const float* cin);


// LREF (column sweeps, PM_SEM_CPU): the packed reference lines of the chain -- window columns chain - pw/2 .. + pw/2,
// all image rows -- are staged in LDS once per workgroup; a step then reads its 11 reference values per lane from
// LDS instead of issuing 11 of its 33 global loads (the steps of a segment re-read almost the same lines).
template <int SEM, int GS, int AXIS, int TPW, int TPH, int DIR, bool LREF>
__global__ void PM_RUNBLK2_BOUNDS k_runblk2(PlaneSet ps, CostParams cp, SweepGeom g, int seg_len) {
  extern __shared__ float lds[];
  const int n = (g.s_last - g.s_first) * g.dir + 1;
  const int n1 = (n + 1 + 3) & ~3;
  float* din = lds;
  float* cin = lds + n1;
  float* dout = lds + 2 * n1;
  float* cout = lds + 3 * n1;
  // [nseg + 1] last values + [2] change flags, sized by the launch: at 1280 columns and 4 wavefronts the
  // block then needs 20 396 B, i.e. EIGHT blocks fit the CU's 160 KB (a fixed-size tail made it seven)
  float* s_last = lds + 4 * n1;
  int* s_changed = (int*)(lds + 4 * n1 + (kWave / GS) * (blockDim.x >> 6) + 1);

  const int chain = g.c_lo + xcd_band_index(blockIdx.x, gridDim.x);
  if (!chain_active(ps, blockIdx.z, chain)) return;  // uniform for the workgroup, before any barrier
  View v = make_view(ps, blockIdx.z);
  if constexpr (LREF && AXIS == 1) {
    // behind the chain arrays and the flags: per image row (a position of the transposed chain) the TPW packed
    // reference values of window columns chain - pw/2 .. + pw/2 as bytes, four columns per dword: NQ colour dwords,
    // NQ gradient dwords, row stride kLref4Stride dwords (odd: lane = row reads without bank conflicts)
    constexpr int NQ = (TPW + 3) / 4;
    static_assert(2 * NQ <= kLref4Stride, "reference row does not fit its LDS stride");
    const int len = ps.rows;
    unsigned* sref4 = (unsigned*)(lds + 4 * n1 + (kWave / GS) * (blockDim.x >> 6) + 1 + 2);
    const uint16_t* src = v.trefpk + (size_t)(chain - TPW / 2) * ps.pitch_t;
    for (int e = threadIdx.x; e < NQ * len; e += blockDim.x) {
      const int q = e / len, row = e - q * len;
      unsigned cw = 0u, gw = 0u;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int t = 4 * q + k;
        if (t < TPW) {
          const unsigned pk = src[(size_t)t * ps.pitch_t + row];
          cw |= (pk & 0xffu) << (8 * k);
          gw |= (pk >> 8) << (8 * k);
        }
      }
      sref4[row * kLref4Stride + q] = cw;
      sref4[row * kLref4Stride + NQ + q] = gw;
    }
    v.lds_ref4 = sref4;
  } else if constexpr (LREF) {
    // row sweeps (off by default): [win][cols] u16 behind the chain arrays, row pitch = cols rounded up to even
    const int len = ps.cols;
    const int rp = (len + 1) & ~1;
    uint16_t* sref = (uint16_t*)(lds + 4 * n1 + (kWave / GS) * (blockDim.x >> 6) + 1 + 2);
    const uint16_t* src = v.refpk + (size_t)(chain - TPH / 2) * ps.pitch;
    for (int t = 0; t < TPH; ++t)
      for (int e = threadIdx.x; e < rp; e += blockDim.x) sref[t * rp + e] = e < len ? src[(size_t)t * ps.pitch + e] : (uint16_t)0;
    v.lds_ref = sref;
    v.lds_ref_pitch = rp;
  }
  const int lane = threadIdx.x & 63;
  constexpr int kPerWave = kWave / GS;
  const int gl = lane & (GS - 1);
  const int gbase = lane & ~(GS - 1);
  const int w = threadIdx.x >> 6;
  const int nw = blockDim.x >> 6;
  const int nseg = kPerWave * nw;
  const int sidx = kPerWave * w + lane / GS;
  const int nd = SEM == 0 ? run2_nd<GS, AXIS, TPW, TPH>(cp) : GS - 2;
  const int stride = AXIS == 0 ? g.dir : g.dir * ps.pitch;
  const ptrdiff_t first =
      AXIS == 0 ? (ptrdiff_t)chain * ps.pitch + g.s_first : (ptrdiff_t)g.s_first * ps.pitch + chain;

  for (int k = threadIdx.x; k <= n; k += blockDim.x) {
    const ptrdiff_t o = first + (ptrdiff_t)(k - 1) * stride;
    const float d = v.disp[o];
    const float cc = k > 0 ? v.cost[o] : 0.f;
    din[k] = d;
    cin[k] = cc;
    dout[k] = d;
    cout[k] = cc;
  }
  __syncthreads();

  const int i0 = sidx * seg_len;
  const int i1 = min(n, i0 + seg_len);
  const bool active = i0 < n;
  unsigned n_steps = 0, n_fix = 0, n_rounds = 0;
#ifdef PM_RUN2_STATS
  unsigned n_eval = 0, n_gsteps = 0, n_geval = 0, adv_sum = 0, n_pred = 0;
#endif
#ifdef PM_RUN2_TIMING
  long long tph[5] = {0, 0, 0, 0, 0}, t_prev_end = 0;
#endif

  // ---- round 1 ------------------------------------------------------------------------------------
  float in_used = active ? din[i0] : 0.f;
  float cand = in_used;
  {
    int i = i0;
    while (__builtin_amdgcn_ballot_w64(active && i < i1) != 0ull) {
      const bool act = active && i < i1;
      const RunStep2 st = run_step2_any<SEM, GS, AXIS, TPW, TPH, DIR, LREF>(v, ps, cp, g, chain, act, i, i1, cand, din, cin);
      ++n_steps;
#ifdef PM_RUN2_STATS
      if constexpr (SEM == 0) {
        n_eval += st.evald;
        n_gsteps += act;
        n_geval += act && st.g_need;
        n_pred += st.pred_ok;
        adv_sum += st.advance;
      }
#endif
#ifdef PM_RUN2_TIMING
      if constexpr (SEM == 0) {
        for (int k = 0; k < 4; ++k) tph[k] += st.t[k + 1] - st.t[k];
        if (t_prev_end) tph[4] += st.t[0] - t_prev_end;  // loop part between two steps (LDS writes, bookkeeping)
        t_prev_end = st.t[4];
      }
#endif
      if (st.mpos >= 0 && st.mpos < st.advance) {
        dout[i + st.mpos + 1] = st.mpos == st.rej_pos ? st.rej_d0 : cand;
        cout[i + st.mpos + 1] = st.adopt ? st.cost : st.c0;
      }
      if (st.rej_pos >= 0) cand = st.rej_d0;
      i += st.advance;
    }
  }
  float lastv = cand;
  if (active && gl == 0) s_last[sidx + 1] = lastv;
  if (threadIdx.x == 0) s_last[0] = in_used;

  // ---- fix-up rounds ---------------------------------------------------------------------------------
  for (int round = 1; round < nseg; ++round) {
    if (threadIdx.x == 0) s_changed[round & 1] = 0;
    __syncthreads();
    const float in = (active && sidx > 0) ? s_last[sidx] : in_used;
    bool redo = active && sidx > 0 && (in != in_used);
    bool new_last = false;
    if (__any(redo)) {
      if (redo) in_used = in;
      float c2 = in;
      int i = i0;
      bool merged = false;
      while (__builtin_amdgcn_ballot_w64(redo && !merged && i < i1) != 0ull) {
        const bool act = redo && !merged && i < i1;
        const RunStep2 st = run_step2_any<SEM, GS, AXIS, TPW, TPH, DIR, LREF>(v, ps, cp, g, chain, act, i, i1, c2, din, cin);
        ++n_fix;
        const bool mine = st.mpos >= 0 && st.mpos < st.advance;
        const float val = st.mpos == st.rej_pos ? st.rej_d0 : c2;
        const float spec = mine ? dout[i + st.mpos + 1] : 0.f;
        const unsigned eq = gballot<GS>(mine && val == spec, gbase);
        int ms = -1;
        if (eq) {  // first merged position in sweep order (lane <-> position mapping of the step function)
          const int lo_lane = __ffs((int)eq) - 1, hi_lane = 31 - __clz((int)eq);
          if (SEM == 0)
            ms = (DIR != 0 ? DIR : g.dir) > 0 ? lo_lane : nd - 1 - hi_lane;
          else
            ms = (DIR != 0 ? DIR : g.dir) > 0 ? lo_lane - 1 : nd - hi_lane;
        }
        const int wlim = ms >= 0 ? ms : st.advance;
        if (st.mpos >= 0 && st.mpos < wlim) {
          dout[i + st.mpos + 1] = val;
          cout[i + st.mpos + 1] = st.adopt ? st.cost : st.c0;
        }
        if (act && ms >= 0) merged = true;
        if (act && st.rej_pos >= 0) c2 = st.rej_d0;
        i += st.advance;
      }
      if (redo && !merged && c2 != lastv) {
        lastv = c2;
        new_last = true;
      }
    }
    __syncthreads();
    if (new_last && gl == 0) {
      s_last[sidx + 1] = lastv;
      s_changed[round & 1] = 1;
    }
    __syncthreads();
    ++n_rounds;
    if (!s_changed[round & 1]) break;
  }
  __syncthreads();
#ifdef PM_RUN2_TIMING
  if (ps.counters && lane == 0 && w == 0 && blockIdx.x % 16 == 0) {  // a sample of wavefronts: phase cycles of round 1
    for (int k = 0; k < 5; ++k) atomicAdd(&ps.counters[8 + k], (unsigned long long)tph[k]);
    atomicAdd(&ps.counters[13], (unsigned long long)n_steps);
  }
#endif
#ifdef PM_RUN2_STATS
  if (ps.counters) {
    if (lane == 0) {
      atomicAdd(&ps.counters[0], (unsigned long long)n_steps);
      atomicAdd(&ps.counters[1], (unsigned long long)n_fix);
      atomicAdd(&ps.counters[2], (unsigned long long)n_eval);
    }
    if (gl == 0) {
      atomicAdd(&ps.counters[3], (unsigned long long)n_gsteps);
      atomicAdd(&ps.counters[4], (unsigned long long)n_geval);
      atomicAdd(&ps.counters[5], (unsigned long long)adv_sum);
      atomicAdd(&ps.counters[6], (unsigned long long)n_pred);
    }
  }
#else
  if (ps.counters && lane == 0) {
    const int base = AXIS * 4;
    atomicAdd(&ps.counters[base + 0], (unsigned long long)n_steps);
    atomicAdd(&ps.counters[base + 1], (unsigned long long)n_fix);
    if (w == 0) atomicAdd(&ps.counters[base + 2], (unsigned long long)n_rounds);
    if (w == 0) atomicAdd(&ps.counters[base + 3], (unsigned long long)n);
  }
#endif

  for (int k = threadIdx.x + 1; k <= n; k += blockDim.x) {
    const float d = dout[k];
    if (d != din[k]) {
      const ptrdiff_t o = first + (ptrdiff_t)(k - 1) * stride;
      v.disp[o] = d;
      v.cost[o] = cout[k];
    }
  }
}

// LDS bytes of the staged reference lines (LREF): column sweeps keep kLref4Stride dwords per image row, row sweeps
// TPW lines of u16.
template <int AXIS, int TPW>
inline size_t run2_lref_bytes(const PlaneSet& ps) {
  if (AXIS == 1) return sizeof(unsigned) * (size_t)kLref4Stride * ps.rows;
  return sizeof(uint16_t) * (size_t)TPW * ((ps.cols + 1) & ~1);
}
template <int SEM, int GS, int AXIS, int TPW, int TPH, int DIR, bool LREF>
inline void launch_run2_kdl(const PlaneSet& ps, const CostParams& cp, const SweepGeom& g, int slots, int waves,
                            hipStream_t stream) {
  const int chains = g.c_hi - g.c_lo + 1;
  const int n = (g.s_last - g.s_first) * g.dir + 1;
  int nwv = waves < 1 ? 1 : (waves > kMaxSegWaves ? kMaxSegWaves : waves);
  const int per_block = (kWave / GS) * nwv;
  int len = (n + per_block - 1) / per_block;
  if (len < 8) len = 8;
  const int n1 = (n + 1 + 3) & ~3;
  size_t lds_bytes = sizeof(float) * (4 * (size_t)n1 + per_block + 1 + 2);
  if (LREF) lds_bytes += run2_lref_bytes<AXIS, TPW>(ps);
  allow_big_lds(k_runblk2<SEM, GS, AXIS, TPW, TPH, DIR, LREF>, lds_bytes);
  hipLaunchKernelGGL((k_runblk2<SEM, GS, AXIS, TPW, TPH, DIR, LREF>), dim3((unsigned)chains, 1, (unsigned)slots),
                     dim3(kWave * nwv), lds_bytes, stream, ps, cp, g, len);
}
// Column sweeps of the benchmark window stage their reference lines in LDS while that leaves at least four
// workgroups per CU (720 rows: 27 KB per workgroup); PM_RUN2_LREF=0 turns it off (A/B knob).
// PM_RUN2_LREF: bit 0 = row sweeps, bit 1 = column sweeps (default 2); PM_RUN2_LREF_KB: LDS budget per workgroup
inline bool run2_lref_enabled(int axis) {
  static const int v = [] {
    const char* e = getenv("PM_RUN2_LREF");
    return e ? atoi(e) : 2;
  }();
  return (v >> axis) & 1;
}
inline size_t run2_lref_limit() {
  static const size_t v = [] {
    const char* e = getenv("PM_RUN2_LREF_KB");
    return (size_t)(e ? atoi(e) : 40) * 1024;
  }();
  return v;
}
template <int SEM, int GS, int AXIS, int TPW, int TPH, int DIR>
inline void launch_run2_kd(const PlaneSet& ps, const CostParams& cp, const SweepGeom& g, int slots, int waves,
                           hipStream_t stream) {
  // (the column sweeps' staged form is read by the pair-plane path only)
  if constexpr (SEM == 0 && TPW == 11 && (AXIS == 0 || (PM_RUN2_PAIRS && PM_RUN2_PK_GRAD))) {
    const int n = (g.s_last - g.s_first) * g.dir + 1;
    const size_t total = sizeof(float) * 4 * (size_t)(n + 4) + run2_lref_bytes<AXIS, TPW>(ps) + 256;
    if (run2_lref_enabled(AXIS) && total <= run2_lref_limit()) {
      launch_run2_kdl<SEM, GS, AXIS, TPW, TPH, DIR, true>(ps, cp, g, slots, waves, stream);
      return;
    }
  }
  launch_run2_kdl<SEM, GS, AXIS, TPW, TPH, DIR, false>(ps, cp, g, slots, waves, stream);
}
// The benchmark window (11x11, PM_SEM_CPU) gets direction-specialised kernels; the others read the direction
// from the geometry (every instantiation costs build time).
template <int SEM, int GS, int AXIS, int TPW, int TPH>
inline void launch_run2_k(const PlaneSet& ps, const CostParams& cp, const SweepGeom& g, int slots, int waves,
                          hipStream_t stream) {
  if constexpr (SEM == 0 && TPW == 11) {
    if (g.dir > 0) launch_run2_kd<SEM, GS, AXIS, TPW, TPH, 1>(ps, cp, g, slots, waves, stream);
    else launch_run2_kd<SEM, GS, AXIS, TPW, TPH, -1>(ps, cp, g, slots, waves, stream);
  } else {
    launch_run2_kd<SEM, GS, AXIS, TPW, TPH, 0>(ps, cp, g, slots, waves, stream);
  }
}

// group: 32 or 16 lanes per segment.  16-lane groups need the window to leave positions in a strip
// (win <= 11); wider or non-square windows use 32.
template <int GS, int AXIS>
inline void launch_run2_axis(const PlaneSet& ps, const CostParams& cp, const SweepGeom& g, int slots, int waves,
                             hipStream_t stream) {
  if (cp.semantics != 0) {
    launch_run2_k<1, GS, AXIS, 3, 3>(ps, cp, g, slots, waves, stream);
    return;
  }
  const int sq = cp.pw == cp.ph ? cp.pw : 0;
  switch (sq) {
    case 3: launch_run2_k<0, GS, AXIS, 3, 3>(ps, cp, g, slots, waves, stream); break;
    case 5: launch_run2_k<0, GS, AXIS, 5, 5>(ps, cp, g, slots, waves, stream); break;
    case 7: launch_run2_k<0, GS, AXIS, 7, 7>(ps, cp, g, slots, waves, stream); break;
    case 9: launch_run2_k<0, GS, AXIS, 9, 9>(ps, cp, g, slots, waves, stream); break;
    case 11: launch_run2_k<0, GS, AXIS, 11, 11>(ps, cp, g, slots, waves, stream); break;
    default: launch_run2_k<0, 32, AXIS, 0, 0>(ps, cp, g, slots, waves, stream); break;
  }
}

}  // namespace pm
#include "pm_run_gpu.hpp"
namespace pm {

template <int SEM, int GS, int AXIS, int TPW, int TPH, int DIR, bool LREF>
__device__ __forceinline__ RunStep2 run_step2_any(const View& v, const PlaneSet& ps, const CostParams& cp,
                                                  const SweepGeom& g, int chain, bool act, int i, int n_end,
                                                  float cand, const float* din, const float* cin) {
  if constexpr (SEM == 0)
    return run_step2<GS, AXIS, TPW, TPH, DIR, LREF>(v, ps, cp, g, chain, act, i, n_end, cand, din, cin);
  else
    return run_step2_gpu<GS, AXIS>(v, ps, cp, g, chain, act, i, n_end, cand, din, cin);
}

// In place.  group = lanes per chain segment: 32 or 16; 8 for PM_SEM_GPU only (its window is 3 lanes).
inline void launch_sweep_run2(const PlaneSet& ps, const CostParams& cp, const SweepGeom& g, int slots, int waves,
                              int group, hipStream_t stream) {
  if (group == 8 && cp.semantics != 0) {
    if (g.axis == 0) launch_run2_k<1, 8, 0, 3, 3>(ps, cp, g, slots, waves, stream);
    else launch_run2_k<1, 8, 1, 3, 3>(ps, cp, g, slots, waves, stream);
    return;
  }
  // (8-lane groups for PM_SEM_CPU 3x3 windows were tried: bit-identical, but 2.82 vs 2.66 ms per frame with 16)
  const bool g16 = group <= 16 && (cp.semantics != 0 || (cp.pw == cp.ph && cp.pw <= 11));
  if (g.axis == 0) {
    if (g16) launch_run2_axis<16, 0>(ps, cp, g, slots, waves, stream);
    else launch_run2_axis<32, 0>(ps, cp, g, slots, waves, stream);
  } else {
    if (g16) launch_run2_axis<16, 1>(ps, cp, g, slots, waves, stream);
    else launch_run2_axis<32, 1>(ps, cp, g, slots, waves, stream);
  }
}

}  // namespace pm
