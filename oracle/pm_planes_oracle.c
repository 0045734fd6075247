/*
 * pm_planes_oracle.c -- CPU definition of PM_MODE_PLANES (see pm_planes_oracle.h: test infrastructure only;
 * this mode has no reference counterpart, the header says what it does share with the reference).
 *
 * Every float operation below is a single IEEE-754 binary32 operation (-ffp-contract=off); the window cost
 * is integer arithmetic, so any tap order gives the same sums.  Pixels of one pass never read what the same
 * pass writes (red pixels read black neighbours, the view pass reads the other view, refinement reads the
 * pixel itself), so the loops may run in any order and with any thread count.
 */
#include "pm_planes_oracle.h"

#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

#include "pm_oracle.h"

enum { ST_INIT = 0, ST_REFINE = 1 };

void pmo_planes_params_default(pmo_planes_params* p) {
  memset(p, 0, sizeof(*p));
  p->n_iters = 8;
  p->patch = 11;
  p->max_disp = 128;
  p->refine_steps = 3;
  for (int i = 0; i < PMO_PL_MAX_ITERS; ++i) p->refine_amp[i] = (float)(32.0 / pow(2.0, (double)i));
  p->slope_max = 1.0f;
  p->slope_init = 0.25f;
  p->slope_per_disp = 1.0f / 64.0f;
  p->alpha = 0.7f;      /* test/stereo_matching/patchmatch_test.cpp:35-37 */
  p->tau_color = 50.0f;
  p->tau_grad = 20.0f;
  p->seed = 123;        /* patchmatch.cpp:146 */
  p->left_right_check = 1;
  p->lr_tol = 1.0f;
  p->state_f16 = 0;
  p->nthreads = 1;
  p->window = PMO_PL_WINDOW_CHECKER;
  p->neighbours = PMO_PL_NEIGH_FOUR;
}

/* One random 32-bit word per (stage, iteration, step, view, draw, pixel): a 64-bit counter key, mixed
 * (splitmix64 finaliser) and advanced by ONE step of cv::RNG's multiply-with-carry generator
 * (OpenCV 3.4 core/operations.hpp: state = (uint32)state * 4164903690 + (state >> 32)), of which the low
 * word is the output -- the generator of patchmatch.cpp:146, addressed by counter instead of run row-major,
 * so that every pixel can draw its numbers independently. */
uint32_t pmo_planes_rand(uint64_t seed, int stage, int it, int k, int view, int draw, int x, int y) {
  const uint64_t tag = (uint64_t)stage | ((uint64_t)it << 4) | ((uint64_t)k << 12) | ((uint64_t)view << 20) |
                       ((uint64_t)draw << 24);
  uint64_t s = seed + 0x9E3779B97F4A7C15ull * (tag + 1);
  s ^= ((uint64_t)(uint32_t)y << 32) | (uint64_t)(uint32_t)x;
  s ^= s >> 30;
  s *= 0xBF58476D1CE4E5B9ull;
  s ^= s >> 27;
  s *= 0x94D049BB133111EBull;
  s ^= s >> 31;
  s = (uint64_t)(uint32_t)s * 4164903690u + (s >> 32);
  return (uint32_t)s;
}
/* uniform [-1, 1]: (float)(int)r * 2^-31, the conversion of randf_32f (OpenCV 3.4 core/rand.cpp) */
static inline float rand_pm1(uint32_t r) { return (float)(int32_t)r * 4.656612873077392578125e-10f; }
/* uniform [0, 1) */
static inline float rand_01(uint32_t r) { return (float)(r >> 8) * 5.9604644775390625e-08f; }

/* binary32 -> binary16 (round to nearest even) -> binary32, in software: gcc 11 has no _Float16 on x86 */
static uint16_t f32_to_f16(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  x &= 0x7fffffffu;
  if (x > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);
  if (x >= 0x47800000u) return (uint16_t)(sign | 0x7c00u);
  if (x >= 0x38800000u) {
    const uint32_t mant = x & 0x7fffffu, e = (x >> 23) - 112u;
    uint32_t h = (e << 10) | (mant >> 13);
    const uint32_t rem = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;
    return (uint16_t)(sign | h);
  }
  if (x < 0x33000000u) return (uint16_t)sign;
  const uint32_t e = x >> 23, m = (x & 0x7fffffu) | 0x800000u;
  const int shift = 126 - (int)e;
  uint32_t h = m >> shift;
  const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
  if (rem > half || (rem == half && (h & 1u))) ++h;
  return (uint16_t)(sign | h);
}
static float f16_to_f32(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3ffu;
  uint32_t x;
  if (e == 0) {
    const float v = (float)m * 5.9604644775390625e-08f;
    memcpy(&x, &v, 4);
    x |= sign;
  } else if (e == 31) {
    x = sign | 0x7f800000u | (m << 13);
  } else {
    x = sign | ((e + 112u) << 23) | (m << 13);
  }
  float f;
  memcpy(&f, &x, 4);
  return f;
}
float pmo_planes_quant_f16(float v) { return f16_to_f32(f32_to_f16(v)); }

static inline float quant(const pmo_planes_params* p, float v) { return p->state_f16 ? pmo_planes_quant_f16(v) : v; }
static inline float slope_bound(const pmo_planes_params* p) { return quant(p, p->slope_max); }
static inline float clamp_slope(float v, float smax) { return fminf(fmaxf(v, -smax), smax); }
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

int pmo_planes_tap(int window, int i, int j) {
  if (window == PMO_PL_WINDOW_CHECKER) return ((i + j) & 1) == 0;
  if (window == PMO_PL_WINDOW_EVEN_COLS) return (j & 1) == 0;
  return 1;
}
int pmo_planes_taps(int window, int P) {
  int n = 0;
  for (int i = 0; i < P; ++i)
    for (int j = 0; j < P; ++j) n += pmo_planes_tap(window, i, j);
  return n;
}

float pmo_planes_cost(const pmo_planes_params* p, const pmo_planes_view* im, int x, int y, float a, float b, float z) {
  const int P = p->patch, h = P / 2, rows = im->rows, cols = im->cols;
  const int Z = (int)rintf(z * 65536.0f), A = (int)rintf(a * 65536.0f), B = (int)rintf(b * 65536.0f);
  int sc = 0, sg = 0;
  for (int i = 0; i < P; ++i) {
    const size_t row = (size_t)clampi(y + i - h, 0, rows - 1) * cols;
    for (int j = 0; j < P; ++j) {
      if (!pmo_planes_tap(p->window, i, j)) continue;
      const int D = Z + A * (j - h) + B * (i - h);
      const int X = (x + j - h) * 65536 - D;
      const int c0 = X >> 16; /* floor: gcc shifts negative ints arithmetically */
      const unsigned w1 = (unsigned)(X >> 8) & 255u, w0 = 256u - w1;
      const size_t o0 = row + clampi(c0, 0, cols - 1), o1 = row + clampi(c0 + 1, 0, cols - 1);
      const unsigned p0 = im->tgt8[o0] | ((unsigned)im->tgtg8[o0] << 16);
      const unsigned p1 = im->tgt8[o1] | ((unsigned)im->tgtg8[o1] << 16);
      const unsigned s = p0 * w0 + p1 * w1 + 0x00800080u;
      const size_t ro = row + clampi(x + j - h, 0, cols - 1);
      sc += abs((int)im->ref8[ro] - (int)((s >> 8) & 255u));
      sg += abs((int)im->refg8[ro] - (int)(s >> 24));
    }
  }
  const float inv_n = 1.0f / (float)pmo_planes_taps(p->window, P);
  const float mc = (float)sc * inv_n, mg = (float)sg * inv_n;
  const float t0 = p->alpha * fminf(mc, p->tau_color);
  const float t1 = (1.0f - p->alpha) * fminf(mg, p->tau_grad);
  return t0 + t1;
}

static inline int admissible(const pmo_planes_params* p, int x, float z) {
  const float zmax = fminf((float)p->max_disp, (float)x);
  return z >= 0.0f && z <= zmax;
}

/* Offer candidate (ca, cb, cz) -- slopes already clamped -- to the pixel. */
static void offer(const pmo_planes_params* p, const pmo_planes_view* im, int x, int y, float ca, float cb, float cz,
                  float* a, float* b, float* z, float* cost) {
  ca = quant(p, ca);
  cb = quant(p, cb);
  cz = quant(p, cz);
  if (!admissible(p, x, cz)) return;
  if (ca == *a && cb == *b && cz == *z) return;
  const float c = quant(p, pmo_planes_cost(p, im, x, y, ca, cb, cz));
  if (c < *cost) {
    *a = ca;
    *b = cb;
    *z = cz;
    *cost = c;
  }
}

void pmo_planes_init(const pmo_planes_params* p, const pmo_planes_view* im, int view, const float* seed,
                     pmo_planes_state* st) {
  const int rows = im->rows, cols = im->cols;
  const float smax = slope_bound(p);
#pragma omp parallel for schedule(dynamic, 4) num_threads(p->nthreads)
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      const size_t o = (size_t)y * cols + x;
      const float zmax = fminf((float)p->max_disp, (float)x);
      const float s = seed ? seed[o] : 0.0f;
      const float u = rand_01(pmo_planes_rand(p->seed, ST_INIT, 0, 0, view, 0, x, y));
      float z = s > 0.0f ? fminf(s, zmax) : u * zmax;
      float a = p->slope_init * rand_pm1(pmo_planes_rand(p->seed, ST_INIT, 0, 0, view, 1, x, y));
      float b = p->slope_init * rand_pm1(pmo_planes_rand(p->seed, ST_INIT, 0, 0, view, 2, x, y));
      a = quant(p, clamp_slope(a, smax));
      b = quant(p, clamp_slope(b, smax));
      z = quant(p, z);
      if (!(z <= zmax)) z = quant(p, 0.0f); /* f16 rounding may step over zmax: such a pixel starts at 0 */
      st->a[o] = a;
      st->b[o] = b;
      st->z[o] = z;
      st->cost[o] = quant(p, pmo_planes_cost(p, im, x, y, a, b, z));
    }
}

void pmo_planes_spatial(const pmo_planes_params* p, const pmo_planes_view* im, pmo_planes_state* st, int arg) {
  const int rows = im->rows, cols = im->cols;
  const int parity = arg & 1, odd_it = (arg >> 1) & 1;
  const int two = p->neighbours == PMO_PL_NEIGH_TWO;
  const int lu = !two || !odd_it, rd = !two || odd_it; /* left + up / right + down */
#pragma omp parallel for schedule(dynamic, 4) num_threads(p->nthreads)
  for (int y = 0; y < rows; ++y)
    for (int x = (y + parity) & 1; x < cols; x += 2) {
      const size_t o = (size_t)y * cols + x;
      float a = st->a[o], b = st->b[o], z = st->z[o], c = st->cost[o];
      /* the neighbour's plane evaluated at this pixel: z_n + a_n*(x - x_n) + b_n*(y - y_n) */
      if (lu && x > 0) offer(p, im, x, y, st->a[o - 1], st->b[o - 1], st->z[o - 1] + st->a[o - 1], &a, &b, &z, &c);
      if (rd && x < cols - 1) offer(p, im, x, y, st->a[o + 1], st->b[o + 1], st->z[o + 1] - st->a[o + 1], &a, &b, &z, &c);
      if (lu && y > 0) offer(p, im, x, y, st->a[o - cols], st->b[o - cols], st->z[o - cols] + st->b[o - cols], &a, &b, &z, &c);
      if (rd && y < rows - 1)
        offer(p, im, x, y, st->a[o + cols], st->b[o + cols], st->z[o + cols] - st->b[o + cols], &a, &b, &z, &c);
      st->a[o] = a;
      st->b[o] = b;
      st->z[o] = z;
      st->cost[o] = c;
    }
}

/* The pixel (x, y) of this view currently matches column x - z of its target image, which is column
 * xo = (cols-1) - (x - z) of the OTHER view's reference image in that view's (mirrored) coordinates.  The
 * plane stored there, d = zo + ao*(u - xo_i) + bo*(v - y), describes the same surface; seen from this view
 * it is  d = zo + a'*(x - xc) + b'*(y' - y)  with  a' = -ao/(1 - ao), b' = bo/(1 - ao)  and
 * xc = (cols-1) - (xo_i - zo)  the column of this view that xo_i matches. */
void pmo_planes_view_prop(const pmo_planes_params* p, const pmo_planes_view* im, pmo_planes_state* st,
                          const pmo_planes_state* other) {
  const int rows = im->rows, cols = im->cols;
  const float smax = slope_bound(p);
#pragma omp parallel for schedule(dynamic, 4) num_threads(p->nthreads)
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      const size_t o = (size_t)y * cols + x;
      float a = st->a[o], b = st->b[o], z = st->z[o], c = st->cost[o];
      const float xo = (float)(cols - 1 - x) + z;
      const int xoi = clampi((int)rintf(xo), 0, cols - 1);
      const size_t oo = (size_t)y * cols + xoi;
      const float ao = other->a[oo], bo = other->b[oo], zo = other->z[oo];
      const float den = 1.0f - ao;
      if (den >= 0.25f) {
        const float na = (-ao) / den, nb = bo / den;
        const float xc = (float)(cols - 1 - xoi) + zo;
        const float dx = (float)x - xc;
        const float t = na * dx;
        const float nz = zo + t;
        offer(p, im, x, y, clamp_slope(na, smax), clamp_slope(nb, smax), nz, &a, &b, &z, &c);
      }
      st->a[o] = a;
      st->b[o] = b;
      st->z[o] = z;
      st->cost[o] = c;
    }
}

void pmo_planes_refine(const pmo_planes_params* p, const pmo_planes_view* im, int view, int it, pmo_planes_state* st) {
  const int rows = im->rows, cols = im->cols;
  const float smax = slope_bound(p);
#pragma omp parallel for schedule(dynamic, 4) num_threads(p->nthreads)
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      const size_t o = (size_t)y * cols + x;
      float a = st->a[o], b = st->b[o], z = st->z[o], c = st->cost[o];
      float dz = p->refine_amp[it];
      for (int k = 0; k < p->refine_steps; ++k) {
        const float ds = dz * p->slope_per_disp;
        const float u0 = rand_pm1(pmo_planes_rand(p->seed, ST_REFINE, it, k, view, 0, x, y));
        const float u1 = rand_pm1(pmo_planes_rand(p->seed, ST_REFINE, it, k, view, 1, x, y));
        const float u2 = rand_pm1(pmo_planes_rand(p->seed, ST_REFINE, it, k, view, 2, x, y));
        const float t0 = dz * u0, t1 = ds * u1, t2 = ds * u2;
        const float nz = z + t0, na = a + t1, nb = b + t2;
        offer(p, im, x, y, clamp_slope(na, smax), clamp_slope(nb, smax), nz, &a, &b, &z, &c);
        dz = dz * 0.5f;
      }
      st->a[o] = a;
      st->b[o] = b;
      st->z[o] = z;
      st->cost[o] = c;
    }
}

void pmo_planes_prepare(const uint8_t* left, const uint8_t* right, int rows, int cols, uint8_t* view0, uint8_t* view1) {
  const size_t n = (size_t)rows * cols;
  float* g = (float*)malloc(sizeof(float) * n);
  const uint8_t* src[2] = {left, right};
  /* view 0: ref = L, tgt = R;  view 1: ref = mirrored R, tgt = mirrored L (patchmatch_gpu.cu:357-368) */
  uint8_t* img_dst0[2] = {view0, view0 + 2 * n};  /* L -> ref of view 0, R -> tgt of view 0 */
  uint8_t* img_dst1[2] = {view1 + 2 * n, view1};  /* L -> tgt of view 1, R -> ref of view 1 */
  for (int s = 0; s < 2; ++s) {
    pmo_gradient_magnitude(src[s], rows, cols, g);
    for (int y = 0; y < rows; ++y)
      for (int x = 0; x < cols; ++x) {
        const size_t o = (size_t)y * cols + x, om = (size_t)y * cols + (cols - 1 - x);
        const float gv = g[o];
        const int gi = (int)rintf(gv);
        const uint8_t g8 = (uint8_t)(gv != gv ? 0 : (gi < 0 ? 0 : (gi > 255 ? 255 : gi)));
        img_dst0[s][o] = src[s][o];
        img_dst0[s][n + o] = g8;
        img_dst1[s][om] = src[s][o];
        img_dst1[s][n + om] = g8;
      }
  }
  free(g);
}

void pmo_planes_match(const pmo_planes_params* p, const uint8_t* left, const uint8_t* right, int rows, int cols,
                      const float* seed_l, const float* seed_r, float* disp_l, float* disp_r, float* planes_out) {
  const size_t n = (size_t)rows * cols;
  const int nv = p->left_right_check ? 2 : 1;
  uint8_t* planes8 = (uint8_t*)malloc(8 * n);
  pmo_planes_prepare(left, right, rows, cols, planes8, planes8 + 4 * n);
  float* stf = (float*)malloc(sizeof(float) * 8 * n);
  float* seed_rm = NULL;
  if (seed_r && nv == 2) {
    seed_rm = (float*)malloc(sizeof(float) * n);
    pmo_flip_h_f32(seed_r, seed_rm, rows, cols);
  }
  pmo_planes_view im[2];
  pmo_planes_state st[2];
  for (int v = 0; v < 2; ++v) {
    const uint8_t* b = planes8 + (size_t)v * 4 * n;
    im[v].rows = rows;
    im[v].cols = cols;
    im[v].ref8 = b;
    im[v].refg8 = b + n;
    im[v].tgt8 = b + 2 * n;
    im[v].tgtg8 = b + 3 * n;
    float* f = stf + (size_t)v * 4 * n;
    st[v].a = f;
    st[v].b = f + n;
    st[v].z = f + 2 * n;
    st[v].cost = f + 3 * n;
  }
  for (int v = 0; v < nv; ++v) pmo_planes_init(p, &im[v], v, v == 0 ? seed_l : seed_rm, &st[v]);
  for (int it = 0; it < p->n_iters; ++it) {
    for (int par = 0; par < 2; ++par)
      for (int v = 0; v < nv; ++v) pmo_planes_spatial(p, &im[v], &st[v], par + 2 * it);
    /* per view: candidates from the other view, then the view's own random refinement */
    for (int v = 0; v < nv; ++v) {
      if (nv == 2) pmo_planes_view_prop(p, &im[v], &st[v], &st[1 - v]);
      pmo_planes_refine(p, &im[v], v, it, &st[v]);
    }
  }
  /* disparity maps; consistency mask on the left map only, as MaskOcclusions does (patchmatch_gpu.cu:273-295) */
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      const size_t o = (size_t)y * cols + x;
      float dl = st[0].z[o];
      if (nv == 2) {
        const float fx = (float)x - dl;
        const int xt = clampi((int)rintf(fx), 0, cols - 1);
        const float dr = st[1].z[(size_t)y * cols + (cols - 1 - xt)];
        const float df = dl - dr;
        if (fabsf(df) > p->lr_tol) dl = 0.0f;
        if (disp_r) disp_r[o] = st[1].z[(size_t)y * cols + (cols - 1 - x)];
      }
      disp_l[o] = dl;
    }
  if (planes_out) memcpy(planes_out, stf, sizeof(float) * 4 * n * nv);
  free(seed_rm);
  free(stf);
  free(planes8);
}
