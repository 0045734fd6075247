/*
 * pm_planes_oracle.h -- CPU definition of the engine's slanted-plane mode (PM_MODE_PLANES).
 *
 * TEST INFRASTRUCTURE ONLY (same rule as pm_oracle.h): used by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg, never by the product.
 *
 * NO REFERENCE COUNTERPART.  BASELINE.json's north_star names random slanted-plane initialisation,
 * red-black spatial propagation, view propagation and random plane refinement; the reference keeps
 * one scalar disparity per pixel and has none of them (its GPU entry point,
 * /root/reference/src/vehicle/patchmatch_gpu/patchmatch_gpu.cu:379-411, is noise + four directional
 * sweeps).  This file therefore DEFINES the algorithm the HIP kernels (csrc/pm_planes.hpp) must
 * reproduce bit for bit; it is not a restatement of reference code and "parity" for this mode means
 * HIP == this definition on the same seeded random numbers, plus quality against synthetic truth.
 * What it shares with the reference path: the images and Sobel gradient (pm_oracle.c), the form of
 * the cost functor alpha*min(mean|dc|, tau_c) + (1-alpha)*min(mean|dg|, tau_g)
 * (test/stereo_matching/patchmatch_test.cpp:30-45), the cv::RNG multiply-with-carry step
 * (OpenCV 3.4 core/operations.hpp RNG::next) and the "right view = the same algorithm on the mirrored
 * (R, L) pair" structure of patchmatch_gpu.cu:357-368.
 *
 * State per pixel and view: plane (a, b, z) in pixel-local form -- z is the disparity AT the pixel,
 * d(x + dx, y + dy) = z + a*dx + b*dy -- and the cost of that plane.  Stored as f32 or as f16
 * (state_f16: every candidate is rounded to f16 BEFORE it is evaluated, so a stored cost is always the
 * cost of the stored plane; costs are rounded to f16 before they are compared and stored).
 *
 * Window cost of plane (a, b, z) at (x, y), window P x P, h = P/2, all in integers:
 *   Z = rint(z*2^16), A = rint(a*2^16), B = rint(b*2^16)
 *   tap (i, j): D = Z + A*(j-h) + B*(i-h);  X = ((x+j-h) << 16) - D;  c0 = X >> 16;  w1 = (X >> 8) & 255
 *   target sample = (P[c0]*(256-w1) + P[c0+1]*w1 + 128) >> 8 per channel (colour, saturated-u8 gradient),
 *   columns and rows clamped to the image (replicated border)
 *   Sc = sum |ref colour - sample|, Sg = sum |ref gradient - sample|
 *   cost = alpha*min(Sc*(1/N), tau_c) + (1-alpha)*min(Sg*(1/N), tau_g)      (float, one rounding per op)
 * A candidate is admissible iff 0 <= z <= min(max_disp, x) and |a|, |b| <= slope_max (slopes are clamped when
 * a candidate is formed).
 */
#ifndef PM_PLANES_ORACLE_H_
#define PM_PLANES_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PMO_PL_MAX_ITERS 16

typedef struct pmo_planes_params {
  int n_iters;                           /* iterations of {red, black, view, refine}                      */
  int patch;                             /* odd window side, 3..15 (11)                                  */
  int max_disp;                          /* disparities live in [0, max_disp] (128)                      */
  int refine_steps;                      /* R candidates per pixel and iteration (3)                     */
  float refine_amp[PMO_PL_MAX_ITERS];    /* dz of the first refinement step of iteration i (32 / 2^i)    */
  float slope_max;                       /* |a|, |b| bound (1.0)                                         */
  float slope_init;                      /* random initial slopes are uniform in +-slope_init (0.25)     */
  float slope_per_disp;                  /* slope range = dz * slope_per_disp (1/64)                     */
  float alpha, tau_color, tau_grad;      /* functor constants (0.7, 50, 20)                              */
  uint64_t seed;                         /* 123                                                          */
  int left_right_check;                  /* right view + consistency mask                                */
  float lr_tol;                          /* |dl - dr| above which the left disparity is zeroed (1.0)     */
  int state_f16;                         /* 0: f32 state, 1: f16 state                                   */
  int nthreads;
  int window;                            /* which taps of the P x P window count (PMO_PL_WINDOW_*)        */
  int neighbours;                        /* spatial stage: PMO_PL_NEIGH_FOUR (default) / PMO_PL_NEIGH_TWO  */
} pmo_planes_params;

/* The window's taps.  FULL: all P*P.  CHECKER (the default since round 4): tap (i, j) counts iff i + j is even -- the
 * centre, and every other tap in both directions: 61 of 121 for 11 x 11.  EVEN_COLS: columns j = 0, 2, 4 ... of every
 * row (66 of 121).  The cost divides by the number of taps that count. */
enum { PMO_PL_WINDOW_FULL = 0, PMO_PL_WINDOW_CHECKER = 1, PMO_PL_WINDOW_EVEN_COLS = 2 };
int pmo_planes_tap(int window, int i, int j);
int pmo_planes_taps(int window, int P);

/* The spatial stage's candidates.  FOUR: a pixel is offered the planes of its left, right, upper and lower neighbour.
 * TWO (an option, round 4): the left and the upper one in the colour passes of an even iteration, the right and the lower
 * one in those of an odd iteration -- planes travel down-right and up-left in turns, half the evaluations of the stage.
 * On the benchmark pairs: the same validity, 99.84-99.88 % of the valid pixels within 1 px (FOUR: 99.86-99.89 %), mean
 * absolute error + 12-13 %. */
enum { PMO_PL_NEIGH_FOUR = 0, PMO_PL_NEIGH_TWO = 1 };

void pmo_planes_params_default(pmo_planes_params* p);

/* Images of one view: reference and target colour and saturated gradient, rows*cols each. */
typedef struct pmo_planes_view {
  int rows, cols;
  const uint8_t* ref8;
  const uint8_t* refg8;
  const uint8_t* tgt8;
  const uint8_t* tgtg8;
} pmo_planes_view;

/* a, b, z, cost: rows*cols floats each (values are f16-representable when state_f16). */
typedef struct pmo_planes_state {
  float* a;
  float* b;
  float* z;
  float* cost;
} pmo_planes_state;

/* random numbers: cv::RNG's MWC step on a per-draw 64-bit counter key */
uint32_t pmo_planes_rand(uint64_t seed, int stage, int it, int k, int view, int draw, int x, int y);
float pmo_planes_quant_f16(float v);

float pmo_planes_cost(const pmo_planes_params* p, const pmo_planes_view* im, int x, int y, float a, float b, float z);

/* seed may be NULL; seed > 0 fixes the initial disparity of that pixel (view coordinates). */
void pmo_planes_init(const pmo_planes_params* p, const pmo_planes_view* im, int view, const float* seed,
                     pmo_planes_state* st);
/* arg = colour + 2 * iteration: colour 0 = the pixels with x + y even, 1 = the others; the iteration only matters with
 * PMO_PL_NEIGH_TWO (which pair of neighbours) */
void pmo_planes_spatial(const pmo_planes_params* p, const pmo_planes_view* im, pmo_planes_state* st, int arg);
/* `other` = the other view's state (its own mirrored coordinates) */
void pmo_planes_view_prop(const pmo_planes_params* p, const pmo_planes_view* im, pmo_planes_state* st,
                          const pmo_planes_state* other);
void pmo_planes_refine(const pmo_planes_params* p, const pmo_planes_view* im, int view, int it, pmo_planes_state* st);

/* Whole Match(): both views, I iterations, disparity maps out (right map in right-image coordinates).
 * seed_l / seed_r in left / right image coordinates or NULL.  planes_out (optional): 8 planes of rows*cols
 * floats: view 0 a, b, z, cost, view 1 (mirrored coordinates) a, b, z, cost. */
void pmo_planes_match(const pmo_planes_params* p, const uint8_t* left, const uint8_t* right, int rows, int cols,
                      const float* seed_l, const float* seed_r, float* disp_l, float* disp_r, float* planes_out);

/* The four u8 planes of both views from an image pair (view 1 = mirrored (R, L)); out: 4 planes of rows*cols bytes
 * per view: ref8, refg8, tgt8, tgtg8. */
void pmo_planes_prepare(const uint8_t* left, const uint8_t* right, int rows, int cols, uint8_t* view0, uint8_t* view1);

#ifdef __cplusplus
}
#endif
#endif
