// pm_launch.hip -- the translation unit of the scalar-mode kernels (pm_kernels.hpp): one launch function per kernel,
// declared in pm_handle.hpp, each enqueuing on the handle's current stream.  No other unit includes pm_kernels.hpp.
#include <algorithm>

#include "pm_handle.hpp"
#include "pm_kernels.hpp"
#include "pm_sweeps.hpp"
#include "pm_texmask.hpp"
#include "pm_tune.hpp"

namespace pm {
namespace eng {

namespace {
dim3 pixel_grid(int cols, int rows, int z) { return dim3((unsigned)((cols + 255) / 256), (unsigned)rows, (unsigned)z); }
}  // namespace

static SetupGrid setup_grid(pm_handle* h, const PlaneSet& ps, int n, int view) {
  SetupGrid sg{};
  sg.view = view;
  sg.tx = (unsigned)((ps.cols + 63) / 64);
  sg.ty = (unsigned)((ps.rows + 63) / 64);
  sg.tz = (unsigned)(n * (view < 0 ? 4 : 2));
  sg.with_lines = pair_planes_wanted(h) ? 1 : 0;  // the line-triple / quad planes of the run engine (pm_run3.hpp)
  if (sg.with_lines) {
    sg.lx = (unsigned)((ps.cols + 255) / 256);
    sg.ly = (unsigned)((ps.nrl + kLinesPerSetupThread - 1) / kLinesPerSetupThread);
    sg.lz = (unsigned)(n * (view < 0 ? 2 : 1));
    sg.cx = (unsigned)((ps.ncl + 31) / 32);
    sg.cy = (unsigned)((ps.rows + 63) / 64);
    sg.cz = (unsigned)(n * (view < 0 ? 2 : 1));
  }
  return sg;
}

void launch_prep(pm_handle* h, const PlaneSet& ps, const uint8_t* d_left, const uint8_t* d_right, int n, size_t stride,
                 int view, const PrepSeedMaps* seeds) {
  if (h->bgr) {
    hipLaunchKernelGGL(k_prep_bgr, dim3((unsigned)((ps.cols + 63) / 64), (unsigned)((ps.rows + kPrepBgrTileH - 1) / kPrepBgrTileH), (unsigned)n),
                       dim3(256), 0, h->stream, ps, *h->bgr);
    if (seeds) launch_seed(h, ps, seeds->l, seeds->r, n, view);
    return;
  }
  const PrepSeeds sd{seeds ? seeds->l : nullptr, seeds ? seeds->r : nullptr, seeds ? 1 : 0};
  hipLaunchKernelGGL(k_prep, pixel_grid(ps.cols, ps.rows, n), dim3(256), 0, h->stream, ps, d_left, d_right, stride, view,
                     sd);
}

void launch_prep_view(pm_handle* h, const PlaneSet& ps, const float* d_iml, const float* d_imr, const float* d_Gl,
                      const float* d_Gr, size_t stride) {
  hipLaunchKernelGGL(k_prep_view, pixel_grid(ps.cols, ps.rows, 1), dim3(256), 0, h->stream, ps, d_iml, d_imr, d_Gl, d_Gr,
                     stride);
}


int run_transpose(pm_handle* h, const PlaneSet& ps, int n, int view) {
  const SetupGrid sg = setup_grid(h, ps, n, view);
  PlaneSet pp = ps;
  if (sg.with_lines && !pp.rpg) {  // (a plane set made before the line planes existed; pm_create allocates them for the handles that use them)
    if (int rc = pair_planes_alloc(h)) return rc;
    pp.rpg = h->rpg;
    pp.rqk = h->rqk;
    pp.cpg = h->cpg;
  }
  hipLaunchKernelGGL(k_setup, dim3(setup_blocks(sg)), dim3(256), 0, h->stream, pp, sg);
  return launch_check(h, "transpose");
}

// seed maps (tightly packed [n][rows][cols], null = zeros) into the disparity planes; view 1 mirrored
void launch_seed(pm_handle* h, const PlaneSet& ps, const float* d_seed_l, const float* d_seed_r, int n, int view) {
  hipLaunchKernelGGL(k_seed, pixel_grid(ps.cols, ps.rows, n), dim3(256), 0, h->stream, ps, d_seed_l, d_seed_r,
                     (size_t)ps.cols, view);
}

int run_sweep(pm_handle* h, const PlaneSet& ps, const CostParams& cp, const SweepGeom& g, int slots,
              float amp) {
  const int chains = g.c_hi - g.c_lo + 1;
  if (chains <= 0 || (g.s_last - g.s_first) * g.dir < 0) return PM_OK;
  Launch l(h, g.axis == 0 ? PM_K_SWEEP_ROW : PM_K_SWEEP_COL);
  launch_sweep(ps, cp, g, slots, h->params.engine, amp, h->stream);  // pm_sweeps.hip
  return launch_check(h, "sweep");
}

// noise + clamp + cost of the current disparity; PM_SEM_CPU square windows use the LDS-tiled kernel
void launch_noise_cost(pm_handle* h, const PlaneSet& ps, const CostParams& cp, const Interior& in, float amount,
                       int slots, int keep_zero) {
  const bool tiled = cp.semantics == PM_SEM_CPU && cp.pw == cp.ph && !h->no_tiled;
  const dim3 tgrid((unsigned)((ps.cols + kTileW - 1) / kTileW), (unsigned)((ps.rows + kTileH - 1) / kTileH),
                   (unsigned)slots);
#ifdef PM_TUNING
  static const int dbg = [] {
    const char* e = pm::tune_env("PM_NOISE_DBG");
    return e ? atoi(e) : 0;
  }();
  keep_zero |= dbg << 8;
#endif
#define PM_NC_CASE(W)                                                                                                 \
  case W:                                                                                                             \
    hipLaunchKernelGGL((k_noise_cost_tiled<W, W>), tgrid, dim3(256), 0, h->stream, ps, cp, in, amount, keep_zero);    \
    return;
  if (tiled) {
    switch (cp.pw) {
      PM_NC_CASE(3)
      PM_NC_CASE(5)
      PM_NC_CASE(7)
      PM_NC_CASE(9)
      PM_NC_CASE(11)
      default: break;
    }
  }
#undef PM_NC_CASE
  hipLaunchKernelGGL(k_noise_cost, pixel_grid(ps.cols, ps.rows, slots), dim3(256), 0, h->stream, ps, cp, in, amount);
}

// RemoveBackground / MaskBackground; PM_SEM_CPU square windows use the LDS-tiled kernel
void launch_background(pm_handle* h, const PlaneSet& ps, const CostParams& cp, const Interior& in, float factor,
                       int cached, int slots) {
  const bool tiled = cp.semantics == PM_SEM_CPU && cp.pw == cp.ph && !h->no_tiled;
  const dim3 tgrid((unsigned)((ps.cols + kTileW - 1) / kTileW), (unsigned)((ps.rows + kTileH - 1) / kTileH),
                   (unsigned)slots);
#define PM_BG_CASE(W)                                                                                        \
  case W:                                                                                                    \
    hipLaunchKernelGGL((k_background_tiled<W, W>), tgrid, dim3(256), 0, h->stream, ps, cp, in, factor, cached); \
    return;
  if (tiled) {
    switch (cp.pw) {
      PM_BG_CASE(3)
      PM_BG_CASE(5)
      PM_BG_CASE(7)
      PM_BG_CASE(9)
      PM_BG_CASE(11)
      default: break;
    }
  }
#undef PM_BG_CASE
  hipLaunchKernelGGL(k_background, pixel_grid(ps.cols, ps.rows, slots), dim3(256), 0, h->stream, ps, cp, in, factor,
                     cached);
}

// the noise step alone (pm_add_noise): an empty interior skips the clamp and the cost
void launch_noise_only(pm_handle* h, const PlaneSet& ps, const CostParams& cp, float amount) {
  const Interior none{1, 0, 1, 0};
  hipLaunchKernelGGL(k_noise_cost, pixel_grid(ps.cols, ps.rows, 1), dim3(256), 0, h->stream, ps, cp, none, amount);
}

// cross-check (when two views ran) + un-mirroring + tight [n][rows][cols] output
void launch_finalize(pm_handle* h, const PlaneSet& ps, float* d_disp_l, float* d_disp_r, int n) {
  hipLaunchKernelGGL(k_finalize, pixel_grid(ps.cols, ps.rows, n), dim3(256), 0, h->stream, ps, d_disp_l, d_disp_r,
                     (size_t)ps.cols);
}

void launch_mask_occlusions(pm_handle* h, float* d_disp_l, const float* d_disp_r, int rows, int cols) {
  hipLaunchKernelGGL(k_mask_occlusions, pixel_grid(cols, rows, 1), dim3(256), 0, h->stream, d_disp_l, d_disp_r, rows,
                     cols);
}

void launch_state_row(pm_handle* h, const PlaneSet& ps, int r, float* d_buf, int to_buf) {
  hipLaunchKernelGGL(k_state_row, dim3((unsigned)((ps.cols + 255) / 256), (unsigned)ps.n_views), dim3(256), 0, h->stream, ps, r,
                     d_buf, to_buf);
}

void launch_tile_round(pm_handle* h, const PlaneSet& ps, const float* snap_disp, const float* snap_cost,
                       const float* d_incoming, const float* d_used, float* d_used_next, int* d_mask, int pred_r, int y_lo,
                       int y_hi) {
  const int chunks = (y_hi - y_lo + kTileRoundRows) / kTileRoundRows;
  hipLaunchKernelGGL(k_tile_round, dim3((unsigned)((ps.cols + 255) / 256), (unsigned)chunks, (unsigned)ps.n_views), dim3(256),
                     0, h->stream, ps, snap_disp, snap_cost, d_incoming, d_used, d_used_next, d_mask, pred_r, y_lo, y_hi);
}

void launch_tile_presweep(pm_handle* h, const PlaneSet& ps, float* snap_disp, float* snap_cost, const float* d_row, int pred_r) {
  const int rows4 = (ps.rows + 3) / 4;  // 16-byte pieces per column (pm_device.hpp::state_at)
  hipLaunchKernelGGL(k_tile_presweep, dim3((unsigned)((ps.pitch + 255) / 256), (unsigned)rows4, (unsigned)ps.n_views), dim3(256),
                     0, h->stream, ps, snap_disp, snap_cost, d_row, pred_r);
}

void launch_state_row_moved(pm_handle* h, const PlaneSet& ps, int r, const float* d_ref, int* d_flag) {
  hipLaunchKernelGGL(k_state_row_moved, dim3((unsigned)((ps.cols + 255) / 256), (unsigned)ps.n_views), dim3(256), 0,
                     h->stream, ps, r, d_ref, d_flag);
}

void launch_restore_cols(pm_handle* h, const PlaneSet& ps, const float* snap_disp, const float* snap_cost,
                         const int* d_mask) {
  hipLaunchKernelGGL(k_restore_cols, pixel_grid(ps.cols, ps.rows, ps.n_views), dim3(256), 0, h->stream, ps, snap_disp,
                     snap_cost, d_mask);
}

void launch_download(pm_handle* h, float* dst_dev, size_t dst_step_floats, const float* d_src, int rows, int cols,
                     hipStream_t stream) {
  hipLaunchKernelGGL(k_download, dim3(kDownloadBlocks), dim3(256), 0, stream, dst_dev, dst_step_floats, d_src, rows, cols);
}

// the same copy kernel the other way: `words` floats from page-locked host memory (device address) into device memory, one
// 16-byte load per lane where the grid allows (every load is a round trip over the bus)
void launch_upload(pm_handle* h, float* d_dst, const float* src_dev, int words, hipStream_t stream) {
  const int blocks = std::min(64, std::max(1, (words / 4 + 255) / 256));
  hipLaunchKernelGGL(k_download, dim3((unsigned)blocks), dim3(256), 0, stream, d_dst, (size_t)words, src_dev, 1, words);
}

void launch_copy_in(pm_handle* h, const PlaneSet& ps, const float* d_src) {
  hipLaunchKernelGGL(k_copy_in, pixel_grid(ps.cols, ps.rows, 1), dim3(256), 0, h->stream, ps, d_src);
}

void launch_copy_out(pm_handle* h, const PlaneSet& ps, float* d_dst, int which) {
  hipLaunchKernelGGL(k_copy_out, pixel_grid(ps.cols, ps.rows, 1), dim3(256), 0, h->stream, ps, d_dst, which);
}

void launch_copy_disp_strided(pm_handle* h, const PlaneSet& ps, float* d_buf, size_t stride, int to_buf) {
  hipLaunchKernelGGL(k_copy_disp_strided, pixel_grid(ps.cols, ps.rows, 1), dim3(256), 0, h->stream, ps, d_buf, stride,
                     to_buf);
}

}  // namespace eng
}  // namespace pm

// ---- ForegroundTextureMask (src/vehicle/stereo_matching/patchmatch.cpp:19-49), device images ------------------------------
extern "C" int pm_foreground_texture_mask(pm_handle* h, const uint8_t* d_gray, int rows, int cols, int ksize,
                                          double min_grad, int downsize, uint8_t* d_mask) {
  using namespace pm;
  using namespace pm::eng;
  if (!h) return PM_ERR_INVALID_ARG;
  if (int rc = refuse_while_capturing(h, "pm_foreground_texture_mask")) return rc;
  if (!d_gray || !d_mask) {
    set_err(h, "pm_foreground_texture_mask: null pointer");
    return PM_ERR_INVALID_ARG;
  }
  if (downsize < 1 || downsize > 8 || ksize / downsize <= 1) {  // the reference CHECKs both (patchmatch.cpp:25-27)
    set_err(h, "pm_foreground_texture_mask: downsize must be within [1, 8] and ksize / downsize > 1");
    return PM_ERR_INVALID_ARG;
  }
  if (int rc = check_size(h, rows, cols, 1)) return rc;
  const int srows = rows / downsize, scols = cols / downsize;
  if (srows < 1 || scols < 1) {
    set_err(h, "pm_foreground_texture_mask: image smaller than downsize");
    return PM_ERR_INVALID_ARG;
  }
  PM_HIP(h, hipSetDevice(h->device));
  const size_t plane = (size_t)h->max_rows * h->max_cols;
  if (!h->texmask_scratch) PM_HIP(h, hipMalloc(&h->texmask_scratch, 4 * plane));
  uint8_t* small = (uint8_t*)h->texmask_scratch;
  uint8_t* lo = small + plane;
  uint8_t* hi = lo + plane;
  uint8_t* bin = hi + plane;
  const int k = ksize / downsize;
  const dim3 block(256);
  const dim3 sgrid((unsigned)((scols + 255) / 256), (unsigned)srows), fgrid((unsigned)((cols + 255) / 256), (unsigned)rows);
  if (downsize > 1) {
    hipLaunchKernelGGL(k_resize_linear_u8, sgrid, block, 0, h->stream, d_gray, rows, cols, small, srows, scols);
    hipLaunchKernelGGL(k_morph_rows, sgrid, block, 0, h->stream, (const uint8_t*)small, srows, scols, k, lo, hi);
    hipLaunchKernelGGL(k_morph_cols_threshold, sgrid, block, 0, h->stream, (const uint8_t*)lo, (const uint8_t*)hi, srows,
                       scols, k, min_grad, bin);
    hipLaunchKernelGGL(k_resize_linear_u8, fgrid, block, 0, h->stream, (const uint8_t*)bin, srows, scols, d_mask, rows, cols);
  } else {
    hipLaunchKernelGGL(k_morph_rows, fgrid, block, 0, h->stream, d_gray, rows, cols, k, lo, hi);
    hipLaunchKernelGGL(k_morph_cols_threshold, fgrid, block, 0, h->stream, (const uint8_t*)lo, (const uint8_t*)hi, rows, cols,
                       k, min_grad, d_mask);
  }
  return launch_check(h, "foreground texture mask");
}
