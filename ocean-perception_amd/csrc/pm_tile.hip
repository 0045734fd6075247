// pm_tile.hip -- the row-tiled phase API pm_tile_* (include/pm/patchmatch.h): one handle holds a band of rows of a
// larger image and runs the phases of a Match one by one, so a driver (pm_tiled.hip in one process, or
// python/tiled.py over RCCL) can exchange boundary rows between the phases.  Host logic only: geometry of the band
// inside the whole image, snapshots, row copies; kernels through the launch functions of pm_handle.hpp.
#include "pm_handle.hpp"

using namespace pm;
using namespace pm::eng;

namespace {

// PlaneSet of the band with the noise pointer moved to the band's slice of the whole-image table.
PlaneSet tile_plane_set(pm_handle* h) {
  const int n_views = h->params.left_right_check ? 2 : 1;
  PlaneSet ps = plane_set(h, h->tile_band_rows, h->tile_cols, n_views);
  ps.noise = h->noise + (size_t)h->tile.band_row0 * ps.pitch;
  return ps;
}

// Rows this tile sweeps: owned rows that the sweeps of the whole image visit, in band coordinates.
Interior tile_interior(pm_handle* h, int pw, int ph) {
  const pm_tile& t = h->tile;
  Interior in = interior(h->params, t.global_rows, h->tile_cols, pw, ph);  // whole-image rows / columns
  const int lo = in.y_lo > t.own_row0 ? in.y_lo : t.own_row0;
  const int hi = in.y_hi < t.own_row0 + t.own_rows - 1 ? in.y_hi : t.own_row0 + t.own_rows - 1;
  in.y_lo = lo - t.band_row0;
  in.y_hi = hi - t.band_row0;
  return in;
}

int tile_check(pm_handle* h, const char* what) {
  if (!h) return PM_ERR_INVALID_ARG;
  if (!h->tile_on) {
    set_err(h, "%s: call pm_tile_begin first", what);
    return PM_ERR_INVALID_ARG;
  }
  PM_HIP(h, hipSetDevice(h->device));
  return PM_OK;
}

}  // namespace

extern "C" {

int pm_tile_begin(pm_handle* h, const pm_tile* tile, const uint8_t* d_left_band, const uint8_t* d_right_band,
                  int band_rows, int cols, const float* d_seed_l_band, const float* d_seed_r_band) {
  if (!h) return PM_ERR_INVALID_ARG;
  if (!tile || !d_left_band || !d_right_band) {
    set_err(h, "pm_tile_begin: null pointer");
    return PM_ERR_INVALID_ARG;
  }
  if (int rc = check_size(h, band_rows, cols, 1)) return rc;
  const pm_params& p = h->params;
  int max_ph = p.bg_patch_h;
  for (int i = 0; i < p.patchmatch_iters; ++i) max_ph = p.patch_h[i] > max_ph ? p.patch_h[i] : max_ph;
  const int halo = (p.semantics == PM_SEM_CPU ? max_ph / 2 : 1) + 1;  // window rows + one more for the Sobel
  const int need_top = tile->own_row0 - halo > 0 ? tile->own_row0 - halo : 0;
  const int own_end = tile->own_row0 + tile->own_rows;
  const int need_end = own_end + halo < tile->global_rows ? own_end + halo : tile->global_rows;
  if (tile->own_rows < 1 || tile->own_row0 < 0 || own_end > tile->global_rows || tile->band_row0 < 0 ||
      tile->band_row0 > need_top || tile->band_row0 + band_rows < need_end ||
      tile->band_row0 + band_rows > tile->global_rows) {
    set_err(h, "pm_tile_begin: band [%d, %d) must cover rows [%d, %d) (owned [%d, %d) + %d halo rows) of %d",
            tile->band_row0, tile->band_row0 + band_rows, need_top, need_end, tile->own_row0, own_end, halo,
            tile->global_rows);
    return PM_ERR_INVALID_ARG;
  }
  PM_HIP(h, hipSetDevice(h->device));
  if (int rc = ensure_noise(h, tile->global_rows, cols)) return rc;
  h->tile = *tile;
  h->tile_band_rows = band_rows;
  h->tile_cols = cols;
  h->tile_on = true;
  const PlaneSet ps = tile_plane_set(h);
  launch_prep(h, ps, d_left_band, d_right_band, 1, (size_t)cols);
  if (int rc = launch_check(h, "prep")) return rc;
  if (int rc = run_transpose(h, ps, 1)) return rc;
  launch_seed(h, ps, d_seed_l_band, d_seed_r_band, 1);
  return launch_check(h, "seed");
}

int pm_tile_noise(pm_handle* h, int it) {
  if (int rc = tile_check(h, "pm_tile_noise")) return rc;
  const pm_params& p = h->params;
  if (it < 0 || it >= p.patchmatch_iters) {
    set_err(h, "pm_tile_noise: iteration %d out of range", it);
    return PM_ERR_INVALID_ARG;
  }
  const PlaneSet ps = tile_plane_set(h);
  const CostParams cp = cost_params(p, p.patch_w[it], p.patch_h[it]);
  const Interior in = tile_interior(h, cp.pw, cp.ph);
  launch_noise_cost(h, ps, cp, in, p.noise_amp[it], ps.n_views, 0);
  return launch_check(h, "noise_cost");
}

static int tile_sweep(pm_handle* h, int it, int k, const int* d_mask);
int pm_tile_sweep(pm_handle* h, int it, int k) { return tile_sweep(h, it, k, nullptr); }
int pm_tile_sweep_masked(pm_handle* h, int it, int k, const int* d_mask) {
  if (!h) return PM_ERR_INVALID_ARG;
  if (!d_mask) {
    set_err(h, "pm_tile_sweep_masked: null mask");
    return PM_ERR_INVALID_ARG;
  }
  return tile_sweep(h, it, k, d_mask);
}

static int tile_sweep(pm_handle* h, int it, int k, const int* d_mask) {
  if (int rc = tile_check(h, "pm_tile_sweep")) return rc;
  const pm_params& p = h->params;
  if (it < 0 || it >= p.patchmatch_iters || k < 0 || k > 3) {
    set_err(h, "pm_tile_sweep: iteration %d / sweep %d out of range", it, k);
    return PM_ERR_INVALID_ARG;
  }
  PlaneSet ps = tile_plane_set(h);
  if (d_mask && (k & 1) == 0) {
    set_err(h, "pm_tile_sweep_masked: the column mask applies to the vertical sweeps (k = 1, 3)");
    return PM_ERR_INVALID_ARG;
  }
  ps.chain_mask = d_mask;
  const CostParams cp = cost_params(p, p.patch_w[it], p.patch_h[it]);
  // Geometry of the sweep on the WHOLE image (PM_SEM_GPU trims one position at the far end of each sweep,
  // patchmatch_gpu.cu:156,214 -- that end is an end of the image, not of a band), then cut to the owned rows.
  const pm_tile& t = h->tile;
  SweepGeom g = sweep_geom(p, interior(p, t.global_rows, h->tile_cols, cp.pw, cp.ph), k);
  const int own_lo = t.own_row0, own_hi = t.own_row0 + t.own_rows - 1;
  if (g.axis == 0) {  // chains are rows
    g.c_lo = (g.c_lo > own_lo ? g.c_lo : own_lo) - t.band_row0;
    g.c_hi = (g.c_hi < own_hi ? g.c_hi : own_hi) - t.band_row0;
    if (g.c_hi < g.c_lo) return PM_OK;
  } else {  // positions along a chain are rows
    if (g.dir > 0) {
      g.s_first = g.s_first > own_lo ? g.s_first : own_lo;
      g.s_last = g.s_last < own_hi ? g.s_last : own_hi;
      if (g.s_last < g.s_first) return PM_OK;
    } else {
      g.s_first = g.s_first < own_hi ? g.s_first : own_hi;
      g.s_last = g.s_last > own_lo ? g.s_last : own_lo;
      if (g.s_last > g.s_first) return PM_OK;
    }
    g.s_first -= t.band_row0;
    g.s_last -= t.band_row0;
  }
  return run_sweep(h, ps, cp, g, ps.n_views, p.noise_amp[it]);
}

int pm_tile_snapshot(pm_handle* h) {
  if (int rc = tile_check(h, "pm_tile_snapshot")) return rc;
  const size_t plane = (size_t)align_up(h->max_rows, 4) * h->max_pitch;  // state planes (pm_device.hpp::state_at)
  if (!h->snap_disp) {
    PM_HIP(h, hipMalloc((void**)&h->snap_disp, sizeof(float) * 2 * plane));
    PM_HIP(h, hipMalloc((void**)&h->snap_cost, sizeof(float) * 2 * plane));
  }
  const PlaneSet ps = tile_plane_set(h);
  const size_t bytes = sizeof(float) * ps.splane * ps.n_views;
  PM_HIP(h, hipMemcpyAsync(h->snap_disp, h->disp, bytes, hipMemcpyDeviceToDevice, h->stream));
  PM_HIP(h, hipMemcpyAsync(h->snap_cost, h->cost, bytes, hipMemcpyDeviceToDevice, h->stream));
  return PM_OK;
}

// pm_tile_set_row (optional) + pm_tile_snapshot in ONE launch: what stands in front of every vertical sweep of a band.
int pm_tile_presweep(pm_handle* h, int pred_image_row, const float* d_row) {
  if (int rc = tile_check(h, "pm_tile_presweep")) return rc;
  const int r = pred_image_row - h->tile.band_row0;
  if (d_row && (r < 0 || r >= h->tile_band_rows)) {
    set_err(h, "pm_tile_presweep: row %d outside the band", pred_image_row);
    return PM_ERR_INVALID_ARG;
  }
  const size_t plane = (size_t)align_up(h->max_rows, 4) * h->max_pitch;  // state planes (pm_device.hpp::state_at)
  if (!h->snap_disp) {
    PM_HIP(h, hipMalloc((void**)&h->snap_disp, sizeof(float) * 2 * plane));
    PM_HIP(h, hipMalloc((void**)&h->snap_cost, sizeof(float) * 2 * plane));
  }
  const PlaneSet ps = tile_plane_set(h);
  launch_tile_presweep(h, ps, h->snap_disp, h->snap_cost, d_row, d_row ? r : -4);
  return launch_check(h, "tile_presweep");
}

int pm_tile_restore(pm_handle* h) {
  if (int rc = tile_check(h, "pm_tile_restore")) return rc;
  if (!h->snap_disp) {
    set_err(h, "pm_tile_restore: no snapshot");
    return PM_ERR_INVALID_ARG;
  }
  const PlaneSet ps = tile_plane_set(h);
  const size_t bytes = sizeof(float) * ps.splane * ps.n_views;
  PM_HIP(h, hipMemcpyAsync(h->disp, h->snap_disp, bytes, hipMemcpyDeviceToDevice, h->stream));
  PM_HIP(h, hipMemcpyAsync(h->cost, h->snap_cost, bytes, hipMemcpyDeviceToDevice, h->stream));
  return PM_OK;
}

int pm_tile_restore_cols(pm_handle* h, const int* d_mask) {
  if (int rc = tile_check(h, "pm_tile_restore_cols")) return rc;
  if (!h->snap_disp || !d_mask) {
    set_err(h, "pm_tile_restore_cols: no snapshot or null mask");
    return PM_ERR_INVALID_ARG;
  }
  const PlaneSet ps = tile_plane_set(h);
  launch_restore_cols(h, ps, h->snap_disp, h->snap_cost, d_mask);
  return launch_check(h, "restore_cols");
}

// One exchange round of vertical sweep k in two launches: k_tile_round (compare the incoming boundary row with the one
// the last sweep used -> mask; flagged columns back to the snapshot; the incoming values into the planes and -- a copy
// for the next round's comparison -- into d_used_next), then the masked sweep.  Equal to
// pm_tile_restore_cols(mask of incoming != used) + pm_tile_set_row + pm_tile_sweep_masked.
int pm_tile_exchange_round(pm_handle* h, int it, int k, int pred_image_row, const float* d_incoming, const float* d_used,
                           float* d_used_next, int* d_mask) {
  if (int rc = tile_check(h, "pm_tile_exchange_round")) return rc;
  const int r = pred_image_row - h->tile.band_row0;
  if (!h->snap_disp || !d_incoming || !d_used || !d_used_next || d_used_next == d_used || !d_mask || r < 0 ||
      r >= h->tile_band_rows) {
    set_err(h, "pm_tile_exchange_round: no snapshot, a null pointer, or row %d outside the band", pred_image_row);
    return PM_ERR_INVALID_ARG;
  }
  if ((k & 1) == 0) {
    set_err(h, "pm_tile_exchange_round: boundary rows belong to the vertical sweeps (k = 1, 3)");
    return PM_ERR_INVALID_ARG;
  }
  const PlaneSet ps = tile_plane_set(h);
  // the rows a sweep of this band can have written: the owned ones
  const int y_lo = h->tile.own_row0 - h->tile.band_row0, y_hi = y_lo + h->tile.own_rows - 1;
  if (r >= y_lo && r <= y_hi) {
    set_err(h, "pm_tile_exchange_round: row %d is one of the band's own rows, not a neighbour's", pred_image_row);
    return PM_ERR_INVALID_ARG;
  }
  launch_tile_round(h, ps, h->snap_disp, h->snap_cost, d_incoming, d_used, d_used_next, d_mask, r, y_lo, y_hi);
  if (int rc = launch_check(h, "tile_round")) return rc;
  return tile_sweep(h, it, k, d_mask);
}

int pm_tile_row_moved(pm_handle* h, int image_row, const float* d_ref_row, int* d_flag) {
  if (int rc = tile_check(h, "pm_tile_row_moved")) return rc;
  const int r = image_row - h->tile.band_row0;
  if (r < 0 || r >= h->tile_band_rows || !d_ref_row || !d_flag) {
    set_err(h, "pm_tile_row_moved: row %d outside the band or null pointer", image_row);
    return PM_ERR_INVALID_ARG;
  }
  const PlaneSet ps = tile_plane_set(h);
  launch_state_row_moved(h, ps, r, d_ref_row, d_flag);
  return launch_check(h, "state_row_moved");
}

static int tile_row_copy(pm_handle* h, int image_row, float* d_dst, const float* d_src, const char* what) {
  if (int rc = tile_check(h, what)) return rc;
  const int r = image_row - h->tile.band_row0;
  if (r < 0 || r >= h->tile_band_rows || (!d_dst && !d_src)) {
    set_err(h, "%s: row %d outside the band or null pointer", what, image_row);
    return PM_ERR_INVALID_ARG;
  }
  const PlaneSet ps = tile_plane_set(h);
  // (the state planes keep four rows interleaved: a row is every fourth word of a stretch -- a small kernel, not a copy)
  launch_state_row(h, ps, r, d_dst ? d_dst : const_cast<float*>(d_src), d_dst ? 1 : 0);
  return launch_check(h, what);
}

int pm_tile_get_row(pm_handle* h, int image_row, float* d_dst) {
  return tile_row_copy(h, image_row, d_dst, nullptr, "pm_tile_get_row");
}
int pm_tile_set_row(pm_handle* h, int image_row, const float* d_src) {
  return tile_row_copy(h, image_row, nullptr, d_src, "pm_tile_set_row");
}

int pm_tile_background(pm_handle* h) {
  if (int rc = tile_check(h, "pm_tile_background")) return rc;
  const pm_params& p = h->params;
  const PlaneSet ps = tile_plane_set(h);
  const CostParams bcp = cost_params(p, p.bg_patch_w, p.bg_patch_h);
  const Interior in = tile_interior(h, bcp.pw, bcp.ph);
  const int last = p.patchmatch_iters - 1;
  const int cached = (last >= 0 && bcp.pw == (p.semantics == PM_SEM_CPU ? p.patch_w[last] : 3) &&
                      bcp.ph == (p.semantics == PM_SEM_CPU ? p.patch_h[last] : 3)) ? 1 : 0;
  const float factor = p.semantics == PM_SEM_CPU ? p.win_by_factor : p.cost_improve_factor;
  launch_background(h, ps, bcp, in, factor, cached, ps.n_views);
  return launch_check(h, "background");
}

int pm_tile_finish(pm_handle* h, float* d_disp_l_own, float* d_disp_r_own) {
  if (int rc = tile_check(h, "pm_tile_finish")) return rc;
  const PlaneSet ps = tile_plane_set(h);
  if (!d_disp_l_own || (ps.n_views > 1 && !d_disp_r_own)) {
    set_err(h, "pm_tile_finish: null output");
    return PM_ERR_INVALID_ARG;
  }
  launch_finalize(h, ps, h->st_disp_l, ps.n_views > 1 ? h->st_disp_r : nullptr, 1);
  if (int rc = launch_check(h, "finalize")) return rc;
  const size_t ofs = (size_t)(h->tile.own_row0 - h->tile.band_row0) * ps.cols;
  const size_t bytes = sizeof(float) * (size_t)h->tile.own_rows * ps.cols;
  PM_HIP(h, hipMemcpyAsync(d_disp_l_own, h->st_disp_l + ofs, bytes, hipMemcpyDeviceToDevice, h->stream));
  if (ps.n_views > 1)
    PM_HIP(h, hipMemcpyAsync(d_disp_r_own, h->st_disp_r + ofs, bytes, hipMemcpyDeviceToDevice, h->stream));
  h->tile_on = false;
  return PM_OK;
}

}  // extern "C"
