// pm_sweeps.hip -- directional sweeps: engine selection, tuning knobs and every sweep kernel instantiation
// (pm_sweeps.hpp).  PatchmatchGpu's PropagateRow / PropagateCol (patchmatch_gpu.cu:116-230) and Patchmatch::Propagate's
// four passes (patchmatch.cpp:248-311).
#include "pm_sweeps.hpp"

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pm/patchmatch.h"
#include "pm_serial.hpp"
#include "pm_wave.hpp"
#include "pm_run2.hpp"
#include "pm_run3.hpp"
#include "pm_tune.hpp"

namespace pm {
namespace {

// Wavefronts per chain in PM_ENGINE_RUNBLK2: 4 up to ~1600 positions per chain, 8 beyond (measured:
// 720p best at 4, tools/sweep_group.sh; 4096x2160 38.7 ms per frame at 8 vs 42.6 at 4), 2 for chains shorter than 400
// positions -- the column chains of a 270-row band of a row-tiled 4096x2160 image: 8 / 16 segments of 17-35 positions are
// mostly speculation boundaries (round 5: eight bands on one device 51.0 -> 48.4 ms in the tuning build) -- but only
// where the launch has chains enough to fill the chip without them: the 376 x 240 pair of the reference's own test is
// short chains on an EMPTY chip, and two wavefronts per chain took its call from 0.65 to 0.79 ms.
// PM_RUNBLK_WAVES overrides.
constexpr int kShortChain = 400, kManyChains = 2048;
int runblk_waves(int chain_len, int n_chains, int axis, int group = 32) {
  struct Knobs {
    int v[2][2];  // [axis][group == 16]
    Knobs() {
      const char* names[2][2] = {{"PM_RUNBLK_WAVES_ROW", "PM_RUNBLK_WAVES_ROW16"},
                                 {"PM_RUNBLK_WAVES_COL", "PM_RUNBLK_WAVES_COL16"}};
      const char* both = pm::tune_env("PM_RUNBLK_WAVES");
      for (int a = 0; a < 2; ++a)
        for (int g = 0; g < 2; ++g) {
          const char* e = pm::tune_env(names[a][g]);
          if (!e && g == 1) e = pm::tune_env(names[a][0]);
          if (!e) e = both;
          const int x = e ? atoi(e) : 0;
          v[a][g] = x < 1 ? 0 : (x > kMaxSegWaves ? kMaxSegWaves : x);
        }
    }
  };
  static const Knobs k;  // initialised once, thread-safe
  const int g16 = group <= 16 ? 1 : 0;
  if (k.v[axis][g16]) return k.v[axis][g16];
  return chain_len > 1600 ? 8 : ((chain_len < kShortChain && n_chains >= kManyChains) ? 2 : 4);
}

// One directional sweep, in place.
// lanes per chain segment of PM_ENGINE_RUNBLK2 (32 or 16); PM_RUNBLK_GROUP overrides.  Measured
// (tools/sweep_group.sh, 720p): PM_SEM_GPU's 3-lane window wins with 16-lane groups (1.60 vs 1.93 ms per
// frame), PM_SEM_CPU's 11-lane window with 32 (a 16-lane strip leaves it only 5-6 positions per step).
// A tuning choice only (results do not depend on it).  Runs of adopted values get shorter as the noise
// amplitude decays, and short runs waste most of a 32-lane strip: measured at 720p / 11x11 / amp 32/2^i
// (tools/sweep_waves.sh) column sweeps win with 16-lane groups from amplitude 4 on, row sweeps (one
// position fewer per strip: the DPP spare lane) only from 0.5 on.  That holds for the FORWARD sweeps, which come
// first after the noise and carry a good value a long way; the BACKWARD sweeps of the same iteration meet what the
// forward ones left -- short runs, 58 % more steps per launch (profiles/r02d_pmc_insts.txt) -- and win with 16-lane
// groups from amplitude 8 (rows) / 16 (columns) on: 294 -> 307 pairs/s (tools/sweep_neg.sh,
// profiles/r02f_sweep_group_thresholds.txt).
int runblk_group(int semantics, int axis, float amp, int win, int dir = 1) {
  static int v = [] {
    const char* e = pm::tune_env("PM_RUNBLK_GROUP");
    const int g = e ? atoi(e) : 0;
    return (g == 8 || g == 16 || g == 32) ? g : 0;
  }();
  if (v) return v;
  if (semantics != PM_SEM_CPU) {
    static const int gpu_g[2] = {[] { const char* e = pm::tune_env("PM_GPU_GROUP_FWD"); return e ? atoi(e) : 16; }(),
                                 [] { const char* e = pm::tune_env("PM_GPU_GROUP_BWD"); return e ? atoi(e) : 16; }()};
    return gpu_g[dir < 0 ? 1 : 0];
  }
  if (win <= 5) return 16;  // small windows leave 11+ positions in a 16-lane strip: 16 wins at every amplitude
  struct Thr {
    float t[2], tn[2];  // forward sweeps, backward sweeps (PM_G16_*_AMP_NEG)
    Thr() {
      const char* er = pm::tune_env("PM_G16_ROW_AMP");
      const char* ec = pm::tune_env("PM_G16_COL_AMP");
      const char* ern = pm::tune_env("PM_G16_ROW_AMP_NEG");
      const char* ecn = pm::tune_env("PM_G16_COL_AMP_NEG");
      t[0] = er ? (float)atof(er) : 0.5f;
      t[1] = ec ? (float)atof(ec) : 4.0f;
      tn[0] = ern ? (float)atof(ern) : 8.0f;
      tn[1] = ecn ? (float)atof(ecn) : 16.0f;
    }
  };
  static const Thr thr_knobs;
  const float* thr = dir < 0 ? thr_knobs.tn : thr_knobs.t;
  return amp <= thr[axis] ? 16 : 32;
}

}  // namespace

void launch_sweep(const PlaneSet& ps, const CostParams& cp, const SweepGeom& g, int slots, int engine, float amp,
                  hipStream_t stream) {
  const int chains = g.c_hi - g.c_lo + 1;
  const int chain_len = (g.s_last - g.s_first) * g.dir + 1;
  if (engine == PM_ENGINE_AUTO) engine = PM_ENGINE_RUNBLK2;
  // the chain engines hold a chain in LDS: beyond the CU's capacity only the serial engine remains
  if (engine != PM_ENGINE_SERIAL && !(engine == PM_ENGINE_WAVE && cp.semantics == PM_SEM_CPU) &&
      chain_lds_bytes(chain_len, 4 * kMaxSegWaves + 4, cp.semantics == PM_SEM_CPU ? 4 : 5) > kChainLdsMax)
    engine = PM_ENGINE_SERIAL;
  // PM_SEM_GPU has two parallel engines: lane-per-segment (WAVE) and the shared-tap run step (RUNBLK2)
  if (engine == PM_ENGINE_SERIAL) {
    hipLaunchKernelGGL(k_sweep_serial, dim3((unsigned)((chains + 63) / 64), 1, (unsigned)slots), dim3(64), 0, stream, ps,
                       cp, g);
  } else if (engine == PM_ENGINE_WAVE) {
    launch_sweep_wave(ps, cp, g, slots, stream);
  } else {
    const int group = runblk_group(cp.semantics, g.axis, amp, g.axis == 0 ? cp.pw : cp.ph, g.dir);
    if (cp.semantics == PM_SEM_CPU)
      launch_sweep_run3(ps, cp, g, slots, runblk_waves(chain_len, chains * slots, g.axis, group), group, stream);
    else {
      static const int gpu_w[2] = {[] { const char* e = pm::tune_env("PM_GPU_WAVES_FWD"); return e ? atoi(e) : 0; }(),
                                   [] { const char* e = pm::tune_env("PM_GPU_WAVES_BWD"); return e ? atoi(e) : 0; }()};
      const int wv = gpu_w[g.dir < 0 ? 1 : 0];
      launch_sweep_run2(ps, cp, g, slots, wv ? wv : runblk_waves(chain_len, chains * slots, g.axis, group), group, stream);
    }
  }
}

}  // namespace pm

#ifdef PM_RUN3_STATS
// stats build only: start / stop the per-chain log, write it out.  File: "RUN3LOG1", launches, then per launch the host
// record (8 x int64: stream, axis, dir, gs, n, chains, waves, 0) and chains x 8 uint32 device words.
extern "C" __attribute__((visibility("default"))) int pm_run3_stats_enable(int on) {
  pm::run3_stats().on = on != 0;
  if (on) pm::run3_stats().recs.clear();
  return 0;
}
extern "C" __attribute__((visibility("default"))) int pm_run3_stats_dump(const char* path) {
  pm::Run3Stats& s = pm::run3_stats();
  if (hipDeviceSynchronize() != hipSuccess) return -3;
  FILE* f = fopen(path, "wb");
  if (!f) return -1;
  const long long nl = (long long)s.recs.size();
  fwrite("RUN3LOG1", 1, 8, f);
  fwrite(&nl, sizeof(nl), 1, f);
  std::vector<unsigned> buf;
  for (size_t i = 0; i < s.recs.size(); ++i) {
    const pm::Run3StatsRec& r = s.recs[i];
    const long long rec[8] = {(long long)r.stream, r.axis, r.dir, r.gs, r.n, r.chains, r.waves, 0};
    fwrite(rec, sizeof(rec), 1, f);
    buf.resize(8 * (size_t)r.chains);
    if (hipMemcpy(buf.data(), s.d_log + 8 * (size_t)pm::Run3Stats::kMaxChains * i, sizeof(unsigned) * buf.size(),
                  hipMemcpyDeviceToHost) != hipSuccess) {
      fclose(f);
      return -3;
    }
    fwrite(buf.data(), sizeof(unsigned), buf.size(), f);
  }
  fclose(f);
  return (int)nl;
}
#endif
