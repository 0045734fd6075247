// BASELINE configs[3] from C++: one large pair matched by n band handles of one process through
// bm::pm::TiledPatchmatchGpu (pm_tiled_* of include/pm/patchmatch.h), and the same pair by the untiled
// bm::pm::PatchmatchGpu::Match -- the construct-and-Match() calls of patchmatch_gpu.h:94-102.  Writes both results for
// tests/test_cpp_tiled.py to compare bit for bit.
// usage: tiled_main <dir> <rows> <cols> <semantics> <patch> <iters> <bands> <rounds>     (rounds = -2: the default schedule, PM_TILED_SCHEDULE_PIPELINED; else SPECULATIVE with that many rounds)
//        tiled_main devices <d0,d1,...>   only construct the row-tiled matcher with band k on device dk (64x96 image)
//                                          and print its topology, or the exception: the multi-device constructor path
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "patchmatch_gpu.hpp"

using namespace bm::pm;
using bm::core::Image1b;
using bm::core::Image1f;

template <typename T>
static bool read_raw(const std::string& path, bm::core::Image<T>& im) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  f.read(reinterpret_cast<char*>(im.data()), sizeof(T) * (size_t)im.rows * im.cols);
  return (bool)f;
}
template <typename T>
static void write_raw(const std::string& path, const bm::core::Image<T>& im) {
  std::ofstream f(path, std::ios::binary);
  f.write(reinterpret_cast<const char*>(im.data()), sizeof(T) * (size_t)im.rows * im.cols);
}

int main(int argc, char** argv) {
  if (argc == 3 && std::string(argv[1]) == "devices") {
    std::vector<int> devices;
    for (const char* q = argv[2]; *q;) {
      devices.push_back(atoi(q));
      while (*q && *q != ',') ++q;
      if (*q == ',') ++q;
    }
    PatchmatchGpu::Params params;
    params.semantics = PM_SEM_CPU;
    try {
      TiledPatchmatchGpu tiled(params, 64, 96, devices);
      const auto t = tiled.Topology();
      std::printf("bands %zu device_boundaries %d peer_links %d\n", devices.size(), t.first, t.second);
      return 0;
    } catch (const std::exception& e) {
      std::printf("exception: %s\n", e.what());
      return 10;
    }
  }
  if (argc < 9) return 2;
  const std::string dir = argv[1];
  const int rows = atoi(argv[2]), cols = atoi(argv[3]), bands = atoi(argv[7]), rounds = atoi(argv[8]);
  PatchmatchGpu::Params params;
  params.semantics = atoi(argv[4]);
  params.patch_size = atoi(argv[5]);
  params.patchmatch_iters = atoi(argv[6]);
  params.max_rows = rows;
  params.max_cols = cols;
  try {
    Image1b il(rows, cols), ir(rows, cols);
    Image1f sl(rows, cols), sr(rows, cols), disp, dispr, tdisp, tdispr;
    if (!read_raw(dir + "/left.u8", il) || !read_raw(dir + "/right.u8", ir) || !read_raw(dir + "/seed_l.f32", sl) ||
        !read_raw(dir + "/seed_r.f32", sr)) {
      std::cerr << "cannot read inputs\n";
      return 3;
    }
    {
      PatchmatchGpu pm(params);
      pm.SetSeeds(sl, sr);
      pm.Match(il, ir, disp, dispr);
    }
    {
      TiledPatchmatchGpu tiled(params, rows, cols, std::vector<int>((size_t)bands, 0));  // every band on device 0
      tiled.SetSeeds(sl, sr);
      // rounds = -2: the default schedule (the bands sweep in order: no rounds at all); else the speculative one
      if (rounds != -2) tiled.SetSchedule(PM_TILED_SCHEDULE_SPECULATIVE);
      tiled.Match(il, ir, tdisp, tdispr, rounds);
      const pm_tiled_info& info = tiled.LastInfo();
      std::printf("rounds_used %d repeated %d exchanges %d\n", info.rounds_used, info.repeated, info.exchanges);
    }
    write_raw(dir + "/disp_l.f32", disp);
    write_raw(dir + "/disp_r.f32", dispr);
    write_raw(dir + "/tiled_l.f32", tdisp);
    write_raw(dir + "/tiled_r.f32", tdispr);
  } catch (const std::exception& e) {
    std::printf("exception: %s\n", e.what());
    return 10;
  }
  return 0;
}
