// Exercises the C++ mirror of bm::pm::PatchmatchGpu the way the reference's own test does
// (test/stereo_matching/patchmatch_gpu_test.cpp:68-88): build Params, construct, call Match() five
// times.  Reads raw inputs written by tests/test_cpp_wrapper.py, writes raw outputs for it to check
// against the oracle.  usage: wrapper_main <dir> <rows> <cols> <semantics> <patch> <iters>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>

#include <type_traits>
#include <vector>

#include "patchmatch_gpu.hpp"
#include "pm/imaging.h"

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
  if (argc < 7) return 2;
  const std::string dir = argv[1];
  const int rows = atoi(argv[2]), cols = atoi(argv[3]);
  PatchmatchGpu::Params params;
  params.matcher_params.templ_cols = 31;  // the reference test sets these; they configure the seeder
  params.matcher_params.templ_rows = 11;
  params.matcher_params.max_disp = 128;
  params.cost_alpha = 0.9f;
  params.semantics = atoi(argv[4]);
  params.patch_size = atoi(argv[5]);
  params.patchmatch_iters = atoi(argv[6]);
  params.max_rows = rows;  // plan in the constructor (the reference allocates lazily in the first Match)
  params.max_cols = cols;
  // Seeder parameters outside their ranges are REFUSED, not replaced: the constructor throws what pm_create said.  The
  // parameter check comes before the device check, so this holds on a box without a GPU too.
  if (argc > 7 && std::string(argv[7]) == "refused") {
    int refused = 0;
    for (int which = 0; which < 2; ++which) {
      PatchmatchGpu::Params bad = params;
      if (which == 0) {
        bad.detector_params.subpixel_corners = true;
        bad.detector_params.subpix_winsize = 0;
      } else {
        bad.detector_params.gftt_k = -1.0;
      }
      try {
        PatchmatchGpu pm(bad);
      } catch (const std::runtime_error& e) {
        const std::string w = e.what();
        if (w.find(which == 0 ? "cornerSubPix" : "gftt") != std::string::npos) ++refused;
      }
    }
    std::cout << "refused " << refused << "\n";
    return refused == 2 ? 0 : 14;
  }
  try {
    PatchmatchGpu pm(params);
    Image1b il(rows, cols), ir(rows, cols);
    Image1f sl(rows, cols), sr(rows, cols), disp, dispr;
    if (!read_raw(dir + "/left.u8", il) || !read_raw(dir + "/right.u8", ir) || !read_raw(dir + "/seed_l.f32", sl) ||
        !read_raw(dir + "/seed_r.f32", sr)) {
      std::cerr << "cannot read inputs\n";
      return 3;
    }
    pm.SetSeeds(sl, sr);
    for (int i = 0; i < 5; ++i) pm.Match(il, ir, disp, dispr);
    write_raw(dir + "/disp_l.f32", disp);
    write_raw(dir + "/disp_r.f32", dispr);
    // Match() without SetSeeds seeds itself, and SparseInit is callable on its own (patchmatch_gpu.h:110-112)
    Image1f none;
    pm.SetSeeds(none, none);
    Image1f auto_l, auto_r;
    pm.Match(il, ir, auto_l, auto_r);
    write_raw(dir + "/auto_l.f32", auto_l);
    write_raw(dir + "/auto_r.f32", auto_r);
    write_raw(dir + "/sparse_init.f32", pm.SparseInit(il, ir, 4));
    // frame loop with the copies off the critical path: Submit()/Collect() return Match()'s maps in order
    {
      PatchmatchGpu::Params p3 = params;
      p3.max_batch = 3;
      PatchmatchGpu seq(p3);
      seq.SetSeeds(sl, sr);
      int collected = 0;
      Image1f ql, qr;
      uint64_t tag = 0;
      for (int i = 0; i < 7; ++i) {
        while (!seq.Submit(il, ir, 1000 + i)) {
          if (!seq.Collect(ql, qr, &tag) || tag != (uint64_t)(1000 + collected)) return 4;
          ++collected;
          if (std::memcmp(ql.data(), disp.data(), sizeof(float) * (size_t)rows * cols) != 0) return 5;
        }
      }
      while (seq.Collect(ql, qr, &tag)) {
        if (tag != (uint64_t)(1000 + collected)) return 4;
        ++collected;
        if (std::memcmp(qr.data(), dispr.data(), sizeof(float) * (size_t)rows * cols) != 0) return 5;
      }
      if (collected != 7) return 6;
    }
    // the same loop on page-locked images with the maps bound at submission: nothing is staged, the DMA engines read
    // the images and write the maps in place
    {
      PatchmatchGpu::Params p4 = params;
      p4.max_batch = 4;
      PatchmatchGpu seq(p4);
      seq.SetSeeds(sl, sr);
      Image1b pl = il, pr = ir;
      std::vector<Image1f> outs_l(4, Image1f(rows, cols)), outs_r(4, Image1f(rows, cols));
      seq.Register(pl);
      seq.Register(pr);
      for (int k = 0; k < 4; ++k) {
        seq.Register(outs_l[(size_t)k]);
        seq.Register(outs_r[(size_t)k]);
      }
      int submitted = 0, collected = 0;
      uint64_t tag = 0;
      const size_t bytes = sizeof(float) * (size_t)rows * cols;
      while (collected < 9) {
        while (submitted < 9 && seq.Submit(pl, pr, outs_l[(size_t)(submitted % 4)], outs_r[(size_t)(submitted % 4)], 2000 + submitted))
          ++submitted;
        if (!seq.Collect(&tag) || tag != (uint64_t)(2000 + collected)) return 15;
        if (std::memcmp(outs_l[(size_t)(collected % 4)].data(), disp.data(), bytes) != 0 ||
            std::memcmp(outs_r[(size_t)(collected % 4)].data(), dispr.data(), bytes) != 0)
          return 16;
        ++collected;
      }
      for (int k = 0; k < 4; ++k) {
        seq.Unregister(outs_l[(size_t)k]);
        seq.Unregister(outs_r[(size_t)k]);
      }
      seq.Unregister(pl);
      seq.Unregister(pr);
    }
    // the Harris response instead of the smaller eigenvalue (feature_detector.hpp:34-35)
    {
      PatchmatchGpu::Params ph = params;
      ph.detector_params.gftt_use_harris_corner_detector = true;
      ph.detector_params.gftt_k = 0.06;
      PatchmatchGpu harris(ph);
      write_raw(dir + "/sparse_init_harris.f32", harris.SparseInit(il, ir, 4));
    }
    // cv::cornerSubPix on the corners and on the matches (feature_detector.hpp:39-44, stereo_matcher.hpp:26)
    {
      PatchmatchGpu::Params ps = params;
      ps.detector_params.subpixel_corners = true;
      ps.detector_params.subpix_winsize = 6;
      ps.matcher_params.subpixel_refinement = true;
      PatchmatchGpu subpix(ps);
      write_raw(dir + "/sparse_init_subpix.f32", subpix.SparseInit(il, ir, 4));
    }
    // a batch in one call: three pairs (the pair, the pair with left and right swapped, the pair again), every map
    // equal to what Match() returns for that pair alone
    {
      PatchmatchGpu::Params p3 = params;
      p3.max_batch = 3;
      PatchmatchGpu batch(p3);
      std::vector<Image1b> ls = {il, ir, il}, rs = {ir, il, ir};
      std::vector<Image1f> dls, drs;
      batch.MatchBatch(ls, rs, dls, drs);
      if (dls.size() != 3 || drs.size() != 3) return 11;
      Image1f sw_l, sw_r;
      pm.Match(ir, il, sw_l, sw_r);
      const size_t bytes = sizeof(float) * (size_t)rows * cols;
      if (std::memcmp(dls[0].data(), auto_l.data(), bytes) != 0 || std::memcmp(drs[0].data(), auto_r.data(), bytes) != 0) return 12;
      if (std::memcmp(dls[1].data(), sw_l.data(), bytes) != 0 || std::memcmp(drs[1].data(), sw_r.data(), bytes) != 0) return 12;
      if (std::memcmp(dls[2].data(), auto_l.data(), bytes) != 0 || std::memcmp(drs[2].data(), auto_r.data(), bytes) != 0) return 12;
      bool threw = false;
      try {
        std::vector<Image1b> four = {il, il, il, il};
        batch.MatchBatch(four, four, dls, drs);
      } catch (const std::invalid_argument&) {
        threw = true;
      }
      if (!threw) return 13;
    }
    // the nested parameter types carry the reference's names (patchmatch_gpu.h:82-83)
    static_assert(std::is_same<decltype(params.detector_params), bm::ft::FeatureDetector::Params>::value, "nested Params");
    static_assert(std::is_same<decltype(params.matcher_params), bm::ft::StereoMatcher::Params>::value, "nested Params");
    // PatchmatchGpu::Match(GpuMat iml, imr, Gl, Gr, GpuMat& disp) (patchmatch_gpu.h:104-108): one view, gradients
    // from the caller, disp = sparse-init map in / result out; device buffers through pm/imaging.h's helpers
    {
      const size_t n = (size_t)rows * cols, bytes = n * sizeof(float);
      std::vector<float> fl(n), fr(n), gl(n), gr(n);
      for (size_t i = 0; i < n; ++i) {
        fl[i] = (float)il.data()[i];
        fr[i] = (float)ir.data()[i];
      }
      if (pm_gradient_magnitude(pm.handle(), il.data(), rows, cols, gl.data()) != PM_OK ||
          pm_gradient_magnitude(pm.handle(), ir.data(), rows, cols, gr.data()) != PM_OK)
        return 7;
      PatchmatchGpu::GpuImage1f dv[5];
      const float* host[5] = {fl.data(), fr.data(), gl.data(), gr.data(), sl.data()};
      for (int k = 0; k < 5; ++k) {
        void* d = nullptr;
        if (pm_device_malloc(pm.handle(), bytes, &d) != PM_OK || pm_upload(pm.handle(), d, host[k], bytes) != PM_OK)
          return 7;
        dv[k].data = (float*)d;
        dv[k].rows = rows;
        dv[k].cols = cols;
        dv[k].step = sizeof(float) * (size_t)cols;
      }
      pm.Match(dv[0], dv[1], dv[2], dv[3], dv[4]);
      Image1f view(rows, cols);
      if (pm_download(pm.handle(), view.data(), dv[4].data, bytes) != PM_OK) return 7;
      write_raw(dir + "/view_l.f32", view);
      for (int k = 0; k < 5; ++k) pm_device_free(pm.handle(), dv[k].data);
    }
    // a larger image re-plans to the envelope of the sizes seen; the old size still works afterwards
    {
      Image1b big_l(rows + 8, cols), big_r(rows + 8, cols);
      std::memset(big_l.data(), 0, (size_t)(rows + 8) * cols);
      std::memset(big_r.data(), 0, (size_t)(rows + 8) * cols);
      Image1f bd, bdr, again, againr;
      pm.SetSeeds(none, none);
      pm.Match(big_l, big_r, bd, bdr);
      if (bd.rows != rows + 8) return 8;
      pm.SetSeeds(sl, sr);
      pm.Match(il, ir, again, againr);
      if (std::memcmp(again.data(), disp.data(), sizeof(float) * (size_t)rows * cols) != 0) return 9;
    }
    std::cout << "ok " << disp.rows << "x" << disp.cols << "\n";
    return 0;
  } catch (const std::exception& e) {
    std::cout << "exception: " << e.what() << "\n";
    return 10;
  }
}
