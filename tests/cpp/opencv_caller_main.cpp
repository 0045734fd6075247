// A caller written against the REFERENCE's types and header name, compiled against this build with zero edits to the
// call lines: cv::Mat1b / cv::Mat1f images (here the stand-in of tests/cpp/fake_opencv), the reference's include line,
// its using-directives, and the call sequence of test/stereo_matching/patchmatch_gpu_test.cpp:68-88 (parameters, the
// constructor, `Image1f disp, dispr;` left EMPTY, Match() five times) without imread / imshow / LOG / Timer.
// tests/test_cpp_wrapper.py writes the 376x240 farmsim pair of tests/golden as raw files and compares what this program
// writes with the fixture's row checksums.   usage: opencv_caller_main <dir> <rows> <cols>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <type_traits>
#include <vector>

#include <opencv2/core.hpp>

// What every translation unit of the vehicle tree has seen before it reaches the PatchMatch header: the image typedefs of
// src/vehicle/vision_core/cv_types.hpp:8,12.  The header below declares them AGAIN (identically), which is legal C++ --
// this block is here so that the compile test proves the two declarations coexist.
namespace bm {
namespace core {
typedef cv::Mat1b Image1b;
typedef cv::Mat1f Image1f;
}  // namespace core
}  // namespace bm

#include "patchmatch_gpu/patchmatch_gpu.h"

using namespace bm;
using namespace core;
using namespace pm;

static_assert(std::is_same<Image1b, cv::Mat1b>::value && std::is_same<Image1f, cv::Mat1f>::value,
              "with OpenCV's header on the include path the image types are OpenCV's");

static bool read_raw(const std::string& path, void* dst, size_t bytes) {
  std::ifstream f(path, std::ios::binary);
  f.read(reinterpret_cast<char*>(dst), (std::streamsize)bytes);
  return (bool)f;
}
static void write_rows(const std::string& path, const Image1f& im) {
  std::ofstream f(path, std::ios::binary);
  for (int r = 0; r < im.rows; ++r) f.write(reinterpret_cast<const char*>(im.ptr(r)), (std::streamsize)sizeof(float) * im.cols);
}

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const std::string dir = argv[1];
  const int rows = atoi(argv[2]), cols = atoi(argv[3]);
  Image1b il(rows, cols), ir(rows, cols);
  if (!read_raw(dir + "/left.u8", il.data, (size_t)rows * cols) || !read_raw(dir + "/right.u8", ir.data, (size_t)rows * cols)) {
    std::cerr << "cannot read inputs\n";
    return 3;
  }
  try {
    // ---- test/stereo_matching/patchmatch_gpu_test.cpp:68-88 -------------------------------------------------------
    PatchmatchGpu::Params params;
    float max_disp = 128;

    params.matcher_params.templ_cols = 31;
    params.matcher_params.templ_rows = 11;
    params.matcher_params.max_disp = max_disp;
    params.matcher_params.max_matching_cost = 0.15;
    params.matcher_params.bidirectional = true;
    params.matcher_params.subpixel_refinement = false;

    params.cost_alpha = 0.9;
    params.patchmatch_iters = 3;

    PatchmatchGpu pm(params);
    Image1f disp, dispr;

    for (int i = 0; i < 5; ++i) {
      pm.Match(il, ir, disp, dispr);
    }
    // ----------------------------------------------------------------------------------------------------------------
    if (disp.rows != rows || disp.cols != cols || dispr.rows != rows || dispr.cols != cols || !disp.isContinuous()) return 4;
    write_rows(dir + "/disp_l.f32", disp);
    write_rows(dir + "/disp_r.f32", dispr);

    // the same pair as ROIs of wider buffers (step > cols), into maps that start out with another size
    const size_t step = (size_t)cols + 40;
    std::vector<uint8_t> wl(step * rows, 7), wr(step * rows, 9);
    for (int r = 0; r < rows; ++r)
      for (int c = 0; c < cols; ++c) {
        wl[r * step + c] = il(r, c);
        wr[r * step + c] = ir(r, c);
      }
    Image1b sl(rows, cols, wl.data(), step), sr(rows, cols, wr.data(), step);
    Image1f d2(5, 9), d2r(rows, cols + 3);
    pm.Match(sl, sr, d2, d2r);
    if (d2.rows != rows || d2.cols != cols || d2r.cols != cols) return 5;
    write_rows(dir + "/strided_l.f32", d2);
    write_rows(dir + "/strided_r.f32", d2r);

    // SparseInit(iml, imr, dilate_factor) (patchmatch_gpu.h:110-112) returns an Image1f = cv::Mat1f
    Image1f seeds = pm.SparseInit(il, ir, 4);
    if (seeds.rows != rows || seeds.cols != cols) return 6;
    write_rows(dir + "/sparse_init.f32", seeds);
    pm.SetSeeds(seeds, Image1f());
    pm.SetSeeds(Image1f(), Image1f());  // cleared again: the sequence below seeds itself like Match()

    // the frame-sequence and batch forms on the same image class: Submit / Collect (the Sequence caller's loop,
    // patchmatch_gpu_test.cpp:118-128, with the copies off the critical path) and MatchBatch
    {
      PatchmatchGpu::Params p2 = params;
      p2.max_batch = 2;
      PatchmatchGpu seq(p2);
      Image1f a, b, c, d;
      if (!seq.Submit(il, ir, 1) || !seq.Submit(il, ir, 2)) return 7;
      uint64_t tag = 0;
      if (!seq.Collect(a, b, &tag) || tag != 1 || !seq.Collect(c, d, &tag) || tag != 2 || seq.Collect(c, d)) return 8;
      write_rows(dir + "/seq_l.f32", c);
      write_rows(dir + "/seq_r.f32", d);
      std::vector<Image1b> ls{il, il}, rs{ir, ir};
      std::vector<Image1f> dls, drs;
      seq.MatchBatch(ls, rs, dls, drs);
      if (dls.size() != 2 || dls[1].rows != rows || drs[1].cols != cols) return 9;
      write_rows(dir + "/batch_l.f32", dls[1]);
      write_rows(dir + "/batch_r.f32", drs[1]);
      Image1f bound_l(rows, cols), bound_r(rows, cols);
      seq.Register(bound_l);
      seq.Register(bound_r);
      if (!seq.Submit(il, ir, bound_l, bound_r, 3) || !seq.Collect(&tag) || tag != 3) return 11;
      write_rows(dir + "/bound_l.f32", bound_l);
      seq.Unregister(bound_l);
      seq.Unregister(bound_r);
    }
  } catch (const std::exception& e) {
    std::cout << "exception: " << e.what() << "\n";
    return 10;
  }
  std::cout << "ok\n";
  return 0;
}
