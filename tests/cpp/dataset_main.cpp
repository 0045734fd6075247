// The Sequence demo of the reference (test/stereo_matching/patchmatch_gpu_test.cpp:95-138): a dataset plays
// stereo pairs back into a callback that matches them -- here through the mirrored EurocDataset, the own PNG
// reader and the pipelined Submit()/Collect().  Modes:
//   dataset_main read <image> <out.raw>                 decode one image, write gray or BGR bytes, print "rows cols ch"
//   dataset_main readgray <jpeg> <out.raw>              IMREAD_GRAYSCALE of a JPEG (luma plane)
//   dataset_main play <euroc_root> <out_dir> <iters>    match every pair, write disp_<i>.f32 (left view)
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <iterator>
#include <string>
#include <vector>

#include "dataset.hpp"

using namespace bm;

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const std::string mode = argv[1];
  try {
    if (mode == "read") {
      core::Image1b gray;
      core::Image3b color;
      const int ch = core::ReadImage(argv[2], &gray, &color);
      std::ofstream f(argv[3], std::ios::binary);
      if (ch == 1) f.write(reinterpret_cast<const char*>(gray.data()), (std::streamsize)gray.rows * gray.cols);
      else f.write(reinterpret_cast<const char*>(color.data()), (std::streamsize)color.rows * color.cols * 3);
      std::printf("%d %d %d\n", ch == 1 ? gray.rows : color.rows, ch == 1 ? gray.cols : color.cols, ch);
      if (ch == 3) {
        const core::Image1b g = core::ConvertToGray(color);
        std::ofstream fg(std::string(argv[3]) + ".gray", std::ios::binary);
        fg.write(reinterpret_cast<const char*>(g.data()), (std::streamsize)g.rows * g.cols);
      }
      return 0;
    }
    if (mode == "readgray") {  // cv::imread(path, IMREAD_GRAYSCALE) of a JPEG: the luma plane
      std::ifstream in(argv[2], std::ios::binary);
      std::vector<char> buf((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
      core::Image1b gray;
      core::DecodeJpeg(reinterpret_cast<const uint8_t*>(buf.data()), buf.size(), false, &gray, nullptr);
      std::ofstream f(argv[3], std::ios::binary);
      f.write(reinterpret_cast<const char*>(gray.data()), (std::streamsize)gray.rows * gray.cols);
      std::printf("%d %d 1\n", gray.rows, gray.cols);
      return 0;
    }
    if (mode == "play" && argc >= 5) {
      dataset::EurocDataset ds(argv[2]);
      const std::string out_dir = argv[3];
      pm::PatchmatchGpu::Params params;
      params.semantics = PM_SEM_CPU;
      params.patch_size = 5;
      params.patchmatch_iters = atoi(argv[4]);
      params.max_batch = 2;
      pm::PatchmatchGpu matcher(params);
      int collected = 0;
      auto save = [&](const core::Image1f& d) {
        std::ofstream f(out_dir + "/disp_" + std::to_string(collected++) + ".f32", std::ios::binary);
        f.write(reinterpret_cast<const char*>(d.data()), (std::streamsize)sizeof(float) * d.rows * d.cols);
      };
      core::Image1f disp, dispr;
      ds.RegisterStereoCallback([&](const core::StereoImage1b& pair) {
        while (!matcher.Submit(pair.left_image, pair.right_image, pair.camera_id))
          if (matcher.Collect(disp, dispr)) save(disp);
      });
      ds.Playback(-1.0f);  // as fast as possible
      while (matcher.Collect(disp, dispr)) save(disp);
      std::printf("played %zu pairs, collected %d\n", ds.NumStereo(), collected);
      return collected == (int)ds.NumStereo() ? 0 : 4;
    }
    return 2;
  } catch (const std::exception& e) {
    std::cout << "exception: " << e.what() << "\n";
    return 10;
  }
}
