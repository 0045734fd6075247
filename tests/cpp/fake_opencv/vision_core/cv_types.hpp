// Stand-in for the reference's src/vehicle/vision_core/cv_types.hpp (the two typedefs of :8 and :12 that the PatchMatch
// callers use), so that the compile test has what every translation unit of the vehicle tree has: bm::core::Image1b /
// Image1f declared as cv::Mat1b / cv::Mat1f BEFORE patchmatch_gpu.hpp declares them again.
#pragma once
#include <opencv2/core.hpp>

namespace bm {
namespace core {
typedef cv::Mat1b Image1b;
typedef cv::Mat1f Image1f;
}  // namespace core
}  // namespace bm
