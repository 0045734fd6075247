// imaging.hpp -- host-side C++ mirror of the bm::imaging functions on the rows either side of the stereo
// hot path, same names and signatures as the reference headers
//   src/vehicle/imaging/normalization.hpp:12,41   Normalize, NormalizeColorIlluminant
//   src/vehicle/imaging/illuminant.hpp:10-15      EstimateIlluminantGaussian
//   src/vehicle/imaging/backscatter.hpp:13,45     FindDarkFast, RemoveBackscatter
//   src/vehicle/imaging/attenuation.hpp:59        CorrectAttenuation
//   src/vehicle/vision_core/image_util.hpp:34     ComputeIntensity
//   src/vehicle/vision_core/image_util.cpp:25     CastImage3bTo3f
// over the C ABI of pm/imaging.h.  Host images in, host images out: each call uploads, runs the device
// kernels and downloads (callers that keep images on the device use the C ABI directly).
// No HIP header, no OpenCV, no Eigen: Image3f is bm::core::Image<Vec3f> (interleaved BGR like cv::Vec3f),
// Vector3f / Vector12f are std::array.  The functions run on a process-wide context (device 0 unless
// SetDevice() is called first) and throw std::runtime_error when no GPU is usable -- there is no CPU path.
#pragma once

#include <array>

#include "patchmatch_gpu.hpp"
#include "pm/imaging.h"

namespace bm {
namespace core {
struct Vec3f {
  float v[3];
  float& operator()(int i) { return v[i]; }
  float operator()(int i) const { return v[i]; }
};
struct Vec3b {
  uint8_t v[3];
};
typedef Image<Vec3f> Image3f;
typedef Image<Vec3b> Image3b;
typedef std::array<float, 3> Vector3f;
typedef std::array<float, 12> Vector12f;
}  // namespace core

namespace imaging {
using namespace core;

// Device used by the free functions below; call before the first of them (default 0).
void SetDevice(int device);

Image3f CastImage3bTo3f(const Image3b& im);
Image1f ComputeIntensity(const Image3f& bgr);
Image3f EstimateIlluminantGaussian(const Image3f& bgr, int ksizeX, int ksizeY, double sigmaX, double sigmaY);
Image3f Normalize(const Image3f& bgr);
Image3f NormalizeColorIlluminant(const Image3f bgr);
float FindDarkFast(const Image1f& intensity, const Image1f& range, float percentile, Image1b& mask);
Image3f RemoveBackscatter(const Image3f& bgr, const Image1f& range, const Vector3f& B, const Vector3f& beta_B);
Image3f CorrectAttenuation(const Image3f& bgr, const Image1f& range, const Vector12f& X);

// Not in the reference as a function: the chain the reference's tests assemble by hand
// (test/imaging/enhance_test.cpp:69-73): 8-bit colour image -> the 8-bit gray image Match() consumes.
Image1b StereoReady(const Image3b& bgr);
// StereoCamera::DispToDepth (vision_core/stereo_camera.cpp:49-53) over a map; 0 where disp <= 0.
Image1f DispToDepth(const Image1f& disp, double fx, double baseline);

}  // namespace imaging
}  // namespace bm
