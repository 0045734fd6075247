/* pm/imaging.h -- C ABI of the range-dependent post-processing that follows the stereo hot path
 * (SURVEY.md section 8, row f-3): disparity -> range -> Sea-thru style correction, per pixel.
 *
 * Replaces, for device-resident images:
 *   - StereoCamera::DispToDepth            src/vehicle/vision_core/stereo_camera.cpp:49-53
 *   - imaging::RemoveBackscatter           src/vehicle/imaging/backscatter.cpp:277-308
 *   - imaging::CorrectAttenuation          src/vehicle/imaging/attenuation.cpp:269-299
 *     (with SetMaxRangeWhereZero, :255-266)
 *   - ComputeIntensity                     src/vehicle/vision_core/image_util.cpp:97-102
 *   - imaging::FindDarkFast                src/vehicle/imaging/backscatter.cpp:41-78
 * i.e. the per-pixel stages of imaging::EnhanceUnderwater (src/vehicle/imaging/enhance.cpp:22-85).
 * The Levenberg-Marquardt parameter fits (EstimateBackscatter, EstimateBeta) and the guided filter
 * stay with the caller: they work on <= a few hundred sampled pixels and hand over B, beta_B, beta_D.
 *
 * Conventions: all image pointers are DEVICE memory, tightly packed; Image3f is interleaved BGR
 * float ([rows][cols][3]) like cv::Mat_<cv::Vec3f>; Image1f is [rows][cols] float.  Every function
 * enqueues on the handle's stream (pm_stream) and returns; pm_synchronize waits.  A disparity map
 * produced by pm_match_device on the same handle can therefore be consumed without a host round trip.
 * Arithmetic is float, in the reference's operation order; exp is the device's correctly-rounded-to-1-ulp
 * expf, so results agree with a host evaluation to a few ulp (tests state 1e-5 relative), not bit for bit.
 */
#ifndef PM_IMAGING_H_
#define PM_IMAGING_H_

#include "pm/patchmatch.h"

#ifdef __cplusplus
extern "C" {
#endif

/* range = (float)(fx * baseline / (double)disp) where disp > 0, else 0 ("no range", the value
 * RemoveBackscatter / CorrectAttenuation treat as background).  DispToDepth CHECK-fails on disp <= 0;
 * its callers skip those pixels (src/vehicle/mesher/object_mesher.cpp uses only tracked features). */
int pm_disp_to_range(pm_handle* h, const float* d_disp, int rows, int cols, double fx, double baseline,
                     float* d_range);

/* out = max(bgr - B * (1 - exp(-beta_B * z)), 0) per channel, z = range where range > 1e-3 else
 * range + 20 m (kBackgroundRange, backscatter.cpp:18). */
int pm_remove_backscatter(pm_handle* h, const float* d_bgr, const float* d_range, int rows, int cols,
                          const float B[3], const float beta_B[3], float* d_out);

/* out = bgr * exp(z * (a * exp(b z) + c * exp(d z))) per channel, X = (a_bgr, b_bgr, c_bgr, d_bgr),
 * z = range where range > 0 else range + max(range) (SetMaxRangeWhereZero). */
int pm_correct_attenuation(pm_handle* h, const float* d_bgr, const float* d_range, int rows, int cols,
                           const float X[12], float* d_out);

/* The three in one pass over the image: disparity map in, corrected image out (and the range map if
 * d_range_out is not NULL).  Equals pm_disp_to_range -> pm_remove_backscatter -> pm_correct_attenuation. */
int pm_range_enhance(pm_handle* h, const float* d_bgr, const float* d_disp, int rows, int cols, double fx,
                     double baseline, const float B[3], const float beta_B[3], const float X[12],
                     float* d_range_out, float* d_out);

/* gray = 0.114 B + 0.587 G + 0.299 R (cv::cvtColor BGR2GRAY on floats). */
int pm_compute_intensity(pm_handle* h, const float* d_bgr, int rows, int cols, float* d_gray);

/* FindDarkFast: the intensity threshold under which `percentile` of the pixels with range > 0.1 lie,
 * found by the reference's 1 + 8 counting steps; writes the mask (255 / 0) of the last step tested and
 * returns the threshold through *threshold.  Synchronises the stream (the counts steer the search). */
int pm_find_dark(pm_handle* h, const float* d_intensity, const float* d_range, int rows, int cols,
                 float percentile, uint8_t* d_mask, float* threshold);

/* ---- range-free enhancement in FRONT of stereo ---------------------------
 * The "stereo-ready" chain of test/imaging/enhance_test.cpp:69-73 and test/stereo_matching/sgbm_test.cpp:66-84:
 *   J    = Normalize(NormalizeColorIlluminant(CastImage3bTo3f(bgr8)))
 *          (src/vehicle/imaging/normalization.cpp:43-69, :178-185; illuminant.cpp:10-21; image_util.cpp:25-31;
 *          as written there: NormalizeColorIlluminant ends with a Normalize of its own, so the value channel is
 *          stretched twice)
 *   gray = cv::cvtColor(J, BGR2GRAY), converted to 8 bit (x 255, saturate_cast) -- the image Match() consumes.
 * d_bgr8: [rows][cols][3] bytes.  d_J ([rows][cols][3] float) and d_gray8 ([rows][cols] bytes) are optional
 * outputs (at least one).  rows, cols >= 8.  Scratch for the image size is allocated on first use. */
int pm_stereo_ready(pm_handle* h, const uint8_t* d_bgr8, int rows, int cols, float* d_J, uint8_t* d_gray8);

/* Match() on 8-bit BGR pairs with that enhancement FOLDED INTO THE LOAD PATH (BASELINE config 5: "underwater enhancement
 * fused into the cost kernel"): per image the two Gaussian passes of the illuminant estimate and two small min / max
 * passes run as kernels; the whole per-pixel tail -- I / (2 blur), both HSV value stretches, gray, 8 bit -- is computed
 * inside the prep kernel that produces the matcher's image / gradient planes, so neither the quotient image, nor the
 * stretched images, nor the gray image is ever written to memory.  The result equals pm_stereo_ready on both images
 * followed by pm_match_device, bit for bit, in every mode of the handle (scalar and PM_MODE_PLANES, f32 / f16 state).
 * d_left_bgr8 / d_right_bgr8: [n][rows][cols][3] bytes on the device; the other arguments as pm_match_device. */
int pm_match_bgr_device(pm_handle* h, int n, const uint8_t* d_left_bgr8, const uint8_t* d_right_bgr8, int rows, int cols,
                        const float* d_seed_l, const float* d_seed_r, float* d_disp_l, float* d_disp_r);

/* The two building blocks on float images, for callers that run them separately:
 * cv::GaussianBlur(src, dst, Size(ksize, ksize), sigma, sigma, BORDER_REPLICATE) for 1-4 interleaved channels
 * (EstimateIlluminantGaussian = 2 x this, illuminant.cpp:10-21), and imaging::Normalize. */
int pm_gaussian_blur(pm_handle* h, const float* d_src, int rows, int cols, int channels, int ksize, double sigma,
                     float* d_dst);
int pm_normalize(pm_handle* h, const float* d_bgr, int rows, int cols, float* d_out);
/* imaging::NormalizeColorIlluminant on a float image (normalization.cpp:178-185). */
int pm_normalize_color_illuminant(pm_handle* h, const float* d_bgr, int rows, int cols, float* d_out);

/* ---- device buffers for host code that does not include HIP (host/imaging.hpp uses them) ------------------
 * pm_device_malloc / pm_device_free wrap hipMalloc / hipFree on the handle's device; pm_upload / pm_download are
 * stream-ordered copies on the handle's stream from / to pageable host memory (pm_download returns after the
 * data has arrived). */
int pm_device_malloc(pm_handle* h, size_t bytes, void** d_ptr);
int pm_device_free(pm_handle* h, void* d_ptr);
int pm_upload(pm_handle* h, void* d_dst, const void* src, size_t bytes);
int pm_download(pm_handle* h, void* dst, const void* d_src, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif
