// dataset.hpp -- the ingest side of the stereo hot path: host-side mirror of the parts of
// bm::dataset the stereo callers use,
//   StereoImage<T>, StereoImage1b / 3b        src/vehicle/vision_core/stereo_image.hpp:13-33
//   StereoDatasetItem, DataProvider           src/vehicle/dataset/data_provider.hpp:46-163 (stereo source only)
//   EurocDataset                              src/vehicle/dataset/euroc_dataset.cpp:12-19,116-165 (ParseStereo)
//   MaybeConvertToGray                        src/vehicle/vision_core/image_util.cpp:52-61
// with an own image reader in place of cv::imread (8-bit PNG through zlib, binary PGM / PPM).  IMU, depth, range
// and ground-truth streams are not on the stereo path and are not mirrored.  JPEG files / LCM image_t payloads
// (lcm_util/decode_image.cpp:11-32) go through the own baseline decoder in jpeg.cpp.
#pragma once

#include <cstdint>
#include <functional>
#include <limits>
#include <string>
#include <vector>

#include "imaging.hpp"

namespace bm {
namespace core {
typedef uint64_t timestamp_t;  // core/timestamp.hpp:11 (nanoseconds)
typedef uint64_t uid_t;        // core/uid.hpp:9
static const timestamp_t kMaxTimestamp = std::numeric_limits<timestamp_t>::max();

template <typename ImageT>
struct StereoImage final {
  explicit StereoImage(timestamp_t timestamp_, uid_t camera_id_, const ImageT& l, const ImageT& r)
      : timestamp(timestamp_), camera_id(camera_id_), left_image(l), right_image(r) {}
  timestamp_t timestamp;
  uid_t camera_id;
  ImageT left_image;
  ImageT right_image;
};
typedef StereoImage<Image1b> StereoImage1b;
typedef StereoImage<Image3b> StereoImage3b;

// cv::imread(path, cv::IMREAD_ANYCOLOR) for PNG / PNM / JPEG: one channel -> gray, otherwise BGR (alpha dropped).
// Exactly one of *gray / *color is filled; returns the channel count (1 or 3).  Throws std::runtime_error on
// unreadable or unsupported files (16-bit, palette, interlaced PNG; ASCII PNM).
int ReadImage(const std::string& path, Image1b* gray, Image3b* color);
// cv::imdecode for a JPEG held in memory (lcm_util/decode_image.cpp:11-32: DecodeJPG) -- baseline / extended
// sequential Huffman, 8 bit, 1 or 3 components.  want_color == false is IMREAD_GRAYSCALE: the luma plane as the
// decoder library delivers it (no colour conversion); true: BGR.  Returns the channel count written.
int DecodeJpeg(const uint8_t* buf, size_t n, bool want_color, Image1b* gray, Image3b* color);
// cv::cvtColor(BGR2GRAY) on 8-bit images: (1868 B + 9617 G + 4899 R + 8192) >> 14.
Image1b ConvertToGray(const Image3b& bgr);
}  // namespace core

namespace dataset {
using namespace core;

typedef std::function<void(const StereoImage1b&)> StereoCallback1b;
typedef std::function<void(const StereoImage3b&)> StereoCallback3b;

struct StereoDatasetItem {
  explicit StereoDatasetItem(timestamp_t timestamp_, const std::string& l, const std::string& r)
      : timestamp(timestamp_), path_left(l), path_right(r) {}
  timestamp_t timestamp;
  std::string path_left;
  std::string path_right;
};

class DataProvider {
 public:
  DataProvider() = default;
  virtual ~DataProvider() = default;

  void RegisterStereoCallback(StereoCallback1b cb) { stereo_callbacks_1b_.emplace_back(cb); }
  void RegisterStereoCallback(StereoCallback3b cb) { stereo_callbacks_3b_.emplace_back(cb); }

  // Loads the next stereo pair and hands it to the callbacks (3-channel callbacks only see colour pairs; the
  // 1-channel callbacks get the gray conversion).  false when no data is left.
  bool Step(bool verbose = false);
  // Plays everything back chronologically on a worker thread, sleeping (t_next - t) / speed between frames;
  // speed < 0: as fast as possible.
  void Playback(float speed = 1.0f, bool verbose = false);
  void Reset();
  timestamp_t FirstTimestamp() const;
  size_t NumStereo() const { return stereo_data.size(); }

 protected:
  std::vector<StereoDatasetItem> stereo_data;

 private:
  void PlaybackWorker(float speed, bool verbose);
  std::vector<StereoCallback1b> stereo_callbacks_1b_;
  std::vector<StereoCallback3b> stereo_callbacks_3b_;
  size_t next_stereo_idx_ = 0;
  timestamp_t last_data_timestamp_ = 0;
};

// <toplevel>/mav0/cam0/data.csv + data/<timestamp>.png, the same for cam1 (euroc_dataset.cpp:116-165).
class EurocDataset : public DataProvider {
 public:
  explicit EurocDataset(const std::string& toplevel_path);

 private:
  void ParseStereo(const std::string& cam0_path, const std::string& cam1_path);
  void ParseImageFolder(const std::string& cam_folder, std::vector<timestamp_t>& output_timestamps,
                        std::vector<std::string>& output_filenames);
};

}  // namespace dataset
}  // namespace bm
