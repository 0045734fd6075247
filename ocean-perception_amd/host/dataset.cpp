// dataset.cpp -- stereo ingest: EuRoC folder layout, PNG / PNM reader, playback (see dataset.hpp).
#include "dataset.hpp"

#include <zlib.h>

#include <chrono>
#include <cstring>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <thread>

namespace bm {
namespace core {
namespace {

std::vector<uint8_t> ReadFile(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("ReadImage: cannot open " + path);
  f.seekg(0, std::ios::end);
  const std::streamoff n = f.tellg();
  f.seekg(0, std::ios::beg);
  std::vector<uint8_t> buf((size_t)(n > 0 ? n : 0));
  if (n > 0) f.read(reinterpret_cast<char*>(buf.data()), n);
  if (!f) throw std::runtime_error("ReadImage: short read on " + path);
  return buf;
}

uint32_t Be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

// Decoded pixels, `ch` interleaved channels in FILE order (PNG / PPM: RGB[A]).
struct Raster {
  int rows = 0, cols = 0, ch = 0;
  std::vector<uint8_t> px;
};

int Paeth(int a, int b, int c) {
  const int p = a + b - c;
  const int pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

Raster DecodePng(const std::vector<uint8_t>& file, const std::string& path) {
  static const uint8_t kSig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (file.size() < 8 + 25 || std::memcmp(file.data(), kSig, 8) != 0) throw std::runtime_error("not a PNG: " + path);
  Raster r;
  std::vector<uint8_t> idat;
  size_t pos = 8;
  bool have_ihdr = false, done = false;
  int color_type = 0;
  while (!done && pos + 12 <= file.size()) {
    const uint32_t len = Be32(&file[pos]);
    const uint8_t* type = &file[pos + 4];
    if (pos + 12 + (size_t)len > file.size()) throw std::runtime_error("truncated PNG chunk: " + path);
    const uint8_t* data = &file[pos + 8];
    const uint32_t crc = Be32(&file[pos + 8 + len]);
    if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), type, 4 + len) != crc) throw std::runtime_error("PNG CRC mismatch: " + path);
    if (std::memcmp(type, "IHDR", 4) == 0) {
      if (len != 13) throw std::runtime_error("bad IHDR: " + path);
      r.cols = (int)Be32(data);
      r.rows = (int)Be32(data + 4);
      const int depth = data[8];
      color_type = data[9];
      if (depth != 8 || data[12] != 0 || (color_type != 0 && color_type != 2 && color_type != 4 && color_type != 6))
        throw std::runtime_error("unsupported PNG (need 8-bit gray / RGB [+alpha], non-interlaced): " + path);
      r.ch = color_type == 0 ? 1 : (color_type == 2 ? 3 : (color_type == 4 ? 2 : 4));
      have_ihdr = true;
    } else if (std::memcmp(type, "IDAT", 4) == 0) {
      idat.insert(idat.end(), data, data + len);
    } else if (std::memcmp(type, "IEND", 4) == 0) {
      done = true;
    }
    pos += 12 + (size_t)len;
  }
  if (!have_ihdr || idat.empty() || r.rows <= 0 || r.cols <= 0) throw std::runtime_error("incomplete PNG: " + path);
  const size_t stride = (size_t)r.cols * r.ch;
  std::vector<uint8_t> raw((stride + 1) * (size_t)r.rows);
  uLongf out_len = (uLongf)raw.size();
  if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size())
    throw std::runtime_error("PNG inflate failed: " + path);
  r.px.resize(stride * (size_t)r.rows);
  const int bpp = r.ch;
  for (int y = 0; y < r.rows; ++y) {
    const uint8_t* in = &raw[(stride + 1) * (size_t)y];
    uint8_t* cur = &r.px[stride * (size_t)y];
    const uint8_t* up = y > 0 ? cur - stride : nullptr;
    const int filter = in[0];
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= (size_t)bpp ? cur[i - bpp] : 0;
      const int b = up ? up[i] : 0;
      const int c = (up && i >= (size_t)bpp) ? up[i - bpp] : 0;
      int v = in[1 + i];
      switch (filter) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) / 2; break;
        case 4: v += Paeth(a, b, c); break;
        default: throw std::runtime_error("bad PNG filter: " + path);
      }
      cur[i] = (uint8_t)v;
    }
  }
  return r;
}

Raster DecodePnm(const std::vector<uint8_t>& file, const std::string& path) {
  size_t pos = 2;
  auto next_int = [&]() -> int {
    for (;;) {
      while (pos < file.size() && (file[pos] == ' ' || file[pos] == '\n' || file[pos] == '\r' || file[pos] == '\t')) ++pos;
      if (pos < file.size() && file[pos] == '#') {
        while (pos < file.size() && file[pos] != '\n') ++pos;
        continue;
      }
      break;
    }
    int v = 0;
    bool any = false;
    while (pos < file.size() && file[pos] >= '0' && file[pos] <= '9') {
      v = v * 10 + (file[pos++] - '0');
      any = true;
    }
    if (!any) throw std::runtime_error("bad PNM header: " + path);
    return v;
  };
  Raster r;
  r.ch = file[1] == '5' ? 1 : 3;
  r.cols = next_int();
  r.rows = next_int();
  const int maxval = next_int();
  ++pos;  // the single whitespace byte after maxval
  const size_t n = (size_t)r.rows * r.cols * r.ch;
  if (maxval != 255 || r.rows <= 0 || r.cols <= 0 || pos + n > file.size())
    throw std::runtime_error("unsupported or truncated PNM (need binary, maxval 255): " + path);
  r.px.assign(file.begin() + (long)pos, file.begin() + (long)(pos + n));
  return r;
}

}  // namespace

int ReadImage(const std::string& path, Image1b* gray, Image3b* color) {
  const std::vector<uint8_t> file = ReadFile(path);
  Raster r;
  if (file.size() >= 2 && file[0] == 0xFF && file[1] == 0xD8) {
    // IMREAD_ANYCOLOR: a 3-component JPEG comes back as BGR, a 1-component one as gray
    bool three = false;
    for (size_t i = 2; i + 9 < file.size() && file[i] == 0xFF;) {
      const int m = file[i + 1];
      if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
        three = file[i + 9] == 3;
        break;
      }
      if (m == 0xDA) break;
      i += 2 + (((size_t)file[i + 2] << 8) | file[i + 3]);
    }
    return DecodeJpeg(file.data(), file.size(), three, gray, color);
  }
  if (file.size() >= 2 && file[0] == 'P' && (file[1] == '5' || file[1] == '6')) r = DecodePnm(file, path);
  else r = DecodePng(file, path);
  if (r.ch <= 2) {  // gray [+ alpha]
    if (!gray) throw std::runtime_error("ReadImage: no gray output given for " + path);
    gray->create(r.rows, r.cols);
    for (size_t i = 0; i < (size_t)r.rows * r.cols; ++i) gray->data()[i] = r.px[i * r.ch];
    return 1;
  }
  if (!color) throw std::runtime_error("ReadImage: no colour output given for " + path);
  color->create(r.rows, r.cols);
  for (size_t i = 0; i < (size_t)r.rows * r.cols; ++i) {  // file order RGB[A] -> BGR
    Vec3b& o = color->data()[i];
    o.v[0] = r.px[i * r.ch + 2];
    o.v[1] = r.px[i * r.ch + 1];
    o.v[2] = r.px[i * r.ch + 0];
  }
  return 3;
}

Image1b ConvertToGray(const Image3b& bgr) {
  Image1b out(bgr.rows, bgr.cols);
  for (size_t i = 0; i < (size_t)bgr.rows * bgr.cols; ++i) {
    const Vec3b& p = bgr.data()[i];
    out.data()[i] = (uint8_t)((p.v[0] * 1868 + p.v[1] * 9617 + p.v[2] * 4899 + 8192) >> 14);
  }
  return out;
}

}  // namespace core

namespace dataset {

static std::string Join(const std::string& a, const std::string& b) {
  if (a.empty()) return b;
  return a.back() == '/' ? a + b : a + "/" + b;
}
static bool Exists(const std::string& path) { return (bool)std::ifstream(path); }

bool DataProvider::Step(bool verbose) {
  if (next_stereo_idx_ >= stereo_data.size()) return false;
  const StereoDatasetItem& item = stereo_data.at(next_stereo_idx_);
  if (verbose) std::cout << "Step() t=" << item.timestamp << " type=STEREO" << std::endl;
  if (!Exists(item.path_left)) throw std::runtime_error("ERROR: Left image filepath is invalid:\n  " + item.path_left);
  if (!Exists(item.path_right)) throw std::runtime_error("ERROR: Right image filepath is invalid:\n  " + item.path_right);
  Image1b gl, gr;
  Image3b cl, cr;
  const int chl = ReadImage(item.path_left, &gl, &cl), chr = ReadImage(item.path_right, &gr, &cr);
  if (chl > 1 && chr > 1) {
    const StereoImage3b stereo3b(item.timestamp, next_stereo_idx_, cl, cr);
    for (const StereoCallback3b& f : stereo_callbacks_3b_) f(stereo3b);
  }
  if (chl > 1) gl = ConvertToGray(cl);  // MaybeConvertToGray
  if (chr > 1) gr = ConvertToGray(cr);
  const StereoImage1b stereo1b(item.timestamp, next_stereo_idx_, gl, gr);
  for (const StereoCallback1b& f : stereo_callbacks_1b_) f(stereo1b);
  last_data_timestamp_ = item.timestamp;
  ++next_stereo_idx_;
  return true;
}

void DataProvider::PlaybackWorker(float speed, bool verbose) {
  while (Step(verbose)) {
    if (next_stereo_idx_ >= stereo_data.size()) break;
    if (speed > 0.f) {
      const float ns_until_next = (float)(stereo_data.at(next_stereo_idx_).timestamp - last_data_timestamp_) / speed;
      std::this_thread::sleep_for(std::chrono::nanoseconds((timestamp_t)ns_until_next));
    }
  }
}

void DataProvider::Playback(float speed, bool verbose) {
  if (speed >= 0.f && speed <= 0.01f) throw std::invalid_argument("Cannot go slower than 1% speed");
  std::thread worker(&DataProvider::PlaybackWorker, this, speed, verbose);
  worker.join();
}

void DataProvider::Reset() {
  last_data_timestamp_ = 0;
  next_stereo_idx_ = 0;
}

timestamp_t DataProvider::FirstTimestamp() const {
  if (stereo_data.empty()) throw std::runtime_error("FirstTimestamp: no data");
  return stereo_data.front().timestamp;
}

EurocDataset::EurocDataset(const std::string& toplevel_path) : DataProvider() {
  const std::string mav0_path = Join(toplevel_path, "mav0");
  ParseStereo(Join(mav0_path, "cam0"), Join(mav0_path, "cam1"));
}

void EurocDataset::ParseStereo(const std::string& cam0_path, const std::string& cam1_path) {
  std::vector<timestamp_t> left_stamps, right_stamps;
  std::vector<std::string> lf, rf;
  ParseImageFolder(cam0_path, left_stamps, lf);
  ParseImageFolder(cam1_path, right_stamps, rf);
  const size_t N = left_stamps.size();
  if (right_stamps.size() != N || lf.size() != N || rf.size() != N)
    throw std::runtime_error("Different number of left/right images and timestamps");
  for (size_t i = 0; i < N; ++i) {
    if (left_stamps.at(i) != right_stamps.at(i)) throw std::runtime_error("Left/right timestamps don't match!");
    if (!Exists(lf.at(i))) throw std::runtime_error("missing image " + lf.at(i));
    if (!Exists(rf.at(i))) throw std::runtime_error("missing image " + rf.at(i));
    stereo_data.emplace_back(StereoDatasetItem(left_stamps.at(i), lf.at(i), rf.at(i)));
  }
}

void EurocDataset::ParseImageFolder(const std::string& cam_folder, std::vector<timestamp_t>& output_timestamps,
                                    std::vector<std::string>& output_filenames) {
  const std::string data_csv_path = Join(cam_folder, "data.csv");
  std::ifstream fin(data_csv_path.c_str());
  if (!fin.is_open()) throw std::runtime_error("Cannot open file: " + data_csv_path);
  std::string item;
  std::getline(fin, item);  // header
  while (std::getline(fin, item)) {
    if (item.empty()) continue;
    const size_t idx = item.find_first_of(',');
    const timestamp_t timestamp = (timestamp_t)std::stoll(item.substr(0, idx));
    // like the reference, the file name is derived from the timestamp column (euroc_dataset.cpp:155-158)
    output_timestamps.emplace_back(timestamp);
    output_filenames.emplace_back(Join(cam_folder, "data/" + item.substr(0, idx) + ".png"));
  }
}

}  // namespace dataset
}  // namespace bm
