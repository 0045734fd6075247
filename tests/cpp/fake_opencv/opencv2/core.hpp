// A STAND-IN for <opencv2/core.hpp>, for compile tests of the drop-in boundary only (tests/test_cpp_wrapper.py):
// OpenCV is not in this image, and the reference's callers hold their images in cv::Mat1b / cv::Mat1f
// (src/vehicle/vision_core/cv_types.hpp:8-12).  cv::Mat_<T> below has the members that
// ocean-perception_amd/host/patchmatch_gpu.hpp touches -- rows, cols, step, data, empty(), create(), the
// (rows, cols) and (rows, cols, data, step) constructors -- with OpenCV's meaning (shared storage on copy, step in
// bytes, create() a no-op when size already matches).  It is not an oracle and computes nothing.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

namespace cv {

typedef unsigned char uchar;

struct MatStep {  // cv::MatStep: converts to the row step in bytes
  size_t p[2] = {0, 0};
  operator size_t() const { return p[0]; }
  size_t operator[](int i) const { return p[i]; }
  MatStep& operator=(size_t s) {
    p[0] = s;
    return *this;
  }
};

struct Size {
  int width = 0, height = 0;
};

template <typename T>
class Mat_ {
 public:
  int rows = 0, cols = 0;
  uchar* data = nullptr;
  MatStep step;

  Mat_() { step.p[1] = sizeof(T); }
  Mat_(int r, int c) : Mat_() { create(r, c); }
  Mat_(int r, int c, const T& value) : Mat_() {
    create(r, c);
    for (int y = 0; y < r; ++y)
      for (int x = 0; x < c; ++x) (*this)(y, x) = value;
  }
  // user-allocated data, `s` = bytes per row (0 = continuous): nothing is copied or owned
  Mat_(int r, int c, T* user, size_t s = 0) : Mat_() {
    rows = r;
    cols = c;
    data = reinterpret_cast<uchar*>(user);
    step = s ? s : sizeof(T) * (size_t)c;
  }
  void create(int r, int c) {
    if (data && r == rows && c == cols) return;
    store_ = std::make_shared<std::vector<uchar>>(sizeof(T) * (size_t)r * (size_t)c);
    rows = r;
    cols = c;
    data = store_->data();
    step = sizeof(T) * (size_t)c;
  }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  bool isContinuous() const { return (size_t)step == sizeof(T) * (size_t)cols; }
  Size size() const { return Size{cols, rows}; }
  T* ptr(int r = 0) { return reinterpret_cast<T*>(data + (size_t)r * (size_t)step); }
  const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + (size_t)r * (size_t)step); }
  T& operator()(int r, int c) { return ptr(r)[c]; }
  const T& operator()(int r, int c) const { return ptr(r)[c]; }

 private:
  std::shared_ptr<std::vector<uchar>> store_;  // copies of a Mat_ share it, as cv::Mat's reference count does
};

typedef Mat_<uchar> Mat1b;
typedef Mat_<float> Mat1f;

}  // namespace cv
