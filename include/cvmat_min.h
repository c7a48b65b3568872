// cvmat_min.h -- a minimal stand-in for the slice of cv::Mat that the reference facade uses
// (reference src/ImageCompressorDll/ImageCompressor.h:12,34,35 takes / returns cv::Mat). This image has no OpenCV;
// when the real headers are available define MIJ_HAVE_OPENCV and <opencv2/core.hpp> is used instead, unchanged.
#ifndef MIJ_CVMAT_MIN_H_
#define MIJ_CVMAT_MIN_H_
#ifdef MIJ_HAVE_OPENCV
#include <opencv2/core.hpp>
#else
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

#define CV_8UC3 16

namespace cv {
// Ref-counted, row-major, 8-bit 3-channel (BGR) image: the only Mat flavour the reference path handles
// (cv::imread(..., IMREAD_COLOR), reference main.cpp:37).
class Mat {
 public:
  int rows = 0, cols = 0;
  size_t step = 0;
  unsigned char *data = nullptr;
  Mat() = default;
  Mat(int r, int c, int type) : rows(r), cols(c), step((size_t)c * 3), buf_(std::make_shared<std::vector<unsigned char>>((size_t)r * c * 3)) {
    (void)type;
    data = buf_->data();
  }
  Mat(int r, int c, int type, void *external, size_t stride = 0) : rows(r), cols(c), step(stride ? stride : (size_t)c * 3), data((unsigned char *)external) { (void)type; }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  int type() const { return CV_8UC3; }
  int channels() const { return 3; }
  size_t total() const { return (size_t)rows * cols; }
  size_t elemSize() const { return 3; }
  bool isContinuous() const { return step == (size_t)cols * 3; }
  template <typename T> T *ptr(int r = 0) { return reinterpret_cast<T *>(data + (size_t)r * step); }
  template <typename T> const T *ptr(int r = 0) const { return reinterpret_cast<const T *>(data + (size_t)r * step); }

 private:
  std::shared_ptr<std::vector<unsigned char>> buf_;
};
}  // namespace cv
#endif
#endif  // MIJ_CVMAT_MIN_H_
