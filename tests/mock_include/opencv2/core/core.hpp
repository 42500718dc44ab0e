// TEST-ONLY MOCK of <opencv2/core/core.hpp> (see ../../README.md): cv::Mat with the members the dvo_amd adaptor uses, with
// OpenCV's signatures.  Owns a contiguous buffer; no reference counting of sub-matrices, no ROI, no types beyond the three below.
#ifndef DVO_AMD_TEST_MOCK_OPENCV_CORE
#define DVO_AMD_TEST_MOCK_OPENCV_CORE
#include <cstddef>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_16U 2
#define CV_32F 5
#define CV_CN_SHIFT 3
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_16UC1 CV_MAKETYPE(CV_16U, 1)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)

namespace cv {

typedef unsigned char uchar;

struct Size {
  int width, height;
  Size() : width(0), height(0) {}
  Size(int w, int h) : width(w), height(h) {}
  bool operator==(const Size &o) const { return width == o.width && height == o.height; }
  bool operator!=(const Size &o) const { return !(*this == o); }
};

struct MatStep {
  size_t bytes;
  MatStep() : bytes(0) {}
  operator size_t() const { return bytes; }
  bool operator!=(const MatStep &o) const { return bytes != o.bytes; }
  bool operator==(const MatStep &o) const { return bytes == o.bytes; }
};

class Mat {
 public:
  int rows, cols;
  uchar *data;
  MatStep step;

  Mat() : rows(0), cols(0), data(0), type_(0) {}
  Mat(int r, int c, int type) : rows(0), cols(0), data(0), type_(0) { create(r, c, type); }
  void create(int r, int c, int type) {
    if (r == rows && c == cols && type == type_ && data) return;
    rows = r, cols = c, type_ = type;
    step.bytes = (size_t)c * elemSize();
    buf_.reset(new std::vector<uchar>((size_t)r * step.bytes));
    data = buf_->empty() ? 0 : &(*buf_)[0];
  }
  int type() const { return type_; }
  int depth() const { return type_ & ((1 << CV_CN_SHIFT) - 1); }
  int channels() const { return (type_ >> CV_CN_SHIFT) + 1; }
  size_t elemSize() const {
    const int d = depth();
    return (size_t)channels() * (d == CV_8U ? 1 : d == CV_16U ? 2 : 4);
  }
  Size size() const { return Size(cols, rows); }
  bool empty() const { return data == 0 || rows == 0 || cols == 0; }
  bool isContinuous() const { return true; }
  Mat clone() const {
    Mat m(rows, cols, type_);
    if (data) std::memcpy(m.data, data, (size_t)rows * step.bytes);
    return m;
  }
  template <class T>
  T *ptr(int y = 0) { return reinterpret_cast<T *>(data + (size_t)y * step.bytes); }
  template <class T>
  const T *ptr(int y = 0) const { return reinterpret_cast<const T *>(data + (size_t)y * step.bytes); }
  template <class T>
  T &at(int y, int x) { return ptr<T>(y)[x]; }
  template <class T>
  const T &at(int y, int x) const { return ptr<T>(y)[x]; }

 private:
  int type_;
  std::shared_ptr<std::vector<uchar> > buf_;  // copies share the pixels, like cv::Mat
};

}  // namespace cv
#endif
