// tests/cv_stub/opencv2/opencv.hpp — DECLARATIONS ONLY, for a -fsyntax-only compile of include/mvo_shim.hpp where OpenCV
// is not installed.  It pins nothing about OpenCV's behaviour (parity with real OpenCV stays unpinned); it only
// keeps the shim's use of the cv:: types it names type-correct.  Layouts of KeyPoint / DMatch follow OpenCV 4.x because
// the shim static_asserts them against mvo_keypoint / mvo_match.
#pragma once
#include <string>
#include <vector>
typedef unsigned char uchar;
#define CV_8U 0
#define CV_32S 4
#define CV_64F 6
#define CV_8UC1 0
#define CV_8UC3 16
#define CV_Assert(expr) do { if (!(expr)) cv::error(0, #expr, "", __FILE__, __LINE__); } while (0)
#define CV_Error(code, msg) cv::error(code, msg, "", __FILE__, __LINE__)
namespace cv {
namespace Error { enum Code { StsError = -2 }; }
[[noreturn]] void error(int code, const std::string& msg, const char* func, const char* file, int line);
struct Point2f { float x, y; };
struct Point3f { float x, y, z; };
struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; };
struct DMatch { int queryIdx, trainIdx, imgIdx; float distance; };
class Mat {
 public:
  Mat();
  Mat(int rows, int cols, int type);
  int rows, cols;
  uchar* data;
  size_t step;
  int type() const;
  int channels() const;
  void create(int rows, int cols, int type);
  Mat rowRange(int a, int b) const;
  Mat clone() const;
  Mat reshape(int cn, int rows = 0) const;
  void convertTo(Mat& m, int rtype) const;
  template <class T> T* ptr(int row = 0);
  template <class T> const T* ptr(int row = 0) const;
  template <class T> T& at(int i);
  template <class T> const T& at(int i) const;
};
}  // namespace cv
