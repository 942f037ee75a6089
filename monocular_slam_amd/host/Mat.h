// Mat.h -- the few OpenCV value types the reference's data model is written in (src/Frame.h, src/DataManager.h,
// src/MapPoint.h use cv::Mat, cv::Point2d/3d, cv::DMatch, cv::KeyPoint), re-stated minimally because OpenCV is
// not available.  Only the observable behaviour the stages rely on is provided: rows/cols/type, at<T>(r,c),
// empty(), clone(), zeros/eye, shallow copies that share storage (cv::Mat semantics).
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

namespace cv {

enum { CV_8U = 0, CV_32F = 5, CV_64F = 6 };
inline int CV_MAKETYPE(int depth, int cn) { return depth + ((cn - 1) << 3); }
inline int cv_depth(int type) { return type & 7; }
inline int cv_channels(int type) { return (type >> 3) + 1; }
inline int cv_elem_size1(int depth) { return depth == CV_8U ? 1 : (depth == CV_32F ? 4 : 8); }

template <typename T> struct Point_ { T x{}, y{}; Point_() = default; Point_(T x_, T y_) : x(x_), y(y_) {}
    template <typename U> Point_(const Point_<U>& o) : x((T)o.x), y((T)o.y) {} };
template <typename T> struct Point3_ { T x{}, y{}, z{}; Point3_() = default; Point3_(T x_, T y_, T z_) : x(x_), y(y_), z(z_) {} };
typedef Point_<float> Point2f;
typedef Point_<double> Point2d;
typedef Point3_<double> Point3d;

struct KeyPoint { Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0; };
struct DMatch { int queryIdx = -1, trainIdx = -1; float distance = 0; };

class Mat {
public:
    int rows = 0, cols = 0;
    Mat() = default;
    Mat(int r, int c, int type) { create(r, c, type); }
    void create(int r, int c, int type) {
        rows = r; cols = c; type_ = type; device.reset();
        step = (size_t)c * elemSize();
        buf_ = std::shared_ptr<uint8_t>(new uint8_t[step * (size_t)r + 8](), std::default_delete<uint8_t[]>());
        data = buf_.get();
    }
    static Mat zeros(int r, int c, int type) { return Mat(r, c, type); }
    static Mat eye(int r, int c, int type) { Mat m(r, c, type); for (int i = 0; i < r && i < c; i++) m.at<double>(i, i) = 1.0; return m; }
    int type() const { return type_; }
    int channels() const { return cv_channels(type_); }
    size_t elemSize() const { return (size_t)cv_elem_size1(cv_depth(type_)) * cv_channels(type_); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    Mat clone() const { Mat m; if (!empty()) { m.create(rows, cols, type_); std::memcpy(m.data, data, step * (size_t)rows); } return m; } // no device twin
    template <typename T> T& at(int r, int c) { return *reinterpret_cast<T*>(data + step * (size_t)r + sizeof(T) * (size_t)c); }
    template <typename T> const T& at(int r, int c) const { return *reinterpret_cast<const T*>(data + step * (size_t)r + sizeof(T) * (size_t)c); }
    template <typename T> T& at(int i) { return reinterpret_cast<T*>(data)[i]; }
    template <typename T> const T& at(int i) const { return reinterpret_cast<const T*>(data)[i]; }
    template <typename T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + step * (size_t)r); }
    template <typename T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + step * (size_t)r); }
    uint8_t* data = nullptr;
    size_t step = 0;
    // Optional device-resident twin of the contents (cv::UMat's role): an opaque owning handle that shallow copies share
    // and clone() drops.  FeatureExtractor attaches the frame's descriptors on the GPU here so that matchFeatures can
    // match without uploading them again (Runtime.h: DeviceFeatures).  The twin records a content hash of the host rows;
    // consumers verify it before trusting the device copy, so in-place edits through at<>/ptr<>/data are safe.
    std::shared_ptr<void> device;
private:
    int type_ = 0;
    std::shared_ptr<uint8_t> buf_;
};

} // namespace cv
