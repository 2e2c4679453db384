// cvlite.h -- the few OpenCV types the reference's PatchMatch / GyroAidedTracker interface is
// written against, for hosts without OpenCV (this image has none).
// When <opencv2/core.hpp> is available, include it instead and this header steps aside.
#pragma once
#if __has_include(<opencv2/core.hpp>)
#include <opencv2/core.hpp>
#else
#include <cmath>
#include <cstddef>
#include <cstring>
#include <memory>
#include <vector>

namespace cv {
typedef unsigned char uchar;

struct Point2f {
    float x = 0, y = 0;
    Point2f() {}
    Point2f(float x_, float y_) : x(x_), y(y_) {}
};
inline Point2f operator+(const Point2f &a, const Point2f &b) { return Point2f(a.x + b.x, a.y + b.y); }
inline Point2f operator-(const Point2f &a, const Point2f &b) { return Point2f(a.x - b.x, a.y - b.y); }
struct Point3f {
    float x = 0, y = 0, z = 0;
    Point3f() {}
    Point3f(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};
inline Point3f operator+(const Point3f &a, const Point3f &b) { return Point3f(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Point3f operator-(const Point3f &a, const Point3f &b) { return Point3f(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Point3f operator*(const Point3f &a, float s) { return Point3f(a.x * s, a.y * s, a.z * s); }

struct KeyPoint {
    Point2f pt;
    float size = 1, angle = -1, response = 0;
    int octave = 0, class_id = -1;
    KeyPoint() {}
    KeyPoint(float x, float y, float size_ = 1) : pt(x, y), size(size_) {}
};

enum { CV_8U = 0, CV_32F = 5, CV_8UC1 = 0, CV_32FC1 = 5 };

// Dense 2-D matrix header: 8UC1 images and small CV_32F matrices only.
class Mat {
public:
    int rows = 0, cols = 0;
    size_t step = 0;
    uchar *data = nullptr;

    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    // header over caller-owned memory (like cv::Mat(rows, cols, type, data, step))
    Mat(int r, int c, int type, void *d, size_t step_ = 0)
        : rows(r), cols(c), step(step_ ? step_ : (size_t)c * esz(type)), data(static_cast<uchar *>(d)), type_(type) {}
    void create(int r, int c, int type)
    {
        rows = r, cols = c, type_ = type, step = (size_t)c * esz(type);
        owner_.reset(new uchar[step * (size_t)r + 16](), std::default_delete<uchar[]>());
        data = owner_.get();
    }
    static Mat zeros(int r, int c, int type) { return Mat(r, c, type); }
    static Mat eye(int r, int c, int type)
    {
        Mat m(r, c, type);
        for (int i = 0; i < (r < c ? r : c); i++) m.at<float>(i, i) = 1.0f;
        return m;
    }
    int type() const { return type_; }
    bool empty() const { return data == nullptr || rows * cols == 0; }
    size_t total() const { return (size_t)rows * cols; }
    template <typename T> T &at(int r, int c) { return *reinterpret_cast<T *>(data + step * r + sizeof(T) * c); }
    template <typename T> const T &at(int r, int c) const { return *reinterpret_cast<const T *>(data + step * r + sizeof(T) * c); }
    template <typename T> T &at(int i) { return rows == 1 ? at<T>(0, i) : at<T>(i, 0); }
    template <typename T> const T &at(int i) const { return rows == 1 ? at<T>(0, i) : at<T>(i, 0); }
    Mat clone() const
    {
        Mat m(rows, cols, type_);
        for (int r = 0; r < rows; r++) std::memcpy(m.data + m.step * r, data + step * r, (size_t)cols * esz(type_));
        return m;
    }
    Mat t() const
    {
        Mat m(cols, rows, CV_32F);
        for (int r = 0; r < rows; r++)
            for (int c = 0; c < cols; c++) m.at<float>(c, r) = at<float>(r, c);
        return m;
    }

private:
    static size_t esz(int type) { return type == CV_32F ? 4 : 1; }
    int type_ = CV_8UC1;
    std::shared_ptr<uchar> owner_;
};

// CV_32F product, accumulated in double per output element like OpenCV's small-matrix gemm
inline Mat operator*(const Mat &a, const Mat &b)
{
    Mat m(a.rows, b.cols, CV_32F);
    for (int r = 0; r < a.rows; r++)
        for (int c = 0; c < b.cols; c++) {
            double s = 0;
            for (int k = 0; k < a.cols; k++) s += (double)a.at<float>(r, k) * (double)b.at<float>(k, c);
            m.at<float>(r, c) = (float)s;
        }
    return m;
}
}  // namespace cv
#endif
