/* cv_compat.h — the handful of OpenCV types the VIDO_SLAM::System / Tracking / Optimizer API surface uses
 * (cv::Mat, cv::KeyPoint, cv::Point2f/3f, cv::Vec2f), for builds where OpenCV is absent (this image).
 * When <opencv2/core.hpp> is available the facade headers include it instead of this file
 * (SURVEY.md §8b).  Member names, type codes and element layout follow OpenCV so that callers written
 * against cv::Mat (vido_slam/demo/run_vido_slam.cc:112-137) compile unchanged for the calls they make. */
#ifndef VIDO_CV_COMPAT_H
#define VIDO_CV_COMPAT_H
#include <cstdint>
#include <cstring>
#include <cstddef>
#include <atomic>
#include <cstdlib>
#include <memory>
#include <new>
#include <vector>
#include <stdexcept>

#define CV_8U 0
#define CV_8S 1
#define CV_16U 2
#define CV_16S 3
#define CV_32S 4
#define CV_32F 5
#define CV_64F 6
#define CV_CN_SHIFT 3
#define CV_MAKETYPE(depth, cn) (((depth) & 7) + (((cn) - 1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_8UC4 CV_MAKETYPE(CV_8U, 4)
#define CV_16UC1 CV_MAKETYPE(CV_16U, 1)
#define CV_32SC1 CV_MAKETYPE(CV_32S, 1)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_32FC2 CV_MAKETYPE(CV_32F, 2)
#define CV_64FC1 CV_MAKETYPE(CV_64F, 1)

namespace cv {

typedef unsigned char uchar;

template <typename T> struct Point_ { T x, y; Point_() : x(0), y(0) {} Point_(T x_, T y_) : x(x_), y(y_) {} };
typedef Point_<float> Point2f; typedef Point_<int> Point2i; typedef Point2i Point;
template <typename T> struct Point3_ { T x, y, z; Point3_() : x(0), y(0), z(0) {} Point3_(T x_, T y_, T z_) : x(x_), y(y_), z(z_) {} };
typedef Point3_<float> Point3f;
struct Size { int width, height; Size() : width(0), height(0) {} Size(int w, int h) : width(w), height(h) {} };
template <typename T, int N> struct Vec { T val[N]; T& operator[](int i) { return val[i]; } const T& operator[](int i) const { return val[i]; } };
typedef Vec<float, 2> Vec2f;

struct KeyPoint {
    Point2f pt; float size, angle, response; int octave, class_id;
    KeyPoint() : size(0), angle(-1), response(0), octave(0), class_id(-1) {}
    KeyPoint(float x, float y, float size_, float angle_ = -1, float response_ = 0, int octave_ = 0, int class_id_ = -1)
        : pt(x, y), size(size_), angle(angle_), response(response_), octave(octave_), class_id(class_id_) {}
};

class Mat {
public:
    int rows, cols; uchar* data; size_t step;
    Mat() : rows(0), cols(0), data(nullptr), step(0), type_(0), buf_(nullptr) {}
    Mat(int r, int c, int type) : rows(0), cols(0), data(nullptr), step(0), type_(0), buf_(nullptr) { create(r, c, type); }
    Mat(int r, int c, int type, void* ext, size_t step_ = 0) : rows(r), cols(c), data((uchar*)ext), step(step_ ? step_ : (size_t)c * esz(type)), type_(type), buf_(nullptr) {}
    Mat(const Mat& o) : rows(o.rows), cols(o.cols), data(o.data), step(o.step), type_(o.type_), buf_(o.buf_) { if (buf_) buf_->fetch_add(1, std::memory_order_relaxed); }
    Mat(Mat&& o) noexcept : rows(o.rows), cols(o.cols), data(o.data), step(o.step), type_(o.type_), buf_(o.buf_) { o.buf_ = nullptr; o.data = nullptr; o.rows = o.cols = 0; }
    Mat& operator=(const Mat& o) { if (this != &o) { if (o.buf_) o.buf_->fetch_add(1, std::memory_order_relaxed); release(); rows = o.rows; cols = o.cols; data = o.data; step = o.step; type_ = o.type_; buf_ = o.buf_; } return *this; }
    Mat& operator=(Mat&& o) noexcept { if (this != &o) { release(); rows = o.rows; cols = o.cols; data = o.data; step = o.step; type_ = o.type_; buf_ = o.buf_; o.buf_ = nullptr; o.data = nullptr; o.rows = o.cols = 0; } return *this; }
    ~Mat() { release(); }
    /* header = shared view, like cv::Mat: one malloc holds the reference count and the zero-filled pixels */
    void create(int r, int c, int type)
    {
        release();
        rows = r; cols = c; type_ = type; step = (size_t)c * esz(type);
        void* blk = calloc(1, kHdr + (size_t)r * step + 16);
        if (!blk) throw std::bad_alloc();
        buf_ = new (blk) std::atomic<int>(1); data = (uchar*)blk + kHdr;
    }
    static Mat zeros(int r, int c, int type) { return Mat(r, c, type); }
    static Mat eye(int r, int c, int type)
    {
        Mat m(r, c, type);
        for (int i = 0; i < (r < c ? r : c); i++) { if (depth_of(type) == CV_32F) m.at<float>(i, i) = 1.f; else if (depth_of(type) == CV_64F) m.at<double>(i, i) = 1.0; else m.data[i * m.step + (size_t)i * esz(type)] = 1; }
        return m;
    }
    int type() const { return type_; }
    int depth() const { return depth_of(type_); }
    int channels() const { return (type_ >> CV_CN_SHIFT) + 1; }
    size_t elemSize() const { return esz(type_); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    bool isContinuous() const { return step == (size_t)cols * esz(type_); }
    size_t total() const { return (size_t)rows * cols; }
    template <typename T> T& at(int r, int c) { return *(T*)(data + (size_t)r * step + (size_t)c * sizeof(T)); }
    template <typename T> const T& at(int r, int c) const { return *(const T*)(data + (size_t)r * step + (size_t)c * sizeof(T)); }
    template <typename T> T& at(int i) { return rows == 1 ? at<T>(0, i) : at<T>(i, 0); }
    template <typename T> const T& at(int i) const { return rows == 1 ? at<T>(0, i) : at<T>(i, 0); }
    template <typename T> T* ptr(int r = 0) { return (T*)(data + (size_t)r * step); }
    template <typename T> const T* ptr(int r = 0) const { return (const T*)(data + (size_t)r * step); }
    Mat clone() const
    {
        Mat m; if (empty()) return m; m.create(rows, cols, type_);
        for (int r = 0; r < rows; r++) memcpy(m.data + (size_t)r * m.step, data + (size_t)r * step, (size_t)cols * esz(type_));
        return m;
    }
    void copyTo(Mat& dst) const { dst = clone(); }
    /* depth conversion with scale, the cases the drivers use (16U/8U/32S -> 32F, 8U/16U -> 32S) */
    void convertTo(Mat& dst, int rtype, double alpha = 1.0) const
    {
        Mat out(rows, cols, CV_MAKETYPE(rtype & 7, channels()));
        const int n = cols * channels();
        for (int r = 0; r < rows; r++) for (int c = 0; c < n; c++) {
            double v;
            switch (depth()) { case CV_8U: v = ptr<uchar>(r)[c]; break; case CV_16U: v = ptr<uint16_t>(r)[c]; break; case CV_32S: v = ptr<int32_t>(r)[c]; break;
                               case CV_32F: v = ptr<float>(r)[c]; break; case CV_64F: v = ptr<double>(r)[c]; break; default: throw std::runtime_error("cv_compat: convertTo source depth"); }
            v *= alpha;
            switch (out.depth()) { case CV_32F: out.ptr<float>(r)[c] = (float)v; break; case CV_32S: out.ptr<int32_t>(r)[c] = (int32_t)(v < 0 ? v - 0.5 : v + 0.5); break;
                                   case CV_64F: out.ptr<double>(r)[c] = v; break; case CV_8U: out.ptr<uchar>(r)[c] = (uchar)(v < 0 ? 0 : v > 255 ? 255 : v + 0.5); break;
                                   default: throw std::runtime_error("cv_compat: convertTo target depth"); }
        }
        dst = out;
    }
    Mat t() const
    {
        Mat m(cols, rows, type_);
        for (int r = 0; r < rows; r++) for (int c = 0; c < cols; c++) memcpy(m.data + (size_t)c * m.step + (size_t)r * esz(type_), data + (size_t)r * step + (size_t)c * esz(type_), esz(type_));
        return m;
    }
    /* Build extension (not an OpenCV call): n 3 x 1 CV_32F matrices — the per-point world coordinates of Frame / Map (vp3DPointSta ...) — as views into ONE reference-counted
     * block instead of n allocations; every element behaves like a Mat(3, 1, CV_32F) of its own (the block goes when the last view goes). */
    static void batch3x1(const float* xyz, int n, std::vector<Mat>& out)
    {
        out.clear(); if (n <= 0) return;
        void* blk = malloc(kHdr + (size_t)n * 12 + 16);
        if (!blk) throw std::bad_alloc();
        std::atomic<int>* rc = new (blk) std::atomic<int>(n);
        memcpy((uchar*)blk + kHdr, xyz, (size_t)n * 12);
        out.resize((size_t)n);
        for (int i = 0; i < n; i++) { Mat& m = out[(size_t)i]; m.rows = 3; m.cols = 1; m.step = 4; m.type_ = CV_32F; m.buf_ = rc; m.data = (uchar*)blk + kHdr + (size_t)i * 12; }
    }
    static size_t esz(int type) { static const size_t d[8] = {1, 1, 2, 2, 4, 4, 8, 2}; return d[type & 7] * ((type >> CV_CN_SHIFT) + 1); }
    static int depth_of(int type) { return type & 7; }
private:
    static constexpr size_t kHdr = 16;
    void release() { if (buf_ && buf_->fetch_sub(1, std::memory_order_acq_rel) == 1) free((void*)buf_); buf_ = nullptr; }
    int type_;
    std::atomic<int>* buf_;
};

/* cv::InputArray / cv::OutputArray as OpenCV declares them (typedefs of const references to proxy classes), reduced to the Mat case: what
 * ORBextractor::operator() takes in the reference (vido_slam/include/ORBextractor.h:49), so that the facade's signature is the same in both builds. */
class _InputArray {
public:
    _InputArray() : m_(nullptr) {}
    _InputArray(const Mat& m) : m_(const_cast<Mat*>(&m)) {}
    Mat getMat(int = -1) const { return m_ ? *m_ : Mat(); }
    bool empty() const { return !m_ || m_->empty(); }
protected:
    Mat* m_;
};
class _OutputArray : public _InputArray {
public:
    _OutputArray() {}
    _OutputArray(Mat& m) : _InputArray(m) {}
    void create(int rows, int cols, int type) const { if (m_) m_->create(rows, cols, type); }
    Mat& getMatRef(int = -1) const { return *m_; }
    bool needed() const { return m_ != nullptr; }
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
inline InputArray noArray() { static const _InputArray none; return none; }

/* CV_32F / CV_64F matrix product; like cv::gemm, float products accumulate in double */
inline Mat operator*(const Mat& a, const Mat& b)
{
    if (a.cols != b.rows || a.type() != b.type()) throw std::runtime_error("cv_compat: operator* shape/type mismatch");
    Mat m(a.rows, b.cols, a.type());
    for (int r = 0; r < a.rows; r++) for (int c = 0; c < b.cols; c++) {
        double s = 0;
        for (int k = 0; k < a.cols; k++) s += a.depth() == CV_32F ? (double)a.at<float>(r, k) * b.at<float>(k, c) : a.at<double>(r, k) * b.at<double>(k, c);
        if (a.depth() == CV_32F) m.at<float>(r, c) = (float)s; else m.at<double>(r, c) = s;
    }
    return m;
}

}  // namespace cv
#endif
