// facade.cpp — host orchestration behind the reference's C++ class surface (include/vido_slam/vido_slam.h).
// Follows, function by function, vido_slam/src/{System,Tracking,Frame,Optimizer,Converter}.cc of the reference; every
// data-parallel stage is a call into the C-ABI (GPU), everything here is bookkeeping on a few thousand points.
// Not reproduced (SURVEY.md §2 out of scope): viewer / imshow / plots, ground-truth metrics, IMU paths.  The addnoise = 1 branches (time(NULL)-seeded cv::RNG draw on the
// depth) are vido_depth_noise (trackhost.cpp); detail::SetDepthNoiseSeed pins the seed.
#include <ctime>
#include "../../include/vido_slam/vido_slam.h"
#include <algorithm>
#include <map>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <future>
#include <mutex>
#include <iostream>
#include <sstream>
#include <stdexcept>

namespace VIDO_SLAM {

// ---------------------------------------------------------------------------------------------------------------
namespace detail {
static vido_ctx* g_ctx = nullptr;
static int g_slot = 0;                 // device slot holding the maps of the frame under construction
static bool g_zero_copy_maps = false;  // GrabImageRGBDDevice: the slots adopt the caller's device buffers (vido_system_set_zero_copy_maps)
void SetZeroCopyMaps(bool on) { g_zero_copy_maps = on; }
static vido_track_params g_tp;
vido_ctx* Context() { return g_ctx; }
// The local window (PartialBatchOptimization) has a context of its own — own stream, own resident ring, own scratch — so that the window solve of frame k can run on a helper
// thread BESIDE the tracking of frame k + 1 (round 5; Tracking::Track): nothing the tracker reads depends on it (the reference's tracker never reads the refined map either:
// Tracking.cc only appends to vmCameraPose, Optimizer.cc:1084-1128 rewrites it), the next window solve and every reader of the Map wait for it (finish_local_ba).
static vido_ctx* g_ctx_ba = nullptr;
static int g_w = 0, g_h = 0;
struct LocalBaJob { std::future<float> fut; bool pending = false; Map* map = nullptr; };
static LocalBaJob g_lba;
static void finish_local_ba();

std::map<std::string, std::string> ParseSettings(const std::string& path)
{
    std::ifstream f(path.c_str());
    if (!f.is_open()) throw std::runtime_error("Failed to open settings file at: " + path);
    std::map<std::string, std::string> kv; std::string line;
    while (std::getline(f, line)) {
        const size_t h = line.find('#'); if (h != std::string::npos) line = line.substr(0, h);
        if (line.empty() || line[0] == '%' || line.compare(0, 3, "---") == 0) continue;
        const size_t c = line.find(':'); if (c == std::string::npos) continue;
        std::string k = line.substr(0, c), v = line.substr(c + 1);
        auto trim = [](std::string& s) { const char* ws = " \t\r\n\""; s.erase(0, s.find_first_not_of(ws)); const size_t e = s.find_last_not_of(ws); if (e == std::string::npos) s.clear(); else s.erase(e + 1); };
        trim(k); trim(v);
        if (!k.empty() && k.find(' ') == std::string::npos) kv[k] = v;
    }
    return kv;
}
static double num(const std::map<std::string, std::string>& kv, const char* key, double def = 0.0)
{
    auto it = kv.find(key); if (it == kv.end() || it->second.empty()) return def;
    return atof(it->second.c_str());
}
// A failed C-ABI call surfaces as an exception that still carries the VIDO_E_* code, so that the C handle (vido_system_*) can hand the same code back to its caller
// instead of guessing it from the message text.
struct VidoFailure : std::runtime_error {
    int code;
    VidoFailure(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
static void check_rc(int rc, const char* what)
{
    if (rc < 0) throw VidoFailure(rc, std::string(what) + ": " + vido_last_error(g_ctx));
}
// VIDO_CALL_PROF=1: wall time of every C-ABI call the facade makes, by name, printed when the process ends — where a tracked frame's host time goes (inside the pipeline
// the same calls take several times what they take on an idle GPU, and which ones is the question)
namespace {
struct CallProf {
    bool on = getenv("VIDO_CALL_PROF") != nullptr; std::map<std::string, std::pair<double, long> > acc;
    ~CallProf() { if (!on) return; std::vector<std::pair<double, std::string> > v; for (auto& kv : acc) v.push_back({kv.second.first, kv.first});
                  std::sort(v.rbegin(), v.rend());
                  for (auto& e : v) fprintf(stderr, "[call prof] %-44s %8.3f ms total %7ld calls %8.3f ms each\n", e.second.c_str(), e.first, acc[e.second].second, e.first / std::max(1L, acc[e.second].second)); }
};
CallProf g_prof;
std::mutex g_prof_mu;                  // (the local-BA helper thread times its calls too)
template <class F> inline int timed_call(F&& f, const char* what)
{
    if (!g_prof.on) return f();
    const auto t0 = std::chrono::steady_clock::now(); const int rc = f();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::lock_guard<std::mutex> g(g_prof_mu);
    auto& a = g_prof.acc[what]; a.first += ms; a.second++;
    return rc;
}
}
struct ProfSection {                  // VIDO_CALL_PROF=1: a named host-side section of the facade in the same table as the C-ABI calls
    const char* name; std::chrono::steady_clock::time_point t0;
    explicit ProfSection(const char* n) : name(n) { if (g_prof.on) t0 = std::chrono::steady_clock::now(); }
    ~ProfSection() { if (!g_prof.on) return; const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                     std::lock_guard<std::mutex> g(g_prof_mu); auto& a = g_prof.acc[name]; a.first += ms; a.second++; }
};
#define check(expr, what) check_rc(timed_call([&]() -> int { return (expr); }, (what)), (what))
static inline double lba_now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static const bool g_lba_trace = getenv("VIDO_LBA_TRACE") != nullptr;      // diagnosis: host timestamps of the window solve's steps and of the tracker's frame boundaries
#define LBA_TRACE(what) do { if (g_lba_trace) fprintf(stderr, "[lba trace] %12.3f %s\n", lba_now_ms(), (what)); } while (0)
static void check_rc_ba(int rc, const char* what) { if (rc < 0) throw VidoFailure(rc, std::string(what) + ": " + vido_last_error(g_ctx_ba)); }
#define check_ba(expr, what) check_rc_ba(timed_call([&]() -> int { return (expr); }, (what)), (what))
int failure_code(const std::exception& e)
{
    if (const VidoFailure* v = dynamic_cast<const VidoFailure*>(&e)) return v->code;
    if (dynamic_cast<const std::invalid_argument*>(&e)) return VIDO_E_INVALID;
    if (dynamic_cast<const std::bad_alloc*>(&e)) return VIDO_E_NOMEM;
    return VIDO_E_INVALID;                                    // the facade's own throws are argument / state errors (wrong sensor, wrong image type, no live System)
}
// The static Optimizer:: methods and the Frame constructor of the reference's interface carry no context argument, so the facade keeps ONE process-wide context (set by
// the tracker's extractor / GrabImageRGBD).  One live System per process, like the reference (Frame's static members, Frame.cc:26-30); calls before the first frame or
// after the System is gone fail with an exception instead of dereferencing a stale pointer.
static vido_ctx* live_ctx(const char* what)
{
    if (!g_ctx) throw std::runtime_error(std::string(what) + ": no live VIDO_SLAM::System (the facade supports one System per process; create it and grab a frame first)");
    return g_ctx;
}
// the local / global BA context (see g_ctx_ba above): created with the tracker's first BA call, on the tracker's device; its extractor state is the smallest the library builds
static vido_ctx* ba_ctx(const char* what)
{
    live_ctx(what);
    if (!g_ctx_ba) {
        vido_config cfg; vido_config_default(&cfg);
        cfg.width = g_w > 0 ? g_w : 640; cfg.height = g_h > 0 ? g_h : 480; cfg.n_levels = 1; cfg.n_features = 64; cfg.max_batch = 1;
        if (const char* d = getenv("VIDO_DEVICE")) cfg.device = atoi(d);
        const char* bp = getenv("VIDO_BA_CTX_PRIO");              // (experiment: the BA context's stream priority / queue slot, see ctx.cpp VIDO_CTX_PRIO)
        if (bp) setenv("VIDO_CTX_PRIO", bp, 1);
        const int crc = vido_create(&cfg, &g_ctx_ba);
        if (bp) unsetenv("VIDO_CTX_PRIO");
        if (crc != VIDO_OK) { g_ctx_ba = nullptr; throw std::runtime_error(std::string(what) + ": vido_create (BA context): " + vido_last_error(nullptr)); }
    }
    return g_ctx_ba;
}
static void toRow16(const cv::Mat& T, double* o) { for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) o[r * 4 + c] = T.at<float>(r, c); }
static cv::Mat fromRow16(const double* o) { cv::Mat T(4, 4, CV_32F); for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) T.at<float>(r, c) = (float)o[r * 4 + c]; return T; }
static cv::Mat vec3(float x, float y, float z) { cv::Mat m(3, 1, CV_32F); m.at<float>(0) = x; m.at<float>(1) = y; m.at<float>(2) = z; return m; }
}  // namespace detail
using namespace detail;

cv::Mat Converter::toInvMatrix(const cv::Mat& T)      // Converter.cc:155-170
{
    cv::Mat Ti = cv::Mat::eye(4, 4, CV_32F);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Ti.at<float>(r, c) = T.at<float>(c, r);
    for (int r = 0; r < 3; r++) { double s = 0; for (int c = 0; c < 3; c++) s += (double)(-Ti.at<float>(r, c)) * T.at<float>(c, 3); Ti.at<float>(r, 3) = (float)s; }
    return Ti;
}

// ---- ORBextractor -------------------------------------------------------------------------------------------------
ORBextractor::ORBextractor(int nf, float sf, int nl, int ini, int mn) : nfeatures(nf), scaleFactor(sf), nlevels(nl), iniThFAST(ini), minThFAST(mn)
{
    mvScaleFactor.resize(nl); mvScaleFactor[0] = 1.0f;
    for (int i = 1; i < nl; i++) mvScaleFactor[i] = mvScaleFactor[i - 1] * sf;
}
ORBextractor::~ORBextractor()
{
    if (!ctx_) return;
    if (g_ctx == ctx_) {
        try { finish_local_ba(); } catch (...) { }              // (a window solve still in flight belongs to the System that is going away)
        g_ctx = nullptr;
        if (g_ctx_ba) { vido_destroy(g_ctx_ba); g_ctx_ba = nullptr; }
    }
    vido_destroy(ctx_);
}
vido_ctx* ORBextractor::context(int width, int height)
{
    if (ctx_ && (width != w_ || height != h_)) throw std::runtime_error("ORBextractor: image size changed after the first frame");
    if (!ctx_) {
        vido_config cfg; vido_config_default(&cfg);
        cfg.width = width; cfg.height = height; cfg.n_features = nfeatures; cfg.scale_factor = scaleFactor; cfg.n_levels = nlevels;
        cfg.ini_th_fast = iniThFAST; cfg.min_th_fast = minThFAST; cfg.max_batch = 1;
        if (const char* d = getenv("VIDO_DEVICE")) cfg.device = atoi(d);
        if (vido_create(&cfg, &ctx_) != VIDO_OK) throw std::runtime_error(std::string("vido_create: ") + vido_last_error(nullptr));
        w_ = width; h_ = height;
        if (!g_ctx) { g_ctx = ctx_; g_w = width; g_h = height; }
    }
    return ctx_;
}
void ORBextractor::PrefetchDevice(const void* dev_pixels, int channels, bool rgb_order, int width, int height, void* ready_event)
{
    vido_ctx* c = context(width, height);
    if (channels != 3 && channels != 4) return;              // (gray device images take the batch entry: no prefetch form)
    if (vido_orb_prefetch_color(c, (const uint8_t*)dev_pixels, channels, rgb_order ? 1 : 0, width * channels, width, height, ready_event) != VIDO_OK) throw std::runtime_error(vido_last_error(c));
}

void ORBextractor::operator()(cv::InputArray image_, cv::InputArray, std::vector<cv::KeyPoint>& keypoints, cv::OutputArray descriptors_)
{
    keypoints.clear();
    if (image_.empty()) return;
    const cv::Mat image = image_.getMat();                     // (a header: shares the caller's pixels)
    if (image.type() != CV_8UC1) throw std::runtime_error("ORBextractor: image must be CV_8UC1");
    vido_ctx* c = context(image.cols, image.rows);
    const int cap = 2 * nfeatures + 256; int n = 0;
    std::vector<vido_keypoint> k(cap); cv::Mat desc(cap, 32, CV_8U);
    int rc;
    if (dev_src_) {                                            // device-resident image (SetDeviceSource): colour -> fused cvtColor ingest, gray -> plain ingest; no host pixels involved
        if (dev_ch_ == 1) rc = vido_orb_extract_batch(c, (const uint8_t*)dev_src_, 1, 1, (size_t)image.cols * image.rows, image.cols, image.cols, image.rows, k.data(), cap, &n, desc.data);
        else rc = timed_call([&]() -> int { return vido_orb_extract_color(c, (const uint8_t*)dev_src_, dev_ch_, rgb_ ? 1 : 0, 1, 1, 0, image.cols * dev_ch_, image.cols, image.rows, nullptr, k.data(), cap, &n, desc.data); }, "orb_extract_color(device)");
        dev_src_ = nullptr;
    }
    else if (color_ && gray_ == image.data && color_->cols == image.cols && color_->rows == image.rows && image.isContinuous())      // cvtColor on the device (SetColorSource)
        rc = vido_orb_extract_color(c, color_->data, color_->channels(), rgb_ ? 1 : 0, 0, 1, 0, (int)color_->step, image.cols, image.rows, image.data, k.data(), cap, &n, desc.data);
    else rc = vido_orb_extract(c, image.data, (int)image.step, image.cols, image.rows, k.data(), cap, &n, desc.data);
    color_ = nullptr; gray_ = nullptr;
    if (rc != VIDO_OK) throw std::runtime_error(vido_last_error(c));
    keypoints.resize(n);
    for (int i = 0; i < n; i++) keypoints[i] = cv::KeyPoint(k[i].x, k[i].y, k[i].size, k[i].angle, k[i].response, k[i].octave);
    descriptors_.create(n, 32, CV_8U);
    cv::Mat descriptors = descriptors_.getMat();
    for (int i = 0; i < n; i++) memcpy(descriptors.ptr<uint8_t>(i), desc.ptr<uint8_t>(i), 32);
}

// ---- Frame ----------------------------------------------------------------------------------------------------------
long unsigned int Frame::nNextId = 0;
static float g_ms_orb = 0, g_ms_lists = 0;      // wall time of the last Frame's extractor call / list stage (vido_system_stats.ms_orb / ms_lists)

Frame::Frame(const cv::Mat& imGray, const cv::Mat& imDepth, const cv::Mat& imFlow, const cv::Mat& maskSEM, const double& timeStamp,
             ORBextractor* extractor, cv::Mat& K, cv::Mat& distCoef, const float& bf, const float& thDepth, const float& thDepthObj, const int& UseSampleFea)
{
    (void)imDepth; (void)imFlow; (void)maskSEM;            // their (patched) copies live in the device slot uploaded by GrabImageRGBD
    mnId = nNextId++; mTimeStamp = timeStamp; mK = K.clone(); mDistCoef = distCoef.clone(); mbf = bf; mThDepth = thDepth; mThDepthObj = thDepthObj;
    fx = K.at<float>(0, 0); fy = K.at<float>(1, 1); cx = K.at<float>(0, 2); cy = K.at<float>(1, 2); invfx = 1.0f / fx; invfy = 1.0f / fy;
    const auto t_orb = std::chrono::steady_clock::now();
    (*extractor)(imGray, cv::Mat(), mvKeys, mDescriptors);                         // Frame.cc:62 ExtractORB
    g_ms_orb = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_orb).count(); g_ms_lists = 0;
    N = (int)mvKeys.size();
    if (mvKeys.empty()) return;
    vido_ctx* c = extractor->context(imGray.cols, imGray.rows);
    // candidates of the static list: the ORB keypoints (Option I, Frame.cc:72-100) or random grid samples (Option II, UseSampleFeature = 1, Frame.cc:101-150) — the same
    // mask / depth / flow filter runs on either (k_static_filter); Option II additionally wants the correspondence inside the image on the low side (:142)
    const std::vector<cv::KeyPoint> sampled = UseSampleFea != 0 ? SampleKeyPoints(imGray.rows, imGray.cols) : std::vector<cv::KeyPoint>();
    const std::vector<cv::KeyPoint>& cand = UseSampleFea != 0 ? sampled : mvKeys;
    const int NC = (int)cand.size();
    const int max_kp = std::max(2 * extractor->nfeatures + 256, NC + 256), max_obj = ((imGray.cols + 3) / 4) * ((imGray.rows + 3) / 4);
    std::vector<vido_keypoint> k(NC);
    for (int i = 0; i < NC; i++) { k[i].x = cand[i].pt.x; k[i].y = cand[i].pt.y; k[i].size = cand[i].size; k[i].angle = cand[i].angle; k[i].response = cand[i].response; k[i].octave = cand[i].octave; }
    k.resize(max_kp);
    std::vector<int32_t> sidx(max_kp), olab(max_obj); std::vector<float> scorr(2 * max_kp), sflow(2 * max_kp), sdep(max_kp), okeys(2 * max_obj), ocorr(2 * max_obj), odep(max_obj), oflow(2 * max_obj);
    int32_t ns = 0, no = 0, nk = NC;
    vido_frame_lists L; L.max_stat = max_kp; L.max_obj = max_obj; L.n_stat = &ns; L.stat_idx = sidx.data(); L.stat_corr = scorr.data(); L.stat_flow = sflow.data(); L.stat_depth = sdep.data();
    L.n_obj = &no; L.obj_keys = okeys.data(); L.obj_corr = ocorr.data(); L.obj_depth = odep.data(); L.obj_label = olab.data(); L.obj_flow = oflow.data();
    vido_track_params tp = g_tp; tp.th_depth_bg = thDepth; tp.th_depth_obj = thDepthObj;
    const auto t_lists = std::chrono::steady_clock::now();
    if (timed_call([&]() -> int { return vido_frame_features(c, g_slot, 1, k.data(), &nk, max_kp, &tp, &L); }, "frame_features") != VIDO_OK) throw std::runtime_error(vido_last_error(c));
    g_ms_lists = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_lists).count();
    for (int i = 0; i < ns; i++) {                                                 // Frame.cc:72-100, 165-177
        const cv::KeyPoint& kp = cand[sidx[i]];
        if (UseSampleFea != 0 && !(scorr[2 * i] > 0 && scorr[2 * i + 1] > 0)) continue;
        mvStatKeysTmp.push_back(kp);
        mvCorres.push_back(cv::KeyPoint(scorr[2 * i], scorr[2 * i + 1], 0, 0, 0, kp.octave, -1));
        mvFlowNext.push_back(cv::Point2f(sflow[2 * i], sflow[2 * i + 1]));
        mvStatDepthTmp.push_back(sdep[i]);
    }
    N_s_tmp = (int)mvStatKeysTmp.size();
    for (int i = 0; i < no; i++) {                                                 // Frame.cc:184-211
        mvObjFlowNext.push_back(cv::Point2f(oflow[2 * i], oflow[2 * i + 1]));
        mvObjCorres.push_back(cv::KeyPoint(ocorr[2 * i], ocorr[2 * i + 1], 0, 0, 0, -1));
        mvObjKeys.push_back(cv::KeyPoint(okeys[2 * i], okeys[2 * i + 1], 0, 0, 0, -1));
        mvObjDepth.push_back(odep[i]); vSemObjLabel.push_back(olab[i]);
    }
    // UndistortKeyPoints (Frame.cc:603-633): cv::undistortPoints(P = K), on the flat host function the parity tests pin (trackhost.cpp)
    mvKeysUn = mvKeys;
    if (mDistCoef.at<float>(0) != 0.0f && N > 0) {
        const float Kf[4] = {fx, fy, cx, cy};
        const float dist[5] = {mDistCoef.at<float>(0), mDistCoef.at<float>(1), mDistCoef.at<float>(2), mDistCoef.at<float>(3), mDistCoef.rows > 4 ? mDistCoef.at<float>(4) : 0.f};
        std::vector<float> in(2 * N), out(2 * N);
        for (int i = 0; i < N; i++) { in[2 * i] = mvKeys[i].pt.x; in[2 * i + 1] = mvKeys[i].pt.y; }
        if (vido_undistort_points(in.data(), N, Kf, dist, out.data()) != VIDO_OK) throw std::runtime_error("UndistortKeyPoints failed");
        for (int i = 0; i < N; i++) { mvKeysUn[i].pt.x = out[2 * i]; mvKeysUn[i].pt.y = out[2 * i + 1]; }
    }
}

std::vector<cv::KeyPoint> Frame::SampleKeyPoints(const int& rows, const int& cols)      // Frame.cc:888-956
{
    const int N_samp = 3000, n_div = 20, x_step = cols / n_div, y_step = rows / n_div;
    std::vector<std::vector<cv::KeyPoint>> grid((size_t)n_div * n_div);
    uint64_t ctr = 0;
    auto next = [&]() {                                      // splitmix64 of (frame id, counter): the reference's generator is cv::RNG(time(NULL))
        uint64_t z = ((uint64_t)mnId << 32) + 0x9e3779b97f4a7c15ull * (++ctr);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); };
    auto uniform = [&](int a, int b) { return b > a ? a + (int)(next() % (uint64_t)(b - a)) : a; };      // cv::RNG::uniform(int, int): an integer in [a, b)
    int key_num = 0;
    if (x_step < 1 || y_step < 1) return {};
    while (key_num < N_samp) {
        const int before = key_num;
        for (int i = 0; i < n_div && key_num < N_samp; ++i)
            for (int j = 0; j < n_div && key_num < N_samp; ++j) {
                const float x = (float)uniform(i * x_step, (i + 1) * x_step), y = (float)uniform(j * y_step, (j + 1) * y_step);
                if (x >= cols || y >= rows || x <= 0 || y <= 0) continue;
                grid[(size_t)i * n_div + j].push_back(cv::KeyPoint(x, y, 0, 0, 0, -1));
                key_num++;
            }
        if (key_num == before) break;                        // (degenerate image sizes: nothing can be sampled)
    }
    std::vector<cv::KeyPoint> out; out.reserve(key_num);
    for (const auto& cell : grid) out.insert(out.end(), cell.begin(), cell.end());
    return out;
}

void Frame::SetPose(cv::Mat Tcw)                          // Frame.cc SetPose / UpdatePoseMatrices
{
    mTcw = Tcw.clone();
    mRcw = cv::Mat(3, 3, CV_32F); mtcw = cv::Mat(3, 1, CV_32F);
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) mRcw.at<float>(r, c) = mTcw.at<float>(r, c); mtcw.at<float>(r) = mTcw.at<float>(r, 3); }
    mRwc = mRcw.t(); mOw = cv::Mat(3, 1, CV_32F);
    for (int r = 0; r < 3; r++) { double s = 0; for (int c = 0; c < 3; c++) s += (double)(-mRwc.at<float>(r, c)) * mtcw.at<float>(c); mOw.at<float>(r) = (float)s; }
}
// Raw forms of Frame::UnprojectStereo* / ObtainFlowDepth* for the facade's own loops: the same float / double arithmetic without a cv::Mat per point.  The per-frame loops
// made ~40 000 of those (a calloc + free each) and inverted the frame's pose once per POINT (Converter::toInvMatrix inside the unprojection): most of the ~2 ms of host time
// between the tracker's kernels.  Twl = [mRwc | mOw] is what Frame::SetPose keeps (the same expression as Converter::toInvMatrix, entry by entry).
static bool unproject_raw(const Frame& F, float u, float v, float z, float* out)
{
    const float x = (u - F.cx) * z * F.invfx, y = (v - F.cy) * z * F.invfy;
    if (F.mRwc.empty() || F.mOw.empty()) {                    // pose never set through SetPose: the general form
        const cv::Mat Twl = Converter::toInvMatrix(F.mTcw);
        for (int r = 0; r < 3; r++) { const double s = (double)Twl.at<float>(r, 0) * x + (double)Twl.at<float>(r, 1) * y + (double)Twl.at<float>(r, 2) * z; out[r] = (float)s + Twl.at<float>(r, 3); }
        return true;
    }
    const float* R = F.mRwc.ptr<float>(); const float* O = F.mOw.ptr<float>();
    for (int r = 0; r < 3; r++) { const double s = (double)R[r * 3] * x + (double)R[r * 3 + 1] * y + (double)R[r * 3 + 2] * z; out[r] = (float)s + O[r]; }
    return true;
}
static inline bool stat3d_raw(const Frame& F, int i, float* out) { const float z = F.mvStatDepth[i]; if (z < 0) return false; return unproject_raw(F, F.mvStatKeys[i].pt.x, F.mvStatKeys[i].pt.y, z, out); }
static inline bool obj3d_raw(const Frame& F, int i, float* out) { const float z = F.mvObjDepth[i]; if (!(z > 0)) return false; return unproject_raw(F, F.mvObjKeys[i].pt.x, F.mvObjKeys[i].pt.y, z, out); }
static cv::Mat unproject(const Frame& F, float u, float v, float z) { float o[3]; unproject_raw(F, u, v, z, o); return vec3(o[0], o[1], o[2]); }
// addnoise = 1 (Frame.cc:711-716, 744-750, 838-845, 865-872): one draw of a fresh cv::RNG((unsigned)time(NULL)) per call, scaled by z*z / (725*0.5) * 0.15 —
// vido_depth_noise (trackhost.cpp).  detail::SetDepthNoiseSeed(s != 0) pins the seed (tests, reproducible runs); 0 = the reference's time(NULL).
static unsigned g_noise_seed = 0;
static inline float noisy(float z, bool addnoise) { return addnoise ? vido_depth_noise(z, g_noise_seed) : z; }
cv::Mat Frame::UnprojectStereoStat(const int& i, const bool& addnoise) { const float z = noisy(mvStatDepth[i], addnoise); if (z < 0) return cv::Mat(); return unproject(*this, mvStatKeys[i].pt.x, mvStatKeys[i].pt.y, z); }
cv::Mat Frame::UnprojectStereoObject(const int& i, const bool& addnoise) { const float z = noisy(mvObjDepth[i], addnoise); if (!(z > 0)) return cv::Mat(); return unproject(*this, mvObjKeys[i].pt.x, mvObjKeys[i].pt.y, z); }
cv::Mat Frame::ObtainFlowDepthCamera(const int& i, const bool& addnoise) { const float z = noisy(mvStatDepth[i], addnoise); if (!(z > 0)) return cv::Mat(); return vec3(mvFlowNext[i].x, mvFlowNext[i].y, z); }
cv::Mat Frame::ObtainFlowDepthObject(const int& i, const bool& addnoise) { const float z = noisy(mvObjDepth[i], addnoise); if (!(z > 0)) return cv::Mat(); return vec3(mvObjFlowNext[i].x, mvObjFlowNext[i].y, z); }

void Map::reset() { *this = Map(); }

// One association table -> tracklets, incrementally.  Row i of TM maps feature j of frame i+1 to feature TM[i][j] of frame i (Tracking.cc:2530-2600).
// The owner tables reproduce what the optimiser's label pass derives from the full list (Optimizer.cc:120-140): a feature is owned by the
// highest-numbered tracklet of length >= 3 that contains it (only a tracklet's first element can be shared).
static void grow_tracklets(const std::vector<std::vector<int> >& TM, const std::vector<std::vector<cv::KeyPoint> >& feat, const std::vector<std::vector<int> >* Lab,
                           std::vector<std::vector<std::pair<int, int> > >& T, std::vector<int>* objid, std::vector<std::vector<int> >& trk, std::vector<std::vector<int> >& pos,
                           std::vector<int>& pre, size_t& rows, std::vector<int>* changes = nullptr)
{
    while (trk.size() < feat.size()) { trk.emplace_back(feat[trk.size()].size(), -1); pos.emplace_back(feat[pos.size()].size(), -1); }
    auto own = [&](int t, int k) { const std::pair<int, int>& e = T[t][k]; if (trk[e.first][e.second] <= t) { trk[e.first][e.second] = t; pos[e.first][e.second] = k;
                                   if (changes) { changes->push_back(e.first); changes->push_back(e.second); changes->push_back(t); changes->push_back(k); } } };
    for (; rows < TM.size(); rows++) {
        const int i = (int)rows;
        std::vector<int> cur(TM[i].size(), -1);
        for (size_t j = 0; j < TM[i].size(); j++) {
            const int m = TM[i][j]; if (m == -1) continue;
            int t;
            if (i > 0 && m < (int)pre.size() && pre[m] != -1) { t = pre[m]; T[t].push_back(std::make_pair(i + 1, (int)j)); }
            else { t = (int)T.size(); T.push_back({std::make_pair(i, m), std::make_pair(i + 1, (int)j)}); if (objid) objid->push_back((*Lab)[i][j]); }
            cur[j] = t;
            const int len = (int)T[t].size();
            if (len == 3) { own(t, 0); own(t, 1); own(t, 2); } else if (len > 3) own(t, len - 1);
        }
        pre.swap(cur);
    }
}
void Map::UpdateTracklets()
{
    grow_tracklets(vnAssoSta, vpFeatSta, nullptr, TrackletSta, nullptr, vnTrkSta, vnPosSta, trkPreSta, trkRowsSta, devWindow ? &trkChangesSta : nullptr);      // (a window that starts later reads the tables)
    grow_tracklets(vnAssoDyn, vpFeatDyn, &vnFeatLabel, TrackletDyn, &nObjID, vnTrkDyn, vnPosDyn, trkPreDyn, trkRowsDyn);
}

void Map::SyncPointsFromDevice() { finish_local_ba(); SyncPointsFromDeviceNow(); }
void Map::SyncPointsFromDeviceNow()                        // (without waiting for a window solve in flight: what the solve itself calls)
{
    if (!devWindow || !g_ctx_ba) return;
    const int N = (int)vp3DPointSta.size();
    std::vector<float> buf;
    for (int f = std::max(0, N - 64); f < devFramesPushed && f < N; f++) {
        const int n = (int)vp3DPointSta[f].size();
        if (!n) continue;
        buf.resize(3 * (size_t)n);
        if (vido_bawin_read_points(g_ctx_ba, f, n, buf.data()) != VIDO_OK) continue;      // (frames that have left the ring were synchronised when they left)
        cv::Mat::batch3x1(buf.data(), n, vp3DPointSta[f]);
    }
}

// ---- Optimizer ---------------------------------------------------------------------------------------------------------
static inline void get3d_world(const cv::KeyPoint& f, float d, const cv::Mat& K, const cv::Mat& Twc, float o[3])
{
    const float invfx = 1.0f / K.at<float>(0, 0), invfy = 1.0f / K.at<float>(1, 1), cx = K.at<float>(0, 2), cy = K.at<float>(1, 2);
    const float z = d, x = (f.pt.x - cx) * z * invfx, y = (f.pt.y - cy) * z * invfy;
    for (int r = 0; r < 3; r++) { const double s = (double)Twc.at<float>(r, 0) * x + (double)Twc.at<float>(r, 1) * y + (double)Twc.at<float>(r, 2) * z; o[r] = (float)s + Twc.at<float>(r, 3); }
}
cv::Mat Optimizer::Get3DinWorld(const cv::KeyPoint& f, const float& d, const cv::Mat& K, const cv::Mat& Twc)
{
    cv::Mat out(3, 1, CV_32F);
    get3d_world(f, d, K, Twc, out.ptr<float>());
    return out;
}
cv::Mat Optimizer::Get3DinCamera(const cv::KeyPoint& f, const float& d, const cv::Mat& K)
{
    const float invfx = 1.0f / K.at<float>(0, 0), invfy = 1.0f / K.at<float>(1, 1), cx = K.at<float>(0, 2), cy = K.at<float>(1, 2);
    return vec3((f.pt.x - cx) * d * invfx, (f.pt.y - cy) * d * invfy, d);
}

static void fill_common(vido_pose_problem& p, const Frame* cur, int n)
{
    memset(&p, 0, sizeof p); p.n = n; p.fx = cur->fx; p.fy = cur->fy; p.cx = cur->cx; p.cy = cur->cy;
    for (int k = 0; k < 16; k++) { p.Twl[k] = (k % 5 == 0); p.T_init[k] = (k % 5 == 0); }
}

int Optimizer::PoseOptimizationNew(Frame* cur, Frame* last, std::vector<int>& TM)      // Optimizer.cc:2180-2334
{
    const int N = (int)TM.size();
    std::vector<double> Xw(3 * N), obs(2 * N);
    const unsigned noise_seed = g_noise_seed ? g_noise_seed : (unsigned)time(NULL);        // (the reference re-reads the clock per point: within one call the same second)
    for (int i = 0; i < N; i++) {
        obs[2 * i] = cur->mvStatKeys[TM[i]].pt.x; obs[2 * i + 1] = cur->mvStatKeys[TM[i]].pt.y;
        // Frame::UnprojectStereoStat(TemperalMatch[i], 1) (Optimizer.cc:2250): the depth carries the addnoise draw (one value of the seed = one offset for the whole call)
        float X[3]; const float z = vido_depth_noise(last->mvStatDepth[TM[i]], noise_seed);
        const bool have = !(z < 0) && unproject_raw(*last, last->mvStatKeys[TM[i]].pt.x, last->mvStatKeys[TM[i]].pt.y, z, X);
        for (int a = 0; a < 3; a++) Xw[3 * i + a] = have ? X[a] : 0.0;
    }
    vido_pose_problem p; fill_common(p, cur, N); p.mode = 0; p.Xw = Xw.data(); p.obs = obs.data(); toRow16(cur->mTcw, p.T_init);
    p.info_edge = 1.0; p.huber_delta = (double)std::sqrt(0.01f); p.use_huber = 1; p.rounds = 1; p.drop_kernel_after_round = 2;
    const int its[4] = {100, 10, 10, 10}; const float th[4] = {0.01f, 5.991f, 5.991f, 5.991f};
    memcpy(p.iters, its, sizeof its); memcpy(p.chi2_th, th, sizeof th);
    vido_pose_result r; std::vector<uint8_t> outl(std::max(N, 1));
    check(vido_pose_optimize(live_ctx("PoseOptimizationNew"), &p, &r, outl.data(), nullptr), "PoseOptimizationNew");
    if (N < 3) return 0;
    cur->SetPose(fromRow16(r.T));
    for (int i = 0; i < N; i++) if (outl[i]) TM[i] = -1;
    return r.n_inliers;
}

int Optimizer::PoseOptimizationFlow2Cam(Frame* cur, Frame* last, std::vector<int>& TM)   // Optimizer.cc:2622-2824
{
    const int N = (int)TM.size();
    std::vector<double> obs(2 * N), flow(2 * N), depth(N);
    for (int i = 0; i < N; i++) {
        const float zd = last->mvStatDepth[TM[i]]; const bool have = zd > 0;      // Frame::ObtainFlowDepthCamera
        flow[2 * i] = have ? last->mvFlowNext[TM[i]].x : 0.0; flow[2 * i + 1] = have ? last->mvFlowNext[TM[i]].y : 0.0; depth[i] = have ? zd : 1.0;
        obs[2 * i] = last->mvStatKeys[TM[i]].pt.x; obs[2 * i + 1] = last->mvStatKeys[TM[i]].pt.y;
    }
    vido_pose_problem p; fill_common(p, cur, N); p.mode = 1; p.obs = obs.data(); p.flow0 = flow.data(); p.depth = depth.data();
    toRow16(Converter::toInvMatrix(last->mTcw), p.Twl); toRow16(cur->mTcw, p.T_init);
    p.info_edge = 0.1; p.info_prior = 0.3; p.huber_delta = (double)std::sqrt(0.04f); p.use_huber = 1; p.rounds = 4; p.drop_kernel_after_round = 2;
    const int its[4] = {100, 100, 100, 100}; const float th[4] = {0.04f, 5.991f, 5.991f, 5.991f};
    memcpy(p.iters, its, sizeof its); memcpy(p.chi2_th, th, sizeof th);
    vido_pose_result r; std::vector<uint8_t> outl(std::max(N, 1)); std::vector<double> f(2 * std::max(N, 1));
    check(vido_pose_optimize(live_ctx("PoseOptimizationFlow2Cam"), &p, &r, outl.data(), f.data()), "PoseOptimizationFlow2Cam");
    if (N < 3) return 0;
    cur->SetPose(fromRow16(r.T));
    for (int i = 0; i < N; i++) {
        if (!outl[i]) {      // :2807-2813 refined optical flow
            cur->mvStatKeys[TM[i]].pt.x = last->mvStatKeys[TM[i]].pt.x + (float)f[2 * i];
            cur->mvStatKeys[TM[i]].pt.y = last->mvStatKeys[TM[i]].pt.y + (float)f[2 * i + 1];
        }
    }
    for (int i = 0; i < N; i++) if (outl[i]) TM[i] = -1;
    return r.n_inliers;
}

// One object's problem for the two per-object optimisers (built once, solved alone or in a batch of all objects of the frame)
namespace {
struct ObjProblem { vido_pose_problem p; std::vector<double> Xw, obs, flow, depth; std::vector<uint8_t> outl; std::vector<double> f; vido_pose_result r; bool joint = false; int N = 0; };
void build_objmot(ObjProblem& o, Frame* cur, Frame* last, const std::vector<int>& ObjId, const cv::Mat& InitModel)       // Optimizer.cc:2826-2935
{
    const int N = (int)ObjId.size(); o.N = N; o.joint = false;
    o.Xw.resize(3 * N); o.obs.resize(2 * N);
    for (int i = 0; i < N; i++) {
        o.obs[2 * i] = cur->mvObjKeys[ObjId[i]].pt.x; o.obs[2 * i + 1] = cur->mvObjKeys[ObjId[i]].pt.y;
        float X[3]; const bool have = obj3d_raw(*last, ObjId[i], X);        // Frame::UnprojectStereoObject
        for (int a = 0; a < 3; a++) o.Xw[3 * i + a] = have ? X[a] : 0.0;
    }
    vido_pose_problem& p = o.p; fill_common(p, cur, N); p.mode = 2; p.Xw = o.Xw.data(); p.obs = o.obs.data();
    toRow16(Converter::toInvMatrix(cur->mTcw) * InitModel, p.T_init);
    const double K[12] = {cur->fx, 0, cur->cx, 0, 0, cur->fy, cur->cy, 0, 0, 0, 1, 0}; double T[16]; toRow16(cur->mTcw, T);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) { double s = 0; for (int k = 0; k < 4; k++) s += K[r * 4 + k] * T[k * 4 + c]; p.P[r * 4 + c] = s; }
    p.info_edge = 1.0; p.use_huber = 0; p.rounds = 1; p.drop_kernel_after_round = 2;
    const int its[4] = {200, 100, 100, 100}; const float th[4] = {0.01f, 5.991f, 5.991f, 5.991f};
    memcpy(p.iters, its, sizeof its); memcpy(p.chi2_th, th, sizeof th);
    o.outl.assign(std::max(N, 1), 0); o.f.clear();
}
void build_flow2(ObjProblem& o, Frame* cur, Frame* last, const std::vector<int>& ObjId, const cv::Mat& InitModel)        // Optimizer.cc:3037-3160
{
    const int N = (int)ObjId.size(); o.N = N; o.joint = true;
    o.obs.resize(2 * N); o.flow.resize(2 * N); o.depth.resize(N);
    for (int i = 0; i < N; i++) {
        const float zd = last->mvObjDepth[ObjId[i]]; const bool have = zd > 0;      // Frame::ObtainFlowDepthObject
        o.flow[2 * i] = have ? last->mvObjFlowNext[ObjId[i]].x : 0.0; o.flow[2 * i + 1] = have ? last->mvObjFlowNext[ObjId[i]].y : 0.0; o.depth[i] = have ? zd : 1.0;
        o.obs[2 * i] = last->mvObjKeys[ObjId[i]].pt.x; o.obs[2 * i + 1] = last->mvObjKeys[ObjId[i]].pt.y;
    }
    vido_pose_problem& p = o.p; fill_common(p, cur, N); p.mode = 1; p.obs = o.obs.data(); p.flow0 = o.flow.data(); p.depth = o.depth.data();
    toRow16(Converter::toInvMatrix(last->mTcw), p.Twl); toRow16(InitModel, p.T_init);
    p.info_edge = 0.1; p.info_prior = 0.5; p.huber_delta = (double)std::sqrt(0.04f); p.use_huber = 1; p.rounds = 1; p.drop_kernel_after_round = 2;
    const int its[4] = {200, 100, 100, 100}; const float th[4] = {0.04f, 5.991f, 5.991f, 5.991f};
    memcpy(p.iters, its, sizeof its); memcpy(p.chi2_th, th, sizeof th);
    o.outl.assign(std::max(N, 1), 0); o.f.assign(2 * std::max(N, 1), 0.0);
}
cv::Mat finish_obj(ObjProblem& o, Frame* cur, Frame* last, const std::vector<int>& ObjId, std::vector<int>& InlierID)     // :2990-3035 / :3215-3253
{
    InlierID.clear();
    for (int i = 0; i < o.N; i++) {
        if (!o.outl[i]) {
            if (o.joint) {      // refined optical flow
                cur->mvObjKeys[ObjId[i]].pt.x = last->mvObjKeys[ObjId[i]].pt.x + (float)o.f[2 * i];
                cur->mvObjKeys[ObjId[i]].pt.y = last->mvObjKeys[ObjId[i]].pt.y + (float)o.f[2 * i + 1];
            }
            InlierID.push_back(ObjId[i]);
        } else cur->vObjLabel[ObjId[i]] = -1;
    }
    return fromRow16(o.r.T);
}
}  // namespace

cv::Mat Optimizer::PoseOptimizationObjMot(Frame* cur, Frame* last, const std::vector<int>& ObjId, std::vector<int>& InlierID)   // :2826-3035
{
    if ((int)ObjId.size() < 3) return cv::Mat::eye(4, 4, CV_32F);
    ObjProblem o; build_objmot(o, cur, last, ObjId, cur->mInitModel);
    check(vido_pose_optimize(live_ctx("PoseOptimizationObjMot"), &o.p, &o.r, o.outl.data(), nullptr), "PoseOptimizationObjMot");
    return finish_obj(o, cur, last, ObjId, InlierID);
}

cv::Mat Optimizer::PoseOptimizationFlow2(Frame* cur, Frame* last, const std::vector<int>& ObjId, std::vector<int>& InlierID)    // :3037-3253
{
    if ((int)ObjId.size() < 3) return cv::Mat::eye(4, 4, CV_32F);
    ObjProblem o; build_flow2(o, cur, last, ObjId, cur->mInitModel);
    check(vido_pose_optimize(live_ctx("PoseOptimizationFlow2"), &o.p, &o.r, o.outl.data(), o.f.data()), "PoseOptimizationFlow2");
    return finish_obj(o, cur, last, ObjId, InlierID);
}

// All dynamic objects of the frame in one launch (extension: the reference calls PoseOptimizationFlow2 / ObjMot once per object inside Tracking::Track's loop,
// Tracking.cc:1268-1274; the objects' point sets are disjoint, so the results equal the per-object calls).
std::vector<cv::Mat> Optimizer::PoseOptimizationObjectsBatch(Frame* cur, Frame* last, const std::vector<std::vector<int> >& ObjIds, const std::vector<cv::Mat>& InitModels,
                                                              std::vector<std::vector<int> >& InlierIDs, bool joint)
{
    const size_t n = ObjIds.size();
    std::vector<cv::Mat> out(n); InlierIDs.assign(n, std::vector<int>());
    std::vector<ObjProblem> O(n); std::vector<vido_pose_problem> P; std::vector<vido_pose_result> R; std::vector<uint8_t*> om; std::vector<double*> fo; std::vector<size_t> which;
    for (size_t i = 0; i < n; i++) {
        if (ObjIds[i].size() < 3) { out[i] = cv::Mat::eye(4, 4, CV_32F); continue; }
        if (joint) build_flow2(O[i], cur, last, ObjIds[i], InitModels[i]); else build_objmot(O[i], cur, last, ObjIds[i], InitModels[i]);
        which.push_back(i);
    }
    for (size_t i : which) { P.push_back(O[i].p); om.push_back(O[i].outl.data()); fo.push_back(joint ? O[i].f.data() : nullptr); }
    R.resize(P.size());
    if (!P.empty()) check(vido_pose_optimize_batch(live_ctx("PoseOptimizationObjectsBatch"), P.data(), (int)P.size(), R.data(), om.data(), fo.data()), "PoseOptimizationObjectsBatch");
    for (size_t k = 0; k < which.size(); k++) { const size_t i = which[k]; O[i].r = R[k]; out[i] = finish_obj(O[i], cur, last, ObjIds[i], InlierIDs[i]); }
    return out;
}

// g2o text dump of the full-batch graph (the reference saves dynamic_slam_graph_{before,after}_opt.g2o, Optimizer.cc:1937-1939) with the
// vendored g2o's tags and field order (types_slam3d.cpp:37-45; VertexSE3 / EdgeSE3 / EdgeSE3Prior / EdgeSE3PointXYZ / VertexPointXYZ ::write,
// LandmarkMotionTernaryEdge::write types_dyn_slam3d.cpp:44-51) and its vertex numbering (one running id in creation order: per frame the
// camera, the static points first seen there, the object motions, the dynamic points).  Written only when VIDO_DUMP_G2O names a directory.
static void rot2quat(const double* T /*3x4*/, double* q /*x y z w*/)
{
    const double m00 = T[0], m01 = T[1], m02 = T[2], m10 = T[4], m11 = T[5], m12 = T[6], m20 = T[8], m21 = T[9], m22 = T[10];
    const double tr = m00 + m11 + m22;
    if (tr > 0) { const double s = std::sqrt(tr + 1.0) * 2; q[3] = 0.25 * s; q[0] = (m21 - m12) / s; q[1] = (m02 - m20) / s; q[2] = (m10 - m01) / s; }
    else if (m00 > m11 && m00 > m22) { const double s = std::sqrt(1.0 + m00 - m11 - m22) * 2; q[3] = (m21 - m12) / s; q[0] = 0.25 * s; q[1] = (m01 + m10) / s; q[2] = (m02 + m20) / s; }
    else if (m11 > m22) { const double s = std::sqrt(1.0 + m11 - m00 - m22) * 2; q[3] = (m02 - m20) / s; q[0] = (m01 + m10) / s; q[1] = 0.25 * s; q[2] = (m12 + m21) / s; }
    else { const double s = std::sqrt(1.0 + m22 - m00 - m11) * 2; q[3] = (m10 - m01) / s; q[0] = (m02 + m20) / s; q[1] = (m12 + m21) / s; q[2] = 0.25 * s; }
}
static void dump_g2o(const std::string& path, const vido_ba_problem& b, const vido_ba_dynamic& d, const std::vector<std::pair<int, int> >& ptOwner, const std::vector<int>& Hfr)
{
    FILE* f = fopen(path.c_str(), "w"); if (!f) return;
    std::vector<int> cu(b.n_cam), pu(b.n_pt), hu(d.n_H), du(d.n_dyn);
    { int uid = 1; size_t ip = 0, ih = 0, id = 0;
      for (int i = 0; i < b.n_cam; i++) {
          cu[i] = uid++;
          while (ip < ptOwner.size() && ptOwner[ip].first == i) pu[ip++] = uid++;
          while (ih < Hfr.size() && Hfr[ih] == i) hu[ih++] = uid++;
          while ((int)id < d.n_dyn && d.dyn_cam[id] == i) du[id++] = uid++;
      } }
    auto se3 = [&](const double* T) { double q[4]; rot2quat(T, q); fprintf(f, "%.9g %.9g %.9g %.9g %.9g %.9g %.9g ", T[3], T[7], T[11], q[0], q[1], q[2], q[3]); };
    auto info = [&](int n, double v) { for (int i = 0; i < n; i++) for (int j = i; j < n; j++) fprintf(f, "%.9g ", i == j ? v : 0.0); };
    fprintf(f, "PARAMS_SE3OFFSET 0 0 0 0 0 0 0 1 \n");
    struct V { int uid, kind, idx; }; std::vector<V> vs;
    for (int i = 0; i < b.n_cam; i++) vs.push_back(V{cu[i], 0, i});
    for (int i = 0; i < b.n_pt; i++) vs.push_back(V{pu[i], 1, i});
    for (int i = 0; i < d.n_H; i++) vs.push_back(V{hu[i], 2, i});
    for (int i = 0; i < d.n_dyn; i++) vs.push_back(V{du[i], 3, i});
    std::sort(vs.begin(), vs.end(), [](const V& a, const V& c) { return a.uid < c.uid; });
    for (const V& v : vs) {
        if (v.kind == 0 || v.kind == 2) { fprintf(f, "VERTEX_SE3:QUAT %d ", v.uid); se3(v.kind == 0 ? b.cam_T + 12 * (size_t)v.idx : d.H_T + 12 * (size_t)v.idx); }
        else { const double* x = v.kind == 1 ? b.pt_xyz + 3 * (size_t)v.idx : d.dyn_xyz + 3 * (size_t)v.idx; fprintf(f, "VERTEX_TRACKXYZ %d %.9g %.9g %.9g ", v.uid, x[0], x[1], x[2]); }
        fprintf(f, "\n");
    }
    if (b.prior_cam >= 0) { fprintf(f, "EDGE_SE3_PRIOR %d 0 ", cu[b.prior_cam]); se3(b.prior_T); info(6, b.info_prior); fprintf(f, "\n"); }
    for (int k = 0; k < b.n_odo; k++) { fprintf(f, "EDGE_SE3:QUAT %d %d ", cu[b.odo_i[k]], cu[b.odo_j[k]]); se3(b.odo_T + 12 * (size_t)k); info(6, b.info_odo); fprintf(f, "\n"); }
    for (int k = 0; k < b.n_obs; k++) { fprintf(f, "EDGE_SE3_TRACKXYZ %d %d 0 %.9g %.9g %.9g ", cu[b.obs_cam[k]], pu[b.obs_pt[k]], b.obs_meas[3 * (size_t)k], b.obs_meas[3 * (size_t)k + 1], b.obs_meas[3 * (size_t)k + 2]); info(3, b.info_obs); fprintf(f, "\n"); }
    static const double I12[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    for (int k = 0; k < d.n_smooth; k++) { fprintf(f, "EDGE_SE3:QUAT %d %d ", hu[d.sm_i[k]], hu[d.sm_j[k]]); se3(I12); info(6, d.info_smooth); fprintf(f, "\n"); }
    for (int k = 0; k < d.n_dyn; k++) { fprintf(f, "EDGE_SE3_TRACKXYZ %d %d 0 %.9g %.9g %.9g ", cu[d.dyn_cam[k]], du[k], d.dyn_meas[3 * (size_t)k], d.dyn_meas[3 * (size_t)k + 1], d.dyn_meas[3 * (size_t)k + 2]); info(3, d.info_dyn); fprintf(f, "\n"); }
    for (int k = 0; k < d.n_tern; k++) { fprintf(f, "EDGE_SE3_MOTION %d %d %d 0 0 0 ", du[d.tern_prev[k]], du[d.tern_cur[k]], hu[d.tern_H[k]]); info(3, d.info_tern); fprintf(f, "\n"); }
    fclose(f);
}

// Map walk of PartialBatchOptimization (Optimizer.cc:56-160, 216-350; STATIC_ONLY graph) and FullBatchOptimization (:1235-2178;
// static + object factors) onto the flat BA problem.
static int g_res_checks = 0, g_res_mismatch = 0;
namespace detail { void SetDepthNoiseSeed(unsigned seed) { g_noise_seed = seed; }
                   void ResidentCheckStats(int* checks, int* mismatches) { if (checks) *checks = g_res_checks; if (mismatches) *mismatches = g_res_mismatch; }
                   float LastFrameStageMs(int which) { return which == 0 ? g_ms_orb : g_ms_lists; } }

// Commits a local-window result to the Map (Optimizer.cc:1084-1128): refined camera poses, the odometry factors re-derived from them
static void commit_window_poses(Map* pMap, int start, int N, const std::vector<double>& cam)
{
    for (int i = start; i < N; i++) {
        cv::Mat T = cv::Mat::eye(4, 4, CV_32F);
        for (int rr = 0; rr < 3; rr++) for (int c = 0; c < 4; c++) T.at<float>(rr, c) = (float)cam[(size_t)(i - start) * 12 + rr * 4 + c];
        pMap->vmCameraPose[i] = T;
        if (i > start) pMap->vmRigidMotion[i - 1][0] = Converter::toInvMatrix(pMap->vmCameraPose[i - 1]) * pMap->vmCameraPose[i];
    }
}

// PartialBatchOptimization on the device-resident window (csrc/bawin.hip): only the new frames' feature rows and the tracklet-label changes go up, the graph is assembled
// and solved on the device, the refined landmarks stay there (Map::SyncPointsFromDevice brings them home when the host needs them).  false: the window cannot take this
// sequence (more features per frame than the ring holds) -> the caller walks the Map like rounds 1-2 did.
static bool batch_optimize_resident(Map* pMap, const cv::Mat& K, int start, int N, int WINDOW_SIZE, const std::vector<double>* check_cam, int check_nobs, int check_npt)
{
    if (pMap->devWindowDisabled) return false;                 // per Map (a process-wide latch switched the resident path off for every later System)
    vido_ctx* c = ba_ctx("PartialBatchOptimization");
    const int cap_f = 24, cap_n = 8192, nc = N - start;      // (the caller sends windows of <= 20 cameras here)
    const float invfx = 1.0f / K.at<float>(0, 0), invfy = 1.0f / K.at<float>(1, 1), kcx = K.at<float>(0, 2), kcy = K.at<float>(1, 2);
    const bool fresh = !pMap->devWindow;
    if (fresh) { check_ba(vido_bawin_create(c, cap_f, cap_n), "bawin_create"); pMap->devWindow = true; pMap->devFramesPushed = 0; }
    std::vector<double> meas; std::vector<float> xyz;
    for (int f = std::max(pMap->devFramesPushed, N - (cap_f - 1)); f < N; f++) {
        const int n = (int)pMap->vpFeatSta[f].size(), fo = f - cap_f;
        if (n > cap_n || (f > 0 && (int)pMap->vnAssoSta[f - 1].size() != n)) { pMap->devWindowDisabled = true; pMap->SyncPointsFromDeviceNow(); pMap->devWindow = false; return false; }
        if (fo >= 0 && !pMap->vp3DPointSta[fo].empty()) {      // the frame this one replaces in the ring: its points go home first
            std::vector<float> buf(3 * pMap->vp3DPointSta[fo].size());
            if (vido_bawin_read_points(c, fo, (int)pMap->vp3DPointSta[fo].size(), buf.data()) == VIDO_OK)
                cv::Mat::batch3x1(buf.data(), (int)pMap->vp3DPointSta[fo].size(), pMap->vp3DPointSta[fo]);      // (one block for the row, not one allocation per point)
        }
        meas.resize(3 * (size_t)n); xyz.resize(3 * (size_t)n);
        for (int j = 0; j < n; j++) {
            const cv::KeyPoint& kp = pMap->vpFeatSta[f][j]; const float z = pMap->vfDepSta[f][j];                   // Optimizer::Get3DinCamera, as the walk inlines it
            meas[3 * (size_t)j] = (kp.pt.x - kcx) * z * invfx; meas[3 * (size_t)j + 1] = (kp.pt.y - kcy) * z * invfy; meas[3 * (size_t)j + 2] = z;
            const cv::Mat& Xw = pMap->vp3DPointSta[f][j]; for (int a = 0; a < 3; a++) xyz[3 * (size_t)j + a] = Xw.at<float>(a);
        }
        LBA_TRACE("ba: push_frame begin");
        check_ba(vido_bawin_push_frame(c, f, n, meas.data(), xyz.data(), f > 0 ? pMap->vnAssoSta[f - 1].data() : nullptr), "bawin_push_frame");
        LBA_TRACE("ba: push_frame end");
    }
    pMap->devFramesPushed = N;
    if (fresh) {                                               // a ring that starts late takes the labels of its frames from the Map's tables, not from the change list
        pMap->trkChangesSta.clear();
        for (int f = std::max(0, N - (cap_f - 1)); f < N; f++) for (size_t j = 0; j < pMap->vnTrkSta[f].size(); j++) if (pMap->vnTrkSta[f][j] != -1) {
            pMap->trkChangesSta.push_back(f); pMap->trkChangesSta.push_back((int)j); pMap->trkChangesSta.push_back(pMap->vnTrkSta[f][j]); pMap->trkChangesSta.push_back(pMap->vnPosSta[f][j]); }
    }
    LBA_TRACE("ba: set_labels begin");
    if (!pMap->trkChangesSta.empty()) { check_ba(vido_bawin_set_labels(c, (int)(pMap->trkChangesSta.size() / 4), pMap->trkChangesSta.data()), "bawin_set_labels"); pMap->trkChangesSta.clear(); }
    std::vector<double> cam((size_t)nc * 12), odo; std::vector<int32_t> oi, oj;
    for (int i = start; i < N; i++) for (int r = 0; r < 3; r++) for (int cc = 0; cc < 4; cc++) cam[(size_t)(i - start) * 12 + r * 4 + cc] = pMap->vmCameraPose[i].at<float>(r, cc);
    for (int i = start + 1; i < N; i++) {
        const cv::Mat& M = pMap->vmRigidMotion[i - 1][0];
        oi.push_back(i - 1 - start); oj.push_back(i - start);
        for (int r = 0; r < 3; r++) for (int cc = 0; cc < 4; cc++) odo.push_back(M.at<float>(r, cc));
    }
    vido_ba_problem b; memset(&b, 0, sizeof b);
    b.n_cam = nc; b.cam_T = cam.data();
    b.n_odo = (int)oi.size(); b.odo_i = oi.data(); b.odo_j = oj.data(); b.odo_T = odo.data();
    b.use_huber = 1; b.huber_obs = b.huber_odo = (double)0.01f; b.info_odo = 1.0 / (double)0.0001f;
    b.prior_cam = (N == WINDOW_SIZE) ? 0 : -1; b.info_prior = 1.0 / 0.0000001; b.info_obs = 1.0 / (double)16.f; b.max_iters = 100; b.gain_threshold = 1e-3;      // Optimizer.cc:183,191-196,226-235
    for (int k = 0; k < 12; k++) b.prior_T[k] = cam[k];
    vido_ba_result r; int32_t no = 0, np = 0;
    LBA_TRACE("ba: solve begin");
    check_ba(vido_bawin_solve(c, start, N, &b, &r, &no, &np), "PartialBatchOptimization (resident window)");
    LBA_TRACE("ba: solve end");
    if (getenv("VIDO_BA_VERBOSE"))
        fprintf(stderr, "[batch partial, resident] cams %d pts %d obs %d | iters %d trials %d chi2 %.6g -> %.6g | setup %.2f ms loop %.2f ms\n", nc, np, no, r.iterations, r.lm_trials, r.chi2_initial, r.chi2_final, r.ms_setup, r.ms_solve_loop);
    if (check_cam) {                                           // VIDO_BA_RESIDENT_CHECK: the Map walk solved the same window on the host-assembled arrays
        g_res_checks++;
        double worst = 0; for (size_t k = 0; k < cam.size() && k < check_cam->size(); k++) worst = std::max(worst, std::fabs(cam[k] - (*check_cam)[k]));
        if (no != check_nobs || np != check_npt || !(worst < 1e-7) || (np && r.iterations == 0)) {
            g_res_mismatch++;
            fprintf(stderr, "[resident check] window [%d,%d): obs %d vs %d, landmarks %d vs %d, max pose difference %.3g\n", start, N, no, check_nobs, np, check_npt, worst);
        }
    }
    if (np == 0) return true;
    commit_window_poses(pMap, start, N, cam);
    return true;
}

static void batch_optimize(Map* pMap, const cv::Mat K, int WINDOW_SIZE, bool global)
{
    const int N = (int)pMap->vpFeatSta.size();
    if (N < 2 || WINDOW_SIZE < 1) return;
    const int start = global ? 0 : std::max(N - WINDOW_SIZE, 0), nc = N - start;
    const float invfx = 1.0f / K.at<float>(0, 0), invfy = 1.0f / K.at<float>(1, 1), kcx = K.at<float>(0, 2), kcy = K.at<float>(1, 2);
    pMap->UpdateTracklets();                                  // no-op when Tracking::Track already did it for this frame
    static const bool host_walk = getenv("VIDO_BA_HOST_WALK") != nullptr, check_walk = getenv("VIDO_BA_RESIDENT_CHECK") != nullptr;
    const bool resident = !global && !host_walk && 6 * nc <= 120;
    if (resident && !check_walk && batch_optimize_resident(pMap, K, start, N, WINDOW_SIZE, nullptr, 0, 0)) return;
    pMap->SyncPointsFromDeviceNow();                          // the walk below reads vp3DPointSta
    if (!(resident && check_walk)) { pMap->devWindow = false; std::vector<int>().swap(pMap->trkChangesSta); }      // nobody consumes the change list on this path (a later resident window starts from the Map's tables): it must not grow with the sequence   // ... and writes it: the ring's copy is stale from here on (a later resident window starts afresh)
    const auto& Tr = pMap->TrackletSta; const auto& lab = pMap->vnTrkSta;
    std::vector<std::vector<int> > mak(N);                     // only the window's frames are touched
    for (int i = start; i < N; i++) mak[i].assign(pMap->vpFeatSta[i].size(), -1);
    std::vector<double> cam((size_t)nc * 12), pts, meas, odo; std::vector<int32_t> oc, op, oi, oj;
    for (int i = start; i < N; i++) for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) cam[(size_t)(i - start) * 12 + r * 4 + c] = pMap->vmCameraPose[i].at<float>(r, c);
    std::vector<std::pair<int, int> > ptOwner;
    { size_t cap = 0; for (int i = start; i < N; i++) cap += lab[i].size(); oc.reserve(cap); op.reserve(cap); meas.reserve(3 * cap); }
    for (int i = start; i < N; i++) {
        if (i != start) {
            const cv::Mat& M = pMap->vmRigidMotion[i - 1][0];
            oi.push_back(i - 1 - start); oj.push_back(i - start);
            for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) odo.push_back(M.at<float>(r, c));
        }
        for (size_t j = 0; j < lab[i].size(); j++) {
            const int t = lab[i][j]; if (t == -1) continue;
            const int pos = pMap->vnPosSta[i][j];
            int pid;
            if (pos == 0) {                                   // tracklet starts inside the window: new point vertex (:290-325)
                pid = (int)ptOwner.size(); ptOwner.push_back(std::make_pair(i, (int)j));
                const cv::Mat& Xw = pMap->vp3DPointSta[i][j]; for (int a = 0; a < 3; a++) pts.push_back(Xw.at<float>(a));
            } else {
                const int pf = Tr[t][pos - 1].first;
                const int pm = pf < start ? -1 : mak[pf][Tr[t][pos - 1].second];
                if (pm == -1) continue;                       // started before the window (:332-333)
                pid = pm;
            }
            mak[i][j] = pid;
            const cv::KeyPoint& f = pMap->vpFeatSta[i][j]; const float z = pMap->vfDepSta[i][j];                  // Optimizer::Get3DinCamera, inlined
            oc.push_back(i - start); op.push_back(pid); meas.push_back((f.pt.x - kcx) * z * invfx); meas.push_back((f.pt.y - kcy) * z * invfy); meas.push_back(z);
        }
    }
    if (pts.empty()) return;
    vido_ba_problem b; memset(&b, 0, sizeof b);
    b.n_cam = nc; b.cam_T = cam.data(); b.n_pt = (int)ptOwner.size(); b.pt_xyz = pts.data();
    b.n_obs = (int)oc.size(); b.obs_cam = oc.data(); b.obs_pt = op.data(); b.obs_meas = meas.data();
    b.n_odo = (int)oi.size(); b.odo_i = oi.data(); b.odo_j = oj.data(); b.odo_T = odo.data();
    b.use_huber = 1; b.huber_obs = b.huber_odo = (double)0.01f; b.info_odo = 1.0 / (double)0.0001f;
    if (!global) { b.prior_cam = (N == WINDOW_SIZE) ? 0 : -1; b.info_prior = 1.0 / 0.0000001; b.info_obs = 1.0 / (double)16.f; b.max_iters = 100; b.gain_threshold = 1e-3; }   // :183,191-196,226-235
    else { b.prior_cam = 0; b.info_prior = 1e5; b.info_obs = 1.0 / (double)80.f; b.max_iters = 300; b.gain_threshold = 1e-4; }                                             // :1325,1333-1338
    for (int k = 0; k < 12; k++) b.prior_T[k] = cam[k];
    vido_ba_result r;
    // ---- object part of FullBatchOptimization (STATIC_ONLY = false, Optimizer.cc:1540-1750)
    std::vector<double> Hs, dxyz, dmeas; std::vector<int32_t> dcam, tp, tc, th, smi, smj; std::vector<int> Hfr;
    std::vector<std::vector<int> > makD(N), Hid(std::max(N - 1, 0));
    if (global) {
        const auto& TrD = pMap->TrackletDyn; const auto& labD = pMap->vnTrkDyn;
        for (int i = 0; i < N; i++) makD[i].assign(pMap->vpFeatDyn[i].size(), -1);
        for (int i = 0; i < N - 1; i++) Hid[i].assign(pMap->vnRMLabel[i].size(), -1);
        auto add_dyn = [&](int i, int j) -> int {          // VertexPointXYZ + EdgeSE3PointXYZ of one dynamic observation (:1560-1582)
            const int id = (int)dcam.size();
            const cv::Mat& Xw = pMap->vp3DPointDyn[i][j]; for (int a = 0; a < 3; a++) dxyz.push_back(Xw.at<float>(a));
            const cv::KeyPoint& f = pMap->vpFeatDyn[i][j]; const float z = pMap->vfDepDyn[i][j];
            dmeas.push_back((f.pt.x - kcx) * z * invfx); dmeas.push_back((f.pt.y - kcy) * z * invfy); dmeas.push_back(z);
            dcam.push_back(i); makD[i][j] = id;
            return id;
        };
        for (int i = 0; i < N; i++) {
            if (i == 0) { for (size_t j = 0; j < labD[0].size(); j++) if (labD[0][j] != -1) add_dyn(0, (int)j); continue; }
            // one motion vertex per object of frame i, initialised to identity (:1583-1592), smoothness edge to the same object's
            // vertex of the previous frame when i > 2 (:1604-1636)
            for (size_t j = 1; j < pMap->vmRigidMotion[i - 1].size(); j++) {
                const int hid = (int)(Hs.size() / 12);
                static const double I12[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
                Hs.insert(Hs.end(), I12, I12 + 12); Hfr.push_back(i);
                if (i > 2) {
                    int trace = -1;
                    for (size_t k = 0; k < pMap->vnRMLabel[i - 2].size(); k++) if (pMap->vnRMLabel[i - 2][k] == pMap->vnRMLabel[i - 1][j]) { trace = (int)k; break; }
                    if (trace > 0 && Hid[i - 2][trace] >= 0) { smi.push_back(Hid[i - 2][trace]); smj.push_back(hid); }
                }
                Hid[i - 1][j] = hid;
            }
            for (size_t j = 0; j < labD[i].size(); j++) {
                const int t = labD[i][j]; if (t == -1) continue;
                const int pos = pMap->vnPosDyn[i][j];
                int hobj = -1;
                for (size_t k = 1; k < pMap->vnRMLabel[i - 1].size(); k++) if (pMap->vnRMLabel[i - 1][k] == pMap->nObjID[t]) { hobj = Hid[i - 1][k]; break; }
                if (hobj == -1 && pos != 0) continue;                         // no motion vertex for this object in this frame (:1668-1671)
                if (pos == 0) { add_dyn(i, (int)j); continue; }
                const int prev = makD[TrD[t][pos - 1].first][TrD[t][pos - 1].second];
                const int id = add_dyn(i, (int)j);
                if (prev >= 0) { tp.push_back(prev); tc.push_back(id); th.push_back(hobj); }    // LandmarkMotionTernaryEdge (:1728-1745)
            }
        }
    }
    vido_ba_dynamic d; memset(&d, 0, sizeof d);
    d.n_H = (int)(Hs.size() / 12); d.H_T = Hs.data(); d.n_dyn = (int)dcam.size(); d.dyn_xyz = dxyz.data(); d.dyn_cam = dcam.data(); d.dyn_meas = dmeas.data();
    d.n_tern = (int)tp.size(); d.tern_prev = tp.data(); d.tern_cur = tc.data(); d.tern_H = th.data(); d.n_smooth = (int)smi.size(); d.sm_i = smi.data(); d.sm_j = smj.data();
    d.info_dyn = 1.0 / (double)80.f; d.info_tern = 1.0 / (double)100.f; d.info_smooth = 1.0 / (double)0.001f;                     // :1333-1338
    d.huber_dyn = d.huber_tern = d.huber_smooth = (double)0.01f;                                                              // :1358
    const char* dump_dir = global ? getenv("VIDO_DUMP_G2O") : nullptr;
    if (dump_dir) dump_g2o(std::string(dump_dir) + "/dynamic_slam_graph_before_opt.g2o", b, d, ptOwner, Hfr);
    if (global && (d.n_H || d.n_dyn)) check_ba(vido_ba_optimize_dynamic(ba_ctx("FullBatchOptimization"), &b, &d, &r, nullptr, nullptr), "FullBatchOptimization");
    else check_ba(vido_ba_optimize(ba_ctx("BatchOptimization"), &b, &r, nullptr, nullptr), global ? "FullBatchOptimization" : "PartialBatchOptimization");
    if (getenv("VIDO_BA_VERBOSE"))
        fprintf(stderr, "[batch %s] cams %d pts %d obs %d | H %d dyn %d tern %d smooth %d | iters %d trials %d chi2 %.6g -> %.6g | setup %.2f ms loop %.2f ms\n", global ? "full" : "partial",
                b.n_cam, b.n_pt, b.n_obs, d.n_H, d.n_dyn, d.n_tern, d.n_smooth, r.iterations, r.lm_trials, r.chi2_initial, r.chi2_final, r.ms_setup, r.ms_solve_loop);
    if (dump_dir) dump_g2o(std::string(dump_dir) + "/dynamic_slam_graph_after_opt.g2o", b, d, ptOwner, Hfr);
    if (resident && check_walk && batch_optimize_resident(pMap, K, start, N, WINDOW_SIZE, &cam, b.n_obs, b.n_pt)) return;      // the resident solve commits; this walk was the check
    auto& poses = global ? pMap->vmCameraPose_RF : pMap->vmCameraPose;
    for (int i = start; i < N; i++) {                      // write-back, Optimizer.cc:1084-1128 / :2098-2137
        if (global && i == 0) continue;                    // the full batch writes vmCameraPose_RF[i+1] only
        cv::Mat T = cv::Mat::eye(4, 4, CV_32F);
        for (int rr = 0; rr < 3; rr++) for (int c = 0; c < 4; c++) T.at<float>(rr, c) = (float)cam[(size_t)(i - start) * 12 + rr * 4 + c];
        poses[i] = T;
        if (i > start && !global) pMap->vmRigidMotion[i - 1][0] = Converter::toInvMatrix(poses[i - 1]) * poses[i];
    }
    for (int i = start; i < N; i++) for (size_t j = 0; j < mak[i].size(); j++) if (mak[i][j] != -1)      // every observation slot of the landmark (:1130-1160, :2140-2154)
        pMap->vp3DPointSta[i][j] = vec3((float)pts[3 * (size_t)mak[i][j]], (float)pts[3 * (size_t)mak[i][j] + 1], (float)pts[3 * (size_t)mak[i][j] + 2]);
    if (global) {
        for (int i = 0; i < N - 1; i++) for (size_t j = 1; j < Hid[i].size(); j++) if (Hid[i][j] >= 0) {   // :2119-2134
            cv::Mat T = cv::Mat::eye(4, 4, CV_32F);
            for (int rr = 0; rr < 3; rr++) for (int c = 0; c < 4; c++) T.at<float>(rr, c) = (float)Hs[(size_t)Hid[i][j] * 12 + rr * 4 + c];
            pMap->vmRigidMotion_RF[i][j] = T;
        }
        for (int i = 0; i < N; i++) for (size_t j = 0; j < makD[i].size(); j++) if (makD[i][j] != -1)       // :2155-2172
            pMap->vp3DPointDyn[i][j] = vec3((float)dxyz[3 * (size_t)makD[i][j]], (float)dxyz[3 * (size_t)makD[i][j] + 1], (float)dxyz[3 * (size_t)makD[i][j] + 2]);
    }
}
void Optimizer::PartialBatchOptimization(Map* pMap, const cv::Mat K, const int W) { finish_local_ba(); batch_optimize(pMap, K, W, false); }
void Optimizer::FullBatchOptimization(Map* pMap, const cv::Mat K) { finish_local_ba(); batch_optimize(pMap, K, (int)pMap->vpFeatSta.size(), true); }

// The window solve of Tracking::Track, beside the next frame (round 5).  The reference calls PartialBatchOptimization at the end of Track() and waits (Tracking.cc:1430-1451);
// nothing the tracker does with the NEXT frame before it appends that frame to the Map reads what the solve writes (refined vmCameraPose / vmRigidMotion[.][0] / landmarks), so
// the solve runs on a helper thread with the BA context's own stream and is joined (a) before the Map grows again, (b) before the next solve, (c) before any reader of the Map
// (FullBatchOptimization, SaveResults, SyncPointsFromDevice, the System's destructor).  Same results, in the same order; 2.3 ms of a 5.7 ms tracker frame leave the tracker's
// critical path of the tracker ALONE.  Opt-in (VIDO_LBA_ASYNC=1, see start_local_ba); the two host-walk check modes always keep the reference's order.
namespace detail {
static void finish_local_ba()
{
    if (!g_lba.pending) return;
    g_lba.pending = false;
    LBA_TRACE("tracker: join begin");
    const float ms = g_lba.fut.get();
    LBA_TRACE("tracker: join end");                          // (rethrows what the solve threw)
    if (g_lba.map) g_lba.map->fLBA_time.push_back(ms);
}
static void start_local_ba(Map* pMap, const cv::Mat& K, int window)
{
    // Default: the reference's order (solve, then return).  VIDO_LBA_ASYNC=1 runs the solve beside the next frame: measured 5.97 -> 4.61 ms per frame for the tracker alone
    // (tools/prof_tracker.py), and NO gain beside the networks (174 frames/s either way for the chain without the detector): there the helper's first stream operation waits
    // 2-5 ms for the GPU while LiteFlowNet's graph runs, the join waits for it, and the solve's reported duration triples — so it stays opt-in.
    static const bool sync_mode = getenv("VIDO_LBA_ASYNC") == nullptr || getenv("VIDO_LBA_SYNC") != nullptr || getenv("VIDO_BA_HOST_WALK") != nullptr || getenv("VIDO_BA_RESIDENT_CHECK") != nullptr;
    finish_local_ba();
    auto job = [pMap, K, window]() -> float {
        const auto t0 = std::chrono::steady_clock::now();
        batch_optimize(pMap, K, window, false);
        return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    if (sync_mode) { pMap->fLBA_time.push_back(job()); return; }
    ba_ctx("PartialBatchOptimization");                         // (created on the tracker's thread, before the helper needs it)
    g_lba.map = pMap; g_lba.pending = true;
    g_lba.fut = std::async(std::launch::async, job);
}
}  // namespace detail

// ---- Tracking ----------------------------------------------------------------------------------------------------------------
Tracking::Tracking(System* pSys, Map* pMap, const std::string& path, const int sensor)        // Tracking.cc:39-172
    : mState(NO_IMAGES_YET), mTestData(KAIST), mSensor(sensor), bJoint(true), f_id(0), max_id(1), StopFrame(0), mScale(1.0f), ransac_seed(20260926u), mpSystem(pSys), mpMap(pMap)
{
    auto kv = ParseSettings(path);
    mK = cv::Mat::eye(3, 3, CV_32F);
    mK.at<float>(0, 0) = (float)num(kv, "Camera.fx"); mK.at<float>(1, 1) = (float)num(kv, "Camera.fy"); mK.at<float>(0, 2) = (float)num(kv, "Camera.cx"); mK.at<float>(1, 2) = (float)num(kv, "Camera.cy");
    const float k3 = (float)num(kv, "Camera.k3");
    mDistCoef = cv::Mat(k3 != 0 ? 5 : 4, 1, CV_32F);
    mDistCoef.at<float>(0) = (float)num(kv, "Camera.k1"); mDistCoef.at<float>(1) = (float)num(kv, "Camera.k2"); mDistCoef.at<float>(2) = (float)num(kv, "Camera.p1"); mDistCoef.at<float>(3) = (float)num(kv, "Camera.p2");
    if (k3 != 0) mDistCoef.at<float>(4) = k3;
    mbf = (float)num(kv, "Camera.bf"); mbRGB = num(kv, "Camera.RGB") != 0;
    mpORBextractorLeft = new ORBextractor((int)num(kv, "ORBextractor.nFeatures", 2000), (float)num(kv, "ORBextractor.scaleFactor", 1.2), (int)num(kv, "ORBextractor.nLevels", 8),
                                          (int)num(kv, "ORBextractor.iniThFAST", 20), (int)num(kv, "ORBextractor.minThFAST", 7));
    const int code = (int)num(kv, "ChooseData", 3);
    mTestData = code == 1 ? OMD : (code == 2 ? KITTI : KAIST);
    mThDepth = (float)num(kv, "ThDepthBG", 80); mThDepthObj = (float)num(kv, "ThDepthOBJ", 60); mDepthMapFactor = (float)num(kv, "DepthMapFactor", 1);
    if (mDepthMapFactor == 0) mDepthMapFactor = 1;
    nMaxTrackPointBG = (int)num(kv, "MaxTrackPointBG", 3000); nMaxTrackPointOBJ = (int)num(kv, "MaxTrackPointOBJ", 800);
    fSFMgThres = (float)num(kv, "SFMgThres", 0.12); fSFDsThres = (float)num(kv, "SFDsThres", 0.3);
    nWINDOW_SIZE = (int)num(kv, "WINDOW_SIZE", 20); nOVERLAP_SIZE = (int)num(kv, "OVERLAP_SIZE", 4); nUseSampleFea = (int)num(kv, "UseSampleFeature", 0);
    if (kv.count("Joint")) bJoint = num(kv, "Joint") != 0;
    if (kv.count("RansacSeed")) ransac_seed = (unsigned)num(kv, "RansacSeed");
    all_timing.assign(5, 0.f);
}
Tracking::~Tracking() { delete mpORBextractorLeft; }


cv::Mat Tracking::GrabImageRGBD(const cv::Mat& imRGB, cv::Mat& imD, const cv::Mat& imFlow, const cv::Mat& maskSEM, const cv::Mat&,
                                const std::vector<std::vector<float> >&, const double& timestamp, cv::Mat&, const int& nImage)
{
    const auto t_grab = std::chrono::steady_clock::now();
    StopFrame = nImage - 1; ms_wait_inputs = 0;
    if (mState == NO_IMAGES_YET) f_id = 0;
    if (imD.type() != CV_32FC1 || imFlow.type() != CV_32FC2 || maskSEM.type() != CV_32SC1 || !imD.isContinuous() || !imFlow.isContinuous() || !maskSEM.isContinuous())
        throw std::runtime_error("GrabImageRGBD: depth CV_32F, flow CV_32FC2, mask CV_32SC1 (continuous) expected");
    vido_ctx* c = mpORBextractorLeft->context(imRGB.cols, imRGB.rows);
    g_ctx = c;
    memset(&g_tp, 0, sizeof g_tp);
    g_tp.dataset = mTestData == OMD ? 0 : (mTestData == KITTI ? 1 : 2); g_tp.depth_map_factor = mDepthMapFactor; g_tp.bf = mbf; g_tp.kaist_scale = mScale;
    g_tp.th_depth_bg = mThDepth; g_tp.th_depth_obj = mThDepthObj; g_tp.dense_step = 4; g_tp.fx = mK.at<float>(0, 0); g_tp.fy = mK.at<float>(1, 1); g_tp.cx = mK.at<float>(0, 2); g_tp.cy = mK.at<float>(1, 2);
    slot_cur_ = (mState == NO_IMAGES_YET) ? 0 : 1 - slot_cur_; g_slot = slot_cur_;
    // depth pre-scale in place on the caller's buffer (:299-322) + maps resident in the slot
    check(vido_frame_upload(c, slot_cur_, 1, imD.ptr<float>(), imFlow.ptr<float>(), maskSEM.ptr<int32_t>(), 0, &g_tp), "frame_upload");
    if (imRGB.channels() == 1) mImGray = imRGB;
    else {                                                     // :327-340 cvtColor: done on the device by the extractor's ingest, which fills mImGray
        if (imRGB.type() != CV_8UC3 && imRGB.type() != CV_8UC4) throw std::runtime_error("GrabImageRGBD: image must be CV_8UC1 / CV_8UC3 / CV_8UC4");
        mImGray = cv::Mat(imRGB.rows, imRGB.cols, CV_8UC1);
        mpORBextractorLeft->SetColorSource(imRGB, mbRGB, mImGray);
    }
    mDepthMap = imD; mFlowMap = imFlow; mSegMap = maskSEM;      // shallow references like the reference keeps (Tracking.cc:343-345); the host-side stages read the device slot
    return GrabCommon(timestamp, nImage, (void*)&t_grab);
}

void Tracking::PrefetchImageDevice(const void* im_dev, int channels, int width, int height, void* image_ready_event)
{
    if (!im_dev || width < 1 || height < 1) throw std::runtime_error("PrefetchImageDevice: device image expected");
    mpORBextractorLeft->PrefetchDevice(im_dev, channels, mbRGB, width, height, image_ready_event);
}

cv::Mat Tracking::GrabImageRGBDDevice(const void* im_dev, int channels, int width, int height, float* depth_dev, const float* flow_dev, const int* mask_dev, void* ready_event,
                                      const double& timestamp, const int& nImage)
{
    const auto t_grab = std::chrono::steady_clock::now();
    StopFrame = nImage - 1;
    if (mState == NO_IMAGES_YET) f_id = 0;
    if (!im_dev || !depth_dev || !flow_dev || !mask_dev || width < 1 || height < 1 || (channels != 1 && channels != 3 && channels != 4))
        throw std::runtime_error("GrabImageRGBDDevice: device image (1 / 3 / 4 channels), depth, flow and mask pointers expected");
    vido_ctx* c = mpORBextractorLeft->context(width, height);
    g_ctx = c;
    ms_wait_inputs = 0;
    if (ready_event) {      // ordered behind the producer (the network stream) on the tracker's stream; the host then waits here, so that the wait for the frame's networks is a number
        check(vido_stream_wait_event(c, ready_event), "stream_wait_event");      // of its own (ms_wait_inputs) instead of sitting inside the first stage that synchronises (round 3: ~5 of the "6.1 ms ORB" were this wait)
        const auto tw = std::chrono::steady_clock::now();
        check(vido_synchronize(c), "synchronize");
        ms_wait_inputs = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - tw).count();
    }
    memset(&g_tp, 0, sizeof g_tp);
    g_tp.dataset = mTestData == OMD ? 0 : (mTestData == KITTI ? 1 : 2); g_tp.depth_map_factor = mDepthMapFactor; g_tp.bf = mbf; g_tp.kaist_scale = mScale;
    g_tp.th_depth_bg = mThDepth; g_tp.th_depth_obj = mThDepthObj; g_tp.dense_step = 4; g_tp.fx = mK.at<float>(0, 0); g_tp.fy = mK.at<float>(1, 1); g_tp.cx = mK.at<float>(0, 2); g_tp.cy = mK.at<float>(1, 2);
    slot_cur_ = (mState == NO_IMAGES_YET) ? 0 : 1 - slot_cur_; g_slot = slot_cur_;
    check(vido_frame_upload(c, slot_cur_, 1, depth_dev, flow_dev, (const int32_t*)mask_dev, detail::g_zero_copy_maps ? 2 : 1, &g_tp), "frame_upload");      // device -> slot (no PCIe), or the slot adopts the buffers; depth pre-scale in place
    mImGray = cv::Mat(height, width, CV_8UC1);                  // size carrier: the pixels stay on the device (ORBextractor::SetDeviceSource)
    mpORBextractorLeft->SetDeviceSource(im_dev, channels, mbRGB);
    mDepthMap = cv::Mat(); mFlowMap = cv::Mat(); mSegMap = cv::Mat();
    return GrabCommon(timestamp, nImage, (void*)&t_grab);
}

cv::Mat Tracking::GrabCommon(const double& timestamp, const int& nImage, void* t_grab_p)
{
    (void)nImage;
    const auto t_grab = *(const std::chrono::steady_clock::time_point*)t_grab_p;
    auto ms_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    vido_ctx* c = g_ctx;
    cv::Mat imD, imFlow;                                        // Frame's map arguments are unused (the slot holds the maps)
    all_timing.assign(5, 0.f);
    auto t_st = std::chrono::steady_clock::now();
    {   // VIDO_DIAG_FIRSTOP=n (diagnosis): n rounds of (trivial stream operation + host wait) in front of the frame's first stage (vido_debug_first_op prints the means at exit)
        static const int diag = [] { const char* e = getenv("VIDO_DIAG_FIRSTOP"); return e ? atoi(e) : 0; }();
        if (diag > 0 && mState != NO_IMAGES_YET) { (void)vido_debug_first_op(c, diag); t_st = std::chrono::steady_clock::now(); }
    }
    if (mState != NO_IMAGES_YET) UpdateMask();
    ms_update_mask = ms_since(t_st); t_st = std::chrono::steady_clock::now();
    mpCurrentFrame = new Frame(mImGray, imD, imFlow, mSegMap, timestamp, mpORBextractorLeft, mK, mDistCoef, mbf, mThDepth, mThDepthObj, nUseSampleFea);
    if (mState != NO_IMAGES_YET) {                             // :369-421
        Frame* F = mpCurrentFrame;
        F->mvStatKeys = mpLastFrame->mvCorres; F->N_s = (int)F->mvStatKeys.size();
        std::vector<float> xy(2 * std::max(F->N_s, 1)); F->mvStatDepth.assign(F->N_s, -1.f);
        for (int i = 0; i < F->N_s; i++) { xy[2 * i] = F->mvStatKeys[i].pt.x; xy[2 * i + 1] = F->mvStatKeys[i].pt.y; }
        if (F->N_s) check(vido_gather_static_depth(c, slot_cur_, xy.data(), F->N_s, F->mvStatDepth.data()), "gather_static_depth");
        mvTmpObjKeys = F->mvObjKeys; mvTmpObjDepth = F->mvObjDepth; mvTmpSemObjLabel = F->vSemObjLabel; mvTmpObjFlowNext = F->mvObjFlowNext; mvTmpObjCorres = F->mvObjCorres;
        F->mvObjKeys = mpLastFrame->mvObjCorres;
        const int no = (int)F->mvObjKeys.size();
        F->mvObjDepth.assign(no, -1.f); F->vSemObjLabel.assign(no, -1);
        std::vector<float> oxy(2 * std::max(no, 1));
        for (int i = 0; i < no; i++) { oxy[2 * i] = F->mvObjKeys[i].pt.x; oxy[2 * i + 1] = F->mvObjKeys[i].pt.y; }
        if (no) check(vido_gather_object_depth_label(c, slot_cur_, oxy.data(), no, mThDepthObj, F->mvObjDepth.data(), F->vSemObjLabel.data()), "gather_object");
        TemperalMatch.assign(F->N_s, -1);
    }
    mpCurrentFrame->vObjLabel.assign(mpCurrentFrame->mvObjKeys.size(), -2);
    ms_frame = ms_since(t_st);
    Track();
    f_id = f_id + 1;
    mImGrayLast = mImGray; mSegMapLast = mSegMap; mFlowMapLast = mFlowMap;      // :777-780
    ms_total = ms_since(t_grab);
    return mpCurrentFrame->mTcw.clone();
}

void Tracking::UpdateMask()                                   // Tracking.cc:3291-3357, device scatter
{
    Frame* L = mpLastFrame; const int n = (int)L->vSemObjLabel.size();
    if (n == 0) return;
    std::vector<float> corr(2 * n);
    for (int i = 0; i < n; i++) { corr[2 * i] = L->mvObjCorres[i].pt.x; corr[2 * i + 1] = L->mvObjCorres[i].pt.y; }
    int32_t rec[64], nrec = 0;
    check(vido_update_mask(g_ctx, 1 - slot_cur_, slot_cur_, L->vSemObjLabel.data(), corr.data(), n, rec, 64, &nrec), "update_mask");
    (void)nrec;                                                 // the recovered labels live in the device slot's mask; RenewFrameInfo samples it there
}

void Tracking::Initialization()                               // Tracking.cc:1512-1580
{
    Frame* F = mpCurrentFrame;
    for (size_t i = 0; i < F->mvStatKeysTmp.size(); i++) F->mvStat3DPointTmp.push_back(Optimizer::Get3DinCamera(F->mvStatKeysTmp[i], F->mvStatDepthTmp[i], mK));
    for (size_t i = 0; i < F->mvObjKeys.size(); i++) F->mvObj3DPoint.push_back(Optimizer::Get3DinCamera(F->mvObjKeys[i], F->mvObjDepth[i], mK));
    mpMap->vpFeatSta.push_back(F->mvStatKeysTmp); mpMap->vfDepSta.push_back(F->mvStatDepthTmp); mpMap->vp3DPointSta.push_back(F->mvStat3DPointTmp);
    mpMap->vpFeatDyn.push_back(F->mvObjKeys); mpMap->vfDepDyn.push_back(F->mvObjDepth); mpMap->vp3DPointDyn.push_back(F->mvObj3DPoint);
    mpMap->vmCameraPose.push_back(cv::Mat::eye(4, 4, CV_32F)); mpMap->vmCameraPose_RF.push_back(cv::Mat::eye(4, 4, CV_32F)); mpMap->vmCameraPose_GT.push_back(cv::Mat::eye(4, 4, CV_32F));
    F->SetPose(cv::Mat::eye(4, 4, CV_32F));
    mpLastFrame = F; mpLastFrame->mvStatKeys = F->mvStatKeysTmp; mpLastFrame->mvStatDepth = F->mvStatDepthTmp; mpLastFrame->N_s = F->N_s_tmp;
    mState = OK;
}

static float reproj(const cv::Mat& T, const cv::Point3f& X, const cv::Point2f& x, float fx, float fy, float cx, float cy)
{
    const float xc = T.at<float>(0, 0) * X.x + T.at<float>(0, 1) * X.y + T.at<float>(0, 2) * X.z + T.at<float>(0, 3);
    const float yc = T.at<float>(1, 0) * X.x + T.at<float>(1, 1) * X.y + T.at<float>(1, 2) * X.z + T.at<float>(1, 3);
    const float invzc = 1.0f / (T.at<float>(2, 0) * X.x + T.at<float>(2, 1) * X.y + T.at<float>(2, 2) * X.z + T.at<float>(2, 3));
    const float u_ = x.x - (fx * xc * invzc + cx), v_ = x.y - (fy * yc * invzc + cy);
    return std::sqrt(u_ * u_ + v_ * v_);
}

cv::Mat Tracking::GetInitModelCam(const std::vector<int>& MatchId, std::vector<int>& MatchId_sub)   // Tracking.cc:1914-2028
{
    const int N = (int)MatchId.size(); Frame *C = mpCurrentFrame, *L = mpLastFrame;
    std::vector<cv::Point2f> cur_2d(N); std::vector<cv::Point3f> pre_3d(N); std::vector<int> outl(N, 0);
    for (int i = 0; i < N; i++) {
        cur_2d[i] = C->mvStatKeys[MatchId[i]].pt;
        float X[3];
        if (!stat3d_raw(*L, MatchId[i], X)) { outl[i] = -1; continue; }      // Frame::UnprojectStereoStat
        pre_3d[i] = cv::Point3f(X[0], X[1], X[2]);
    }
    std::vector<float> g3, g2; std::vector<int> gidx;
    for (int i = 0; i < N; i++) if (outl[i] == 0) { g3.push_back(pre_3d[i].x); g3.push_back(pre_3d[i].y); g3.push_back(pre_3d[i].z); g2.push_back(cur_2d[i].x); g2.push_back(cur_2d[i].y); gidx.push_back(i); }
    double T[16]; int32_t ninl = 0; std::vector<uint8_t> mask(std::max<size_t>(gidx.size(), 1));
    check(vido_pnp_ransac(live_ctx("GetInitModelCam"), g3.data(), g2.data(), (int)gidx.size(), C->fx, C->fy, C->cx, C->cy, 500, 0.4, 0.98, ransac_seed + (unsigned)f_id, T, mask.data(), &ninl), "pnp_ransac");
    cv::Mat Mod = fromRow16(T);
    cv::Mat MotionModel = mVelocity.empty() ? L->mTcw.clone() : mVelocity * L->mTcw;
    std::vector<int> MM_inlier;
    for (int i = 0; i < N; i++) if (reproj(MotionModel, pre_3d[i], cur_2d[i], C->fx, C->fy, C->cx, C->cy) < 0.4f) MM_inlier.push_back(i);
    MatchId_sub.clear();
    if (ninl > (int)MM_inlier.size()) { for (size_t k = 0; k < gidx.size(); k++) if (mask[k]) MatchId_sub.push_back(MatchId[gidx[k]]); return Mod; }
    for (int i : MM_inlier) MatchId_sub.push_back(MatchId[i]);
    return MotionModel;
}

// GetInitModelObj in two halves so that the RANSAC of all objects of a frame can run as one batch: inputs, then the comparison with the previous motion
namespace {
struct ObjInit { std::vector<cv::Point2f> cur_2d; std::vector<cv::Point3f> pre_3d; std::vector<float> g3, g2; std::vector<uint8_t> mask; double T[16]; int32_t ninl = 0; };
void init_obj_inputs(ObjInit& o, Frame* C, Frame* L, const std::vector<int>& ObjId)
{
    const int N = (int)ObjId.size();
    o.cur_2d.resize(N); o.pre_3d.resize(N); o.g3.resize(3 * N); o.g2.resize(2 * N); o.mask.assign(std::max(N, 1), 0);
    for (int i = 0; i < N; i++) {
        o.cur_2d[i] = C->mvObjKeys[ObjId[i]].pt;
        float X[3];
        o.pre_3d[i] = obj3d_raw(*L, ObjId[i], X) ? cv::Point3f(X[0], X[1], X[2]) : cv::Point3f(0, 0, 1);      // Frame::UnprojectStereoObject
        o.g3[3 * i] = o.pre_3d[i].x; o.g3[3 * i + 1] = o.pre_3d[i].y; o.g3[3 * i + 2] = o.pre_3d[i].z; o.g2[2 * i] = o.cur_2d[i].x; o.g2[2 * i + 1] = o.cur_2d[i].y;
    }
}
cv::Mat init_obj_decide(const ObjInit& o, Frame* C, Frame* L, const std::vector<int>& ObjId, std::vector<int>& ObjId_sub, int objid)    // Tracking.cc:2076-2162
{
    const int N = (int)ObjId.size();
    cv::Mat Mod = fromRow16(o.T);
    const int CurObjLab = C->nModLabel[objid]; int PreObjID = -1;
    for (size_t i = 0; i < L->nModLabel.size(); i++) if (L->nModLabel[i] == CurObjLab) { PreObjID = (int)i; break; }
    ObjId_sub.clear();
    if (PreObjID != -1 && PreObjID < (int)L->vObjMod.size() && !L->vObjMod[PreObjID].empty()) {
        cv::Mat MotionModel = C->mTcw * L->vObjMod[PreObjID];
        std::vector<int> MM;
        for (int i = 0; i < N; i++) if (reproj(MotionModel, o.pre_3d[i], o.cur_2d[i], C->fx, C->fy, C->cx, C->cy) < 0.4f) MM.push_back(i);
        if (o.ninl > (int)MM.size()) { for (int i = 0; i < N; i++) if (o.mask[i]) ObjId_sub.push_back(ObjId[i]); return Mod; }
        for (int i : MM) ObjId_sub.push_back(ObjId[i]);
        return MotionModel;
    }
    for (int i = 0; i < N; i++) if (o.mask[i]) ObjId_sub.push_back(ObjId[i]);      // no previous motion: RANSAC model (:2136-2147)
    return Mod;
}
}  // namespace

cv::Mat Tracking::GetInitModelObj(const std::vector<int>& ObjId, std::vector<int>& ObjId_sub, const int objid)   // Tracking.cc:2030-2162
{
    Frame *C = mpCurrentFrame, *L = mpLastFrame;
    ObjInit o; init_obj_inputs(o, C, L, ObjId);
    check(vido_pnp_ransac(live_ctx("GetInitModelObj"), o.g3.data(), o.g2.data(), (int)ObjId.size(), C->fx, C->fy, C->cx, C->cy, 500, 0.4, 0.98, ransac_seed + 7919u * (unsigned)(objid + 1) + (unsigned)f_id, o.T, o.mask.data(), &o.ninl), "pnp_ransac(obj)");
    return init_obj_decide(o, C, L, ObjId, ObjId_sub, objid);
}

// GetInitModelObj for every object of the frame: one batched RANSAC launch, one synchronisation; same seeds, same results as the per-object calls
std::vector<cv::Mat> Tracking::GetInitModelObjBatch(const std::vector<std::vector<int> >& ObjIds, std::vector<std::vector<int> >& ObjIds_sub)
{
    Frame *C = mpCurrentFrame, *L = mpLastFrame; const size_t n = ObjIds.size();
    std::vector<ObjInit> O(n); std::vector<const float*> p3(n), p2(n); std::vector<uint8_t*> mk(n); std::vector<int32_t> nn(n), ninl(n); std::vector<uint64_t> seeds(n); std::vector<double> T(16 * std::max<size_t>(n, 1));
    for (size_t i = 0; i < n; i++) {
        init_obj_inputs(O[i], C, L, ObjIds[i]);
        p3[i] = O[i].g3.data(); p2[i] = O[i].g2.data(); mk[i] = O[i].mask.data(); nn[i] = (int32_t)ObjIds[i].size();
        seeds[i] = ransac_seed + 7919u * (unsigned)(i + 1) + (unsigned)f_id;
    }
    if (n) check(vido_pnp_ransac_batch(live_ctx("GetInitModelObjBatch"), (int)n, p3.data(), p2.data(), nn.data(), C->fx, C->fy, C->cx, C->cy, 500, 0.4, 0.98, seeds.data(), T.data(), mk.data(), ninl.data()), "pnp_ransac_batch");
    std::vector<cv::Mat> out(n); ObjIds_sub.assign(n, std::vector<int>());
    for (size_t i = 0; i < n; i++) {
        memcpy(O[i].T, &T[16 * i], sizeof O[i].T); O[i].ninl = ninl[i];
        if (nn[i] < 4) { for (int k = 0; k < 16; k++) O[i].T[k] = (k % 5 == 0); O[i].ninl = 0; }
        out[i] = init_obj_decide(O[i], C, L, ObjIds[i], ObjIds_sub[i], (int)i);
    }
    return out;
}

void Tracking::GetSceneFlowObj()                              // Tracking.cc:1582-1668
{
    Frame *C = mpCurrentFrame, *L = mpLastFrame; const int N = (int)C->mvObjKeys.size();
    C->vFlow_3d.assign(N, cv::Point3f(0, 0, 0));
    if (N == 0) return;
    std::vector<float> kc(2 * N), kl(2 * N), xl(3 * N), xc(3 * N), f3(3 * N);
    for (int i = 0; i < N; i++) { kc[2 * i] = C->mvObjKeys[i].pt.x; kc[2 * i + 1] = C->mvObjKeys[i].pt.y; kl[2 * i] = L->mvObjKeys[i].pt.x; kl[2 * i + 1] = L->mvObjKeys[i].pt.y; }
    check(vido_unproject_world(g_ctx, kl.data(), L->mvObjDepth.data(), N, &g_tp, L->mTcw.ptr<float>(), xl.data()), "unproject(last)");
    check(vido_unproject_world(g_ctx, kc.data(), C->mvObjDepth.data(), N, &g_tp, C->mTcw.ptr<float>(), xc.data()), "unproject(cur)");
    check(vido_scene_flow(g_ctx, xl.data(), xc.data(), L->vSemObjLabel.data(), C->vSemObjLabel.data(), N, f3.data(), C->vObjLabel.data()), "scene_flow");
    for (int i = 0; i < N; i++) C->vFlow_3d[i] = cv::Point3f(f3[3 * i], f3[3 * i + 1], f3[3 * i + 2]);
}

std::vector<std::vector<int> > Tracking::DynObjTracking()     // Tracking.cc:1670-1912: vido_dyn_obj_tracking (trackhost.cpp) on the frame's flat lists
{
    Frame *C = mpCurrentFrame, *L = mpLastFrame; const int n = (int)C->vSemObjLabel.size();
    std::vector<float> xy(2 * std::max(n, 1)), f3(3 * std::max(n, 1));
    for (int i = 0; i < n; i++) { xy[2 * i] = C->mvObjKeys[i].pt.x; xy[2 * i + 1] = C->mvObjKeys[i].pt.y; f3[3 * i] = C->vFlow_3d[i].x; f3[3 * i + 1] = C->vFlow_3d[i].y; f3[3 * i + 2] = C->vFlow_3d[i].z; }
    const int nl = (int)L->nSemPosition.size();
    std::vector<uint8_t> lstat(std::max(nl, 1)); for (int k = 0; k < nl; k++) lstat[k] = k < (int)L->bObjStat.size() && L->bObjStat[k] ? 1 : 0;
    std::vector<int32_t> off(n + 2), ids(n + 1), ml(n + 1), sp(n + 1); int32_t nobj = 0, mid = max_id;
    const int rc = vido_dyn_obj_tracking(C->vSemObjLabel.data(), C->vObjLabel.data(), xy.data(), C->mvObjDepth.data(), f3.data(), L->vSemObjLabel.data(), n,
                                         L->nSemPosition.data(), lstat.data(), L->nModLabel.data(), nl, mImGray.rows, mImGray.cols, fSFMgThres, fSFDsThres, mThDepthObj, f_id, &mid,
                                         off.data(), ids.data(), ml.data(), sp.data(), n + 1, &nobj);
    if (rc != VIDO_OK) throw std::runtime_error("DynObjTracking failed");
    max_id = mid;
    std::vector<std::vector<int> > ObjIdNew(nobj);
    for (int i = 0; i < nobj; i++) ObjIdNew[i].assign(ids.begin() + off[i], ids.begin() + off[i + 1]);
    C->nModLabel.assign(ml.begin(), ml.begin() + nobj); C->nSemPosition.assign(sp.begin(), sp.begin() + nobj);
    return ObjIdNew;
}

std::vector<std::vector<std::pair<int, int> > > Tracking::GetStaticTrack()     // Tracking.cc:2514-2613
{
    const auto& TM = mpMap->vnAssoSta; const int N = (int)TM.size();
    std::vector<int> pre; std::vector<std::vector<std::pair<int, int> > > T;
    for (int i = 0; i < N; i++) {
        std::vector<int> cur(TM[i].size(), -1);
        for (size_t j = 0; j < TM[i].size(); j++) {
            const int m = TM[i][j]; if (m == -1) continue;
            if (i > 0 && m < (int)pre.size() && pre[m] != -1) { T[pre[m]].push_back(std::make_pair(i + 1, (int)j)); cur[j] = pre[m]; }
            else { T.push_back({std::make_pair(i, m), std::make_pair(i + 1, (int)j)}); cur[j] = (int)T.size() - 1; }
        }
        pre = cur;
    }
    return T;
}
std::vector<std::vector<std::pair<int, int> > > Tracking::GetDynamicTrackNew()  // Tracking.cc:2615-2720
{
    const auto& TM = mpMap->vnAssoDyn; const auto& Lab = mpMap->vnFeatLabel; const int N = (int)TM.size();
    std::vector<int> pre, ObjectID; std::vector<std::vector<std::pair<int, int> > > T;
    for (int i = 0; i < N; i++) {
        std::vector<int> cur(TM[i].size(), -1);
        for (size_t j = 0; j < TM[i].size(); j++) {
            const int m = TM[i][j]; if (m == -1) continue;
            if (i > 0 && m < (int)pre.size() && pre[m] != -1) { T[pre[m]].push_back(std::make_pair(i + 1, (int)j)); cur[j] = pre[m]; }
            else { T.push_back({std::make_pair(i, m), std::make_pair(i + 1, (int)j)}); ObjectID.push_back(Lab[i][j]); cur[j] = (int)T.size() - 1; }
        }
        pre = cur;
    }
    mpMap->nObjID = ObjectID;          // the same list UpdateTracklets() maintains
    return T;
}

void Tracking::RenewFrameInfo(const std::vector<int>& TM_sta)  // Tracking.cc:2959-3289: vido_renew_static / vido_renew_objects (trackhost.cpp) + depth / 3-D point assembly
{
    Frame* C = mpCurrentFrame; const int W = mImGray.cols, H = mImGray.rows;
    // ---- static features (:2973-3105)
    const std::vector<cv::KeyPoint> sample_src = nUseSampleFea == 1 ? C->mvStatKeysTmp : C->mvKeys;      // Tracking.cc:3013-3018 (a copy: mvStatKeysTmp is rewritten below)
    const int ns = (int)C->mvStatKeys.size(), nk = (int)sample_src.size();
    std::vector<float> sxy(2 * std::max(ns, 1)), kxy(2 * std::max(nk, 1));
    for (int i = 0; i < ns; i++) { sxy[2 * i] = C->mvStatKeys[i].pt.x; sxy[2 * i + 1] = C->mvStatKeys[i].pt.y; }
    for (int i = 0; i < nk; i++) { kxy[2 * i] = sample_src[i].pt.x; kxy[2 * i + 1] = sample_src[i].pt.y; }
    // the values of mSegMap / mDepthMap / mFlowMap at the candidate points of both parts, from the device slot in ONE gather: [static list | sample source | object points]
    // (the maps themselves never come to the host; the slot's mask carries UpdateMask's recovered labels, its depth is the pre-scaled one)
    const int no_pts = (int)C->mvObjKeys.size(), n_all = ns + nk + no_pts;
    std::vector<float> axy(2 * (size_t)std::max(n_all, 1)), sdep(std::max(n_all, 1)), sflo(2 * (size_t)std::max(n_all, 1)); std::vector<int32_t> smsk(std::max(n_all, 1));
    memcpy(axy.data(), sxy.data(), sizeof(float) * 2 * ns); memcpy(axy.data() + 2 * (size_t)ns, kxy.data(), sizeof(float) * 2 * nk);
    for (int i = 0; i < no_pts; i++) { axy[2 * (size_t)(ns + nk + i)] = C->mvObjKeys[i].pt.x; axy[2 * (size_t)(ns + nk + i) + 1] = C->mvObjKeys[i].pt.y; }
    check(vido_gather_point_samples(g_ctx, slot_cur_, axy.data(), n_all, smsk.data(), sdep.data(), sflo.data()), "gather_point_samples");
    const vido_point_samples ss = {smsk.data(), sdep.data(), sflo.data()}, ks = {smsk.data() + ns, sdep.data() + ns, sflo.data() + 2 * (size_t)ns},
                             os = {smsk.data() + ns + nk, sdep.data() + ns + nk, sflo.data() + 2 * (size_t)(ns + nk)};
    const int cap = (int)TM_sta.size() + nk + 8;
    std::vector<int32_t> src(cap), inl(cap); std::vector<float> fl(2 * (size_t)cap); int32_t n = 0;
    { ProfSection ps_("renew: vido_renew_static_sampled (host)");
    if (vido_renew_static_sampled(W, H, sxy.data(), ns, &ss, TM_sta.data(), (int)TM_sta.size(), kxy.data(), nk, &ks, nMaxTrackPointBG, src.data(), inl.data(), fl.data(), cap, &n) != VIDO_OK)
        throw std::runtime_error("RenewFrameInfo: vido_renew_static failed"); }
    ProfSection* ps_a = new ProfSection("renew: static lists + 3-D points (host)");
    std::vector<cv::KeyPoint> keys(n), corres(n); std::vector<cv::Point2f> flows(n); std::vector<int> inlierID(n);
    for (int k = 0; k < n; k++) {
        keys[k] = inl[k] >= 0 ? C->mvStatKeys[src[k]] : sample_src[src[k]];
        flows[k] = cv::Point2f(fl[2 * k], fl[2 * k + 1]); inlierID[k] = inl[k];
        corres[k] = cv::KeyPoint(keys[k].pt.x + fl[2 * k], keys[k].pt.y + fl[2 * k + 1], 0, 0, 0, -1);
    }
    C->N_s_tmp = n;
    std::vector<float> depth(n, -1.f); std::vector<cv::Mat> p3d; std::vector<float> xyz3(3 * (size_t)std::max(n, 1));
    const cv::Mat Twc = Converter::toInvMatrix(C->mTcw);
    for (int i = 0; i < n; i++) { const float d = inl[i] >= 0 ? ss.depth[src[i]] : ks.depth[src[i]]; if (d > 0) depth[i] = d; get3d_world(keys[i], depth[i], mK, Twc, &xyz3[3 * (size_t)i]); }      // mDepthMap.at(key); Get3DinWorld per point
    cv::Mat::batch3x1(xyz3.data(), n, p3d);                                       // (one block for the frame's points instead of n allocations)
    C->nStaInlierID = std::move(inlierID); C->mvStatKeysTmp = std::move(keys); C->mvStatDepthTmp = std::move(depth); C->mvStat3DPointTmp = std::move(p3d); C->mvFlowNext = std::move(flows); C->mvCorres = std::move(corres);
    delete ps_a;

    // ---- objects (:3116-3289)
    const int no = (int)C->mvObjKeys.size(), nobj = (int)C->vnObjInlierID.size(), nt = (int)mvTmpSemObjLabel.size();
    std::vector<float> oxy(2 * std::max(no, 1)), txy(2 * std::max(nt, 1)), tfl(2 * std::max(nt, 1)), tco(2 * std::max(nt, 1));
    for (int i = 0; i < no; i++) { oxy[2 * i] = C->mvObjKeys[i].pt.x; oxy[2 * i + 1] = C->mvObjKeys[i].pt.y; }
    for (int j = 0; j < nt; j++) { txy[2 * j] = mvTmpObjKeys[j].pt.x; txy[2 * j + 1] = mvTmpObjKeys[j].pt.y; tfl[2 * j] = mvTmpObjFlowNext[j].x; tfl[2 * j + 1] = mvTmpObjFlowNext[j].y;
                                   tco[2 * j] = mvTmpObjCorres[j].pt.x; tco[2 * j + 1] = mvTmpObjCorres[j].pt.y; }
    std::vector<int32_t> ioff(nobj + 1, 0), iids; std::vector<uint8_t> ostat(std::max(nobj, 1));
    for (int i = 0; i < nobj; i++) { ioff[i + 1] = ioff[i] + (int)C->vnObjInlierID[i].size(); iids.insert(iids.end(), C->vnObjInlierID[i].begin(), C->vnObjInlierID[i].end()); ostat[i] = C->bObjStat[i] ? 1 : 0; }
    if (iids.empty()) iids.push_back(0);
    const int ocap = (int)iids.size() + (nobj + 1) * nt + 8;
    // output scratch sized for the worst case (every dense sample kept for every object: ~0.5 MB per array) — kept across frames: as fresh vectors each was an mmap + page
    // faults + a fill per frame (glibc serves blocks > 128 KB from mmap), 0.2 ms of this function (round 5)
    static thread_local std::vector<float> okx, odep, ofl, oco; static thread_local std::vector<int32_t> osem, oinl, olab; int32_t on = 0;
    if ((int)odep.size() < ocap) { okx.resize(2 * (size_t)ocap); odep.resize(ocap); ofl.resize(2 * (size_t)ocap); oco.resize(2 * (size_t)ocap); osem.resize(ocap); oinl.resize(ocap); olab.resize(ocap); }
    ProfSection* ps_b = new ProfSection("renew: vido_renew_objects_sampled (host)");
    if (vido_renew_objects_sampled(W, H, oxy.data(), C->vObjLabel.data(), no, &os, nobj, ioff.data(), iids.data(), ostat.data(), C->nSemPosition.data(), C->nModLabel.data(),
                           txy.data(), mvTmpObjDepth.data(), mvTmpSemObjLabel.data(), tfl.data(), tco.data(), nt, nMaxTrackPointOBJ,
                           okx.data(), odep.data(), osem.data(), ofl.data(), oco.data(), oinl.data(), olab.data(), ocap, &on) != VIDO_OK)
        throw std::runtime_error("RenewFrameInfo: vido_renew_objects failed");
    delete ps_b;
    ProfSection ps_c("renew: object lists + 3-D points (host)");
    std::vector<cv::KeyPoint> okeys(on), ocorr(on); std::vector<float> odepth(odep.begin(), odep.begin() + on); std::vector<cv::Point2f> oflow(on);
    std::vector<cv::Mat> o3d; std::vector<float> oxyz3(3 * (size_t)std::max(on, 1));
    for (int i = 0; i < on; i++) {
        okeys[i] = cv::KeyPoint(okx[2 * i], okx[2 * i + 1], 0, 0, 0, -1); ocorr[i] = cv::KeyPoint(oco[2 * i], oco[2 * i + 1], 0, 0, 0, -1); oflow[i] = cv::Point2f(ofl[2 * i], ofl[2 * i + 1]);
        get3d_world(okeys[i], odepth[i], mK, Twc, &oxyz3[3 * (size_t)i]);
    }
    cv::Mat::batch3x1(oxyz3.data(), on, o3d);
    C->mvObjKeys = std::move(okeys); C->mvObjDepth = std::move(odepth); C->mvObj3DPoint = std::move(o3d); C->mvObjCorres = std::move(ocorr); C->mvObjFlowNext = std::move(oflow);
    C->vSemObjLabel.assign(osem.begin(), osem.begin() + on); C->nDynInlierID.assign(oinl.begin(), oinl.begin() + on); C->vObjLabel.assign(olab.begin(), olab.begin() + on);
}

void Tracking::Track()                                        // Tracking.cc:1081-1509
{
    LBA_TRACE("tracker: Track begin");
    if (mState == NO_IMAGES_YET) mState = NOT_INITIALIZED;
    Frame* C = mpCurrentFrame;
    if (mState == NOT_INITIALIZED) { Initialization(); if (mState != OK) return; }
    else {
        Frame* L = mpLastFrame;
        for (int i = 0; i < C->N_s; i++) TemperalMatch[i] = i;
        if (TemperalMatch.size() < 2) { C->SetPose(L->mTcw); return; }
        auto t0 = std::chrono::steady_clock::now();
        cv::Mat iniTcw = GetInitModelCam(TemperalMatch, TemperalMatch_subset);
        C->SetPose(iniTcw);
        if (bJoint) Optimizer::PoseOptimizationFlow2Cam(C, L, TemperalMatch_subset); else Optimizer::PoseOptimizationNew(C, L, TemperalMatch_subset);
        all_timing[1] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (!L->mTcw.empty()) mVelocity = C->mTcw * Converter::toInvMatrix(L->mTcw);
        GetSceneFlowObj();
        t0 = std::chrono::steady_clock::now();
        std::vector<std::vector<int> > ObjIdNew = DynObjTracking();
        all_timing[2] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        const size_t no = ObjIdNew.size();
        C->bObjStat.assign(no, true); C->vObjMod.resize(no); C->vSpeed.resize(no); C->vObjCentre3D.resize(no); C->vnObjID.resize(no); C->vnObjInlierID.resize(no);
        t0 = std::chrono::steady_clock::now();
        // the reference's per-object loop (:1192-1305) as three batched stages: initial models of all objects (one RANSAC launch), motion
        // optimisation of all objects that kept >= 50 inliers (one LM launch), then the per-object bookkeeping
        std::vector<cv::Mat> centres(no);
        for (size_t i = 0; i < no; i++) {
            cv::Mat centre = vec3(0, 0, 0);
            for (int id : ObjIdNew[i]) { float X[3]; if (!obj3d_raw(*L, id, X)) continue; for (int a = 0; a < 3; a++) centre.at<float>(a) += X[a]; }
            for (int a = 0; a < 3; a++) centre.at<float>(a) /= (float)ObjIdNew[i].size();
            centres[i] = centre; C->vObjCentre3D[i] = centre; C->vnObjID[i] = ObjIdNew[i];
        }
        std::vector<std::vector<int> > in_ids_all;
        std::vector<cv::Mat> init_models = GetInitModelObjBatch(ObjIdNew, in_ids_all);
        std::vector<std::vector<int> > opt_ids; std::vector<cv::Mat> opt_init; std::vector<size_t> opt_obj;
        for (size_t i = 0; i < no; i++) {
            if (in_ids_all[i].size() < 50) { C->bObjStat[i] = false; C->vObjMod[i] = cv::Mat::eye(4, 4, CV_32F); C->vObjCentre3D[i] = vec3(0, 0, 0); C->vSpeed[i] = cv::Point2f(0, 0); C->vnObjInlierID[i] = in_ids_all[i]; continue; }
            opt_ids.push_back(in_ids_all[i]); opt_init.push_back(init_models[i]); opt_obj.push_back(i);
        }
        std::vector<std::vector<int> > inl_all;
        std::vector<cv::Mat> mods = Optimizer::PoseOptimizationObjectsBatch(C, L, opt_ids, opt_init, inl_all, bJoint);
        if (!init_models.empty()) C->mInitModel = init_models.back();
        for (size_t k = 0; k < opt_obj.size(); k++) {
            const size_t i = opt_obj[k]; const cv::Mat& centre = centres[i];
            C->vObjMod[i] = bJoint ? Converter::toInvMatrix(C->mTcw) * mods[k] : mods[k];
            C->vnObjInlierID[i] = inl_all[k];
            const cv::Mat& Hm = C->vObjMod[i]; float v[3];       // speed = |t - (I - R) c| * 36   (:1295-1302)
            for (int r = 0; r < 3; r++) { float s = Hm.at<float>(r, 3); for (int c2 = 0; c2 < 3; c2++) s -= ((r == c2 ? 1.f : 0.f) - Hm.at<float>(r, c2)) * centre.at<float>(c2); v[r] = s; }
            C->vSpeed[i].x = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) * 36 * 36;
        }
        ms_obj_motion_sum = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        all_timing[3] = no ? ms_obj_motion_sum / no : 0.f;
        t0 = std::chrono::steady_clock::now();
        RenewFrameInfo(TemperalMatch_subset);
        all_timing[4] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        finish_local_ba();                                 // the previous frame's window solve: everything below appends to the Map it read
        mpMap->vfAll_time.push_back(all_timing);
        C->mpPrevFrame = L; L->mpNextFrame = C;
        mpLastFrame = C; mpLastFrame->mvStatKeys = C->mvStatKeysTmp; mpLastFrame->mvStatDepth = C->mvStatDepthTmp; mpLastFrame->N_s = C->N_s_tmp;
        mpMap->vpFeatSta.push_back(C->mvStatKeysTmp); mpMap->vfDepSta.push_back(C->mvStatDepthTmp); mpMap->vp3DPointSta.push_back(C->mvStat3DPointTmp); mpMap->vnAssoSta.push_back(C->nStaInlierID);
        mpMap->vpFeatDyn.push_back(C->mvObjKeys); mpMap->vfDepDyn.push_back(C->mvObjDepth); mpMap->vp3DPointDyn.push_back(C->mvObj3DPoint); mpMap->vnAssoDyn.push_back(C->nDynInlierID); mpMap->vnFeatLabel.push_back(C->vObjLabel);
        mpMap->UpdateTracklets();                      // incremental form of GetStaticTrack() / GetDynamicTrackNew() (Tracking.cc:1395-1400)
        cv::Mat Twc = Converter::toInvMatrix(C->mTcw);
        mpMap->vmCameraPose.push_back(Twc); mpMap->vmCameraPose_RF.push_back(Twc);
        std::vector<cv::Mat> Mot, Cen; std::vector<int> MotLab, SemLab; std::vector<bool> Stat;
        Mot.push_back(Converter::toInvMatrix(mVelocity)); MotLab.push_back(0); SemLab.push_back(0); Stat.push_back(true); Cen.push_back(vec3(0, 0, 0));
        for (size_t i = 0; i < C->vObjMod.size(); i++) { if (!C->bObjStat[i]) continue; Stat.push_back(true); Mot.push_back(C->vObjMod[i]); MotLab.push_back(C->nModLabel[i]); SemLab.push_back(C->nSemPosition[i]); Cen.push_back(C->vObjCentre3D[i]); }
        mpMap->vmRigidMotion.push_back(Mot); mpMap->vmRigidMotion_RF.push_back(Mot); mpMap->vnRMLabel.push_back(MotLab); mpMap->vnSMLabel.push_back(SemLab); mpMap->vbObjStat.push_back(Stat); mpMap->vmRigidCentre.push_back(Cen);
        mpMap->AddFrame(C);
        // local batch optimisation every frame (:1430-1451)
        const int window = f_id < nWINDOW_SIZE ? f_id : nWINDOW_SIZE;
        LBA_TRACE("tracker: start_local_ba");
        start_local_ba(mpMap, mK, window);                 // (fLBA_time gets this solve's duration when it is joined)
    }
    if (f_id == StopFrame && mTestData == KITTI) { Optimizer::FullBatchOptimization(mpMap, mK); f_id = 0; }   // :1489-1498
    mState = OK;
}

// ---- System ------------------------------------------------------------------------------------------------------------------------
System::~System() { try { finish_local_ba(); } catch (...) { } delete mpTracker; delete mpMap; }
void System::Init(const std::string& strSettingsFile, const eSensor sensor)    // System.cc:23-48
{
    mSensor = sensor;
    if (sensor != RGBD) throw std::runtime_error("System::Init: only the RGBD sensor path is built (IMU_RGBD / VIO is out of scope)");
    mpMap = new Map();
    mpTracker = new Tracking(this, mpMap, strSettingsFile, (int)sensor);
}
cv::Mat System::TrackRGBD(const cv::Mat& im, cv::Mat& depthmap, const cv::Mat& flowmap, const cv::Mat& masksem, const cv::Mat& Tgt,
                          const std::vector<std::vector<float> >& vObjPose_gt, const double& ts, cv::Mat& imTraj, const int& nImage)
{
    if (mSensor != RGBD) throw std::runtime_error("ERROR: you called TrackRGBD but input sensor was not set to RGBD.");
    // the realtime demo hands over the services' wire types (run_vido.cc:57-110: depth MONO16, mask MONO8); the offline one converts first
    // (run_vido_slam.cc:96-118).  Both are accepted: 16U depth -> 32F (converted copy becomes the caller's Mat, like convertTo in place), 8U mask -> 32S.
    if (depthmap.type() == CV_16UC1) { cv::Mat d32; depthmap.convertTo(d32, CV_32F); depthmap = d32; }
    if (masksem.type() == CV_8UC1) { cv::Mat m32; masksem.convertTo(m32, CV_32SC1); return mpTracker->GrabImageRGBD(im, depthmap, flowmap, m32, Tgt, vObjPose_gt, ts, imTraj, nImage); }
    return mpTracker->GrabImageRGBD(im, depthmap, flowmap, masksem, Tgt, vObjPose_gt, ts, imTraj, nImage);
}
cv::Mat System::TrackRGBDDevice(const void* im_dev, int channels, int width, int height, float* depth_dev, const float* flow_dev, const int* mask_dev, void* ready_event,
                                const double& ts, const int& nImage)
{
    if (mSensor != RGBD) throw std::runtime_error("ERROR: you called TrackRGBDDevice but input sensor was not set to RGBD.");
    return mpTracker->GrabImageRGBDDevice(im_dev, channels, width, height, depth_dev, flow_dev, mask_dev, ready_event, ts, nImage);
}
void System::PrefetchImageDevice(const void* im_dev, int channels, int width, int height, void* image_ready_event)
{
    if (mSensor != RGBD) throw std::runtime_error("ERROR: you called PrefetchImageDevice but input sensor was not set to RGBD.");
    mpTracker->PrefetchImageDevice(im_dev, channels, width, height, image_ready_event);
}
void System::SaveResultsIJRR2020(const std::string& prefix)   // System.cc:80-240 (pose / motion files; GT files are not produced)
{
    finish_local_ba();
    auto dump = [](const std::string& path, const std::vector<cv::Mat>& poses) {
        FILE* f = fopen(path.c_str(), "w"); if (!f) return;
        for (size_t i = 0; i < poses.size(); i++) { fprintf(f, "%zu", i); for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) fprintf(f, " %.9f", poses[i].at<float>(r, c)); fprintf(f, "\n"); }
        fclose(f);
    };
    dump(prefix + "initial_rgbd_new.txt", mpMap->vmCameraPose); dump(prefix + "refined_rgbd_new.txt", mpMap->vmCameraPose_RF);
    FILE* f = fopen((prefix + "obj_mot_rgbd_new.txt").c_str(), "w");
    if (f) {
        for (size_t i = 0; i < mpMap->vmRigidMotion.size(); i++) for (size_t j = 1; j < mpMap->vmRigidMotion[i].size(); j++) {
            fprintf(f, "%zu %d", i, mpMap->vnRMLabel[i][j]); const cv::Mat& M = mpMap->vmRigidMotion[i][j];
            for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) fprintf(f, " %.9f", M.at<float>(r, c));
            fprintf(f, " 0 0 0 1\n");
        }
        fclose(f);
    }
    // extension (not written by the reference): the object motions refined by FullBatchOptimization (vmRigidMotion_RF)
    f = fopen((prefix + "obj_mot_refined.txt").c_str(), "w");
    if (f) {
        for (size_t i = 0; i < mpMap->vmRigidMotion_RF.size(); i++) for (size_t j = 1; j < mpMap->vmRigidMotion_RF[i].size(); j++) {
            fprintf(f, "%zu %d", i, mpMap->vnRMLabel[i][j]); const cv::Mat& M = mpMap->vmRigidMotion_RF[i][j];
            for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) fprintf(f, " %.9f", M.at<float>(r, c));
            fprintf(f, " 0 0 0 1\n");
        }
        fclose(f);
    }
}

}  // namespace VIDO_SLAM

// ---- C handle over System (include/vido_c.h "The whole per-frame pipeline behind one C handle") -----------------------------------------
struct vido_system {
    VIDO_SLAM::System sys; std::string err; bool inited = false;
    cv::Mat im, depth, flow, mask, traj;                      // headers over the caller's buffers of the current call (kept: the tracker holds shallow references)
};
static std::string g_sys_create_error;

extern "C" {

int vido_system_create(const char* yaml, vido_system** out)
{
    if (!yaml || !out) { g_sys_create_error = "vido_system_create: null argument"; return VIDO_E_INVALID; }
    *out = nullptr;
    vido_system* s = new vido_system();
    try { s->sys.Init(yaml, VIDO_SLAM::System::RGBD); s->inited = true; }
    catch (const std::exception& e) { g_sys_create_error = e.what(); delete s; return VIDO_E_INVALID; }
    *out = s;
    return VIDO_OK;
}
void vido_system_destroy(vido_system* s) { delete s; }
const char* vido_system_last_error(const vido_system* s) { return s ? s->err.c_str() : g_sys_create_error.c_str(); }

int vido_system_track_rgbd(vido_system* s, const uint8_t* im, int channels, int width, int height, float* depth, const float* flow, const int32_t* mask,
                           double timestamp, int n_image, float Tcw_out[16])
{
    if (!s || !s->inited) return VIDO_E_INVALID;
    if (!im || !depth || !flow || !mask || !Tcw_out || width <= 0 || height <= 0 || (channels != 1 && channels != 3 && channels != 4)) { s->err = "vido_system_track_rgbd: bad argument"; return VIDO_E_INVALID; }
    try {
        s->im = cv::Mat(height, width, CV_MAKETYPE(CV_8U, channels), (void*)im);
        s->depth = cv::Mat(height, width, CV_32FC1, (void*)depth);
        s->flow = cv::Mat(height, width, CV_32FC2, (void*)flow);
        s->mask = cv::Mat(height, width, CV_32SC1, (void*)mask);
        if (s->traj.empty()) s->traj = cv::Mat::zeros(600, 800, CV_8UC3);
        const cv::Mat id = cv::Mat::eye(4, 4, CV_32F); const std::vector<std::vector<float> > gt;
        cv::Mat T = s->sys.TrackRGBD(s->im, s->depth, s->flow, s->mask, id, gt, timestamp, s->traj, n_image);
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) Tcw_out[r * 4 + c] = T.at<float>(r, c);
    } catch (const std::exception& e) {
        s->err = e.what();
        return VIDO_SLAM::failure_code(e);
    }
    return VIDO_OK;
}

int vido_system_track_rgbd_device(vido_system* s, const void* im_dev, int channels, int width, int height, float* depth_dev, const float* flow_dev, const int32_t* mask_dev,
                                  void* ready_event, double timestamp, int n_image, float Tcw_out[16])
{
    if (!s || !s->inited) return VIDO_E_INVALID;
    if (!im_dev || !depth_dev || !flow_dev || !mask_dev || !Tcw_out || width <= 0 || height <= 0 || (channels != 1 && channels != 3 && channels != 4)) { s->err = "vido_system_track_rgbd_device: bad argument"; return VIDO_E_INVALID; }
    try {
        cv::Mat T = s->sys.TrackRGBDDevice(im_dev, channels, width, height, depth_dev, flow_dev, (const int*)mask_dev, ready_event, timestamp, n_image);
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) Tcw_out[r * 4 + c] = T.at<float>(r, c);
    } catch (const std::exception& e) {
        s->err = e.what();
        return VIDO_SLAM::failure_code(e);
    }
    return VIDO_OK;
}

int vido_system_get_stats(const vido_system* s, vido_system_stats* o)
{
    if (!s || !o || !s->inited) return VIDO_E_INVALID;
    memset(o, 0, sizeof *o);
    try { VIDO_SLAM::finish_local_ba(); } catch (...) { }      // (VIDO_LBA_ASYNC: the helper's window solve writes the Map and the timings read below)
    VIDO_SLAM::Tracking* T = const_cast<vido_system*>(s)->sys.GetTracker(); VIDO_SLAM::Map* M = const_cast<vido_system*>(s)->sys.GetMap();
    if (!T || !T->mpCurrentFrame) return VIDO_OK;
    const VIDO_SLAM::Frame* F = T->mpCurrentFrame;
    o->frame_id = T->f_id - 1; o->n_keypoints = F->N; o->n_static = (int)F->mvStatKeysTmp.size(); o->n_static_inliers = 0;
    for (int id : F->nStaInlierID) if (id >= 0) o->n_static_inliers++;
    int no = 0; for (size_t i = 0; i < F->bObjStat.size(); i++) if (F->bObjStat[i]) no++;
    o->n_objects = no; o->n_object_points = (int)F->mvObjKeys.size(); o->ba_window = std::min(std::max(T->f_id - 1, 0), T->nWINDOW_SIZE);
    o->ms_total = T->ms_total; o->ms_update_mask = T->ms_update_mask; o->ms_frame = T->ms_frame; o->ms_wait_inputs = T->ms_wait_inputs; o->ms_orb = VIDO_SLAM::detail::LastFrameStageMs(0); o->ms_lists = VIDO_SLAM::detail::LastFrameStageMs(1);
    if (T->all_timing.size() >= 5) { o->ms_cam_pose = T->all_timing[1]; o->ms_obj_tracking = T->all_timing[2]; o->ms_renew = T->all_timing[4]; }
    o->ms_obj_motion = T->ms_obj_motion_sum;
    o->ms_local_ba = M && !M->fLBA_time.empty() && T->f_id > 1 ? M->fLBA_time.back() : 0.f;
    return VIDO_OK;
}

int vido_system_save_results(vido_system* s, const char* prefix)
{
    if (!s || !s->inited) return VIDO_E_INVALID;
    try { s->sys.SaveResultsIJRR2020(prefix ? prefix : ""); } catch (const std::exception& e) { s->err = e.what(); return VIDO_E_INVALID; }
    return VIDO_OK;
}

vido_ctx* vido_system_context(vido_system* s) { return s && s->inited ? VIDO_SLAM::detail::Context() : nullptr; }

int vido_system_prefetch_image_device(vido_system* s, const void* im_dev, int channels, int width, int height, void* image_ready_event)
{
    if (!s || !s->inited) return VIDO_E_INVALID;
    if (!im_dev || width <= 0 || height <= 0 || (channels != 1 && channels != 3 && channels != 4)) { s->err = "vido_system_prefetch_image_device: bad argument"; return VIDO_E_INVALID; }
    try { s->sys.PrefetchImageDevice(im_dev, channels, width, height, image_ready_event); }
    catch (const std::exception& e) { s->err = e.what(); return VIDO_SLAM::failure_code(e); }
    return VIDO_OK;
}

int vido_system_set_zero_copy_maps(vido_system* s, int zero_copy)
{
    if (!s) return VIDO_E_INVALID;
    VIDO_SLAM::detail::SetZeroCopyMaps(zero_copy != 0);
    return VIDO_OK;
}

int vido_system_set_depth_noise_seed(vido_system* s, unsigned seed)
{
    if (!s) return VIDO_E_INVALID;
    VIDO_SLAM::detail::SetDepthNoiseSeed(seed);
    return VIDO_OK;
}

}  // extern "C"

// The incremental tracklet store (Map::UpdateTracklets) fed one association row at a time, as Tracking::Track does once per frame, flattened for the parity test against
// the reference's full rebuild (Tracking::GetStaticTrack / GetDynamicTrackNew, Tracking.cc:2514-2720).  n_feat0: feature count of frame 0.
extern "C" int vido_tracklets_incremental(int n_rows, const int32_t* row_off, const int32_t* row_n, const int32_t* TM, const int32_t* labels, int n_feat0,
                                          int32_t* trk_off, int32_t* pairs, int32_t* obj_id, int32_t* owner_trk, int32_t* owner_pos, int cap_trk, int cap_pairs, int32_t* n_trk)
{
    if (n_rows < 0 || !n_trk || !trk_off || (n_rows && (!row_off || !row_n || !TM))) return VIDO_E_INVALID;
    try {
        VIDO_SLAM::Map M;
        M.vpFeatSta.push_back(std::vector<cv::KeyPoint>((size_t)std::max(n_feat0, 0)));
        for (int i = 0; i < n_rows; i++) {                    // one "frame" per row: features of the new frame, its association row, then the incremental update
            M.vpFeatSta.push_back(std::vector<cv::KeyPoint>((size_t)row_n[i]));
            M.vnAssoSta.push_back(std::vector<int>(TM + row_off[i], TM + row_off[i] + row_n[i]));
            if (labels) { M.vpFeatDyn = M.vpFeatSta; M.vnAssoDyn = M.vnAssoSta; M.vnFeatLabel.push_back(std::vector<int>(labels + row_off[i], labels + row_off[i] + row_n[i])); }
            M.UpdateTracklets();
        }
        const auto& T = labels ? M.TrackletDyn : M.TrackletSta;
        const auto& trk = labels ? M.vnTrkDyn : M.vnTrkSta; const auto& pos = labels ? M.vnPosDyn : M.vnPosSta;
        *n_trk = (int32_t)T.size();
        if ((int)T.size() > cap_trk) return VIDO_E_CAPACITY;
        int o = 0;
        for (size_t t = 0; t < T.size(); t++) {
            trk_off[t] = o;
            for (const auto& e : T[t]) { if (o >= cap_pairs) return VIDO_E_CAPACITY; pairs[2 * o] = e.first; pairs[2 * o + 1] = e.second; o++; }
            if (labels && obj_id) obj_id[t] = M.nObjID[t];
        }
        trk_off[T.size()] = o;
        if (owner_trk && owner_pos) {                         // per-feature owner tables, frames concatenated (frame 0, then each row's frame)
            int q = 0;
            for (size_t f = 0; f < trk.size(); f++) for (size_t k = 0; k < trk[f].size(); k++) { owner_trk[q] = trk[f][k]; owner_pos[q] = pos[f][k]; q++; }
        }
    } catch (const std::exception&) { return VIDO_E_INVALID; }
    return VIDO_OK;
}
