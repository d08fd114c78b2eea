// run_vido_slam.cpp — offline driver, counterpart of the reference's vido_slam/demo/run_vido_slam.cc:67-137:
// main(config.yaml) -> System::Init(yaml, RGBD) -> per frame: load image, flow (.flo), depth, mask -> System::TrackRGBD(...) -> poses.
// Two on-disk layouts:
//  (1) the REFERENCE's (run_vido_slam.cc:47-65, 113-122), chosen when <image_path>/../vTimestampsImage.txt exists: one header line, then one
//      integer-nanosecond time stamp per line; frame name = first 19 characters of std::to_string((long double) stamp); <image_path>/<name>.png is a Bayer-RG u8 PNG
//      (cv::COLOR_BayerRG2BGR), ../flow_image/<name>.flo (Middlebury "PIEH"), ../depth_image/<name>.png 16-bit gray (ANYDEPTH -> CV_32F), ../mask_image/<name>.png
//      (UNCHANGED -> CV_32SC1); frames start_index .. end;
//  (2) raw dumps (tests that predate the PNG reader): <image_path>/<idx>.gray (u8 HxW), ../flow_image/<idx>.flo, ../depth_image/<idx>.depth (f32), ../mask_image/<idx>.mask (i32).
// PNG decoding: chunk walk + zlib inflate + the five scan-line filters, 8/16-bit gray and 8-bit RGB(A), non-interlaced (what the reference's dataset holds); OpenCV is not
// in this image, and cv::imread / cv::cvtColor(COLOR_BayerRG2BGR) are third-party code: the bilinear demosaic below restates OpenCV's published scheme (parity unpinned).
#include "../include/vido_slam/vido_slam.h"
#include <cmath>
#include <zlib.h>
#include <sstream>
#include <chrono>
#include <cstdio>
#include <algorithm>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>
using namespace VIDO_SLAM;

static std::vector<char> slurp(const std::string& p, size_t expect)
{
    std::ifstream f(p.c_str(), std::ios::binary); if (!f) throw std::runtime_error("cannot open " + p);
    std::vector<char> b((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (expect && b.size() != expect) throw std::runtime_error("unexpected size of " + p);
    return b;
}
static cv::Mat read_flo(const std::string& p, int w, int h)                 // cv::optflow::readOpticalFlow
{
    std::vector<char> b = slurp(p, 12 + (size_t)w * h * 8);
    float magic; int fw, fh; memcpy(&magic, b.data(), 4); memcpy(&fw, b.data() + 4, 4); memcpy(&fh, b.data() + 8, 4);
    if (magic != 202021.25f || fw != w || fh != h) throw std::runtime_error("bad .flo header in " + p);
    cv::Mat m(h, w, CV_32FC2); memcpy(m.data, b.data() + 12, (size_t)w * h * 8); return m;
}

// ---- minimal PNG reader --------------------------------------------------------------------------------------------------------------------------
struct Png { int w = 0, h = 0, depth = 0, channels = 0; std::vector<uint8_t> px; };      // px: rows of w * channels samples, 16-bit samples big-endian as in the file
static Png read_png(const std::string& path)
{
    std::vector<char> f = slurp(path, 0);
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (f.size() < 33 || memcmp(f.data(), sig, 8) != 0) throw std::runtime_error("not a PNG: " + path);
    auto be32 = [&](size_t o) { return ((uint32_t)(uint8_t)f[o] << 24) | ((uint32_t)(uint8_t)f[o + 1] << 16) | ((uint32_t)(uint8_t)f[o + 2] << 8) | (uint32_t)(uint8_t)f[o + 3]; };
    Png P; int color = 0, interlace = 0; std::vector<uint8_t> z;
    for (size_t o = 8; o + 12 <= f.size();) {
        const uint32_t len = be32(o); const std::string type(f.data() + o + 4, 4);
        if (o + 12 + len > f.size()) throw std::runtime_error("truncated PNG: " + path);
        const uint8_t* d = (const uint8_t*)f.data() + o + 8;
        if (type == "IHDR") { P.w = (int)be32(o + 8); P.h = (int)be32(o + 12); P.depth = d[8]; color = d[9]; interlace = d[12]; }
        else if (type == "IDAT") z.insert(z.end(), d, d + len);
        else if (type == "IEND") break;
        o += 12 + len;
    }
    P.channels = color == 0 ? 1 : (color == 2 ? 3 : (color == 6 ? 4 : (color == 4 ? 2 : 0)));
    if (!P.channels || interlace || (P.depth != 8 && P.depth != 16)) throw std::runtime_error("unsupported PNG flavour (need 8/16-bit gray / RGB(A), non-interlaced): " + path);
    const size_t bpp = (size_t)P.channels * P.depth / 8, stride = bpp * P.w;
    std::vector<uint8_t> raw((stride + 1) * P.h);
    uLongf out_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &out_len, z.data(), (uLong)z.size()) != Z_OK || out_len != raw.size()) throw std::runtime_error("PNG inflate failed: " + path);
    P.px.assign(stride * P.h, 0);
    for (int y = 0; y < P.h; y++) {                          // scan-line filters (PNG spec 9.2)
        const uint8_t ft = raw[(stride + 1) * y]; const uint8_t* in = raw.data() + (stride + 1) * y + 1; uint8_t* cur = P.px.data() + stride * y; const uint8_t* up = y ? cur - stride : nullptr;
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0; int pred = 0;
            if (ft == 1) pred = a; else if (ft == 2) pred = b; else if (ft == 3) pred = (a + b) >> 1;
            else if (ft == 4) { const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
            else if (ft != 0) throw std::runtime_error("bad PNG filter in " + path);
            cur[i] = (uint8_t)(in[i] + pred);
        }
    }
    return P;
}
// cv::cvtColor(COLOR_BayerRG2BGR), bilinear (OpenCV imgproc demosaicing.cpp, Bayer2RGB_: the missing colours of a site are the means of its 2 or 4 nearest
// neighbours of that colour; the outermost rows / columns are copies of their inner neighbours).  "RG" names the second row's 2nd and 3rd pixels: row 1 = G R G R, row 0 = B G B G.
static cv::Mat bayer_rg_to_bgr(const Png& P)
{
    const int w = P.w, h = P.h; cv::Mat out(h, w, CV_8UC3);
    auto at = [&](int y, int x) { return (int)P.px[(size_t)y * w + x]; };
    for (int y = 1; y < h - 1; y++) for (int x = 1; x < w - 1; x++) {
        int b, g, r; const bool odd_y = y & 1, odd_x = x & 1;
        if (!odd_y && !odd_x) { b = at(y, x); g = (at(y - 1, x) + at(y + 1, x) + at(y, x - 1) + at(y, x + 1) + 2) >> 2; r = (at(y - 1, x - 1) + at(y - 1, x + 1) + at(y + 1, x - 1) + at(y + 1, x + 1) + 2) >> 2; }
        else if (odd_y && odd_x) { r = at(y, x); g = (at(y - 1, x) + at(y + 1, x) + at(y, x - 1) + at(y, x + 1) + 2) >> 2; b = (at(y - 1, x - 1) + at(y - 1, x + 1) + at(y + 1, x - 1) + at(y + 1, x + 1) + 2) >> 2; }
        else if (!odd_y) { g = at(y, x); b = (at(y, x - 1) + at(y, x + 1) + 1) >> 1; r = (at(y - 1, x) + at(y + 1, x) + 1) >> 1; }      // green site in a blue row
        else { g = at(y, x); r = (at(y, x - 1) + at(y, x + 1) + 1) >> 1; b = (at(y - 1, x) + at(y + 1, x) + 1) >> 1; }                    // green site in a red row
        uint8_t* o = out.ptr<uint8_t>(y) + 3 * x; o[0] = (uint8_t)b; o[1] = (uint8_t)g; o[2] = (uint8_t)r;
    }
    for (int y = 0; y < h; y++) { const int yy = std::min(std::max(y, 1), h - 2);
        for (int x = 0; x < w; x++) { const int xx = std::min(std::max(x, 1), w - 2); if (yy != y || xx != x) memcpy(out.ptr<uint8_t>(y) + 3 * x, out.ptr<uint8_t>(yy) + 3 * xx, 3); } }
    return out;
}
static cv::Mat png_to_f32(const Png& P)           // imread(ANYDEPTH) + convertTo(CV_32F)
{
    if (P.channels != 1) throw std::runtime_error("depth PNG must be single-channel");
    cv::Mat m(P.h, P.w, CV_32F);
    for (int y = 0; y < P.h; y++) for (int x = 0; x < P.w; x++)
        m.at<float>(y, x) = P.depth == 16 ? (float)((P.px[2 * ((size_t)y * P.w + x)] << 8) | P.px[2 * ((size_t)y * P.w + x) + 1]) : (float)P.px[(size_t)y * P.w + x];
    return m;
}
static cv::Mat png_to_i32(const Png& P)           // imread(UNCHANGED) + convertTo(CV_32SC1)
{
    if (P.channels != 1) throw std::runtime_error("mask PNG must be single-channel");
    cv::Mat m(P.h, P.w, CV_32SC1);
    for (int y = 0; y < P.h; y++) for (int x = 0; x < P.w; x++)
        m.at<int32_t>(y, x) = P.depth == 16 ? ((P.px[2 * ((size_t)y * P.w + x)] << 8) | P.px[2 * ((size_t)y * P.w + x) + 1]) : P.px[(size_t)y * P.w + x];
    return m;
}
// LoadKaistImg (run_vido_slam.cc:47-65)
static bool load_kaist_stamps(const std::string& image_dir, std::vector<std::string>& names, std::vector<double>& stamps)
{
    std::ifstream fin((image_dir + "/../vTimestampsImage.txt").c_str());
    if (!fin.is_open()) return false;
    std::string line; std::getline(fin, line);
    while (std::getline(fin, line) && !line.empty()) {
        std::stringstream ss; ss << line; long double s = 0; ss >> s;
        names.push_back(std::to_string(s).substr(0, 19)); stamps.push_back((double)(s / 1e9));
    }
    return true;
}

int main(int argc, char** argv)
{
    if (argc < 2) { std::cerr << "usage: run_vido_slam config.yaml [out_poses.txt]" << std::endl; return 1; }
    try {
        auto kv = detail::ParseSettings(argv[1]);
        const std::string dir = kv["image_path"]; const int w = atoi(kv["Camera.width"].c_str()), h = atoi(kv["Camera.height"].c_str());
        std::vector<std::string> names; std::vector<double> stamps;
        const bool kaist_layout = load_kaist_stamps(dir, names, stamps);
        const int start = kv.count("start_index") ? atoi(kv["start_index"].c_str()) : 0;
        const int n = kaist_layout ? (kv.count("n_frames") ? std::min((int)names.size(), atoi(kv["n_frames"].c_str())) : (int)names.size()) : atoi(kv["n_frames"].c_str());
        System SLAM; SLAM.Init(argv[1], System::RGBD);
        cv::Mat id = cv::Mat::eye(4, 4, CV_32F), imTraj = cv::Mat::zeros(600, 800, CV_8UC3);
        std::vector<std::vector<float> > vObjPose_gt;
        FILE* out = fopen(argc > 2 ? argv[2] : "poses.txt", "w");
        std::vector<double> frame_ms;
        for (int idx = start; idx < n; idx++) {
            cv::Mat gray, flow, depth, mask; double stamp = (double)idx;
            if (kaist_layout) {                                // run_vido_slam.cc:113-122
                const std::string& nm = names[idx]; stamp = stamps[idx];
                const Png raw = read_png(dir + "/" + nm + ".png");
                if (raw.w != w || raw.h != h || raw.depth != 8) throw std::runtime_error("image " + nm + ".png: size / depth mismatch");
                if (raw.channels == 1) gray = bayer_rg_to_bgr(raw);                      // cv::cvtColor(raw, bgr, COLOR_BayerRG2BGR): TrackRGBD gets BGR
                else { gray = cv::Mat(h, w, CV_MAKETYPE(CV_8U, raw.channels)); memcpy(gray.data, raw.px.data(), raw.px.size()); }
                flow = read_flo(dir + "/../flow_image/" + nm + ".flo", w, h);
                depth = png_to_f32(read_png(dir + "/../depth_image/" + nm + ".png"));
                mask = png_to_i32(read_png(dir + "/../mask_image/" + nm + ".png"));
                if (depth.cols != w || depth.rows != h || mask.cols != w || mask.rows != h) throw std::runtime_error("depth / mask size mismatch at " + nm);
            } else {
                char name[64]; snprintf(name, sizeof name, "%06d", idx);
                std::vector<char> g = slurp(dir + "/" + name + ".gray", (size_t)w * h);
                gray = cv::Mat(h, w, CV_8UC1); memcpy(gray.data, g.data(), g.size());
                flow = read_flo(dir + "/../flow_image/" + name + ".flo", w, h);
                std::vector<char> d = slurp(dir + "/../depth_image/" + name + ".depth", (size_t)w * h * 4), m = slurp(dir + "/../mask_image/" + name + ".mask", (size_t)w * h * 4);
                depth = cv::Mat(h, w, CV_32F); mask = cv::Mat(h, w, CV_32SC1); memcpy(depth.data, d.data(), d.size()); memcpy(mask.data, m.data(), m.size());
            }
            const auto t0 = std::chrono::steady_clock::now();
            cv::Mat Tcw = SLAM.TrackRGBD(gray, depth, flow, mask, id, vObjPose_gt, stamp, imTraj, kaist_layout ? 10000 : n);
            frame_ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
            fprintf(out, "%d", idx); for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) fprintf(out, " %.9g", Tcw.at<float>(r, c)); fprintf(out, "\n");
        }
        fclose(out);
        SLAM.SaveResultsIJRR2020(argc > 3 ? argv[3] : "");
        Map* M = SLAM.GetMap(); double lba = 0; for (float t : M->fLBA_time) lba += t;
        std::cout << "frames " << n - start << " local-BA mean ms " << (M->fLBA_time.empty() ? 0.0 : lba / M->fLBA_time.size()) << std::endl;
        {   // the incremental tracklet store against the reference-style full rebuild from the association tables
            const std::vector<int> ids = M->nObjID;
            const bool ok = SLAM.GetTracker()->GetStaticTrack() == M->TrackletSta && SLAM.GetTracker()->GetDynamicTrackNew() == M->TrackletDyn && ids == M->nObjID;
            std::cout << "tracklets static " << M->TrackletSta.size() << " dynamic " << M->TrackletDyn.size() << " incremental_equals_rebuild " << (ok ? 1 : 0) << std::endl;
        }
        { int ck = 0, mm = 0; detail::ResidentCheckStats(&ck, &mm);      // VIDO_BA_RESIDENT_CHECK=1: every local window solved on the device-resident window AND through the Map walk
          std::cout << "resident_window checks " << ck << " mismatches " << mm << std::endl; }
        {   // Frame::UndistortKeyPoints on the last frame (Frame.cc:603-633): largest displacement mvKeys -> mvKeysUn (0 when Camera.k1 == 0)
            const Frame* F = SLAM.GetTracker()->mpLastFrame ? SLAM.GetTracker()->mpLastFrame : SLAM.GetTracker()->mpCurrentFrame;
            double mx = 0; if (F) for (size_t i = 0; i < F->mvKeys.size() && i < F->mvKeysUn.size(); i++)
                mx = std::max(mx, (double)std::hypot(F->mvKeys[i].pt.x - F->mvKeysUn[i].pt.x, F->mvKeys[i].pt.y - F->mvKeysUn[i].pt.y));
            std::cout << "undistort keys " << (F ? F->mvKeysUn.size() : 0) << " max_shift_px " << mx << std::endl;
        }
        if (!M->vfAll_time.empty()) {   // Tracking::Track stage means (the reference's all_timing layout: [0] feature, [1] camera pose, [2] scene flow/object tracking, [3] per-object motion, [4] renew + map)
            std::vector<double> acc(5, 0.0); int cnt = 0;
            for (size_t i = 2; i < M->vfAll_time.size(); i++) { for (int k = 0; k < 5 && k < (int)M->vfAll_time[i].size(); k++) acc[k] += M->vfAll_time[i][k]; cnt++; }
            if (cnt) { std::cout << "stage_ms"; for (double v : acc) std::cout << " " << v / cnt; std::cout << std::endl; }
        }
        if (frame_ms.size() > 3) {      // TrackRGBD wall time per frame (host buffers in, pose out), first two frames (initialisation, allocations) excluded
            std::vector<double> t(frame_ms.begin() + 2, frame_ms.end()); std::sort(t.begin(), t.end());
            double mean = 0; for (double v : t) mean += v; mean /= t.size();
            std::cout << "track_ms mean " << mean << " median " << t[t.size() / 2] << " max " << t.back() << std::endl;
        }
    } catch (const std::exception& e) { std::cerr << "run_vido_slam: " << e.what() << std::endl; return 2; }
    return 0;
}
