// run_vido_slam.cpp — offline driver, counterpart of the reference's vido_slam/demo/run_vido_slam.cc:67-137:
// main(config.yaml) -> System::Init(yaml, RGBD) -> per frame: load gray/bgr, flow (.flo), depth, mask ->
// System::TrackRGBD(...) -> poses.  On-disk layout (this image has no PNG codec for C++, so
// images are raw dumps): <image_path>/<idx>.gray (u8 HxW), ../flow_image/<idx>.flo (Middlebury "PIEH"),
// ../depth_image/<idx>.depth (f32 HxW, sensor units), ../mask_image/<idx>.mask (i32 HxW).
#include "../include/vido_slam/vido_slam.h"
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

int main(int argc, char** argv)
{
    if (argc < 2) { std::cerr << "usage: run_vido_slam config.yaml [out_poses.txt]" << std::endl; return 1; }
    try {
        auto kv = detail::ParseSettings(argv[1]);
        const std::string dir = kv["image_path"]; const int w = atoi(kv["Camera.width"].c_str()), h = atoi(kv["Camera.height"].c_str());
        const int n = atoi(kv["n_frames"].c_str()), start = kv.count("start_index") ? atoi(kv["start_index"].c_str()) : 0;
        System SLAM; SLAM.Init(argv[1], System::RGBD);
        cv::Mat id = cv::Mat::eye(4, 4, CV_32F), imTraj = cv::Mat::zeros(600, 800, CV_8UC3);
        std::vector<std::vector<float> > vObjPose_gt;
        FILE* out = fopen(argc > 2 ? argv[2] : "poses.txt", "w");
        std::vector<double> frame_ms;
        for (int idx = start; idx < n; idx++) {
            char name[64]; snprintf(name, sizeof name, "%06d", idx);
            std::vector<char> g = slurp(dir + "/" + name + ".gray", (size_t)w * h);
            cv::Mat gray(h, w, CV_8UC1); memcpy(gray.data, g.data(), g.size());
            cv::Mat flow = read_flo(dir + "/../flow_image/" + name + ".flo", w, h);
            std::vector<char> d = slurp(dir + "/../depth_image/" + name + ".depth", (size_t)w * h * 4), m = slurp(dir + "/../mask_image/" + name + ".mask", (size_t)w * h * 4);
            cv::Mat depth(h, w, CV_32F), mask(h, w, CV_32SC1); memcpy(depth.data, d.data(), d.size()); memcpy(mask.data, m.data(), m.size());
            const auto t0 = std::chrono::steady_clock::now();
            cv::Mat Tcw = SLAM.TrackRGBD(gray, depth, flow, mask, id, vObjPose_gt, (double)idx, imTraj, n);
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
