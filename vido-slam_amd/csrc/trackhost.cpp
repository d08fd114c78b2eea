// trackhost.cpp — the tracker's host-side bookkeeping stages on flat arrays (C-ABI, no device work, no ctx): what Tracking::DynObjTracking,
// Tracking::RenewFrameInfo, Tracking::GetStaticTrack / GetDynamicTrackNew and Frame::UndistortKeyPoints do to a few thousand points per frame
// (reference vido_slam/src/Tracking.cc:1670-1912, 2959-3289, 2514-2720; Frame.cc:603-633).  The C++ facade (facade.cpp) calls exactly these
// functions, and the parity tests compare them with literal restatements of the reference's own O(N*M) loops (test infrastructure, not linked here) — the one place where the build's algorithm differs from the reference's is the "already used" test of the re-seeding (a 1-px cell hash
// instead of a scan over the whole kept set), and that difference is what those tests pin.
#include <cfloat>
#include <ctime>
#include "../../include/vido_c.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <utility>
#include <vector>

namespace {

// "is some point of a fixed set closer than 1 px to q" — the reference scans the whole set for every sample (Tracking.cc:3030-3040, 3198-3208: O(N*M));
// a 1-px cell grid restricts the scan to the 3x3 neighbourhood and evaluates the same float expression, so the answer is identical.
struct NearSet {
    // The cell heads live in ONE per-thread array that is all -1 between uses: a set touches only the cells of its own points and puts them back in its destructor.
    // (Round 5: as a fresh (W + 2) x (H + 2) vector per call the grid was 1.2 MB of allocation + fill at 640 x 480, twice per frame — most of RenewFrameInfo's host time.)
    // At most ONE set may use the shared grid at a time (two live sets would splice their chains through different next[] arrays): a second set alive on the same thread
    // gets a private grid instead.  The touched cells are remembered, so the caller's point buffer may change or go away before the set does.
    int W, H; std::vector<int> own; bool shared; std::vector<int>& head; std::vector<int> next, touched; const float* xy; int n_pts;
    static bool& in_use() { static thread_local bool u = false; return u; }
    static std::vector<int>& grid(size_t cells) { static thread_local std::vector<int> g; if (g.size() < cells) g.assign(cells, -1); return g; }
    NearSet(const float* p, int n, int w, int h) : W(w + 2), H(h + 2), own(in_use() ? (size_t)(w + 2) * (h + 2) : 0, -1), shared(!in_use()),
                                                     head(shared ? grid((size_t)(w + 2) * (h + 2)) : own), next((size_t)std::max(n, 1), -1), touched((size_t)n), xy(p), n_pts(n) {
        if (shared) in_use() = true;
        for (int i = 0; i < n; i++) { const int c = cell(p[2 * i], p[2 * i + 1]); touched[i] = c; next[i] = head[c]; head[c] = i; }
    }
    ~NearSet() { if (shared) { for (int c : touched) head[c] = -1; in_use() = false; } }
    NearSet(const NearSet&) = delete; NearSet& operator=(const NearSet&) = delete;
    int clampx(float x) const { return std::min(std::max((int)std::floor(x) + 1, 0), W - 1); }
    int clampy(float y) const { return std::min(std::max((int)std::floor(y) + 1, 0), H - 1); }
    int cell(float x, float y) const { return clampy(y) * W + clampx(x); }
    bool near(float qx, float qy) const {
        const int cx = clampx(qx), cy = clampy(qy);
        for (int yy = std::max(cy - 1, 0); yy <= std::min(cy + 1, H - 1); yy++) for (int xx = std::max(cx - 1, 0); xx <= std::min(cx + 1, W - 1); xx++)
            for (int i = head[(size_t)yy * W + xx]; i >= 0; i = next[i]) {
                const float dx = xy[2 * i] - qx, dy = xy[2 * i + 1] - qy;
                if (std::sqrt(dx * dx + dy * dy) < 1.0f) return true;
            }
        return false;
    }
};

int most_frequent(std::vector<int> v)                  // std::map count + SortPairInt (descending count; the lowest key wins a tie)
{
    std::sort(v.begin(), v.end()); int best = v[0], bc = 0, run = 0;
    for (size_t j = 0; j < v.size(); j++) { run = (j > 0 && v[j] == v[j - 1]) ? run + 1 : 1; if (run > bc) { bc = run; best = v[j]; } }
    return best;
}

}  // namespace

extern "C" {

int vido_undistort_points(const float* xy, int n, const float K[4], const float dist[5], float* xy_out)
{
    if (n < 0 || !K || !dist || (n && (!xy || !xy_out))) return VIDO_E_INVALID;
    // OpenCV converts the CV_32F camera matrix / coefficients to double and works in double throughout, with the inverse focal lengths (cvUndistortPointsInternal)
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3], ifx = 1. / fx, ify = 1. / fy;
    const double k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3], k3 = dist[4];
    for (int i = 0; i < n; i++) {
        if (dist[0] == 0.0f) { xy_out[2 * i] = xy[2 * i]; xy_out[2 * i + 1] = xy[2 * i + 1]; continue; }   // Frame.cc:605-609: no distortion, keys copied
        const double x0 = ((double)xy[2 * i] - cx) * ifx, y0 = ((double)xy[2 * i + 1] - cy) * ify; double x = x0, y = y0;
        for (int it = 0; it < 5; it++) {
            const double r2 = x * x + y * y, icdist = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
            const double dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x), dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
            x = (x0 - dx) * icdist; y = (y0 - dy) * icdist;
        }
        xy_out[2 * i] = (float)(x * fx + cx); xy_out[2 * i + 1] = (float)(y * fy + cy);
    }
    return VIDO_OK;
}

// values of the three maps at ((int)x, (int)y) of every point of a list: what the renew stages read from mSegMap / mDepthMap / mFlowMap.  Points outside the image
// are never looked at by the stages (their bounds test comes first), so their entries may hold anything.
static void sample_maps(const vido_host_maps* m, const float* xy, int n, std::vector<int32_t>& mask, std::vector<float>& depth, std::vector<float>& flow)
{
    mask.assign((size_t)std::max(n, 1), 0); depth.assign((size_t)std::max(n, 1), 0.f); flow.assign(2 * (size_t)std::max(n, 1), 0.f);
    for (int i = 0; i < n; i++) {
        const int x = (int)xy[2 * i], y = (int)xy[2 * i + 1];
        if (x >= m->width || y >= m->height || x < 0 || y < 0) continue;
        const size_t o = (size_t)y * m->width + x;
        mask[i] = m->mask[o]; depth[i] = m->depth[o]; flow[2 * i] = m->flow[2 * o]; flow[2 * i + 1] = m->flow[2 * o + 1];
    }
}

int vido_renew_static_sampled(int W, int H, const float* stat_xy, int n_stat, const vido_point_samples* ss, const int32_t* TM_sta, int n_tm,
                              const float* sample_xy, int n_sample, const vido_point_samples* ks, int max_num, int32_t* src_out, int32_t* inlier_out, float* flow_out, int cap, int32_t* n_out)
{
    if (W < 1 || H < 1 || !n_out || n_tm < 0 || n_sample < 0 || cap < 0 || (n_tm && (!TM_sta || !stat_xy || !ss || !ss->mask || !ss->depth || !ss->flow)) ||
        (n_sample && (!sample_xy || !ks || !ks->mask || !ks->depth || !ks->flow)) || (cap && (!src_out || !inlier_out || !flow_out))) return VIDO_E_INVALID;
    int n = 0; bool overflow = false;
    std::vector<float> kept;                                  // positions of the kept points (the inlier part is the "already used" check set)
    auto try_add = [&](float px, float py, int src, int inl, const vido_point_samples* v) -> bool {
        const int x = (int)px, y = (int)py;
        if (x >= W || y >= H || x <= 0 || y <= 0) return false;
        if (v->mask[src] != 0) return false;
        const float d = v->depth[src]; if (d > 40 || d <= 0) return false;
        const float fxe = v->flow[2 * (size_t)src], fye = v->flow[2 * (size_t)src + 1];
        if (fxe != 0 && fye != 0 && px + fxe < W && py + fye < H && px + fxe > 0 && py + fye > 0) {
            if (n < cap) { src_out[n] = src; inlier_out[n] = inl; flow_out[2 * n] = fxe; flow_out[2 * n + 1] = fye; } else overflow = true;
            kept.push_back(px); kept.push_back(py); n++;
            return true;
        }
        return false;
    };
    for (int i = 0; i < n_tm; i++) {                          // (1) the inliers of the last frame (:2977-3012)
        if (TM_sta[i] == -1) continue;
        if (TM_sta[i] < 0 || TM_sta[i] >= n_stat) return VIDO_E_INVALID;
        try_add(stat_xy[2 * TM_sta[i]], stat_xy[2 * TM_sta[i] + 1], TM_sta[i], TM_sta[i], ss);
        if (n > max_num) break;
    }
    int tot = n, start_id = 0; const int step = 20;         // (2) top-up from the detected keypoints in stride-20 passes (:3014-3075)
    const std::vector<float> check(kept);                    // mvKeysTmpCheck: copied once, not extended by the top-up
    const NearSet near_set(check.data(), (int)check.size() / 2, W, H);
    while (tot < max_num) {
        if (start_id == step) break;
        for (int i = start_id; i < n_sample; i += step) {
            if (near_set.near(sample_xy[2 * i], sample_xy[2 * i + 1])) continue;
            if (try_add(sample_xy[2 * i], sample_xy[2 * i + 1], i, -1, ks)) tot++;
            if (tot >= max_num) break;
        }
        start_id++;
    }
    *n_out = n;
    return overflow ? VIDO_E_CAPACITY : VIDO_OK;
}

int vido_renew_static(const vido_host_maps* m, const float* stat_xy, int n_stat, const int32_t* TM_sta, int n_tm, const float* sample_xy, int n_sample,
                      int max_num, int32_t* src_out, int32_t* inlier_out, float* flow_out, int cap, int32_t* n_out)
{
    if (!m || !m->mask || !m->depth || !m->flow || n_stat < 0 || n_sample < 0 || (n_stat && !stat_xy) || (n_sample && !sample_xy)) return VIDO_E_INVALID;
    std::vector<int32_t> m1, m2; std::vector<float> d1, d2, f1, f2;
    sample_maps(m, stat_xy, n_stat, m1, d1, f1); sample_maps(m, sample_xy, n_sample, m2, d2, f2);
    const vido_point_samples ss = {m1.data(), d1.data(), f1.data()}, ks = {m2.data(), d2.data(), f2.data()};
    return vido_renew_static_sampled(m->width, m->height, stat_xy, n_stat, &ss, TM_sta, n_tm, sample_xy, n_sample, &ks, max_num, src_out, inlier_out, flow_out, cap, n_out);
}

int vido_renew_objects(const vido_host_maps* m, const float* obj_xy, const int32_t* obj_label, int n_obj_pts,
                       int n_objects, const int32_t* inl_off, const int32_t* inl_ids, const uint8_t* obj_stat, const int32_t* sem_position, const int32_t* mod_label,
                       const float* tmp_xy, const float* tmp_depth, const int32_t* tmp_sem, const float* tmp_flow, const float* tmp_corr, int n_tmp, int max_num_obj,
                       float* keys_out, float* depth_out, int32_t* sem_out, float* flow_out, float* corr_out, int32_t* inlier_out, int32_t* label_out, int cap, int32_t* n_out)
{
    if (!m || !m->mask || !m->depth || !m->flow || n_obj_pts < 0 || (n_obj_pts && !obj_xy)) return VIDO_E_INVALID;
    std::vector<int32_t> m1; std::vector<float> d1, f1;
    sample_maps(m, obj_xy, n_obj_pts, m1, d1, f1);
    const vido_point_samples os = {m1.data(), d1.data(), f1.data()};
    return vido_renew_objects_sampled(m->width, m->height, obj_xy, obj_label, n_obj_pts, &os, n_objects, inl_off, inl_ids, obj_stat, sem_position, mod_label,
                                      tmp_xy, tmp_depth, tmp_sem, tmp_flow, tmp_corr, n_tmp, max_num_obj, keys_out, depth_out, sem_out, flow_out, corr_out, inlier_out, label_out, cap, n_out);
}

int vido_renew_objects_sampled(int W, int H, const float* obj_xy, const int32_t* obj_label, int n_obj_pts, const vido_point_samples* os,
                               int n_objects, const int32_t* inl_off, const int32_t* inl_ids, const uint8_t* obj_stat, const int32_t* sem_position, const int32_t* mod_label,
                               const float* tmp_xy, const float* tmp_depth, const int32_t* tmp_sem, const float* tmp_flow, const float* tmp_corr, int n_tmp, int max_num_obj,
                               float* keys_out, float* depth_out, int32_t* sem_out, float* flow_out, float* corr_out, int32_t* inlier_out, int32_t* label_out, int cap, int32_t* n_out)
{
    if (W < 1 || H < 1 || !n_out || n_objects < 0 || n_tmp < 0 || cap < 0 || (n_objects && (!inl_off || !obj_stat || !sem_position || !mod_label)) ||
        (n_obj_pts && (!os || !os->mask || !os->depth || !os->flow)) ||
        (n_tmp && (!tmp_xy || !tmp_depth || !tmp_sem || !tmp_flow || !tmp_corr)) || (cap && (!keys_out || !depth_out || !sem_out || !flow_out || !corr_out || !inlier_out || !label_out)))
        return VIDO_E_INVALID;
    int n = 0; bool overflow = false;
    std::vector<float> kept;
    auto push = [&](float kx, float ky, float d, int sem, float fx_, float fy_, float cx_, float cy_, int inl, int lab) {
        if (n < cap) { keys_out[2 * n] = kx; keys_out[2 * n + 1] = ky; depth_out[n] = d; sem_out[n] = sem; flow_out[2 * n] = fx_; flow_out[2 * n + 1] = fy_;
                       corr_out[2 * n] = cx_; corr_out[2 * n + 1] = cy_; inlier_out[n] = inl; label_out[n] = lab; } else overflow = true;
        n++;
    };
    std::vector<int> cnt(n_objects, 0);
    for (int i = 0; i < n_objects; i++) {                     // (1) inliers of the tracked objects, snapped to their integer pixel (:3127-3165)
        if (!obj_stat[i]) { cnt[i] = -1; continue; }
        int count = 0;
        for (int q = inl_off[i]; q < inl_off[i + 1]; q++) {
            const int id = inl_ids[q];
            if (id < 0 || id >= n_obj_pts) return VIDO_E_INVALID;
            const int x = (int)obj_xy[2 * id], y = (int)obj_xy[2 * id + 1];
            if (x >= W || y >= H || x <= 0 || y <= 0) continue;
            const float d = os->depth[id]; const int sem = os->mask[id];
            if (sem != 0 && d < 25 && d > 0) {
                const float fl0 = os->flow[2 * (size_t)id], fl1 = os->flow[2 * (size_t)id + 1];
                if (x + fl0 < W && y + fl1 < H && x + fl0 > 0 && y + fl1 > 0) {
                    push((float)x, (float)y, d, sem, fl0, fl1, x + fl0, y + fl1, id, obj_label[id]);
                    kept.push_back((float)x); kept.push_back((float)y); count++;
                }
            }
        }
        cnt[i] = count;
    }
    const std::vector<float> check(kept);                    // mvObjKeysTmpCheck
    const NearSet near_set(check.data(), (int)check.size() / 2, W, H);
    for (int i = 0; i < n_objects; i++) {                     // (2) top-up per object from this frame's dense samples of the same semantic label, stride 15 (:3168-3228)
        if (!obj_stat[i]) continue;
        const int SemLabel = sem_position[i]; int tot = cnt[i], sid = 0; const int ostep = 15;
        while (tot < max_num_obj) {
            if (sid == ostep) break;
            for (int j = sid; j < n_tmp; j += ostep) {
                if (tmp_sem[j] != SemLabel) continue;
                if (near_set.near(tmp_xy[2 * j], tmp_xy[2 * j + 1])) continue;
                push(tmp_xy[2 * j], tmp_xy[2 * j + 1], tmp_depth[j], tmp_sem[j], tmp_flow[2 * j], tmp_flow[2 * j + 1], tmp_corr[2 * j], tmp_corr[2 * j + 1], -1, mod_label[i]);
                tot++;
                if (tot >= max_num_obj) break;
            }
            sid++;
        }
    }
    // (3) all samples of labels that belong to no tracked object: new objects, label -2 (:3230-3270)
    std::vector<int> uni(tmp_sem, tmp_sem + n_tmp); std::sort(uni.begin(), uni.end()); uni.erase(std::unique(uni.begin(), uni.end()), uni.end());
    std::vector<char> known(uni.size(), 0);
    for (int i = 0; i < n_objects; i++) for (size_t j = 0; j < uni.size(); j++) if (uni[j] == sem_position[i] && obj_stat[i]) { known[j] = 1; break; }
    for (size_t i = 0; i < known.size(); i++) if (!known[i]) for (int j = 0; j < n_tmp; j++) if (uni[i] == tmp_sem[j])
        push(tmp_xy[2 * j], tmp_xy[2 * j + 1], tmp_depth[j], tmp_sem[j], tmp_flow[2 * j], tmp_flow[2 * j + 1], tmp_corr[2 * j], tmp_corr[2 * j + 1], -1, -2);
    *n_out = n;
    return overflow ? VIDO_E_CAPACITY : VIDO_OK;
}

int vido_dyn_obj_tracking(const int32_t* sem_label, int32_t* obj_label, const float* obj_xy, const float* obj_depth, const float* flow3d, const int32_t* last_sem_label, int n,
                          const int32_t* last_sem_position, const uint8_t* last_obj_stat, const int32_t* last_mod_label, int n_last, int rows, int cols,
                          float sf_mg_thres, float sf_ds_thres, float th_depth_obj, int f_id, int32_t* max_id,
                          int32_t* obj_off, int32_t* obj_ids, int32_t* mod_label_out, int32_t* sem_position_out, int max_objects, int32_t* n_objects)
{
    if (n < 0 || !max_id || !n_objects || max_objects < 0 || (n && (!sem_label || !obj_label || !obj_xy || !obj_depth || !flow3d || !last_sem_label || !obj_ids)) ||
        (n_last && (!last_sem_position || !last_obj_stat || !last_mod_label)) || !obj_off || (max_objects && (!mod_label_out || !sem_position_out))) return VIDO_E_INVALID;
    std::vector<int> uni(sem_label, sem_label + n); std::sort(uni.begin(), uni.end()); uni.erase(std::unique(uni.begin(), uni.end()), uni.end());
    std::vector<std::vector<int> > Posi(uni.size());
    for (int i = 0; i < n; i++) {                             // (:1681-1700) points grouped by semantic label, rejected points (-1) left out
        if (obj_label[i] == -1) continue;
        Posi[std::lower_bound(uni.begin(), uni.end(), sem_label[i]) - uni.begin()].push_back(i);
    }
    std::vector<std::vector<int> > ObjId; std::vector<int> sem_posi;
    const int shr_row = 10, shr_col = 20;
    for (size_t i = 0; i < Posi.size(); i++) {                // (:1706-1735) objects mostly on the image border are dropped
        if (Posi[i].empty()) continue;                        // (the reference divides 0 / 0 here: NaN > 0.5 is false and it would keep an empty object; none can reach the 150-point test)
        float count = 0;
        for (int id : Posi[i]) { const float u = obj_xy[2 * id], v = obj_xy[2 * id + 1]; if (v < shr_row || v > (rows - shr_row) || u < shr_col || u > (cols - shr_col)) count += 1; }
        if (count / Posi[i].size() > 0.5f) { for (int id : Posi[i]) obj_label[id] = -1; continue; }
        ObjId.push_back(Posi[i]); sem_posi.push_back(uni[i]);
    }
    std::vector<std::vector<int> > ObjIdNew; std::vector<int> SemPosNew;
    for (size_t i = 0; i < ObjId.size(); i++) {               // (:1742-1822) static / far / small objects
        float depth_sum = 0, sf_count = 0;
        for (int id : ObjId[i]) {
            depth_sum += obj_depth[id];
            const float sf = std::sqrt(flow3d[3 * id] * flow3d[3 * id] + flow3d[3 * id + 2] * flow3d[3 * id + 2]);
            if (sf < sf_mg_thres) sf_count += 1;
        }
        if (sf_count / ObjId[i].size() > sf_ds_thres) { for (int id : ObjId[i]) obj_label[id] = 0; continue; }
        if (depth_sum / ObjId[i].size() > th_depth_obj || ObjId[i].size() < 150) { for (int id : ObjId[i]) obj_label[id] = -1; continue; }
        ObjIdNew.push_back(ObjId[i]); SemPosNew.push_back(sem_posi[i]);
    }
    if (f_id == 1) *max_id = 1;                               // (:1843-1896) identity: the last frame's object with the points' dominant last label, else a new id
    if ((int)ObjIdNew.size() > max_objects) return VIDO_E_CAPACITY;
    int off = 0;
    for (size_t i = 0; i < ObjIdNew.size(); i++) {
        std::vector<int> Lb_last; for (int id : ObjIdNew[i]) Lb_last.push_back(last_sem_label[id]);
        const int New_lab = most_frequent(Lb_last);
        bool exist = false; int lab = 0;
        if (*max_id != 1) for (int k = 0; k < n_last; k++) if (last_sem_position[k] == New_lab && last_obj_stat[k]) { lab = last_mod_label[k]; exist = true; break; }
        if (!exist) { lab = *max_id; *max_id = *max_id + 1; }
        for (int id : ObjIdNew[i]) obj_label[id] = lab;
        mod_label_out[i] = lab; sem_position_out[i] = SemPosNew[i];
        obj_off[i] = off; for (int id : ObjIdNew[i]) obj_ids[off++] = id;
    }
    obj_off[ObjIdNew.size()] = off;
    *n_objects = (int)ObjIdNew.size();
    return VIDO_OK;
}


/* Frame::UnprojectStereoStat / UnprojectStereoObject / ObtainFlowDepth*, addnoise = 1 (Frame.cc:706-716, 737-750, 773-790, 833-845, 860-872): every call builds a FRESH
 * cv::RNG seeded with time(NULL) and adds ONE draw  z + rng.gaussian(z*z / (725*0.5) * 0.15)  to the depth — a deterministic function of (z, seed), the same offset for every
 * point measured within one second.  cv::RNG (OpenCV 3.4, third-party, not in /root/reference: restated from its published source, parity unpinned): the state is the
 * seed itself (0 -> 0xffffffff), next() is the multiply-with-carry  state = (uint32)state * 4164903690 + (state >> 32); gaussian() is the 128-strip ziggurat of
 * modules/core/src/rand.cpp::randn_0_1_32f on the state's low word, scaled by sigma in double.  seed 0 here = "now" (time(NULL)), as the reference. */
float vido_depth_noise(float z, unsigned seed)
{
    struct Zig { unsigned kn[128]; float wn[128], fn[128]; };
    static const Zig Z = [] {                                 // (function-local static: built once, thread-safe — two Systems may draw their first sample at the same time)
        Zig z{};
        const double m1 = 2147483648.0; double dn = 3.442619855899, tn = dn; const double vn = 9.91256303526217e-3;
        const double q = vn / std::exp(-.5 * dn * dn);
        z.kn[0] = (unsigned)((dn / q) * m1); z.kn[1] = 0;
        z.wn[0] = (float)(q / m1); z.wn[127] = (float)(dn / m1);
        z.fn[0] = 1.f; z.fn[127] = (float)std::exp(-.5 * dn * dn);
        for (int i = 126; i >= 1; i--) {
            dn = std::sqrt(-2. * std::log(vn / dn + std::exp(-.5 * dn * dn)));
            z.kn[i + 1] = (unsigned)((dn / tn) * m1); tn = dn;
            z.fn[i] = (float)std::exp(-.5 * dn * dn); z.wn[i] = (float)(dn / m1);
        }
        return z;
    }();
    const unsigned* kn = Z.kn; const float *wn = Z.wn, *fn = Z.fn;
    if (seed == 0) seed = (unsigned)time(NULL);
    uint64_t st = seed ? (uint64_t)seed : 0xffffffffull;
    auto next = [&]() { st = (uint64_t)(unsigned)st * 4164903690u + (unsigned)(st >> 32); return (unsigned)st; };
    const float r = 3.442620f, rng_flt = 2.3283064365386962890625e-10f;
    float x, y;
    for (;;) {
        const int hz = (int)st; next();
        const int iz = hz & 127;
        x = hz * wn[iz];
        if ((unsigned)std::abs(hz) < kn[iz]) break;
        if (iz == 0) {                                        // the tail
            do {
                x = (unsigned)st * rng_flt; next();
                y = (unsigned)st * rng_flt; next();
                x = (float)(-std::log(x + FLT_MIN) * 0.2904764);
                y = (float)-std::log(y + FLT_MIN);
            } while (y + y < x * x);
            x = hz > 0 ? r + x : -r - x;
            break;
        }
        y = (unsigned)st * rng_flt; next();                   // a wedge
        if (fn[iz] + y * (fn[iz - 1] - fn[iz]) < (float)std::exp(-.5 * x * x)) break;
    }
    const double sigma = z * z / (725 * 0.5) * 0.15;          // (float * float) / double * double, as written in Frame.cc
    return (float)(z + (double)x * sigma);
}
}  // extern "C"
