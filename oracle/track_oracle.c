/* track_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE ONLY; see vido_oracle.h).
 *
 * Restates the data-parallel stages of the per-frame tracking front-end:
 *   Tracking.cc:299-322   depth pre-scale (in place on the caller's buffer)
 *   Frame.cc:72-100       static-candidate filter over the ORB keypoints (UseSampleFeature==0)
 *   Frame.cc:165-177      depth gather for the candidates
 *   Frame.cc:184-211      dense object sampling on a 4-px lattice
 *   Tracking.cc:369-421   cross-frame hand-over gathers (static depth; object depth + label)
 *   Tracking.cc:3291-3357 UpdateMask (per lost label: scatter last mask through last flow)
 *   Frame.cc:706-771      UnprojectStereoStat/Object (addnoise = 0 branch)
 *   Tracking.cc:1582-1668 GetSceneFlowObj
 * All in the reference's float arithmetic.
 */
#include "vido_oracle.h"
#include <stdlib.h>
#include <string.h>

/* Tracking.cc:299-322.  mode: 0 OMD (d/factor), 1 KITTI (bf/(d/factor)), 2 KAIST (scale*bf/(d/factor)). */
void vo_depth_prescale(float* d, int n, int mode, float factor, float bf, float scale)
{
    for (int i = 0; i < n; i++) {
        if (d[i] < 0) d[i] = 0;
        else if (mode == 0) d[i] = d[i] / factor;
        else if (mode == 1) d[i] = bf / (d[i] / factor);
        else d[i] = scale * bf / (d[i] / factor);
    }
}

/* Frame.cc:72-100 + :165-177.  Outputs: index of the surviving keypoint, correspondence (kp+flow),
 * flow, depth (-1 if not > 0).  Returns count. */
int vo_static_candidates(const vo_keypoint* kps, int n, const float* depth, const float* flow, const int32_t* mask,
                         int w, int h, float th_depth, int* out_idx, float* out_corr, float* out_flow, float* out_depth)
{
    int m = 0;
    for (int i = 0; i < n; i++) {
        int x = (int)kps[i].x, y = (int)kps[i].y;
        if (mask[(size_t)y * w + x] != 0) continue;
        float dd = depth[(size_t)y * w + x];
        if (dd > th_depth || dd <= 0) continue;
        float fx = flow[((size_t)y * w + x) * 2], fy = flow[((size_t)y * w + x) * 2 + 1];
        if (fx != 0 && fy != 0) {
            if (kps[i].x + fx < (float)w && kps[i].y + fy < (float)h && kps[i].x < (float)w && kps[i].y < (float)h) {
                out_idx[m] = i;
                out_corr[2 * m] = kps[i].x + fx; out_corr[2 * m + 1] = kps[i].y + fy;
                out_flow[2 * m] = fx; out_flow[2 * m + 1] = fy;
                /* :165-177: d = imDepth.at<float>(v,u) with float->int truncation of the key position */
                float d2 = depth[(size_t)(int)kps[i].y * w + (int)kps[i].x];
                out_depth[m] = d2 > 0 ? d2 : -1.f;
                m++;
            }
        }
    }
    return m;
}

/* Frame.cc:184-211 */
int vo_dense_object_samples(const float* depth, const float* flow, const int32_t* mask, int w, int h, float th_obj, int step,
                            float* keys, float* corr, float* odepth, int32_t* label, float* oflow, int cap)
{
    int m = 0;
    for (int i = 0; i < h; i += step)
        for (int j = 0; j < w; j += step) {
            size_t p = (size_t)i * w + j;
            if (mask[p] != 0 && depth[p] < th_obj && depth[p] > 0) {
                const float fx = flow[2 * p], fy = flow[2 * p + 1];
                if (j + fx < (float)w && j + fx > 0 && i + fy < (float)h && i + fy > 0) {
                    if (m < cap) {
                        oflow[2 * m] = fx; oflow[2 * m + 1] = fy;
                        corr[2 * m] = j + fx; corr[2 * m + 1] = i + fy;
                        keys[2 * m] = (float)j; keys[2 * m + 1] = (float)i;
                        odepth[m] = depth[p]; label[m] = mask[p];
                    }
                    m++;
                }
            }
        }
    return m;
}

/* Tracking.cc:369-391 */
void vo_gather_static_depth(const float* keys, int n, const float* depth, int w, int h, float* out)
{
    for (int i = 0; i < n; i++) {
        const int v = (int)keys[2 * i + 1], u = (int)keys[2 * i];
        out[i] = -1.f;
        if (u < (w - 1) && u > 0 && v < (h - 1) && v > 0) { float d = depth[(size_t)v * w + u]; if (d > 0) out[i] = d; }
    }
}

/* Tracking.cc:398-421 */
void vo_gather_object_depth_label(const float* keys, int n, const float* depth, const int32_t* mask, int w, int h, float th_obj,
                                  float* out_d, int32_t* out_label)
{
    for (int i = 0; i < n; i++) {
        const int u = (int)keys[2 * i], v = (int)keys[2 * i + 1];
        if (u < (w - 1) && u > 0 && v < (h - 1) && v > 0 && depth[(size_t)v * w + u] < th_obj && depth[(size_t)v * w + u] > 0) {
            out_d[i] = depth[(size_t)v * w + u]; out_label[i] = mask[(size_t)v * w + u];
        } else { out_d[i] = 0.1f; out_label[i] = 0; }
    }
}

/* Tracking.cc:3291-3357.  last_label / last_corr: per dense object point of the last frame.
 * mask_cur is modified in place.  Returns the number of labels that were recovered. */
static int cmp_int(const void* a, const void* b) { int x = *(const int*)a, y = *(const int*)b; return x < y ? -1 : x > y; }
int vo_update_mask(const int32_t* last_label, const float* last_corr, int n, const int32_t* mask_last, const float* flow_last,
                   int32_t* mask_cur, int w, int h, int32_t* recovered, int cap)
{
    int* uni = (int*)malloc(sizeof(int) * (n + 1)); int nu = 0, nrec = 0;
    memcpy(uni, last_label, sizeof(int) * n); qsort(uni, n, sizeof(int), cmp_int);
    for (int i = 0; i < n; i++) if (nu == 0 || uni[nu - 1] != uni[i]) uni[nu++] = uni[i];
    int* tmp = (int*)malloc(sizeof(int) * (n + 1));
    for (int a = 0; a < nu; a++) {
        int nt = 0;
        for (int j = 0; j < n; j++) {
            if (last_label[j] != uni[a]) continue;
            const int u = (int)last_corr[2 * j], v = (int)last_corr[2 * j + 1];
            if (u < w && u > 0 && v < h && v > 0) tmp[nt++] = mask_cur[(size_t)v * w + u];
        }
        if (nt < 100) continue;
        /* most frequent label; SortPairInt sorts by count descending — ties are resolved by std::sort's
         * unspecified order in the reference; restated as: highest count, then smallest label */
        qsort(tmp, nt, sizeof(int), cmp_int);
        int best = tmp[0], bc = 0, run = 0;
        for (int j = 0; j < nt; j++) { run = (j > 0 && tmp[j] == tmp[j - 1]) ? run + 1 : 1; if (run > bc) { bc = run; best = tmp[j]; } }
        if (best == 0) {
            for (int j = 0; j < h; j++)
                for (int k = 0; k < w; k++)
                    if (mask_last[(size_t)j * w + k] == uni[a]) {
                        const int fx = (int)flow_last[((size_t)j * w + k) * 2], fy = (int)flow_last[((size_t)j * w + k) * 2 + 1];
                        if (k + fx < w && k + fx > 0 && j + fy < h && j + fy > 0) mask_cur[(size_t)(j + fy) * w + (k + fx)] = uni[a];
                    }
            if (nrec < cap) recovered[nrec] = uni[a];
            nrec++;
        }
    }
    free(uni); free(tmp);
    return nrec;
}

/* Frame.cc:706-771 (addnoise=0): back-project (u,v,z) and move to the world with Twc = inv(Tcw).
 * Tcw: row-major 4x4 float.  cv::Mat float products accumulate in double (cv::gemm) then round to float. */
void vo_unproject_world(const float* keys, const float* z, int n, float fx, float fy, float cx, float cy, const float* Tcw, float* out)
{
    const float invfx = 1.0f / fx, invfy = 1.0f / fy;
    float Rwl[9], twl[3];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Rwl[r * 3 + c] = Tcw[c * 4 + r];
    for (int r = 0; r < 3; r++) {
        double s = 0; for (int c = 0; c < 3; c++) s += (double)(-Rwl[r * 3 + c]) * (double)Tcw[c * 4 + 3];
        twl[r] = (float)s;
    }
    for (int i = 0; i < n; i++) {
        const float zz = z[i];
        if (!(zz > 0)) { out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = 0; continue; }
        const float x = (keys[2 * i] - cx) * zz * invfx, y = (keys[2 * i + 1] - cy) * zz * invfy;
        for (int r = 0; r < 3; r++) {
            double s = (double)Rwl[r * 3] * x + (double)Rwl[r * 3 + 1] * y + (double)Rwl[r * 3 + 2] * zz;
            out[3 * i + r] = (float)s + twl[r];
        }
    }
}

/* GetSceneFlowObj, Tracking.cc:1582-1668: flow3d = X_w(cur) - X_w(last) where both labels > 0, else label -1. */
void vo_scene_flow(const float* Xw_last, const float* Xw_cur, const int32_t* sem_last, const int32_t* sem_cur, int n,
                   float* flow3d, int32_t* obj_label_inout)
{
    for (int i = 0; i < n; i++) {
        if (sem_cur[i] <= 0 || sem_last[i] <= 0) { obj_label_inout[i] = -1; flow3d[3 * i] = flow3d[3 * i + 1] = flow3d[3 * i + 2] = 0; continue; }
        for (int r = 0; r < 3; r++) flow3d[3 * i + r] = Xw_cur[3 * i + r] - Xw_last[3 * i + r];
    }
}
