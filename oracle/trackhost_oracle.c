/* trackhost_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE ONLY): literal restatements of the reference's host-side tracking bookkeeping,
 * statement by statement, with the reference's own data-structure walks (std::vector push_back order, linear label searches, the O(N*M) "already used"
 * scans) — deliberately NOT the algorithms the product uses (grid hash, binary searches).  Only tests/ may call this.
 *   vo_undistort_points     Frame::UndistortKeyPoints           vido_slam/src/Frame.cc:603-633   (cv::undistortPoints is OpenCV 3.4, third party, not in tree:
 *                                                                 its published 5-iteration fixed-point algorithm is restated -> "parity unpinned" for that call)
 *   vo_renew_static         Tracking::RenewFrameInfo (static)   vido_slam/src/Tracking.cc:2973-3075
 *   vo_renew_objects        Tracking::RenewFrameInfo (objects)  vido_slam/src/Tracking.cc:3116-3270
 *   vo_dyn_obj_tracking     Tracking::DynObjTracking            vido_slam/src/Tracking.cc:1670-1912
 *   vo_static_tracklets / vo_dynamic_tracklets                  Tracking::GetStaticTrack / GetDynamicTrackNew   vido_slam/src/Tracking.cc:2514-2613, 2615-2720
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { const int32_t* mask; const float* depth; const float* flow; int32_t width, height; } vo_host_maps;

/* cv::undistortPoints(src, dst, K, dist, noArray(), K), OpenCV 3.4 calib3d undistort.cpp cvUndistortPointsInternal: camera matrix and coefficients converted to double, x0 = (u - cx) * (1/fx); 5 iterations of
 * icdist = 1/(1 + ((k3 r2 + k2) r2 + k1) r2), delta = tangential terms, x = (x0 - deltaX) icdist; then u' = x fx + cx.  Double arithmetic, float in/out. */
void vo_undistort_points(const float* xy, int n, const float* K, const float* dist, float* out)
{
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
    const double k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3], k3 = dist[4];
    for (int i = 0; i < n; i++) {
        if (dist[0] == 0.0f) { out[2 * i] = xy[2 * i]; out[2 * i + 1] = xy[2 * i + 1]; continue; }     /* Frame.cc:605-609 */
        const double ifx = 1. / fx, ify = 1. / fy;
        double x = (xy[2 * i] - cx) * ifx, y = (xy[2 * i + 1] - cy) * ify;      /* x = (x - cx)*ifx */
        const double x0 = x, y0 = y;
        for (int j = 0; j < 5; j++) {
            const double r2 = x * x + y * y;
            const double icdist = 1. / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
            const double deltaX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
            const double deltaY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        out[2 * i] = (float)(x * fx + cx); out[2 * i + 1] = (float)(y * fy + cy);
    }
}

/* Tracking.cc:2973-3075.  Returns the number of kept static features; element k: src[k] index into stat_xy (inlier[k] >= 0) or sample_xy (inlier[k] == -1). */
int vo_renew_static(const vo_host_maps* m, const float* stat_xy, const int32_t* TM_sta, int n_tm, const float* sample_xy, int n_sample, int max_num_sta,
                    int32_t* src, int32_t* inlier, float* flow, int cap)
{
    const int cols = m->width, rows = m->height;
    float* keys = (float*)malloc(sizeof(float) * 2 * (size_t)(cap + 1)); int n = 0;
    /* (1) Save the inliers from last frame */
    for (int i = 0; i < n_tm; ++i) {
        if (TM_sta[i] == -1) continue;
        const float px = stat_xy[2 * TM_sta[i]], py = stat_xy[2 * TM_sta[i] + 1];
        int x = (int)px, y = (int)py;
        if (x >= cols || y >= rows || x <= 0 || y <= 0) continue;
        if (m->mask[(size_t)y * cols + x] != 0) continue;
        if (m->depth[(size_t)y * cols + x] > 40 || m->depth[(size_t)y * cols + x] <= 0) continue;
        float flow_xe = m->flow[2 * ((size_t)y * cols + x)], flow_ye = m->flow[2 * ((size_t)y * cols + x) + 1];
        if (flow_xe != 0 && flow_ye != 0) {
            if (px + flow_xe < cols && py + flow_ye < rows && px + flow_xe > 0 && py + flow_ye > 0) {
                if (n < cap) { keys[2 * n] = px; keys[2 * n + 1] = py; src[n] = TM_sta[i]; inlier[n] = TM_sta[i]; flow[2 * n] = flow_xe; flow[2 * n + 1] = flow_ye; }
                n++;
            }
        }
        if (n > max_num_sta) break;
    }
    /* (2) Save extra keypoints to make it a fixed number */
    int tot_num = n, start_id = 0, step = 20;
    const int n_check = n < cap ? n : cap;                       /* mvKeysTmpCheck = mvKeysTmp (copied here, never extended) */
    while (tot_num < max_num_sta) {
        if (start_id == step) break;
        for (int i = start_id; i < n_sample; i = i + step) {
            float min_dist = 100; int used = 0;
            for (int j = 0; j < n_check; ++j) {
                float cur_dist = sqrtf((keys[2 * j] - sample_xy[2 * i]) * (keys[2 * j] - sample_xy[2 * i]) + (keys[2 * j + 1] - sample_xy[2 * i + 1]) * (keys[2 * j + 1] - sample_xy[2 * i + 1]));
                if (cur_dist < min_dist) min_dist = cur_dist;
                if (min_dist < 1.0) { used = 1; break; }
            }
            if (used) continue;
            int x = (int)sample_xy[2 * i], y = (int)sample_xy[2 * i + 1];
            if (x >= cols || y >= rows || x <= 0 || y <= 0) continue;
            if (m->mask[(size_t)y * cols + x] != 0) continue;
            if (m->depth[(size_t)y * cols + x] > 40 || m->depth[(size_t)y * cols + x] <= 0) continue;
            float flow_xe = m->flow[2 * ((size_t)y * cols + x)], flow_ye = m->flow[2 * ((size_t)y * cols + x) + 1];
            if (flow_xe != 0 && flow_ye != 0) {
                if (sample_xy[2 * i] + flow_xe < cols && sample_xy[2 * i + 1] + flow_ye < rows && sample_xy[2 * i] + flow_xe > 0 && sample_xy[2 * i + 1] + flow_ye > 0) {
                    if (n < cap) { src[n] = i; inlier[n] = -1; flow[2 * n] = flow_xe; flow[2 * n + 1] = flow_ye; }
                    n++;
                    tot_num = tot_num + 1;
                }
            }
            if (tot_num >= max_num_sta) break;
        }
        start_id = start_id + 1;
    }
    free(keys);
    return n;
}

/* Tracking.cc:3116-3270.  Returns the number of object features written (all seven output lists have that length). */
int vo_renew_objects(const vo_host_maps* m, const float* obj_xy, const int32_t* obj_label,
                     int n_objects, const int32_t* inl_off, const int32_t* inl_ids, const uint8_t* obj_stat, const int32_t* sem_position, const int32_t* mod_label,
                     const float* tmp_xy, const float* tmp_depth, const int32_t* tmp_sem, const float* tmp_flow, const float* tmp_corr, int n_tmp, int max_num_obj,
                     float* keys, float* depth, int32_t* sem, float* flow, float* corr, int32_t* inlier, int32_t* label, int cap)
{
    const int cols = m->width, rows = m->height;
    int n = 0;
    int* ObjFeaCount = (int*)malloc(sizeof(int) * (size_t)(n_objects + 1));
#define VO_PUSH(kx, ky, d, s, fx_, fy_, cx_, cy_, inl, lab) do { if (n < cap) { keys[2 * n] = (kx); keys[2 * n + 1] = (ky); depth[n] = (d); sem[n] = (s); flow[2 * n] = (fx_); \
        flow[2 * n + 1] = (fy_); corr[2 * n] = (cx_); corr[2 * n + 1] = (cy_); inlier[n] = (inl); label[n] = (lab); } n++; } while (0)
    /* (1) Again, save the inliers from last frame */
    for (int i = 0; i < n_objects; ++i) {
        if (!obj_stat[i]) { ObjFeaCount[i] = -1; continue; }
        int count = 0;
        for (int j = inl_off[i]; j < inl_off[i + 1]; ++j) {
            const int x = (int)obj_xy[2 * inl_ids[j]], y = (int)obj_xy[2 * inl_ids[j] + 1];
            if (x >= cols || y >= rows || x <= 0 || y <= 0) continue;
            if (m->mask[(size_t)y * cols + x] != 0 && m->depth[(size_t)y * cols + x] < 25 && m->depth[(size_t)y * cols + x] > 0) {
                const float flow_x = m->flow[2 * ((size_t)y * cols + x)], flow_y = m->flow[2 * ((size_t)y * cols + x) + 1];
                if (x + flow_x < cols && y + flow_y < rows && x + flow_x > 0 && y + flow_y > 0) {
                    VO_PUSH((float)x, (float)y, m->depth[(size_t)y * cols + x], m->mask[(size_t)y * cols + x], flow_x, flow_y, x + flow_x, y + flow_y, inl_ids[j], obj_label[inl_ids[j]]);
                    count = count + 1;
                }
            }
        }
        ObjFeaCount[i] = count;
    }
    /* (2) Save extra key points to make each object having a fixed number */
    const int n_check = n < cap ? n : cap;                       /* mvObjKeysTmpCheck */
    float* check = (float*)malloc(sizeof(float) * 2 * (size_t)(n_check + 1));
    memcpy(check, keys, sizeof(float) * 2 * (size_t)n_check);
    for (int i = 0; i < n_objects; ++i) {
        if (!obj_stat[i]) continue;
        int SemLabel = sem_position[i];
        int tot_num = ObjFeaCount[i];
        int start_id = 0, step = 15;
        while (tot_num < max_num_obj) {
            if (start_id == step) break;
            for (int j = start_id; j < n_tmp; j = j + step) {
                if (tmp_sem[j] != SemLabel) continue;
                float min_dist = 100; int used = 0;
                for (int k = 0; k < n_check; ++k) {
                    float cur_dist = sqrtf((check[2 * k] - tmp_xy[2 * j]) * (check[2 * k] - tmp_xy[2 * j]) + (check[2 * k + 1] - tmp_xy[2 * j + 1]) * (check[2 * k + 1] - tmp_xy[2 * j + 1]));
                    if (cur_dist < min_dist) min_dist = cur_dist;
                    if (min_dist < 1.0) { used = 1; break; }
                }
                if (used) continue;
                VO_PUSH(tmp_xy[2 * j], tmp_xy[2 * j + 1], tmp_depth[j], tmp_sem[j], tmp_flow[2 * j], tmp_flow[2 * j + 1], tmp_corr[2 * j], tmp_corr[2 * j + 1], -1, mod_label[i]);
                tot_num = tot_num + 1;
                if (tot_num >= max_num_obj) break;
            }
            start_id = start_id + 1;
        }
    }
    /* (3) Update new appearing objects: (3.1) unique labels (sorted), (3.2) labels owned by a live object, (3.3) every sample of the other labels */
    int* UniLab = (int*)malloc(sizeof(int) * (size_t)(n_tmp + 1)); int nu = 0;
    for (int j = 0; j < n_tmp; j++) { int k = 0; while (k < nu && UniLab[k] != tmp_sem[j]) k++; if (k == nu) UniLab[nu++] = tmp_sem[j]; }
    for (int a = 1; a < nu; a++) { int v = UniLab[a], b = a - 1; while (b >= 0 && UniLab[b] > v) { UniLab[b + 1] = UniLab[b]; b--; } UniLab[b + 1] = v; }
    char* NewLab = (char*)calloc((size_t)nu + 1, 1);
    for (int i = 0; i < n_objects; ++i) {
        int CurSemLabel = sem_position[i];
        for (int j = 0; j < nu; ++j) if (UniLab[j] == CurSemLabel && obj_stat[i]) { NewLab[j] = 1; break; }
    }
    for (int i = 0; i < nu; ++i) if (!NewLab[i]) for (int j = 0; j < n_tmp; j++) if (UniLab[i] == tmp_sem[j])
        VO_PUSH(tmp_xy[2 * j], tmp_xy[2 * j + 1], tmp_depth[j], tmp_sem[j], tmp_flow[2 * j], tmp_flow[2 * j + 1], tmp_corr[2 * j], tmp_corr[2 * j + 1], -1, -2);
#undef VO_PUSH
    free(ObjFeaCount); free(check); free(UniLab); free(NewLab);
    return n;
}

/* Tracking.cc:1670-1912.  obj_label is updated in place.  Returns the number of dynamic objects; obj_ids[obj_off[i] .. obj_off[i+1]) are object i's points. */
int vo_dyn_obj_tracking(const int32_t* sem_label, int32_t* obj_label, const float* obj_xy, const float* obj_depth, const float* flow3d, const int32_t* last_sem_label, int n,
                        const int32_t* last_sem_position, const uint8_t* last_obj_stat, const int32_t* last_mod_label, int n_last, int rows, int cols,
                        float fSFMgThres, float fSFDsThres, float mThDepthObj, int f_id, int32_t* max_id,
                        int32_t* obj_off, int32_t* obj_ids, int32_t* mod_label_out, int32_t* sem_position_out)
{
    /* find the unique labels in semantic label (sorted) */
    int* UniLab = (int*)malloc(sizeof(int) * (size_t)(n + 1)); int nu = 0;
    for (int i = 0; i < n; i++) { int k = 0; while (k < nu && UniLab[k] != sem_label[i]) k++; if (k == nu) UniLab[nu++] = sem_label[i]; }
    for (int a = 1; a < nu; a++) { int v = UniLab[a], b = a - 1; while (b >= 0 && UniLab[b] > v) { UniLab[b + 1] = UniLab[b]; b--; } UniLab[b + 1] = v; }
    /* collect the predicted labels and semantic labels in vector: Posi[j] = indices with label UniLab[j] (linear search per point, :1689-1700) */
    int* cnt = (int*)calloc((size_t)nu + 1, sizeof(int)); int* which = (int*)malloc(sizeof(int) * (size_t)(n + 1));
    for (int i = 0; i < n; ++i) {
        which[i] = -1;
        if (obj_label[i] == -1) continue;
        for (int j = 0; j < nu; ++j) if (sem_label[i] == UniLab[j]) { which[i] = j; cnt[j]++; break; }
    }
    int** Posi = (int**)malloc(sizeof(int*) * (size_t)(nu + 1)); int* fill = (int*)calloc((size_t)nu + 1, sizeof(int));
    for (int j = 0; j < nu; j++) Posi[j] = (int*)malloc(sizeof(int) * (size_t)(cnt[j] + 1));
    for (int i = 0; i < n; ++i) if (which[i] >= 0) Posi[which[i]][fill[which[i]]++] = i;
    /* save objects only from Posi() -> ObjId() */
    int* ObjSel = (int*)malloc(sizeof(int) * (size_t)(nu + 1)); int nobj = 0;              /* ObjId[k] = Posi[ObjSel[k]], sem_posi[k] = UniLab[ObjSel[k]] */
    const int shrin_thr_row = 10, shrin_thr_col = 20;
    for (int i = 0; i < nu; ++i) {
        if (cnt[i] == 0) continue;                              /* the reference divides by zero here (0/0 = NaN, comparison false, the empty object is kept and dropped at the 150-point test) */
        float count = 0, count_thres = 0.5f;
        for (int j = 0; j < cnt[i]; ++j) {
            const float u = obj_xy[2 * Posi[i][j]], v = obj_xy[2 * Posi[i][j] + 1];
            if (v < shrin_thr_row || v > (rows - shrin_thr_row) || u < shrin_thr_col || u > (cols - shrin_thr_col)) count = count + 1;
        }
        if (count / cnt[i] > count_thres) { for (int k = 0; k < cnt[i]; ++k) obj_label[Posi[i][k]] = -1; continue; }
        ObjSel[nobj++] = i;
    }
    /* check scene flow distribution of each object and keep the dynamic object */
    int* NewSel = (int*)malloc(sizeof(int) * (size_t)(nobj + 1)); int nnew = 0;
    for (int i = 0; i < nobj; ++i) {
        const int* ids = Posi[ObjSel[i]]; const int sz = cnt[ObjSel[i]];
        float obj_center_depth = 0, sf_count = 0;
        for (int j = 0; j < sz; ++j) {
            obj_center_depth = obj_center_depth + obj_depth[ids[j]];
            float sf_norm = sqrtf(flow3d[3 * ids[j]] * flow3d[3 * ids[j]] + flow3d[3 * ids[j] + 2] * flow3d[3 * ids[j] + 2]);
            if (sf_norm < fSFMgThres) sf_count = sf_count + 1;
        }
        if (sf_count / sz > fSFDsThres) { for (int k = 0; k < sz; ++k) obj_label[ids[k]] = 0; continue; }
        else if (obj_center_depth / sz > mThDepthObj || sz < 150) { for (int k = 0; k < sz; ++k) obj_label[ids[k]] = -1; continue; }
        else NewSel[nnew++] = ObjSel[i];
    }
    /* relabel the objects that associate with the objects in last frame */
    if (f_id == 1) *max_id = 1;
    int off = 0;
    for (int i = 0; i < nnew; ++i) {
        const int* ids = Posi[NewSel[i]]; const int sz = cnt[NewSel[i]];
        /* std::map<int,int> dups over the last frame's labels of the points, then sort by count descending (SortPairInt): the most frequent label, the smallest
         * label among equally frequent ones (std::map iterates keys ascending and the sort of these few pairs keeps their order) */
        int New_lab = 0, best = -1;
        for (int a = 0; a < sz; a++) {
            const int lab = last_sem_label[ids[a]]; int c = 0;
            for (int b = 0; b < sz; b++) if (last_sem_label[ids[b]] == lab) c++;
            if (c > best || (c == best && lab < New_lab)) { best = c; New_lab = lab; }
        }
        int LabId;
        if (*max_id == 1) { LabId = *max_id; *max_id = *max_id + 1; }
        else {
            int exist = 0; LabId = 0;
            for (int k = 0; k < n_last; ++k) if (last_sem_position[k] == New_lab && last_obj_stat[k]) { LabId = last_mod_label[k]; exist = 1; break; }
            if (!exist) { LabId = *max_id; *max_id = *max_id + 1; }
        }
        for (int k = 0; k < sz; ++k) obj_label[ids[k]] = LabId;
        mod_label_out[i] = LabId; sem_position_out[i] = UniLab[NewSel[i]];
        obj_off[i] = off; for (int k = 0; k < sz; ++k) obj_ids[off++] = ids[k];
    }
    obj_off[nnew] = off;
    for (int j = 0; j < nu; j++) free(Posi[j]);
    free(Posi); free(fill); free(cnt); free(which); free(UniLab); free(ObjSel); free(NewSel);
    return nnew;
}

/* Tracking::GetStaticTrack / GetDynamicTrackNew (Tracking.cc:2514-2613, 2615-2720): tracklets from the per-frame association rows.  Row i (frame i+1) has
 * row_n[i] entries TM[row_off[i] + j] = index of the matched feature in frame i, or -1.  Output: tracklet t = (frame, feature) pairs
 * pairs[2 * trk_off[t] .. 2 * trk_off[t+1]); obj_id[t] = label of the tracklet's second element (dynamic; labels may be NULL).  Returns the tracklet count. */
int vo_tracklets(int n_rows, const int32_t* row_off, const int32_t* row_n, const int32_t* TM, const int32_t* labels, int32_t* trk_off, int32_t* pairs, int32_t* obj_id, int cap_trk, int cap_pairs)
{
    /* the reference keeps std::vector<std::vector<pair>>; modelled with per-tracklet linked chunks: first count lengths with the same walk, then fill */
    int ntrk = 0; int max_row = 0;
    for (int i = 0; i < n_rows; i++) if (row_n[i] > max_row) max_row = row_n[i];
    int* pre = (int*)malloc(sizeof(int) * (size_t)(max_row + 1)); int* cur = (int*)malloc(sizeof(int) * (size_t)(max_row + 1)); int npre = 0;
    int* len = (int*)calloc((size_t)cap_trk + 1, sizeof(int));
    for (int pass = 0; pass < 2; pass++) {
        ntrk = 0; npre = 0;
        int* fillp = NULL;
        if (pass == 1) { fillp = (int*)malloc(sizeof(int) * (size_t)(cap_trk + 1)); int o = 0; for (int t = 0; t < cap_trk; t++) { trk_off[t] = o; fillp[t] = o; o += len[t]; if (o > cap_pairs) o = cap_pairs; } trk_off[cap_trk] = o; }
        for (int i = 0; i < n_rows; ++i) {
            for (int j = 0; j < row_n[i]; ++j) {
                cur[j] = -1;
                const int m = TM[row_off[i] + j];
                if (m == -1) continue;
                int t;
                if (i > 0 && m < npre && pre[m] != -1) t = pre[m];          /* the match already belongs to a tracklet: append (frame i+1, j) */
                else { t = ntrk++; if (pass == 1 && t < cap_trk) { if (obj_id && labels) obj_id[t] = labels[row_off[i] + j]; if (fillp[t] < cap_pairs) { pairs[2 * fillp[t]] = i; pairs[2 * fillp[t] + 1] = m; } fillp[t]++; }
                       else if (pass == 0 && t < cap_trk) len[t]++; }           /* new tracklet: (frame i, m) first */
                if (t < cap_trk) { if (pass == 0) len[t]++; else { if (fillp[t] < cap_pairs) { pairs[2 * fillp[t]] = i + 1; pairs[2 * fillp[t] + 1] = j; } fillp[t]++; } }
                cur[j] = t;
            }
            npre = row_n[i]; memcpy(pre, cur, sizeof(int) * (size_t)npre);
        }
        if (pass == 1) free(fillp);
    }
    if (ntrk < cap_trk) trk_off[ntrk] = trk_off[ntrk];    /* offsets beyond ntrk are the running total */
    free(pre); free(cur); free(len);
    return ntrk;
}
