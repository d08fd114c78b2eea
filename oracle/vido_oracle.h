/* vido_oracle.h — CPU ORACLE (TEST INFRASTRUCTURE ONLY).
 *
 * Plain-C restatement of the reference's per-frame hot path (bxh1/VIDO-SLAM), written from the
 * reference sources cited next to every function.  It exists to CHECK the HIP path:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link/load it.
 * The product (vido-slam_amd/, include/) never includes or calls anything in oracle/.
 *
 * PARITY STATUS: the reference has no tests / golden vectors for this path and its C++ cannot be
 * built here (OpenCV, Eigen, CXSparse absent) => the OpenCV-side primitives (cvtColor, resize,
 * GaussianBlur, FAST, fastAtan2) are restated from OpenCV-3.4 semantics: "parity unpinned".
 * Everything taken from the reference's own files follows them line by line in meaning.
 */
#ifndef VIDO_ORACLE_H
#define VIDO_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define VO_MAX_LEVELS 16
#define VO_EDGE_THRESHOLD 19
#define VO_HALF_PATCH 15
#define VO_PATCH 31

typedef struct { float x, y, size, angle, response; int octave; } vo_keypoint;

typedef struct {
    int   n_features, n_levels, ini_th, min_th;
    float scale_factor;
    float scale[VO_MAX_LEVELS], inv_scale[VO_MAX_LEVELS];
    int   n_per_level[VO_MAX_LEVELS];
    int   umax[VO_HALF_PATCH + 1];
} vo_orb_params;

/* ---- ORB front-end (orb_oracle.c) ---- */
void vo_orb_params_init(vo_orb_params* p, int n_features, float scale_factor, int n_levels, int ini_th, int min_th);
void vo_level_size(const vo_orb_params* p, int w, int h, int level, int* lw, int* lh);
void vo_bgr2gray(const uint8_t* src, int sstride, int w, int h, int channels, int rgb_order, uint8_t* dst, int dstride);
void vo_resize_linear_u8(const uint8_t* src, int sstride, int sw, int sh, uint8_t* dst, int dstride, int dw, int dh);
void vo_gaussian_blur7(const uint8_t* src, int sstride, int w, int h, uint8_t* dst, int dstride);
float vo_fast_atan2(float y, float x);
/* cv::FAST(TYPE_9_16) on one sub-image; returns count, writes (x,y,score) triplets as ints */
int  vo_fast9_16(const uint8_t* img, int stride, int w, int h, int threshold, int nonmax, int* out_xys, int cap);
/* threshold-free corner score S(p) (=cornerScore with threshold 0 for a corner; 0 if none) */
int  vo_fast_score_map(const uint8_t* img, int stride, int w, int h, uint8_t* score, int sstride);
/* ComputeKeyPointsOctTree FAST stage for one level: candidates in the reference's order, coordinates
 * relative to (minBorderX,minBorderY) exactly as handed to DistributeOctTree. */
int  vo_level_candidates(const vo_orb_params* p, const uint8_t* img, int stride, int w, int h,
                         float* cx, float* cy, float* cresp, int cap);
int  vo_distribute_octree(const float* cx, const float* cy, const float* cresp, int n,
                          int minX, int maxX, int minY, int maxY, int N, int* out_idx, int cap);
float vo_ic_angle(const uint8_t* img, int stride, int x, int y, const int* umax);
void vo_brief(const uint8_t* blurred, int stride, int x, int y, float angle_deg, uint8_t desc[32]);
/* full ORBextractor::operator(): gray u8 in, keypoints (level-0 coordinates) + rBRIEF descriptors out.
 * Also optionally returns the pyramid (tightly packed levels, concatenated) for debugging. */
int  vo_orb_extract(const vo_orb_params* p, const uint8_t* gray, int stride, int w, int h,
                    vo_keypoint* kps, uint8_t* desc, int cap, int* n_cand_per_level);
int  vo_orb_pyramid(const vo_orb_params* p, const uint8_t* gray, int stride, int w, int h, uint8_t* out, int* offsets);
void vo_hamming_match(const uint8_t* a, int na, const uint8_t* b, int nb, int* idx, int* dist);

#ifdef __cplusplus
}
#endif
#endif
