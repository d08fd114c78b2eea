/* orb_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE ONLY; see vido_oracle.h).
 *
 * Restates, in plain C, the ORB front-end of the reference:
 *   vido_slam/src/ORBextractor.cc  :67-94 IC_Angle, :98-137 computeOrbDescriptor, :400-460 ctor,
 *   :471-527 DivideNode, :529-753 DistributeOctTree, :755-843 ComputeKeyPointsOctTree,
 *   :1034-1105 operator(), :1107-1132 ComputePyramid
 * plus the OpenCV-3.4 primitives those lines call (cvtColor, resize INTER_LINEAR, GaussianBlur,
 * FAST 9/16, fastAtan2, cvRound) restated from their documented integer semantics — OpenCV is a
 * third-party dependency absent from /root/reference (find_package(OpenCV 3.3.4),
 * vido_slam/CMakeLists.txt:20): PARITY UNPINNED for those, see SURVEY.md App. B.
 *
 * Build with -ffp-contract=off: float expressions must not be fused (the HIP side is built the same).
 */
#include "vido_oracle.h"
#include "../include/vido_orb_pattern.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

static inline int cv_round_f(float v) { return (int)lrintf(v); }   /* cvRound: round-half-even */
static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline int cv_floor_f(float v) { return (int)floorf(v); }
static inline int cv_ceil_f(float v) { return (int)ceilf(v); }

/* ---- ORBextractor ctor, ORBextractor.cc:400-460 ---- */
void vo_orb_params_init(vo_orb_params* p, int n_features, float scale_factor, int n_levels, int ini_th, int min_th)
{
    memset(p, 0, sizeof *p);
    p->n_features = n_features; p->scale_factor = scale_factor; p->n_levels = n_levels;
    p->ini_th = ini_th; p->min_th = min_th;
    p->scale[0] = 1.0f;
    for (int i = 1; i < n_levels; i++) p->scale[i] = p->scale[i - 1] * scale_factor;        /* :409-413 */
    for (int i = 0; i < n_levels; i++) p->inv_scale[i] = 1.0f / p->scale[i];                /* :418-422 */
    float factor = 1.0f / scale_factor;                                                      /* :426-436 */
    float desired = n_features * (1 - factor) / (1 - (float)pow((double)factor, (double)n_levels));
    int sum = 0;
    for (int l = 0; l < n_levels - 1; l++) {
        p->n_per_level[l] = cv_round_f(desired);
        sum += p->n_per_level[l];
        desired *= factor;
    }
    p->n_per_level[n_levels - 1] = n_features - sum > 0 ? n_features - sum : 0;
    /* umax, :444-459 */
    int v, v0, vmax = cv_floor_f(VO_HALF_PATCH * sqrtf(2.f) / 2 + 1);
    int vmin = cv_ceil_f(VO_HALF_PATCH * sqrtf(2.f) / 2);
    const double hp2 = VO_HALF_PATCH * VO_HALF_PATCH;
    for (v = 0; v <= vmax; ++v) p->umax[v] = cv_round_d(sqrt(hp2 - v * v));
    for (v = VO_HALF_PATCH, v0 = 0; v >= vmin; --v) {
        while (p->umax[v0] == p->umax[v0 + 1]) ++v0;
        p->umax[v] = v0;
        ++v0;
    }
}

/* ComputePyramid level size, ORBextractor.cc:1111-1112 */
void vo_level_size(const vo_orb_params* p, int w, int h, int level, int* lw, int* lh)
{
    float s = p->inv_scale[level];
    *lw = cv_round_f((float)w * s);
    *lh = cv_round_f((float)h * s);
}

/* cvtColor BGR2GRAY / RGB2GRAY (u8), OpenCV 3.4 fixed point: coefficients 1868/9617/4899 >>14.
 * Call site: Tracking.cc:327-340. */
void vo_bgr2gray(const uint8_t* src, int sstride, int w, int h, int channels, int rgb_order, uint8_t* dst, int dstride)
{
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * sstride; uint8_t* d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++, s += channels) {
            int b = rgb_order ? s[2] : s[0], g = s[1], r = rgb_order ? s[0] : s[2];
            d[x] = (uint8_t)((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14);
        }
    }
}

/* cv::resize(..., INTER_LINEAR) for CV_8UC1, OpenCV 3.4 generic (non-IPP) path:
 * 11-bit coefficient tables, int32 horizontal pass, ">>4, *b >>16, +2 >>2" vertical pass.
 * Call site: ORBextractor.cc:1120. */
void vo_resize_linear_u8(const uint8_t* src, int sstride, int sw, int sh, uint8_t* dst, int dstride, int dw, int dh)
{
    double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
    int* xofs = (int*)malloc(sizeof(int) * dw); short* ialpha = (short*)malloc(sizeof(short) * 2 * dw);
    int* row0 = (int*)malloc(sizeof(int) * dw); int* row1 = (int*)malloc(sizeof(int) * dw);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor_f(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ialpha[2 * dx] = (short)cv_round_f((1.f - fx) * 2048.f);
        ialpha[2 * dx + 1] = (short)cv_round_f(fx * 2048.f);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor_f(fy);
        fy -= sy;
        short b0 = (short)cv_round_f((1.f - fy) * 2048.f), b1 = (short)cv_round_f(fy * 2048.f);
        int y0 = sy < 0 ? 0 : (sy > sh - 1 ? sh - 1 : sy);
        int y1 = sy + 1 < 0 ? 0 : (sy + 1 > sh - 1 ? sh - 1 : sy + 1);
        const uint8_t* S0 = src + (size_t)y0 * sstride; const uint8_t* S1 = src + (size_t)y1 * sstride;
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx], sx1 = sx + 1 < sw ? sx + 1 : sx;
            row0[dx] = S0[sx] * ialpha[2 * dx] + S0[sx1] * ialpha[2 * dx + 1];
            row1[dx] = S1[sx] * ialpha[2 * dx] + S1[sx1] * ialpha[2 * dx + 1];
        }
        uint8_t* D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; dx++) {
            int v = (((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2;
            D[dx] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
    free(xofs); free(ialpha); free(row0); free(row1);
}

static inline int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * n - 2 - i; }
    return i;
}

/* GaussianBlur 7x7 sigma=2 BORDER_REFLECT_101 on u8 (call site ORBextractor.cc:1079).
 * Restatement choice (parity unpinned): separable Q0.8 kernel {18,34,49,54,49,34,18}/256
 * (getGaussianKernel(7,2) rounded, centre adjusted so the taps sum to 256), exact 16-bit
 * intermediate, single round-to-nearest at the end. */
static const int VO_GK7[7] = {18, 34, 49, 54, 49, 34, 18};
void vo_gaussian_blur7(const uint8_t* src, int sstride, int w, int h, uint8_t* dst, int dstride)
{
    uint16_t* tmp = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int acc = 0;
            for (int k = -3; k <= 3; k++) acc += VO_GK7[k + 3] * src[(size_t)y * sstride + reflect101(x + k, w)];
            tmp[(size_t)y * w + x] = (uint16_t)acc;                  /* <= 255*256 */
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int acc = 0;
            for (int k = -3; k <= 3; k++) acc += VO_GK7[k + 3] * tmp[(size_t)reflect101(y + k, h) * w + x];
            dst[(size_t)y * dstride + x] = (uint8_t)((acc + 32768) >> 16);
        }
    free(tmp);
}

/* cv::fastAtan2 (degrees), OpenCV 3.4 scalar polynomial. Call site ORBextractor.cc:93. */
float vo_fast_atan2(float y, float x)
{
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON); c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON); c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* ---- cv::FAST TYPE_9_16 -------------------------------------------------------------- */
static const int RING_DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int RING_DY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

/* cornerScore<16>: largest threshold for which p is still a 9-contiguous corner, minus... exactly
 * OpenCV's iterative min/max over the 16 arcs, started at `threshold`. */
static int corner_score16(const uint8_t* ptr, const int* pixel, int threshold)
{
    const int K = 8, N = K * 3 + 1;
    int k, v = ptr[0];
    short d[25];
    for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
    int a0 = threshold;
    for (k = 0; k < 16; k += 2) {
        int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
        a = a < d[k + 3] ? a : d[k + 3];
        if (a <= a0) continue;
        for (int t = 4; t <= 8; t++) a = a < d[k + t] ? a : d[k + t];
        int m = a < d[k] ? a : d[k]; a0 = a0 > m ? a0 : m;
        m = a < d[k + 9] ? a : d[k + 9]; a0 = a0 > m ? a0 : m;
    }
    int b0 = -a0;
    for (k = 0; k < 16; k += 2) {
        int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
        b = b > d[k + 3] ? b : d[k + 3];
        for (int t = 4; t <= 5; t++) b = b > d[k + t] ? b : d[k + t];
        if (b >= b0) continue;
        for (int t = 6; t <= 8; t++) b = b > d[k + t] ? b : d[k + t];
        int m = b > d[k] ? b : d[k]; b0 = b0 < m ? b0 : m;
        m = b > d[k + 9] ? b : d[k + 9]; b0 = b0 < m ? b0 : m;
    }
    return -b0 - 1;
}

int vo_fast9_16(const uint8_t* img, int stride, int w, int h, int threshold, int nonmax, int* out, int cap)
{
    const int K = 8, N = 25;
    int pixel[25];
    for (int k = 0; k < 16; k++) pixel[k] = RING_DX[k] + RING_DY[k] * stride;
    for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
    if (threshold < 0) threshold = 0; if (threshold > 255) threshold = 255;
    uint8_t tab[512];
    for (int i = -255; i <= 255; i++) tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);
    if (w < 7 || h < 7) return 0;
    uint8_t* buf = (uint8_t*)calloc((size_t)3 * w, 1);
    int* cp = (int*)calloc((size_t)3 * (w + 1), sizeof(int));
    int n = 0;
    for (int i = 3; i < h - 2; i++) {
        const uint8_t* ptr = img + (size_t)i * stride + 3;
        uint8_t* curr = buf + (size_t)((i - 3) % 3) * w;
        int* cornerpos = cp + (size_t)((i - 3) % 3) * (w + 1) + 1;
        memset(curr, 0, w);
        int nc = 0;
        if (i < h - 3) {
            for (int j = 3; j < w - 3; j++, ptr++) {
                int v = ptr[0];
                const uint8_t* t = tab - v + 255;
                int d = t[ptr[pixel[0]]] | t[ptr[pixel[8]]];
                if (d == 0) continue;
                d &= t[ptr[pixel[2]]] | t[ptr[pixel[10]]];
                d &= t[ptr[pixel[4]]] | t[ptr[pixel[12]]];
                d &= t[ptr[pixel[6]]] | t[ptr[pixel[14]]];
                if (d == 0) continue;
                d &= t[ptr[pixel[1]]] | t[ptr[pixel[9]]];
                d &= t[ptr[pixel[3]]] | t[ptr[pixel[11]]];
                d &= t[ptr[pixel[5]]] | t[ptr[pixel[13]]];
                d &= t[ptr[pixel[7]]] | t[ptr[pixel[15]]];
                int found = 0;
                if (d & 1) {
                    int vt = v - threshold, count = 0;
                    for (int k = 0; k < N; k++) {
                        int x = ptr[pixel[k]];
                        if (x < vt) { if (++count > K) { found = 1; break; } } else count = 0;
                    }
                }
                if (!found && (d & 2)) {
                    int vt = v + threshold, count = 0;
                    for (int k = 0; k < N; k++) {
                        int x = ptr[pixel[k]];
                        if (x > vt) { if (++count > K) { found = 1; break; } } else count = 0;
                    }
                }
                if (found) {
                    cornerpos[nc++] = j;
                    if (nonmax) curr[j] = (uint8_t)corner_score16(ptr, pixel, threshold);
                }
            }
        }
        cornerpos[-1] = nc;
        if (i == 3) continue;
        const uint8_t* prev = buf + (size_t)((i - 4 + 3) % 3) * w;
        const uint8_t* pprev = buf + (size_t)((i - 5 + 3) % 3) * w;
        cornerpos = cp + (size_t)((i - 4 + 3) % 3) * (w + 1) + 1;
        nc = cornerpos[-1];
        for (int k = 0; k < nc; k++) {
            int j = cornerpos[k];
            int score = prev[j];
            if (!nonmax || (score > prev[j + 1] && score > prev[j - 1] &&
                            score > pprev[j - 1] && score > pprev[j] && score > pprev[j + 1] &&
                            score > curr[j - 1] && score > curr[j] && score > curr[j + 1])) {
                if (n < cap) { out[3 * n] = j; out[3 * n + 1] = i - 1; out[3 * n + 2] = score; }
                n++;
            }
        }
    }
    free(buf); free(cp);
    return n;
}

/* Threshold-free score map: S(p) = max over the 16 bright/dark 9-arcs of the arc's min |diff|, minus 1
 * (0 when no arc is one-signed).  "p is a FAST corner at threshold t"  <=>  S(p) >= t (t>=1).
 * Used only to cross-check the reformulation the HIP kernel relies on. */
int vo_fast_score_map(const uint8_t* img, int stride, int w, int h, uint8_t* score, int sstride)
{
    int n = 0;
    for (int y = 0; y < h; y++) memset(score + (size_t)y * sstride, 0, w);
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            const uint8_t* p = img + (size_t)y * stride + x;
            int d[25];
            for (int k = 0; k < 25; k++) d[k] = (int)p[0] - (int)p[RING_DX[k & 15] + RING_DY[k & 15] * stride];
            int best = 0;
            for (int s = 0; s < 16; s++) {
                int mn = 1 << 20, mx = -(1 << 20);
                for (int t = 0; t < 9; t++) { int v = d[s + t]; if (v < mn) mn = v; if (v > mx) mx = v; }
                if (mn > best) best = mn;          /* all ring darker than centre by >= mn */
                if (-mx > best) best = -mx;        /* all ring brighter */
            }
            if (best > 0) { score[(size_t)y * sstride + x] = (uint8_t)(best - 1); n++; }
        }
    return n;
}

/* FAST stage of ComputeKeyPointsOctTree for ONE level, ORBextractor.cc:759-819. */
int vo_level_candidates(const vo_orb_params* p, const uint8_t* img, int stride, int w, int h,
                        float* cx, float* cy, float* cresp, int cap)
{
    const float W = 30;
    const int minBorderX = VO_EDGE_THRESHOLD - 3, minBorderY = minBorderX;
    const int maxBorderX = w - VO_EDGE_THRESHOLD + 3, maxBorderY = h - VO_EDGE_THRESHOLD + 3;
    const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
    const int nCols = (int)(width / W), nRows = (int)(height / W);
    if (nCols <= 0 || nRows <= 0) return 0;
    const int wCell = (int)ceilf(width / nCols), hCell = (int)ceilf(height / nRows);
    int tmpcap = (wCell + 6) * (hCell + 6);
    int* tmp = (int*)malloc(sizeof(int) * 3 * tmpcap);
    int n = 0;
    for (int i = 0; i < nRows; i++) {
        const float iniY = (float)(minBorderY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBorderY - 3) continue;
        if (maxY > maxBorderY) maxY = (float)maxBorderY;
        for (int j = 0; j < nCols; j++) {
            const float iniX = (float)(minBorderX + j * wCell);
            float maxX = iniX + wCell + 6;
            if (iniX >= maxBorderX - 6) continue;
            if (maxX > maxBorderX) maxX = (float)maxBorderX;
            const uint8_t* sub = img + (size_t)(int)iniY * stride + (int)iniX;
            int sw = (int)maxX - (int)iniX, sh = (int)maxY - (int)iniY;
            int m = vo_fast9_16(sub, stride, sw, sh, p->ini_th, 1, tmp, tmpcap);
            if (m == 0) m = vo_fast9_16(sub, stride, sw, sh, p->min_th, 1, tmp, tmpcap);
            for (int k = 0; k < m; k++) {
                if (n < cap) {
                    cx[n] = (float)tmp[3 * k] + (float)(j * wCell);
                    cy[n] = (float)tmp[3 * k + 1] + (float)(i * hCell);
                    cresp[n] = (float)tmp[3 * k + 2];
                }
                n++;
            }
        }
    }
    free(tmp);
    return n;
}

/* ---- DistributeOctTree, ORBextractor.cc:529-753 (+DivideNode :471-527) ----------------------
 * std::list<ExtractorNode> is modelled by an index-linked list over a node pool.  The reference
 * sorts (size, node pointer) pairs (:674): pointer order is allocation-dependent there; the
 * restatement breaks size ties by node creation order (later-created = "larger pointer"). */
typedef struct {
    int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
    int* keys; int nkeys; int no_more; int prev, next; int seq;
} onode;
typedef struct { onode* n; int cnt, cap; int head, tail, size; } olist;

static int ol_new(olist* L)
{
    if (L->cnt == L->cap) { L->cap *= 2; L->n = (onode*)realloc(L->n, sizeof(onode) * L->cap); }
    onode* q = &L->n[L->cnt]; memset(q, 0, sizeof *q); q->prev = q->next = -1; q->seq = L->cnt;
    return L->cnt++;
}
static void ol_push_front(olist* L, int i)
{ L->n[i].prev = -1; L->n[i].next = L->head; if (L->head >= 0) L->n[L->head].prev = i; else L->tail = i; L->head = i; L->size++; }
static void ol_push_back(olist* L, int i)
{ L->n[i].next = -1; L->n[i].prev = L->tail; if (L->tail >= 0) L->n[L->tail].next = i; else L->head = i; L->tail = i; L->size++; }
static int ol_erase(olist* L, int i)
{
    int p = L->n[i].prev, nx = L->n[i].next;
    if (p >= 0) L->n[p].next = nx; else L->head = nx;
    if (nx >= 0) L->n[nx].prev = p; else L->tail = p;
    L->size--; free(L->n[i].keys); L->n[i].keys = NULL;
    return nx;
}

static void divide_node(olist* L, int self, const float* cx, const float* cy, int ch[4])
{
    for (int c = 0; c < 4; c++) ch[c] = ol_new(L);            /* may realloc: take pointers after */
    onode* s = &L->n[self];
    const int halfX = (int)ceilf((float)(s->URx - s->ULx) / 2);
    const int halfY = (int)ceilf((float)(s->BRy - s->ULy) / 2);
    onode *n1 = &L->n[ch[0]], *n2 = &L->n[ch[1]], *n3 = &L->n[ch[2]], *n4 = &L->n[ch[3]];
    n1->ULx = s->ULx; n1->ULy = s->ULy; n1->URx = s->ULx + halfX; n1->URy = s->ULy;
    n1->BLx = s->ULx; n1->BLy = s->ULy + halfY; n1->BRx = s->ULx + halfX; n1->BRy = s->ULy + halfY;
    n2->ULx = n1->URx; n2->ULy = n1->URy; n2->URx = s->URx; n2->URy = s->URy;
    n2->BLx = n1->BRx; n2->BLy = n1->BRy; n2->BRx = s->URx; n2->BRy = s->ULy + halfY;
    n3->ULx = n1->BLx; n3->ULy = n1->BLy; n3->URx = n1->BRx; n3->URy = n1->BRy;
    n3->BLx = s->BLx; n3->BLy = s->BLy; n3->BRx = n1->BRx; n3->BRy = s->BLy;
    n4->ULx = n3->URx; n4->ULy = n3->URy; n4->URx = n2->BRx; n4->URy = n2->BRy;
    n4->BLx = n3->BRx; n4->BLy = n3->BRy; n4->BRx = s->BRx; n4->BRy = s->BRy;
    for (int c = 0; c < 4; c++) { L->n[ch[c]].keys = (int*)malloc(sizeof(int) * (s->nkeys ? s->nkeys : 1)); L->n[ch[c]].nkeys = 0; }
    for (int i = 0; i < s->nkeys; i++) {
        int k = s->keys[i];
        onode* d;
        if (cx[k] < (float)n1->URx) d = (cy[k] < (float)n1->BRy) ? n1 : n3;
        else d = (cy[k] < (float)n1->BRy) ? n2 : n4;
        d->keys[d->nkeys++] = k;
    }
    for (int c = 0; c < 4; c++) if (L->n[ch[c]].nkeys == 1) L->n[ch[c]].no_more = 1;
}

typedef struct { int size, seq, node; } sp_pair;
static int sp_cmp(const void* a, const void* b)
{
    const sp_pair* x = (const sp_pair*)a; const sp_pair* y = (const sp_pair*)b;
    if (x->size != y->size) return x->size < y->size ? -1 : 1;
    return x->seq < y->seq ? -1 : (x->seq > y->seq ? 1 : 0);
}

int vo_distribute_octree(const float* cx, const float* cy, const float* cresp, int n,
                         int minX, int maxX, int minY, int maxY, int N, int* out_idx, int cap)
{
    olist L; L.cap = 64 + 8 * (n > N ? n : N); L.cnt = 0; L.n = (onode*)malloc(sizeof(onode) * L.cap);
    L.head = L.tail = -1; L.size = 0;
    int nIni = (int)roundf((float)(maxX - minX) / (maxY - minY));
    if (nIni < 1) nIni = 1;            /* reference divides by zero here for very tall images */
    const float hX = (float)(maxX - minX) / nIni;
    int* ini = (int*)malloc(sizeof(int) * nIni);
    for (int i = 0; i < nIni; i++) {
        int q = ol_new(&L); onode* ni = &L.n[q];
        ni->ULx = (int)(hX * (float)i); ni->ULy = 0;
        ni->URx = (int)(hX * (float)(i + 1)); ni->URy = 0;
        ni->BLx = ni->ULx; ni->BLy = maxY - minY;
        ni->BRx = ni->URx; ni->BRy = maxY - minY;
        ni->keys = (int*)malloc(sizeof(int) * (n ? n : 1)); ni->nkeys = 0;
        ol_push_back(&L, q); ini[i] = q;
    }
    for (int i = 0; i < n; i++) {
        int b = (int)(cx[i] / hX);
        if (b >= nIni) b = nIni - 1;   /* UB guard; unreachable for in-range keys */
        onode* q = &L.n[ini[b]]; q->keys[q->nkeys++] = i;
    }
    for (int it = L.head; it >= 0;) {
        if (L.n[it].nkeys == 1) { L.n[it].no_more = 1; it = L.n[it].next; }
        else if (L.n[it].nkeys == 0) it = ol_erase(&L, it);
        else it = L.n[it].next;
    }
    int finish = 0;
    sp_pair* vs = (sp_pair*)malloc(sizeof(sp_pair) * (4 * (size_t)(n + 4) + 16));
    sp_pair* vprev = (sp_pair*)malloc(sizeof(sp_pair) * (4 * (size_t)(n + 4) + 16));
    int nvs = 0;
    while (!finish) {
        int prevSize = L.size, nToExpand = 0;
        nvs = 0;
        for (int it = L.head; it >= 0;) {
            if (L.n[it].no_more) { it = L.n[it].next; continue; }
            int ch[4]; divide_node(&L, it, cx, cy, ch);
            for (int c = 0; c < 4; c++) {
                if (L.n[ch[c]].nkeys > 0) {
                    ol_push_front(&L, ch[c]);
                    if (L.n[ch[c]].nkeys > 1) { nToExpand++; vs[nvs].size = L.n[ch[c]].nkeys; vs[nvs].seq = L.n[ch[c]].seq; vs[nvs].node = ch[c]; nvs++; }
                } else { free(L.n[ch[c]].keys); L.n[ch[c]].keys = NULL; }
            }
            it = ol_erase(&L, it);
        }
        if (L.size >= N || L.size == prevSize) finish = 1;
        else if (L.size + nToExpand * 3 > N) {
            while (!finish) {
                prevSize = L.size;
                int nprev = nvs; memcpy(vprev, vs, sizeof(sp_pair) * nvs); nvs = 0;
                qsort(vprev, nprev, sizeof(sp_pair), sp_cmp);
                for (int j = nprev - 1; j >= 0; j--) {
                    int ch[4]; divide_node(&L, vprev[j].node, cx, cy, ch);
                    for (int c = 0; c < 4; c++) {
                        if (L.n[ch[c]].nkeys > 0) {
                            ol_push_front(&L, ch[c]);
                            if (L.n[ch[c]].nkeys > 1) { vs[nvs].size = L.n[ch[c]].nkeys; vs[nvs].seq = L.n[ch[c]].seq; vs[nvs].node = ch[c]; nvs++; }
                        } else { free(L.n[ch[c]].keys); L.n[ch[c]].keys = NULL; }
                    }
                    ol_erase(&L, vprev[j].node);
                    if (L.size >= N) break;
                }
                if (L.size >= N || L.size == prevSize) finish = 1;
            }
        }
    }
    int m = 0;
    for (int it = L.head; it >= 0; it = L.n[it].next) {
        onode* q = &L.n[it];
        int best = q->keys[0]; float mr = cresp[best];
        for (int k = 1; k < q->nkeys; k++) if (cresp[q->keys[k]] > mr) { best = q->keys[k]; mr = cresp[best]; }
        if (m < cap) out_idx[m] = best;
        m++;
    }
    for (int i = 0; i < L.cnt; i++) free(L.n[i].keys);
    free(L.n); free(ini); free(vs); free(vprev);
    return m;
}

/* IC_Angle, ORBextractor.cc:67-94 */
float vo_ic_angle(const uint8_t* img, int stride, int x, int y, const int* umax)
{
    int m_01 = 0, m_10 = 0;
    const uint8_t* center = img + (size_t)y * stride + x;
    for (int u = -VO_HALF_PATCH; u <= VO_HALF_PATCH; ++u) m_10 += u * center[u];
    for (int v = 1; v <= VO_HALF_PATCH; ++v) {
        int v_sum = 0, d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return vo_fast_atan2((float)m_01, (float)m_10);
}

/* computeOrbDescriptor, ORBextractor.cc:98-137 (dead in the reference; built per north_star). */
void vo_brief(const uint8_t* img, int stride, int x, int y, float angle_deg, uint8_t desc[32])
{
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    float angle = angle_deg * factorPI;
    float a = (float)cos((double)angle), b = (float)sin((double)angle);
    const uint8_t* center = img + (size_t)y * stride + x;
    for (int i = 0; i < 32; i++) {
        int val = 0;
        for (int k = 0; k < 8; k++) {
            const signed char* pt = VIDO_ORB_PATTERN[i * 8 + k];
            int t0 = center[cv_round_f(pt[0] * b + pt[1] * a) * stride + cv_round_f(pt[0] * a - pt[1] * b)];
            int t1 = center[cv_round_f(pt[2] * b + pt[3] * a) * stride + cv_round_f(pt[2] * a - pt[3] * b)];
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

/* pyramid with materialised reflect-101 border (EDGE_THRESHOLD+3 so the blur of a bordered level
 * equals the reference's blur-with-reflect of the bare level).  Returns total bytes. */
#define VO_BORDER (VO_EDGE_THRESHOLD)
int vo_orb_pyramid(const vo_orb_params* p, const uint8_t* gray, int stride, int w, int h, uint8_t* out, int* offsets)
{
    int off = 0; const uint8_t* prev = gray; int pw = w, ph = h, ps = stride;
    for (int l = 0; l < p->n_levels; l++) {
        int lw, lh; vo_level_size(p, w, h, l, &lw, &lh);
        if (offsets) offsets[l] = off;
        if (out) {
            uint8_t* dst = out + off;
            if (l == 0) for (int y = 0; y < h; y++) memcpy(dst + (size_t)y * lw, gray + (size_t)y * stride, w);
            else vo_resize_linear_u8(prev, ps, pw, ph, dst, lw, lw, lh);
            prev = dst; pw = lw; ph = lh; ps = lw;
        }
        off += lw * lh;
    }
    return off;
}

/* ORBextractor::operator(), ORBextractor.cc:1034-1105, with computeDescriptors (:1086) enabled. */
int vo_orb_extract(const vo_orb_params* p, const uint8_t* gray, int stride, int w, int h,
                   vo_keypoint* kps, uint8_t* desc, int cap, int* n_cand_per_level)
{
    int offsets[VO_MAX_LEVELS];
    int total = vo_orb_pyramid(p, gray, stride, w, h, NULL, offsets);
    uint8_t* pyr = (uint8_t*)malloc(total);
    vo_orb_pyramid(p, gray, stride, w, h, pyr, offsets);
    int nout = 0;
    for (int l = 0; l < p->n_levels; l++) {
        int lw, lh; vo_level_size(p, w, h, l, &lw, &lh);
        const uint8_t* img = pyr + offsets[l];
        int ccap = lw * lh / 4 + 16;
        float* cx = (float*)malloc(sizeof(float) * ccap); float* cy = (float*)malloc(sizeof(float) * ccap);
        float* cr = (float*)malloc(sizeof(float) * ccap);
        int nc = vo_level_candidates(p, img, lw, lw, lh, cx, cy, cr, ccap);
        if (n_cand_per_level) n_cand_per_level[l] = nc;
        const int minBX = VO_EDGE_THRESHOLD - 3, minBY = minBX, maxBX = lw - VO_EDGE_THRESHOLD + 3, maxBY = lh - VO_EDGE_THRESHOLD + 3;
        int* sel = (int*)malloc(sizeof(int) * (nc + 1));
        int ns = nc > 0 ? vo_distribute_octree(cx, cy, cr, nc, minBX, maxBX, minBY, maxBY, p->n_per_level[l], sel, nc) : 0;
        if (ns > 0) {
            uint8_t* blur = (uint8_t*)malloc((size_t)lw * lh);
            vo_gaussian_blur7(img, lw, lw, lh, blur, lw);
            const int scaledPatch = (int)(VO_PATCH * p->scale[l]);
            for (int i = 0; i < ns; i++) {
                float x = cx[sel[i]] + (float)minBX, y = cy[sel[i]] + (float)minBY;   /* :829-836 */
                int xi = cv_round_f(x), yi = cv_round_f(y);
                float ang = vo_ic_angle(img, lw, xi, yi, p->umax);                     /* :841-842 */
                if (nout < cap) {
                    vo_keypoint* k = &kps[nout];
                    k->angle = ang; k->response = cr[sel[i]]; k->octave = l; k->size = (float)scaledPatch;
                    if (desc) vo_brief(blur, lw, xi, yi, ang, desc + (size_t)32 * nout);
                    if (l != 0) { x *= p->scale[l]; y *= p->scale[l]; }               /* :1094-1100 */
                    k->x = x; k->y = y;
                }
                nout++;
            }
            free(blur);
        }
        free(cx); free(cy); free(cr); free(sel);
    }
    free(pyr);
    return nout;
}

/* Brute-force 256-bit Hamming matcher.  NO reference call site exists: defined
 * by north_star; smallest distance, lowest index on ties.  PARITY UNPINNED. */
void vo_hamming_match(const uint8_t* a, int na, const uint8_t* b, int nb, int* idx, int* dist)
{
    for (int i = 0; i < na; i++) {
        int best = -1, bd = 1 << 30;
        for (int j = 0; j < nb; j++) {
            int d = 0;
            for (int k = 0; k < 32; k++) d += __builtin_popcount((unsigned)(a[i * 32 + k] ^ b[j * 32 + k]));
            if (d < bd) { bd = d; best = j; }
        }
        idx[i] = best; dist[i] = nb > 0 ? bd : -1;
    }
}


/* (float)cos((double)x) / (float)sin((double)x) as csrc/orb.hip::sincos_0_2pi computes them for the steered-BRIEF rotation (one Cody-Waite reduction by pi/2 + fdlibm's
 * __kernel_sin / __kernel_cos polynomials, restated from fdlibm's published k_sin.c / k_cos.c), checked against THIS host's libm — the arithmetic the descriptor code
 * above uses (ORBextractor.cc:103).  Returns the number of floats u in [0, 6.2833] (bit patterns first, first + stride, ...) where either value differs.
 * TEST INFRASTRUCTURE: the product's copy of this formula lives in the kernel; this one exists so that a CPU test can sweep the whole domain. */
static void vo_sincos_0_2pi(double x, float* sn, float* cs)
{
    const double k = rint(x * 6.36619772367581382433e-01);
    double r = fma(-k, 1.57079632679489655800e+00, x);
    r = fma(-k, 6.12323399573676603587e-17, r);
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06); ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03); ps = fma(z, ps, -1.66666666666666324348e-01);
    const double s = fma(r * z, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07); pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03); pc = fma(z, pc, 4.16666666666666019037e-02);
    const double c = fma(z * z, pc, fma(z, -0.5, 1.0));
    const int q = (int)k & 3;
    const double so = q == 0 ? s : (q == 1 ? c : (q == 2 ? -s : -c)), co = q == 0 ? c : (q == 1 ? -s : (q == 2 ? -c : s));
    *sn = (float)so; *cs = (float)co;
}
long long vo_sincos_0_2pi_mismatches(unsigned first, unsigned stride)
{
    const float lim = 6.2833f; unsigned hi; memcpy(&hi, &lim, 4);
    long long bad = 0;
    if (stride == 0) stride = 1;
    for (unsigned long long u = first; u <= hi; u += stride) {
        const unsigned uu = (unsigned)u; float x; memcpy(&x, &uu, 4);
        float s, c; vo_sincos_0_2pi((double)x, &s, &c);
        if (s != (float)sin((double)x) || c != (float)cos((double)x)) bad++;
    }
    return bad;
}
