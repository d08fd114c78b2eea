/* nets_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE ONLY; see vido_oracle.h).
 *
 * Restates the native GPU ops on the inference path of the three network nodes:
 *   src/thirdparty/flow_net/src/correlation/correlation.py:7-102   kernel_Correlation_rearrange + _updateOutput
 *   src/thirdparty/mask_rcnn/maskrcnn_benchmark/csrc/cuda/ROIAlign_cuda.cu:15-122  RoIAlignForward
 *   src/thirdparty/mask_rcnn/maskrcnn_benchmark/csrc/cuda/nms.cu:13-131           nms_kernel + host sweep
 *       (CUDA semantics: IoU with the "+1" pixel convention, suppress when IoU > thresh; the CPU fallback
 *        csrc/cpu/nms_cpu.cpp:60 uses >= — the node runs the CUDA op)
 *   src/thirdparty/mask_rcnn/maskrcnn_benchmark/modeling/box_coder.py:52-95        BoxCoder.decode
 * Pinned by the reference's own known-answer tests (tests/golden/maskrcnn_kats.npz, extracted from
 * src/tests/test_nms.py and src/tests/test_box_coder.py by tools/gen_golden_maskrcnn_kats.py).
 */
#include "vido_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* out[b, (p+3)*7+(o+3), y, x] = mean_c f1[b,c,y*s,x*s] * f2[b,c,y*s+p*s,x*s+o*s] (zero outside); out H' = ceil(H/s) */
void vo_correlation(const float* f1, const float* f2, int B, int C, int H, int W, int stride, float* out)
{
    const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
    for (int b = 0; b < B; b++)
        for (int ch = 0; ch < 49; ch++) {
            const int o = (ch % 7 - 3) * stride, p = (ch / 7 - 3) * stride;
            for (int y = 0; y < Ho; y++)
                for (int x = 0; x < Wo; x++) {
                    const int y1 = y * stride, x1 = x * stride, y2 = y1 + p, x2 = x1 + o;
                    /* the reference accumulates 32 strided partial sums, then adds them in order */
                    float part[32]; memset(part, 0, sizeof part);
                    if (y2 >= 0 && y2 < H && x2 >= 0 && x2 < W)
                        for (int c = 0; c < C; c++)
                            part[c % 32] += f1[(((size_t)b * C + c) * H + y1) * W + x1] * f2[(((size_t)b * C + c) * H + y2) * W + x2];
                    float tot = 0; for (int i = 0; i < 32; i++) tot += part[i];
                    out[(((size_t)b * 49 + ch) * Ho + y) * Wo + x] = tot / (float)C;
                }
        }
}

static float bilinear(const float* d, int h, int w, float y, float x)
{
    if (y < -1.0 || y > h || x < -1.0 || x > w) return 0;
    if (y <= 0) y = 0;
    if (x <= 0) x = 0;
    int yl = (int)y, xl = (int)x, yh, xh;
    if (yl >= h - 1) { yh = yl = h - 1; y = (float)yl; } else yh = yl + 1;
    if (xl >= w - 1) { xh = xl = w - 1; x = (float)xl; } else xh = xl + 1;
    float ly = y - yl, lx = x - xl, hy = 1.f - ly, hx = 1.f - lx;
    float v1 = d[yl * w + xl], v2 = d[yl * w + xh], v3 = d[yh * w + xl], v4 = d[yh * w + xh];
    float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
    return (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
}

/* rois [n,5] = (batch, x1, y1, x2, y2); out [n, C, PH, PW] */
void vo_roi_align(const float* feat, int B, int C, int H, int W, const float* rois, int n, float scale, int PH, int PW, int sampling, float* out)
{
    (void)B;
    for (int i = 0; i < n; i++) {
        const float* r = rois + 5 * i;
        const int bi = (int)r[0];
        const float sw = r[1] * scale, sh = r[2] * scale, ew = r[3] * scale, eh = r[4] * scale;
        const float rw = fmaxf(ew - sw, 1.f), rh = fmaxf(eh - sh, 1.f);
        const float bh = rh / (float)PH, bw = rw / (float)PW;
        const int gh = sampling > 0 ? sampling : (int)ceilf(rh / PH), gw = sampling > 0 ? sampling : (int)ceilf(rw / PW);
        const float count = (float)(gh * gw);
        for (int c = 0; c < C; c++) {
            const float* d = feat + ((size_t)bi * C + c) * H * W;
            for (int ph = 0; ph < PH; ph++)
                for (int pw = 0; pw < PW; pw++) {
                    float acc = 0;
                    for (int iy = 0; iy < gh; iy++) {
                        const float y = sh + ph * bh + (float)(iy + .5f) * bh / (float)gh;
                        for (int ix = 0; ix < gw; ix++) {
                            const float x = sw + pw * bw + (float)(ix + .5f) * bw / (float)gw;
                            acc += bilinear(d, H, W, y, x);
                        }
                    }
                    out[(((size_t)i * C + c) * PH + ph) * PW + pw] = acc / count;
                }
        }
    }
}

/* maskrcnn_benchmark.layers.nms on CUDA: returns kept ORIGINAL indices in ascending order */
typedef struct { float s; int i; } sc_idx;
static int cmp_desc(const void* a, const void* b) { const sc_idx* x = a; const sc_idx* y = b; if (x->s != y->s) return x->s > y->s ? -1 : 1; return x->i < y->i ? -1 : (x->i > y->i); }
static int cmp_int_asc(const void* a, const void* b) { int x = *(const int*)a, y = *(const int*)b; return x < y ? -1 : x > y; }
int vo_nms(const float* boxes, const float* scores, int n, float thresh, int* keep)
{
    sc_idx* ord = (sc_idx*)malloc(sizeof(sc_idx) * (n + 1)); unsigned char* sup = (unsigned char*)calloc(n + 1, 1);
    for (int i = 0; i < n; i++) { ord[i].s = scores[i]; ord[i].i = i; }
    qsort(ord, n, sizeof(sc_idx), cmp_desc);
    int m = 0;
    for (int a = 0; a < n; a++) {
        if (sup[a]) continue;
        const float* A = boxes + 4 * ord[a].i; keep[m++] = ord[a].i;
        const float areaA = (A[2] - A[0] + 1) * (A[3] - A[1] + 1);
        for (int b = a + 1; b < n; b++) {
            if (sup[b]) continue;
            const float* Bx = boxes + 4 * ord[b].i;
            const float left = fmaxf(A[0], Bx[0]), right = fminf(A[2], Bx[2]), top = fmaxf(A[1], Bx[1]), bottom = fminf(A[3], Bx[3]);
            const float w = fmaxf(right - left + 1, 0.f), h = fmaxf(bottom - top + 1, 0.f), inter = w * h;
            const float areaB = (Bx[2] - Bx[0] + 1) * (Bx[3] - Bx[1] + 1);
            if (inter / (areaA + areaB - inter) > thresh) sup[b] = 1;
        }
    }
    qsort(keep, m, sizeof(int), cmp_int_asc);
    free(ord); free(sup);
    return m;
}

/* BoxCoder.decode, box_coder.py:52-95; deltas [n, 4k], boxes [n,4], clip = log(1000/16) */
void vo_box_decode(const float* deltas, const float* boxes, int n, int k, const float* wts, float* out)
{
    const float clip = (float)log(1000. / 16);
    for (int i = 0; i < n; i++) {
        const float* b = boxes + 4 * i;
        const float w = b[2] - b[0] + 1, h = b[3] - b[1] + 1, cx = b[0] + 0.5f * w, cy = b[1] + 0.5f * h;
        for (int j = 0; j < k; j++) {
            const float* d = deltas + (size_t)i * 4 * k + 4 * j; float* o = out + (size_t)i * 4 * k + 4 * j;
            float dx = d[0] / wts[0], dy = d[1] / wts[1], dw = d[2] / wts[2], dh = d[3] / wts[3];
            if (dw > clip) dw = clip;
            if (dh > clip) dh = clip;
            const float pcx = dx * w + cx, pcy = dy * h + cy, pw = expf(dw) * w, phh = expf(dh) * h;
            o[0] = pcx - 0.5f * pw; o[1] = pcy - 0.5f * phh; o[2] = pcx + 0.5f * pw - 1; o[3] = pcy + 0.5f * phh - 1;
        }
    }
}
