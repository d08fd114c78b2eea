// nets.hip — the native GPU ops of the three network nodes, on gfx950 (the dense conv/GEMM layers of the nets
// themselves run on PyTorch-ROCm; these are the hand-written ops the reference ships as CUDA):
//   vido_correlation   kernel_Correlation_rearrange + kernel_Correlation_updateOutput
//                      (reference src/thirdparty/flow_net/src/correlation/correlation.py:7-102), fused: no padded
//                      NHWC copies, both inputs read once per tile straight from NCHW
//   vido_roi_align     RoIAlignForward (src/thirdparty/mask_rcnn/maskrcnn_benchmark/csrc/cuda/ROIAlign_cuda.cu:15-122)
//   vido_nms           nms_kernel + the HOST sweep of nms.cu:13-131 — here the sweep runs on the device too
//                      (one wave keeps the 64-bit suppression words in its lanes), no mask download
//   vido_box_decode    BoxCoder.decode (maskrcnn_benchmark/modeling/box_coder.py:52-95)
#include "common.hpp"
#include <numeric>

// ---- correlation ---------------------------------------------------------------------------------------
// out[b, (p+3)*7+(o+3), y, x] = (1/C) sum_c f1[b,c,y*s,x*s] * f2[b,c,(y+p)*s,(x+o)*s]   (zero outside the image)
// 16x16 output tile per workgroup; per channel chunk the (16+6)^2 neighbourhood of f2 is staged in LDS and every
// thread keeps its 49 displacement sums in registers.
#define CORR_CH 8
// Small pyramid levels have only a handful of 16x16 tiles: the channel range is then split over `nsplit` workgroups per tile (grid.z =
// batch x nsplit) that write unscaled partial sums, reduced in a fixed order by k_correlation_reduce (deterministic, no float atomics).
__global__ __launch_bounds__(256) void k_correlation(const float* __restrict__ f1, const float* __restrict__ f2, int C, int H, int W, int s,
                                                     int Ho, int Wo, float* __restrict__ out, int nsplit, int cper)
{
    __shared__ float t2[CORR_CH][22][23];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4, b = blockIdx.z / nsplit, split = blockIdx.z - b * nsplit;
    const int c_lo = split * cper, c_hi = min(C, c_lo + cper);
    const int x = blockIdx.x * 16 + tx, y = blockIdx.y * 16 + ty;
    const bool valid = x < Wo && y < Ho;
    float acc[49];
#pragma unroll
    for (int k = 0; k < 49; k++) acc[k] = 0.f;
    const size_t plane = (size_t)H * W;
    for (int c0 = c_lo; c0 < c_hi; c0 += CORR_CH) {
        const int nc = min(CORR_CH, c_hi - c0);
        for (int i = threadIdx.x; i < nc * 22 * 22; i += 256) {
            const int cc = i / (22 * 22), r = (i / 22) % 22, q = i % 22;
            const int yy = (blockIdx.y * 16 + r - 3), xx = (blockIdx.x * 16 + q - 3);
            float v = 0.f;
            if (yy >= 0 && yy < Ho && xx >= 0 && xx < Wo) v = f2[((size_t)b * C + c0 + cc) * plane + (size_t)(yy * s) * W + xx * s];
            t2[cc][r][q] = v;
        }
        float a[CORR_CH];
#pragma unroll
        for (int cc = 0; cc < CORR_CH; cc++) a[cc] = (valid && cc < nc) ? f1[((size_t)b * C + c0 + cc) * plane + (size_t)(y * s) * W + x * s] : 0.f;
        __syncthreads();
#pragma unroll
        for (int cc = 0; cc < CORR_CH; cc++) {
            if (cc < nc) {
#pragma unroll
                for (int p = 0; p < 7; p++)
#pragma unroll
                    for (int o = 0; o < 7; o++) acc[p * 7 + o] += a[cc] * t2[cc][ty + p][tx + o];
            }
        }
        __syncthreads();
    }
    if (valid) {
        float* o = out + (size_t)split * (gridDim.z / nsplit) * 49 * Ho * Wo;      // partial plane of this split (nsplit == 1: the output itself)
#pragma unroll
        for (int k = 0; k < 49; k++) o[(((size_t)b * 49 + k) * Ho + y) * Wo + x] = nsplit == 1 ? acc[k] / (float)C : acc[k];
    }
}
__global__ __launch_bounds__(256) void k_correlation_reduce(const float* __restrict__ part, int nsplit, size_t n, float inv_c_dummy, int C, float* __restrict__ out)
{
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    float sum = 0.f;
    for (int k = 0; k < nsplit; k++) sum += part[(size_t)k * n + i];
    out[i] = sum / (float)C;
}

// The tail of the detector's static head (nets/maskrcnn.py::analyse_image_static; predictor.py:select_top_predictions + run_mask_rcnn.py:create_pixel_masks) in one launch:
// live[i] = score[i] > confidence and i < n_det; order = stable descending sort of (live ? score : -1); labels_out[r] = live[order[r]] ? label[order[r]] : 0; n_live.
// cap <= 1024 slots, one workgroup, rank by counting (a stable sort: ties keep their slot order, like torch.sort(stable=True)).
__global__ __launch_bounds__(1024) void k_det_order(const float* __restrict__ scores, const long long* __restrict__ labels, const int* __restrict__ n_det, float confidence, int cap,
                                                   long long* __restrict__ order, long long* __restrict__ labels_out, long long* __restrict__ n_live)
{
    __shared__ float key[1024];
    __shared__ int cnt;
    const int i = threadIdx.x;
    if (i == 0) cnt = 0;
    const int nd = *n_det;
    bool live = false;
    if (i < cap) { live = scores[i] > confidence && i < nd; key[i] = live ? scores[i] : -1.0f; }
    __syncthreads();
    if (i < cap) {
        const float k = key[i]; int rank = 0;
        for (int j = 0; j < cap; j++) { const float kj = key[j]; rank += (kj > k) || (kj == k && j < i); }
        order[rank] = i; labels_out[rank] = live ? labels[i] : 0;
        if (live) atomicAdd(&cnt, 1);
    }
    __syncthreads();
    if (i == 0) *n_live = cnt;
}

// LevelMapper of the FPN pooler (maskrcnn_benchmark/modeling/poolers.py:11-45) for every box in ONE launch — the same fp32 operations in the same order as the torch
// expression it replaces (floor(4 + log2(sqrt(area) / 224 + 1e-6)) clamped to [k_min, k_max], minus k_min), which was fourteen element-wise launches per pooler call
__global__ void k_roi_levels(const float* __restrict__ boxes, int n, float k_min, float k_max, int* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x1 = boxes[4 * i], y1 = boxes[4 * i + 1], x2 = boxes[4 * i + 2], y2 = boxes[4 * i + 3];
    const float area = __fmul_rn(__fadd_rn(__fsub_rn(x2, x1), 1.f), __fadd_rn(__fsub_rn(y2, y1), 1.f));
    const float t = __fadd_rn(__fdiv_rn(__fsqrt_rn(area), 224.f), 1e-6f);
    float l = floorf(__fadd_rn(4.f, log2f(t)));
    l = fminf(fmaxf(l, k_min), k_max);                         // (a NaN area — never produced by the box decoder — maps to k_min like torch's clamp of NaN would not; boxes are finite)
    out[i] = (int)l - (int)k_min;
}

// one thread per pixel of one detection: 256 (c) multiply-adds down the channels, the detection's weight row in LDS; reads are coalesced over the pixels
__global__ __launch_bounds__(256) void k_mask_logit_select(const float* __restrict__ feat, const float* __restrict__ w, const float* __restrict__ b, const long long* __restrict__ labels,
                                                           float* __restrict__ out, int c, int hw, int classes)
{
    __shared__ float wl[1024];
    const int n = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
    long long lab = labels[n]; lab = lab < 0 ? 0 : (lab >= classes ? classes - 1 : lab);
    for (int i = threadIdx.x; i < c; i += 256) wl[i] = w[(size_t)lab * c + i];
    __syncthreads();
    if (p >= hw) return;
    const float* f = feat + (size_t)n * c * hw + p;
    // (float64 sums: the kernel is bound by its 80 MB of reads, and its result then differs from the library's convolution by the LIBRARY's rounding only)
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    int i = 0;
    for (; i + 4 <= c; i += 4) { a0 += (double)wl[i] * f[(size_t)i * hw]; a1 += (double)wl[i + 1] * f[(size_t)(i + 1) * hw]; a2 += (double)wl[i + 2] * f[(size_t)(i + 2) * hw]; a3 += (double)wl[i + 3] * f[(size_t)(i + 3) * hw]; }
    for (; i < c; i++) a0 += (double)wl[i] * f[(size_t)i * hw];
    const float v = (float)((a0 + a1) + (a2 + a3) + (b ? (double)b[lab] : 0.0));
    out[(size_t)n * hw + p] = 1.f / (1.f + expf(-v));
}

// ---- bias + LeakyReLU epilogue -------------------------------------------------------------------------------
// y[n,c,:,:] = leaky(x[n,c,:,:] + bias[c]) in place: the convolution library runs without its bias so that bias add and activation are
// one pass over the tensor instead of two extra kernels (float4 when the plane size allows it).
__global__ __launch_bounds__(256) void k_bias_act(float* __restrict__ x, const float* __restrict__ bias, int C, size_t hw, size_t total, float slope)
{
    const size_t i = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) * 4;
    if (i >= total) return;
    if ((hw & 3) == 0) {
        const float b = bias[(i / hw) % C];
        float4 v = *(float4*)(x + i);
        v.x += b; v.y += b; v.z += b; v.w += b;
        v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope; v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
        *(float4*)(x + i) = v;
    } else {
        for (size_t k = i; k < min(i + 4, total); k++) { float v = x[k] + bias[(k / hw) % C]; x[k] = v > 0.f ? v : v * slope; }
    }
}

// y = f(x + bias[c]) in place, f = ELU (alpha 1) or the logistic function: the decoder blocks of MonoDepth2 (mono_depth2/src/networks/depth_decoder.py:51-66, layers.py ConvBlock /
// Conv3x3) — the library convolution runs without its bias, bias add and activation are one pass
template <int KIND>
__global__ __launch_bounds__(256) void k_bias_unary(float* __restrict__ x, const float* __restrict__ bias, int C, size_t hw, size_t total)
{
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float v = x[i] + bias[(i / hw) % C];
    x[i] = KIND == 1 ? (v > 0.f ? v : expf(v) - 1.f) : 1.f / (1.f + expf(-v));
}

// torch.cat([interpolate(x, scale_factor=2, mode="nearest"), skip], 1) followed by ReflectionPad2d(1) (depth_decoder.py:58-61 + Conv3x3's pad) as ONE pass:
// x [C1][h][w], skip [C2][2h][2w] (C2 may be 0) -> out [C1 + C2][2h + 2][2w + 2]
__global__ __launch_bounds__(256) void k_upcat_reflect(const float* __restrict__ x, const float* __restrict__ skip, int C1, int C2, int h, int w, float* __restrict__ out)
{
    const int H2 = 2 * h, W2 = 2 * w, OW = W2 + 2, OH = H2 + 2;
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, total = (size_t)(C1 + C2) * OH * OW;
    if (i >= total) return;
    const int X = (int)(i % OW), Y = (int)((i / OW) % OH), c = (int)(i / ((size_t)OW * OH));
    int yy = Y - 1, xx = X - 1;
    yy = yy < 0 ? -yy : (yy >= H2 ? 2 * H2 - 2 - yy : yy); xx = xx < 0 ? -xx : (xx >= W2 ? 2 * W2 - 2 - xx : xx);
    out[i] = c < C1 ? x[((size_t)c * h + (yy >> 1)) * w + (xx >> 1)] : skip[((size_t)(c - C1) * H2 + yy) * W2 + xx];
}

// run_mono_depth.py:137-145: (disp - min) / (max - min + 1e-12) * 65536, clamped to the MONO16 range, as int32; mm = {min, max} on the device
__global__ __launch_bounds__(256) void k_minmax_norm_u16(const float* __restrict__ d, const float* __restrict__ mm, size_t n, int* __restrict__ out)
{
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float lo = mm[0], hi = mm[1];
    float v = (d[i] - lo) / (hi - lo + 1e-12f) * 65536.0f;
    v = fminf(fmaxf(v, 0.f), 65535.f);
    out[i] = (int)v;
}

// x = act(x + bias[c] + res): the FrozenBatchNorm shift (folded into a bias), the shortcut add and the ReLU that close a bottleneck
// (maskrcnn_benchmark/modeling/backbone/resnet.py:352-372), one pass over the tensor instead of four
__global__ __launch_bounds__(256) void k_bias_res_act(float* __restrict__ x, const float* __restrict__ bias, const float* __restrict__ res, int C, size_t hw, size_t total, float slope)
{
    const size_t i = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) * 4;
    if (i >= total) return;
    if ((hw & 3) == 0) {
        const float b = bias[(i / hw) % C];
        float4 v = *(float4*)(x + i); const float4 r = *(const float4*)(res + i);
        v.x += b + r.x; v.y += b + r.y; v.z += b + r.z; v.w += b + r.w;
        v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope; v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
        *(float4*)(x + i) = v;
    } else {
        for (size_t k = i; k < min(i + 4, total); k++) { float v = x[k] + bias[(k / hw) % C] + res[k]; x[k] = v > 0.f ? v : v * slope; }
    }
}

// ---- ROI-Align ---------------------------------------------------------------------------------------------
// MI355X form of ROIAlign_forward (maskrcnn_benchmark/csrc/cuda/ROIAlign_cuda.cu:15-122 defines the VALUES; the mapping below is this build's own):
//  * the feature maps are read CHANNELS-LAST ([B][H][W][C]; k_nchw_to_nhwc makes that copy once per frame, shared by the box and the mask pooler): a bilinear tap of 64
//    channels is ONE coalesced 256-byte wave load, where the channel-planar layout gave every lane its own cache line;
//  * a workgroup owns (roi, 64 channels); its 4 waves split the PH x PW bins, lane = channel.  Sample coordinates, tap indices and the four weights depend on
//    (roi, bin, sample) only, i.e. they are wave-uniform: computed once per sample for all 64 channels (no per-output-element index arithmetic);
//  * per channel the arithmetic order is the reference's — w1*v1 + w2*v2 + w3*v3 + w4*v4 per sample, samples accumulated row-major, one division by the sample count —
//    so the result is bit-identical to the oracle's restatement;
//  * results go to an LDS tile [64 channels][bins] (odd pitch) and leave as ONE contiguous run of out[roi][c0 .. c0+63][PH][PW]: full-line stores although lanes run over channels.
struct RoiLevels { const float* feat[4]; int H[4], W[4]; float scale[4]; };     // channels-last maps
template <int SR /* sampling ratio known at compile time (2: the node's setting; the 4 x 4 taps of a bin are then 16 independent loads in flight), 0: run-time / adaptive */>
__global__ __launch_bounds__(256) void k_roi_align_nhwc(RoiLevels L, int C, const float* __restrict__ rois, int roi_stride /* 5: (batch, x1, y1, x2, y2); 4: (x1, y1, x2, y2) */,
                                                        const int* __restrict__ level /* null: level 0 */, int PH, int PW, int sampling, float* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float rl_tile[];
    const int i = blockIdx.x, c0 = blockIdx.y * 64, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nbin = PH * PW, pitch = nbin | 1;
    const float* r = rois + (size_t)roi_stride * i;
    const int l = level ? level[i] : 0, bi = roi_stride == 5 ? (int)r[0] : 0;
    const float* box = r + (roi_stride == 5 ? 1 : 0);
    const float scale = L.scale[l]; const int H = L.H[l], W = L.W[l];
    const float sw = box[0] * scale, sh = box[1] * scale, ew = box[2] * scale, eh = box[3] * scale;
    const float rw = fmaxf(ew - sw, 1.f), rh = fmaxf(eh - sh, 1.f);
    const float bh = rh / (float)PH, bw = rw / (float)PW;
    const int gh = SR > 0 ? SR : (sampling > 0 ? sampling : (int)ceilf(rh / PH)), gw = SR > 0 ? SR : (sampling > 0 ? sampling : (int)ceilf(rw / PW));
    const float count = (float)(gh * gw);
    const int c = min(c0 + lane, C - 1);                                      // lanes past C repeat the last channel (their column of the tile is never written out)
    const float* d = L.feat[l] + (size_t)bi * H * W * C + c;
    // one sample: tap offsets (in pixels) and weights; outside the map by more than a pixel -> weights 0 (bilinear_interpolate returns 0: adding +0.f leaves the sum unchanged)
    auto taps = [&](float y, float x, int& o1, int& o2, int& o3, int& o4, float& w1, float& w2, float& w3, float& w4) {
        const bool out = y < -1.0 || y > H || x < -1.0 || x > W;
        if (y <= 0) y = 0;
        if (x <= 0) x = 0;
        int yl = out ? 0 : (int)y, xl = out ? 0 : (int)x, yh, xh;
        if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
        if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
        const float ly = y - yl, lx = x - xl, hy = 1.f - ly, hx = 1.f - lx;
        o1 = yl * W + xl; o2 = yl * W + xh; o3 = yh * W + xl; o4 = yh * W + xh;
        w1 = hy * hx; w2 = hy * lx; w3 = ly * hx; w4 = ly * lx;
        return out;
    };
    for (int bin = wave; bin < nbin; bin += 4) {
        const int ph = bin / PW, pw = bin - ph * PW;
        float acc = 0.f;
        if (SR == 2) {
            int o[4][4]; float w[4][4], v[4][4]; bool out[4];
#pragma unroll
            for (int s2 = 0; s2 < 4; s2++) {
                const int iy = s2 >> 1, ix = s2 & 1;
                const float y = sh + ph * bh + (float)(iy + .5f) * bh / (float)gh, x = sw + pw * bw + (float)(ix + .5f) * bw / (float)gw;
                out[s2] = taps(y, x, o[s2][0], o[s2][1], o[s2][2], o[s2][3], w[s2][0], w[s2][1], w[s2][2], w[s2][3]);
            }
#pragma unroll
            for (int s2 = 0; s2 < 4; s2++)
#pragma unroll
                for (int t = 0; t < 4; t++) v[s2][t] = d[(size_t)o[s2][t] * C];               // 16 coalesced 256-byte wave loads in flight
#pragma unroll
            for (int s2 = 0; s2 < 4; s2++) acc += out[s2] ? 0.f : (w[s2][0] * v[s2][0] + w[s2][1] * v[s2][1] + w[s2][2] * v[s2][2] + w[s2][3] * v[s2][3]);
        } else {
            for (int iy = 0; iy < gh; iy++) {
                const float y = sh + ph * bh + (float)(iy + .5f) * bh / (float)gh;
                for (int ix = 0; ix < gw; ix++) {
                    const float x = sw + pw * bw + (float)(ix + .5f) * bw / (float)gw;
                    int o1, o2, o3, o4; float w1, w2, w3, w4;
                    if (taps(y, x, o1, o2, o3, o4, w1, w2, w3, w4)) { acc += 0.f; continue; }
                    const float v1 = d[(size_t)o1 * C], v2 = d[(size_t)o2 * C], v3 = d[(size_t)o3 * C], v4 = d[(size_t)o4 * C];
                    acc += (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
                }
            }
        }
        rl_tile[lane * pitch + bin] = acc / count;
    }
    __syncthreads();
    const int nc = min(64, C - c0), total = nc * nbin;
    float* o = out + ((size_t)i * C + c0) * nbin;
    for (int e = threadIdx.x; e < total; e += 256) { const int cc = e / nbin; o[e] = rl_tile[cc * pitch + (e - cc * nbin)]; }
}
// [B][C][H][W] -> [B][H][W][C] through a 64 x 64 LDS tile: coalesced 256-byte runs on both sides
__global__ __launch_bounds__(256) void k_nchw_to_nhwc(const float* __restrict__ src, int C, int HW, float* __restrict__ dst)
{
    __shared__ float t[64][65];
    const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64, b = blockIdx.z, lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const float* s = src + (size_t)b * C * HW; float* d = dst + (size_t)b * HW * C;
    for (int k = q; k < 64; k += 4) if (c0 + k < C && p0 + lane < HW) t[k][lane] = s[(size_t)(c0 + k) * HW + p0 + lane];
    __syncthreads();
    for (int k = q; k < 64; k += 4) if (p0 + k < HW && c0 + lane < C) d[(size_t)(p0 + k) * C + c0 + lane] = t[lane][k];
}

// ---- NMS (boxes sorted by descending score) ----------------------------------------------------------------------
// Suppression bit matrix (the VALUES are nms.cu:13-67's: bit j of mask[i][cb] <=> IoU(box i, box cb*64+j) > thresh, "+1" box convention, j > i on the diagonal tile).
// MI355X form: one WAVE per 64 x 64 tile, upper-triangle tiles only (the sweeps never read below the diagonal: a box is only suppressed by a better-scored one);
// lane j keeps COLUMN box j and its area in registers for the whole tile, the 64 row boxes come in one at a time through wave-uniform (scalar) loads, and the
// row's 64-bit word is the wave's BALLOT of "IoU > thresh" — one v_cmp per row instead of a 64-step per-lane loop over an LDS copy of the column boxes.
__device__ __forceinline__ void tile_of(int t, int nb, int& rb, int& cb)
{
    // linear index t over the upper triangle (row-major: row rb holds nb - rb tiles) -> (rb, cb)
    int r = (int)((2.0f * nb + 1.0f - sqrtf((2.0f * nb + 1.0f) * (2.0f * nb + 1.0f) - 8.0f * (float)t)) * 0.5f);
    r = max(0, min(r, nb - 1));
    while (r > 0 && r * nb - r * (r - 1) / 2 > t) r--;
    while ((r + 1) * nb - (r + 1) * r / 2 <= t) r++;
    rb = r; cb = r + (t - (r * nb - r * (r - 1) / 2));
}
__device__ __forceinline__ void nms_tile(const float* __restrict__ boxes, const int* __restrict__ group, int n, float thresh, unsigned long long* __restrict__ mask, int col_blocks, int rb, int cb, int lane)
{
    const int col = cb * 64 + lane; const bool cv = col < n;
    const float4 b = cv ? *(const float4*)(boxes + 4 * (size_t)col) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float Sb = (b.z - b.x + 1) * (b.w - b.y + 1);
    const int gb = (group && cv) ? group[col] : 0;
    const int rows = min(n - rb * 64, 64);
    unsigned long long mine = 0;
    for (int i = 0; i < rows; i++) {
        const float* a = boxes + 4 * (size_t)(rb * 64 + i);                      // wave-uniform address: scalar loads
        const float a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
        const int ga = group ? group[rb * 64 + i] : 0;                           // boxes of different groups (classes) never suppress each other
        const float left = fmaxf(a0, b.x), right = fminf(a2, b.z), top = fmaxf(a1, b.y), bottom = fminf(a3, b.w);
        const float w = fmaxf(right - left + 1, 0.f), h = fmaxf(bottom - top + 1, 0.f), inter = w * h;
        const float Sa = (a2 - a0 + 1) * (a3 - a1 + 1);
        const bool hit = cv && ga == gb && inter / (Sa + Sb - inter) > thresh;
        unsigned long long word = __ballot(hit);
        if (rb == cb) word &= i >= 63 ? 0ull : ~((2ull << i) - 1ull);           // diagonal tile: only the boxes after i
        if (lane == i) mine = word;
    }
    if (lane < rows) mask[(size_t)(rb * 64 + lane) * col_blocks + cb] = mine;
}
__global__ __launch_bounds__(256) void k_nms_mask(const float* __restrict__ boxes, const int* __restrict__ group /* may be null */, int n, float thresh,
                                                 unsigned long long* __restrict__ mask, int col_blocks)
{
    const int t = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), nb = col_blocks;      // wave-uniform: the row-box loads below become scalar loads
    if (t >= nb * (nb + 1) / 2) return;
    int rb, cb; tile_of(t, nb, rb, cb);
    nms_tile(boxes, group, n, thresh, mask, col_blocks, rb, cb, threadIdx.x & 63);
}
// the reference does this sweep on the host after a D2H copy; here one wave owns the `remv` words (lane w <-> word w)
__global__ __launch_bounds__(64) void k_nms_sweep(const unsigned long long* __restrict__ mask, int n, int col_blocks, int* __restrict__ keep, int* __restrict__ n_keep)
{
    const int lane = threadIdx.x;
    int cnt = 0;
    // col_blocks suppression words spread round-robin over the lanes (word w lives in lane w%64, slot w/64)
    unsigned long long remv[16];
    const int wpl = (col_blocks + 63) / 64;                      // words per lane (<= 16 -> n <= 65536)
    for (int k = 0; k < 16; k++) remv[k] = 0;
    for (int i = 0; i < n; i++) {
        const int nblock = i >> 6, inblock = i & 63;
        const int owner = nblock & 63, slot = nblock >> 6;
        unsigned long long word = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) if (k == slot) word = remv[k];
        word = __shfl(word, owner, 64);
        if (!(word & (1ULL << inblock))) {
            if (lane == 0) keep[cnt] = i;
            cnt++;
            const unsigned long long* p = mask + (size_t)i * col_blocks;
#pragma unroll
            for (int k = 0; k < 16; k++) if (k < wpl) { const int w = k * 64 + lane; if (w >= nblock && w < col_blocks) remv[k] |= p[w]; }
        }
    }
    if (lane == 0) *n_keep = cnt;
}

// Block-wise sweep, one wave per segment (a segment = one independent NMS problem: an RPN level, or the whole grouped box-head problem).  The sequential
// dependency of greedy NMS is only INSIDE a block of 64 boxes: that part runs on scalar registers over the block's 64x64 diagonal bit matrix (v_readlane);
// the suppression rows of the kept boxes are then OR-ed into the lanes' `remv` words with independent loads.  ~40 us for five 1000-box levels in ONE launch
// instead of five dependent single-wave loops of ~215 us each.  Row stride of `mask` is col_blocks words; segment g covers rows [seg_off[g], seg_off[g]+seg_n[g]),
// whose bit columns are relative to the segment start (the mask kernel is launched per segment with the same layout).
__global__ __launch_bounds__(64) void k_nms_sweep_seg(const unsigned long long* __restrict__ mask, const int* __restrict__ seg_off, const int* __restrict__ seg_n, int col_blocks,
                                                      int* __restrict__ keep, int* __restrict__ n_keep, int keep_stride)
{
    const int g = blockIdx.x, lane = threadIdx.x;
    const int n = seg_n[g];
    const unsigned long long* M = mask + (size_t)seg_off[g] * col_blocks;
    int* kp = keep + (size_t)g * keep_stride;
    const int cb = (n + 63) >> 6;
    unsigned long long remv[16];
#pragma unroll
    for (int k = 0; k < 16; k++) remv[k] = 0;
    int cnt = 0;
    for (int blk = 0; blk < cb; blk++) {
        const int owner = blk & 63, slot = blk >> 6, rows = min(n - blk * 64, 64);
        unsigned long long word = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) if (k == slot) word = remv[k];
        word = __shfl(word, owner, 64);
        const unsigned long long diag = lane < rows ? M[(size_t)(blk * 64 + lane) * col_blocks + blk] : 0ull;
        const uint32_t dlo = (uint32_t)diag, dhi = (uint32_t)(diag >> 32);
        unsigned long long alive = ~word;
        if (rows < 64) alive &= (1ull << rows) - 1ull;
        unsigned long long keepmask = 0;
        while (alive) {                                                          // wave-uniform scalar loop over the surviving boxes of the block
            const int j = __builtin_ctzll(alive);
            keepmask |= 1ull << j;
            const unsigned long long dj = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)dlo, j) | ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)dhi, j) << 32);   // (the builtin returns a signed int: no sign extension into the high word)
            alive &= ~dj; alive &= ~(1ull << j);
        }
        if ((keepmask >> lane) & 1ull) kp[cnt + __popcll(keepmask & ((1ull << lane) - 1ull))] = blk * 64 + lane;
        cnt += __popcll(keepmask);
        // OR the kept rows into the words of the later blocks.  Round 2 walked the kept rows one after the other (a dependent chain of up to 64 memory round trips per block);
        // here lane j loads ITS row's word of column block w (all 64 loads of a column block in flight together, the column blocks unrolled four at a time) and the wave
        // OR-reduces them: one round trip per four column blocks.
        const bool mine = (keepmask >> lane) & 1ull;
        const unsigned long long* prow = M + (size_t)(blk * 64 + lane) * col_blocks;
        for (int w0 = blk + 1; w0 < cb; w0 += 4) {
            unsigned long long v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = (mine && w0 + u < cb) ? prow[w0 + u] : 0ull;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                uint32_t lo = (uint32_t)v[u], hi = (uint32_t)(v[u] >> 32);
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) { lo |= (uint32_t)__shfl_xor((int)lo, o, 64); hi |= (uint32_t)__shfl_xor((int)hi, o, 64); }
                const int w = w0 + u;
                if (w < cb && lane == (w & 63)) {
                    const unsigned long long r = (unsigned long long)lo | ((unsigned long long)hi << 32);
#pragma unroll
                    for (int k = 0; k < 16; k++) if (k == (w >> 6)) remv[k] |= r;
                }
            }
        }
    }
    for (int i = cnt + lane; i < keep_stride; i += 64) kp[i] = -1;
    if (lane == 0) n_keep[g] = cnt;
}
// the same for independent segments (RPN levels, box-head classes): grid (tile quads, segment); a segment's rows start at seg_off[g], its bit columns are relative to that
__global__ __launch_bounds__(256) void k_nms_mask_seg(const float* __restrict__ boxes, const int* __restrict__ group, const int* __restrict__ seg_off, const int* __restrict__ seg_n,
                                                     float thresh, unsigned long long* __restrict__ mask, int col_blocks)
{
    const int g = blockIdx.y, n = seg_n[g], off = seg_off[g];
    const int nb = (n + 63) >> 6, t = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (t >= nb * (nb + 1) / 2) return;
    int rb, cb; tile_of(t, nb, rb, cb);
    nms_tile(boxes + 4 * (size_t)off, group ? group + off : nullptr, n, thresh, mask + (size_t)off * col_blocks, col_blocks, rb, cb, threadIdx.x & 63);
}

// ---- Masker + label image in one pass (mask_head/inference.py:87-160 paste_mask_in_image per detection on the host, then run_mask_rcnn.py:112-118
// blank_mask += mask * class_index per detection): every output pixel walks the kept detections in order, samples the 1-px padded MxM mask probability with
// the bilinear rule of F.interpolate(align_corners=False) at the detection's truncated box, thresholds, and accumulates class indices in u8 (wraps on overlap
// like the reference's numpy loop).
struct PasteDet { int x0, y0, w, h, label; float rw, rh; };
__global__ __launch_bounds__(256) void k_paste_prepare(const float* __restrict__ boxes, const int64_t* __restrict__ labels, int n, int M, int padding, PasteDet* __restrict__ det)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float scale = (float)((double)(M + 2 * padding) / (double)M);
    const float* b = boxes + 4 * (size_t)i;
    const float wh = (b[2] - b[0]) * 0.5f * scale, hh = (b[3] - b[1]) * 0.5f * scale;
    const float xc = (b[2] + b[0]) * 0.5f, yc = (b[3] + b[1]) * 0.5f;
    const int x0 = (int)(xc - wh), y0 = (int)(yc - hh), x1 = (int)(xc + wh), y1 = (int)(yc + hh);      // .to(torch.int32): truncation toward zero
    PasteDet d; d.x0 = x0; d.y0 = y0; d.w = max(x1 - x0 + 1, 1); d.h = max(y1 - y0 + 1, 1); d.label = (int)labels[i];
    const int Mp = M + 2 * padding;
    d.rw = (float)Mp / (float)d.w; d.rh = (float)Mp / (float)d.h;                                      // area_pixel_compute_scale, align_corners = false
    det[i] = d;
}
__global__ __launch_bounds__(256) void k_paste_label(const float* __restrict__ masks /*[n,1,M,M]*/, const PasteDet* __restrict__ det, int n, int M, int padding, float thresh,
                                                     int H, int W, uint8_t* __restrict__ out)
{
    __shared__ PasteDet sd[128];
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    const int Mp = M + 2 * padding;
    unsigned acc = 0;
    for (int d0 = 0; d0 < n; d0 += 128) {
        __syncthreads();
        if ((int)threadIdx.x < min(128, n - d0)) sd[threadIdx.x] = det[d0 + threadIdx.x];
        __syncthreads();
        if (x < W && y < H) {
            for (int k = 0; k < min(128, n - d0); k++) {
                const PasteDet d = sd[k];
                const int u = x - d.x0, v = y - d.y0;
                if (u < 0 || v < 0 || u >= d.w || v >= d.h) continue;
                float sy = d.rh * ((float)v + 0.5f) - 0.5f; if (sy < 0.f) sy = 0.f;
                float sx = d.rw * ((float)u + 0.5f) - 0.5f; if (sx < 0.f) sx = 0.f;
                const int iy = (int)sy, ix = (int)sx; const int iyp = iy < Mp - 1 ? 1 : 0, ixp = ix < Mp - 1 ? 1 : 0;
                const float ly = sy - (float)iy, lx = sx - (float)ix, hy = 1.f - ly, hx = 1.f - lx;
                const float* m = masks + (size_t)(d0 + k) * M * M;
                auto at = [&](int yy, int xx) -> float { yy -= padding; xx -= padding; return (yy < 0 || xx < 0 || yy >= M || xx >= M) ? 0.f : m[yy * M + xx]; };
                const float val = hy * (hx * at(iy, ix) + lx * at(iy, ix + ixp)) + ly * (hx * at(iy + iyp, ix) + lx * at(iy + iyp, ix + ixp));
                if (val > thresh) acc += (unsigned)d.label;
            }
        }
    }
    if (x < W && y < H) out[(size_t)y * W + x] = (uint8_t)acc;
}

// ---- box decode ------------------------------------------------------------------------------------------------
__global__ void k_box_decode(const float* __restrict__ deltas, const float* __restrict__ boxes, int n, int k, float wx, float wy, float ww, float wh, float* __restrict__ out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * k) return;
    const int i = t / k;
    const float clip = (float)log(1000. / 16);
    const float* b = boxes + 4 * (size_t)i; const float* d = deltas + 4 * (size_t)t; float* o = out + 4 * (size_t)t;
    const float w = b[2] - b[0] + 1, h = b[3] - b[1] + 1, cx = b[0] + 0.5f * w, cy = b[1] + 0.5f * h;
    float dx = d[0] / wx, dy = d[1] / wy, dw = d[2] / ww, dh = d[3] / wh;
    if (dw > clip) dw = clip;
    if (dh > clip) dh = clip;
    const float pcx = dx * w + cx, pcy = dy * h + cy, pw = expf(dw) * w, phh = expf(dh) * h;
    o[0] = pcx - 0.5f * pw; o[1] = pcy - 0.5f * phh; o[2] = pcx + 0.5f * pw - 1; o[3] = pcy + 0.5f * phh - 1;
}

// ---- host ------------------------------------------------------------------------------------------------------
struct NetState { char* d = nullptr; char* h = nullptr; size_t cap = 0; };     // device scratch + pinned mirror
void net_state_destroy(vido_ctx* ctx)
{
    NetState* S = ctx->net; if (!S) return;
    hipFree(S->d); hipHostFree(S->h); delete S; ctx->net = nullptr;
}
static int net_scratch(vido_ctx* ctx, size_t bytes, NetState** out)
{
    if (!ctx->net) ctx->net = new NetState();
    NetState* S = ctx->net;
    if (bytes > S->cap) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (S->d) { hipFree(S->d); hipHostFree(S->h); S->d = nullptr; S->h = nullptr; }
        S->cap = bytes + bytes / 4 + 4096;
        HIP_TRY(ctx, hipMalloc((void**)&S->d, S->cap)); HIP_TRY(ctx, hipHostMalloc((void**)&S->h, S->cap));
    }
    *out = S;
    return VIDO_OK;
}
static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

// ---- node pre-processing and LiteFlowNet's warp as single passes ------------------------------------------------------------------------------
// u8 HxWx3 interleaved (BGR) -> f32 [3][OH][OW] planar in REVERSED channel order (RGB), resized like torch's interpolate(mode="area") (= adaptive average pooling,
// aten/src/ATen/native/cuda/AdaptiveAveragePooling.cu: window [floor(o*I/O), ceil((o+1)*I/O)) evaluated in float, row-major float sum, sum / kH / kW), then divided by
// `div` when div != 1.  Replaces flip + permute + float + interpolate(area) + div: five launches, one of them 0.37 ms (the pooling kernel walks one output row per thread).
__global__ __launch_bounds__(256) void k_area_feed(const uint8_t* __restrict__ src, int H, int W, float* __restrict__ dst, int OH, int OW, float div)
{
    const int ow = blockIdx.x * 64 + (threadIdx.x & 63), oh = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ow >= OW || oh >= OH) return;
    const int h0 = (int)floorf((float)(oh * H) / OH), h1 = (int)ceilf((float)((oh + 1) * H) / OH);
    const int w0 = (int)floorf((float)(ow * W) / OW), w1 = (int)ceilf((float)((ow + 1) * W) / OW);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int ih = h0; ih < h1; ih++) {
        const uint8_t* row = src + ((size_t)ih * W + w0) * 3;
        for (int iw = w0; iw < w1; iw++, row += 3) { s0 += (float)row[2]; s1 += (float)row[1]; s2 += (float)row[0]; }
    }
    const float kh = (float)(h1 - h0), kw = (float)(w1 - w0);
    float o0 = s0 / kh / kw, o1 = s1 / kh / kw, o2 = s2 / kh / kw;
    if (div != 1.0f) { const float inv = 1.0f / div; o0 *= inv; o1 *= inv; o2 *= inv; }      // torch's div-by-scalar on the device multiplies by the reciprocal: same values as the torch form
    const size_t plane = (size_t)OH * OW, o = (size_t)oh * OW + ow;
    dst[o] = o0; dst[plane + o] = o1; dst[2 * plane + o] = o2;
}
// flow_net/src/layers.py:25-37 (Backward): out[b,c,h,w] = bilinear sample of x[b,c] at the grid point g = (-1 + (2w+1)/W + u / ((W-1)/2), -1 + (2h+1)/H + v / ((H-1)/2)),
// grid_sample(mode bilinear, padding zeros, align_corners False): pixel coordinate ((g + 1) * size - 1) / 2.  One thread per (b, h, w), the channels in a loop
// (the four taps and weights are shared by all channels).  Replaces 2 linspace + expand + 2 div + 2 add + cat + permute + grid_sample.
#define BW_CG 8
__global__ __launch_bounds__(256) void k_backwarp(const float* __restrict__ x, const float* __restrict__ flow, int B, int C, int H, int W, float* __restrict__ out)
{
    // blockIdx.z = (image, group of BW_CG channels): the maps are small (<= 120 x 160 at 640 x 480), one thread per pixel alone leaves most CUs idle
    const int w = blockIdx.x * 64 + (threadIdx.x & 63), h = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int ncg = (C + BW_CG - 1) / BW_CG, b = blockIdx.z / ncg, c0 = (blockIdx.z - b * ncg) * BW_CG, c1 = min(C, c0 + BW_CG);
    if (w >= W || h >= H) return;
    const size_t hw = (size_t)H * W, o = (size_t)h * W + w;
    const float u = flow[(size_t)b * 2 * hw + o], v = flow[(size_t)b * 2 * hw + hw + o];
    const float stepx = ((1.0f - 1.0f / W) - (-1.0f + 1.0f / W)) / (float)(W - 1), stepy = ((1.0f - 1.0f / H) - (-1.0f + 1.0f / H)) / (float)(H - 1);
    const float gx = ((-1.0f + 1.0f / W) + stepx * (float)w) + u / (((float)W - 1.0f) / 2.0f);
    const float gy = ((-1.0f + 1.0f / H) + stepy * (float)h) + v / (((float)H - 1.0f) / 2.0f);
    const float ix = ((gx + 1.0f) * (float)W - 1.0f) / 2.0f, iy = ((gy + 1.0f) * (float)H - 1.0f) / 2.0f;
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float ax = ix - fx, ay = iy - fy;
    const float wnw = (1.0f - ax) * (1.0f - ay), wne = ax * (1.0f - ay), wsw = (1.0f - ax) * ay, wse = ax * ay;
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    const float* xb = x + (size_t)b * C * hw; float* ob = out + (size_t)b * C * hw + o;
    const size_t inw = (size_t)(vy0 ? y0 : 0) * W + (vx0 ? x0 : 0), ine = (size_t)(vy0 ? y0 : 0) * W + (vx1 ? x1 : 0);
    const size_t isw = (size_t)(vy1 ? y1 : 0) * W + (vx0 ? x0 : 0), ise = (size_t)(vy1 ? y1 : 0) * W + (vx1 ? x1 : 0);
    const float mnw = (vy0 && vx0) ? wnw : 0.f, mne = (vy0 && vx1) ? wne : 0.f, msw = (vy1 && vx0) ? wsw : 0.f, mse = (vy1 && vx1) ? wse : 0.f;
#pragma unroll 4
    for (int c = c0; c < c1; c++) {
        const float* xc = xb + (size_t)c * hw;
        // grid_sampler_2d_kernel accumulates nw, ne, sw, se in this order, skipping taps outside the image
        float acc = 0.f;
        acc += xc[inw] * mnw; acc += xc[ine] * mne; acc += xc[isw] * msw; acc += xc[ise] * mse;
        ob[(size_t)c * hw] = acc;
    }
}

// ---- LiteFlowNet's regularisation stage as two passes (flow_net/src/layers.py:213-262, Regularization.forward) -----------------------------------------------------
// front: diff = sqrt(sum_c (first - Backward(second, flow * scale))^2), centred = flow - mean(flow) -> channels 0..2 of the stage's input tensor (the netFeat features
// are copied behind them); replaces mul + Backward + sub + pow + sum + sqrt + sub + cat (the spatial mean comes in as a device scalar pair).
__global__ __launch_bounds__(256) void k_lfn_reg_front(const float* __restrict__ im1, const float* __restrict__ im2, const float* __restrict__ flow, const float* __restrict__ mean /*[B,2]*/,
                                                       float scale, int B, int C, int H, int W, float* __restrict__ out, int out_ch)
{
    const int w = blockIdx.x * 64 + (threadIdx.x & 63), h = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
    if (w >= W || h >= H) return;
    const size_t hw = (size_t)H * W, o = (size_t)h * W + w;
    const float fu = flow[(size_t)b * 2 * hw + o], fv = flow[(size_t)b * 2 * hw + hw + o];
    const float u = fu * scale, v = fv * scale;
    // the sampling grid of k_backwarp (layers.py:25-37), same expression order
    const float stepx = ((1.0f - 1.0f / W) - (-1.0f + 1.0f / W)) / (float)(W - 1), stepy = ((1.0f - 1.0f / H) - (-1.0f + 1.0f / H)) / (float)(H - 1);
    const float gx = ((-1.0f + 1.0f / W) + stepx * (float)w) + u / (((float)W - 1.0f) / 2.0f);
    const float gy = ((-1.0f + 1.0f / H) + stepy * (float)h) + v / (((float)H - 1.0f) / 2.0f);
    const float ix = ((gx + 1.0f) * (float)W - 1.0f) / 2.0f, iy = ((gy + 1.0f) * (float)H - 1.0f) / 2.0f;
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float ax = ix - fx, ay = iy - fy;
    const float wnw = (1.0f - ax) * (1.0f - ay), wne = ax * (1.0f - ay), wsw = (1.0f - ax) * ay, wse = ax * ay;
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    const size_t inw = (size_t)(vy0 ? y0 : 0) * W + (vx0 ? x0 : 0), ine = (size_t)(vy0 ? y0 : 0) * W + (vx1 ? x1 : 0);
    const size_t isw = (size_t)(vy1 ? y1 : 0) * W + (vx0 ? x0 : 0), ise = (size_t)(vy1 ? y1 : 0) * W + (vx1 ? x1 : 0);
    const float mnw = (vy0 && vx0) ? wnw : 0.f, mne = (vy0 && vx1) ? wne : 0.f, msw = (vy1 && vx0) ? wsw : 0.f, mse = (vy1 && vx1) ? wse : 0.f;
    float acc2 = 0.f;
    for (int c = 0; c < C; c++) {
        const float* xc = im2 + ((size_t)b * C + c) * hw;
        float wv = 0.f;
        wv += xc[inw] * mnw; wv += xc[ine] * mne; wv += xc[isw] * msw; wv += xc[ise] * mse;
        const float dlt = im1[((size_t)b * C + c) * hw + o] - wv;
        acc2 += dlt * dlt;
    }
    float* ob = out + (size_t)b * out_ch * hw + o;
    ob[0] = sqrtf(acc2); ob[hw] = fu - mean[2 * b]; ob[2 * hw] = fv - mean[2 * b + 1];
}
// tail: dist = netDist(...) [B, K*K, H, W] -> d = exp(-dist^2 - max_c(-dist^2)), div = 1 / sum_c d, out_x = (netScaleX(d * unfold(flow_x)) ) * div, out_y likewise, where
// unfold is the K x K neighbourhood (zero padded) and netScaleX/Y are 1x1 convolutions K*K -> 1 (weights wx / wy, biases bx / by).  One thread per pixel; replaces
// pow + neg + max + sub + exp + sum + reciprocal + 2 unfold + 2 mul + 2 (im2col + GEMM + bias) + 2 mul + cat.
template <int K>
__global__ __launch_bounds__(256) void k_lfn_reg_tail(const float* __restrict__ dist, const float* __restrict__ flow, const float* __restrict__ wx, const float* __restrict__ bx,
                                                      const float* __restrict__ wy, const float* __restrict__ by, int B, int H, int W, float* __restrict__ out)
{
    constexpr int ND = K * K, PAD = (K - 1) / 2;
    const int w = blockIdx.x * 64 + (threadIdx.x & 63), h = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
    if (w >= W || h >= H) return;
    const size_t hw = (size_t)H * W, o = (size_t)h * W + w;
    const float* dp = dist + (size_t)b * ND * hw + o;
    float e[ND]; float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < ND; c++) { const float v = dp[(size_t)c * hw]; e[c] = -(v * v); m = fmaxf(m, e[c]); }
    const float* fxp = flow + (size_t)b * 2 * hw; const float* fyp = fxp + hw;
    float s = 0.f, ax = 0.f, ay = 0.f;
#pragma unroll
    for (int c = 0; c < ND; c++) {
        const float p = expf(e[c] - m);
        s += p;
        const int yy = h + c / K - PAD, xx = w + c % K - PAD;
        const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
        const size_t q = in ? (size_t)yy * W + xx : 0;
        const float ux = in ? fxp[q] : 0.f, uy = in ? fyp[q] : 0.f;
        ax += wx[c] * (p * ux); ay += wy[c] * (p * uy);
    }
    const float div = 1.0f / s;
    out[(size_t)b * 2 * hw + o] = (ax + bx[0]) * div;
    out[(size_t)b * 2 * hw + hw + o] = (ay + by[0]) * div;
}

// Depthwise ConvTranspose2d(C, C, kernel 4, stride 2, padding 1, groups = C, no bias): LiteFlowNet's netUpflow (2 channels) and netUpcorr (49 channels)
// (flow_net/src/layers.py:105-108).  out[c][oy][ox] = sum over the (at most) 2 x 2 input pixels iy = (oy + 1 - ky) / 2, ix = (ox + 1 - kx) / 2 with ky = (oy + 1) % 2 + {0, 2},
// kx likewise, of act(in[c][iy][ix]) * w[c][ky][kx]; act = LeakyReLU(slope) of the INPUT (slope 1: none) — the matching stage applies it to the cost volume just before
// netUpcorr, so that pass is folded in.  One thread per output pixel, memory-bound (MIOpen runs this as a grouped backward-data convolution: 70 us for 49 x 120 x 160).
__global__ __launch_bounds__(256) void k_deconv4s2_dw(const float* __restrict__ in, const float* __restrict__ wgt /*[C][4][4]*/, int C, int H, int W, float slope, float* __restrict__ out)
{
    const int OW = 2 * W, OH = 2 * H;
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63), oy = blockIdx.y * 4 + (threadIdx.x >> 6), bc = blockIdx.z, c = bc % C;
    if (ox >= OW || oy >= OH) return;
    const float* ip = in + (size_t)bc * H * W; const float* wp = wgt + 16 * c;
    const int ky0 = (oy + 1) & 1, kx0 = (ox + 1) & 1, iy0 = (oy + 1 - ky0) >> 1, ix0 = (ox + 1 - kx0) >> 1;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 2; a++) {
        const int iy = iy0 - a, ky = ky0 + 2 * a;
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int ix = ix0 - b, kx = kx0 + 2 * b;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) { float v = ip[(size_t)iy * W + ix]; v = v > 0.f ? v : v * slope; acc += v * wp[ky * 4 + kx]; }
        }
    }
    out[((size_t)bc * OH + oy) * OW + ox] = acc;
}

// dynamic-LDS limit of k_roi_align_nhwc: raised only when a call needs more than any earlier one (the attribute call is kept out of hipGraph captures, whose
// replays run with the limit the warm-up calls have set)
static int roi_lds_limit(vido_ctx* ctx, size_t lds)
{
    static size_t have = 48 * 1024;
    if (lds <= have) return VIDO_OK;
    HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_roi_align_nhwc<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_roi_align_nhwc<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    have = lds;
    return VIDO_OK;
}

extern "C" {

int vido_correlation(vido_ctx* ctx, const float* first, const float* second, int B, int C, int H, int W, int stride, float* out, int on_device)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!first || !second || !out || B < 1 || C < 1 || H < 1 || W < 1 || (stride != 1 && stride != 2)) return vido_set_error(ctx, VIDO_E_INVALID, "correlation: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (on_device && ctx->has_ext_stream) ? ctx->ext_stream : ctx->stream;
    const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
    const size_t nin = (size_t)B * C * H * W * 4, nout = (size_t)B * 49 * Ho * Wo * 4;
    const float *d1 = first, *d2 = second; float* dout = out;
    const int tiles = ((Wo + 15) / 16) * ((Ho + 15) / 16) * B, nchunk = (C + CORR_CH - 1) / CORR_CH;
    int nsplit = std::max(1, std::min(nchunk, 512 / std::max(tiles, 1)));
    const int cper = ((nchunk + nsplit - 1) / nsplit) * CORR_CH; nsplit = (C + cper - 1) / cper;
    const size_t nel = (size_t)B * 49 * Ho * Wo, io_bytes = on_device ? 0 : 2 * al256(nin) + al256(nout), pbytes = nsplit > 1 ? al256(nel * 4 * nsplit) : 0;
    NetState* S = nullptr;
    if (io_bytes + pbytes) { int rc = net_scratch(ctx, io_bytes + pbytes, &S); if (rc) return rc; }
    if (!on_device) {
        memcpy(S->h, first, nin); memcpy(S->h + al256(nin), second, nin);
        HIP_TRY(ctx, hipMemcpyAsync(S->d, S->h, 2 * al256(nin), hipMemcpyHostToDevice, st));
        d1 = (float*)S->d; d2 = (float*)(S->d + al256(nin)); dout = (float*)(S->d + 2 * al256(nin));
    }
    if (nsplit == 1) hipLaunchKernelGGL(k_correlation, dim3((Wo + 15) / 16, (Ho + 15) / 16, B), dim3(256), 0, st, d1, d2, C, H, W, stride, Ho, Wo, dout, 1, C);
    else {
        float* part = (float*)(S->d + io_bytes);
        hipLaunchKernelGGL(k_correlation, dim3((Wo + 15) / 16, (Ho + 15) / 16, B * nsplit), dim3(256), 0, st, d1, d2, C, H, W, stride, Ho, Wo, part, nsplit, cper);
        hipLaunchKernelGGL(k_correlation_reduce, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, st, (const float*)part, nsplit, nel, 0.f, C, dout);
    }
    HIP_TRY(ctx, hipGetLastError());
    if (!on_device) {
        HIP_TRY(ctx, hipStreamSynchronize(st));
        HIP_TRY(ctx, hipMemcpyAsync(S->h, dout, nout, hipMemcpyDeviceToHost, st)); HIP_TRY(ctx, hipStreamSynchronize(st));
        memcpy(out, S->h, nout);
    }
    return VIDO_OK;
}

/* The static head's tail in one launch (see k_det_order): scores f32 [cap], labels int64 [cap], n_det int32 [1] -> order int64 [cap] (slots in descending score order, the
 * slots that fail the confidence test or lie past n_det behind them in slot order), labels_out int64 [cap] (0 for those), n_live int64 [1].  cap <= 1024. */
int vido_det_order(vido_ctx* ctx, const float* scores, const long long* labels, const int* n_det, float confidence, int cap, long long* order, long long* labels_out, long long* n_live)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!scores || !labels || !n_det || !order || !labels_out || !n_live || cap < 1 || cap > 1024) return vido_set_error(ctx, VIDO_E_INVALID, "det_order: bad arguments (cap 1 .. 1024)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    hipLaunchKernelGGL(k_det_order, dim3(1), dim3(1024), 0, st, scores, labels, n_det, confidence, cap, order, labels_out, n_live);
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

/* out[i] = the FPN level (0 .. k_max - k_min) of box i: LevelMapper of modeling/poolers.py:11-45 in one launch; boxes [n][4] f32 (x1, y1, x2, y2), out int32 [n]. */
int vido_roi_levels(vido_ctx* ctx, const float* boxes, int n, float k_min, float k_max, int* out)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!boxes || !out || n < 0 || !(k_min <= k_max)) return vido_set_error(ctx, VIDO_E_INVALID, "roi_levels: bad arguments");
    if (n == 0) return VIDO_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    hipLaunchKernelGGL(k_roi_levels, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, boxes, n, k_min, k_max, out);
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

/* The mask head's tail for the ONE class channel a detection needs (maskrcnn_benchmark/modeling/roi_heads/mask_head/roi_mask_predictors.py:27-31 + inference.py:29-47:
 * mask_fcn_logits, sigmoid, the detection's label channel): out[n][p] = sigmoid(sum_c w[label[n]][c] feat[n][c][p] + b[label[n]]) — instead of all num_classes logit maps,
 * a bias pass, a sigmoid pass and a gather (81 x the multiply-adds, four launches).  feat [n][c][hw] f32, w [classes][c], labels int64 [n] (clamped to the class range), out [n][hw]. */
int vido_mask_logit_select(vido_ctx* ctx, const float* feat, const float* w, const float* b, const long long* labels, float* out, int n, int c, int hw, int classes)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!feat || !w || !labels || !out || n < 1 || c < 1 || c > 1024 || hw < 1 || classes < 1) return vido_set_error(ctx, VIDO_E_INVALID, "mask_logit_select: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    hipLaunchKernelGGL(k_mask_logit_select, dim3((unsigned)((hw + 255) / 256), (unsigned)n), dim3(256), 0, st, feat, w, b, labels, out, c, hw, classes);
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

/* In-place conv epilogue on a DEVICE tensor x[N,C,H,W] (f32, contiguous): x = leaky_relu(x + bias[c], slope); slope = 1 is a plain bias add. */
int vido_bias_act(vido_ctx* ctx, float* x, const float* bias, int N, int C, int H, int W, float slope)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!x || !bias || N < 1 || C < 1 || H < 1 || W < 1) return vido_set_error(ctx, VIDO_E_INVALID, "bias_act: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    const size_t hw = (size_t)H * W, total = (size_t)N * C * hw;
    hipLaunchKernelGGL(k_bias_act, dim3((unsigned)((total / 4 + 256) / 256)), dim3(256), 0, st, x, bias, C, hw, total, slope);
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

/* x = f(x + bias[c]) in place on a DEVICE tensor x[N,C,H,W]; kind 1: ELU (alpha = 1), 2: logistic function. */
int vido_bias_unary(vido_ctx* ctx, float* x, const float* bias, int N, int C, int H, int W, int kind)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!x || !bias || N < 1 || C < 1 || H < 1 || W < 1 || (kind != 1 && kind != 2)) return vido_set_error(ctx, VIDO_E_INVALID, "bias_unary: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    const size_t hw = (size_t)H * W, total = (size_t)N * C * hw;
    if (kind == 1) hipLaunchKernelGGL(k_bias_unary<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, bias, C, hw, total);
    else hipLaunchKernelGGL(k_bias_unary<2>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, bias, C, hw, total);
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

/* ReflectionPad2d(1)(cat([nearest-upsample x2 of x [C1][h][w], skip [C2][2h][2w]], channel axis)) -> out [C1 + C2][2h + 2][2w + 2]; DEVICE tensors, skip may be NULL (C2 = 0). */
int vido_upcat_reflect(vido_ctx* ctx, const float* x, const float* skip, int C1, int C2, int h, int w, float* out)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!x || !out || C1 < 1 || C2 < 0 || (C2 && !skip) || h < 1 || w < 1) return vido_set_error(ctx, VIDO_E_INVALID, "upcat_reflect: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    const size_t total = (size_t)(C1 + C2) * (2 * h + 2) * (2 * w + 2);
    hipLaunchKernelGGL(k_upcat_reflect, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, skip, C1, C2, h, w, out);
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

/* out[i] = (int) clamp((d[i] - mm[0]) / (mm[1] - mm[0] + 1e-12) * 65536, 0, 65535): the MONO16 normalisation of the depth node; d [n] f32, mm [2] f32 = {min, max}, out [n] i32, DEVICE. */
int vido_minmax_norm_u16(vido_ctx* ctx, const float* d, const float* mm, int64_t n, int32_t* out)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!d || !mm || !out || n < 1) return vido_set_error(ctx, VIDO_E_INVALID, "minmax_norm_u16: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    hipLaunchKernelGGL(k_minmax_norm_u16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d, mm, (size_t)n, (int*)out);
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

int vido_area_feed(vido_ctx* ctx, const uint8_t* bgr, int H, int W, float* out, int OH, int OW, float div)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!bgr || !out || H < 1 || W < 1 || OH < 1 || OW < 1 || !(div > 0)) return vido_set_error(ctx, VIDO_E_INVALID, "area_feed: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    hipLaunchKernelGGL(k_area_feed, dim3((OW + 63) / 64, (OH + 3) / 4), dim3(256), 0, st, bgr, H, W, out, OH, OW, div);
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

int vido_backwarp(vido_ctx* ctx, const float* x, const float* flow, int B, int C, int H, int W, float* out)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!x || !flow || !out || B < 1 || C < 1 || H < 2 || W < 2 || (long long)B * ((C + 7) / 8) > 65535) return vido_set_error(ctx, VIDO_E_INVALID, "backwarp: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    hipLaunchKernelGGL(k_backwarp, dim3((W + 63) / 64, (H + 3) / 4, B * ((C + BW_CG - 1) / BW_CG)), dim3(256), 0, st, x, flow, B, C, H, W, out);
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

int vido_deconv4s2_depthwise(vido_ctx* ctx, const float* x, const float* weight, int B, int C, int H, int W, float input_slope, float* out)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!x || !weight || !out || B < 1 || C < 1 || H < 1 || W < 1 || (long long)B * C > 65535) return vido_set_error(ctx, VIDO_E_INVALID, "deconv4s2_depthwise: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    hipLaunchKernelGGL(k_deconv4s2_dw, dim3((2 * W + 63) / 64, (2 * H + 3) / 4, B * C), dim3(256), 0, st, x, weight, C, H, W, input_slope, out);
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

int vido_lfn_reg_front(vido_ctx* ctx, const float* im1, const float* im2, const float* flow, const float* mean, float scale, int B, int C, int H, int W, float* out, int out_channels)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!im1 || !im2 || !flow || !mean || !out || B < 1 || B > 65535 || C < 1 || H < 2 || W < 2 || out_channels < 3) return vido_set_error(ctx, VIDO_E_INVALID, "lfn_reg_front: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    hipLaunchKernelGGL(k_lfn_reg_front, dim3((W + 63) / 64, (H + 3) / 4, B), dim3(256), 0, st, im1, im2, flow, mean, scale, B, C, H, W, out, out_channels);
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

int vido_lfn_reg_tail(vido_ctx* ctx, const float* dist, const float* flow, const float* wx, const float* bx, const float* wy, const float* by, int B, int K, int H, int W, float* out)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!dist || !flow || !wx || !bx || !wy || !by || !out || B < 1 || B > 65535 || H < 1 || W < 1 || (K != 3 && K != 5 && K != 7)) return vido_set_error(ctx, VIDO_E_INVALID, "lfn_reg_tail: bad arguments (K must be 3, 5 or 7)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    const dim3 grid((W + 63) / 64, (H + 3) / 4, B);
    if (K == 3) hipLaunchKernelGGL(k_lfn_reg_tail<3>, grid, dim3(256), 0, st, dist, flow, wx, bx, wy, by, B, H, W, out);
    else if (K == 5) hipLaunchKernelGGL(k_lfn_reg_tail<5>, grid, dim3(256), 0, st, dist, flow, wx, bx, wy, by, B, H, W, out);
    else hipLaunchKernelGGL(k_lfn_reg_tail<7>, grid, dim3(256), 0, st, dist, flow, wx, bx, wy, by, B, H, W, out);
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

int vido_bias_res_act(vido_ctx* ctx, float* x, const float* bias, const float* res, int N, int C, int H, int W, float slope)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!res) return vido_bias_act(ctx, x, bias, N, C, H, W, slope);
    if (!x || !bias || N < 1 || C < 1 || H < 1 || W < 1) return vido_set_error(ctx, VIDO_E_INVALID, "bias_res_act: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    const size_t hw = (size_t)H * W, total = (size_t)N * C * hw;
    hipLaunchKernelGGL(k_bias_res_act, dim3((unsigned)((total / 4 + 256) / 256)), dim3(256), 0, st, x, bias, res, C, hw, total, slope);
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

/* [B][C][H][W] -> [B][H][W][C] on DEVICE tensors (f32): the layout the ROI-Align kernel reads (lanes across channels). */
int vido_nchw_to_nhwc(vido_ctx* ctx, const float* src, int B, int C, int H, int W, float* dst)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!src || !dst || B < 1 || C < 1 || H < 1 || W < 1 || B > 65535) return vido_set_error(ctx, VIDO_E_INVALID, "nchw_to_nhwc: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    hipLaunchKernelGGL(k_nchw_to_nhwc, dim3((H * W + 63) / 64, (C + 63) / 64, B), dim3(256), 0, st, src, C, H * W, dst);
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

int vido_roi_align(vido_ctx* ctx, const float* feat, int B, int C, int H, int W, const float* rois, int n_rois, float spatial_scale,
                   int pooled_h, int pooled_w, int sampling_ratio, float* out, int on_device)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!feat || (n_rois && (!rois || !out)) || B < 1 || C < 1 || H < 1 || W < 1 || n_rois < 0 || pooled_h < 1 || pooled_w < 1) return vido_set_error(ctx, VIDO_E_INVALID, "roi_align: bad arguments");
    if (n_rois == 0) return VIDO_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (on_device && ctx->has_ext_stream) ? ctx->ext_stream : ctx->stream;
    const size_t nf = (size_t)B * C * H * W * 4, nr = (size_t)n_rois * 5 * 4, nout = (size_t)n_rois * C * pooled_h * pooled_w * 4;
    const float *df = feat, *dr = rois; float* dout = out; NetState* S = nullptr;
    // scratch: [host staging of map | rois | out]  (host callers only)  +  the channels-last copy of the map
    const size_t base = on_device ? 0 : al256(nf) + al256(nr) + al256(nout);
    { int rc = net_scratch(ctx, base + al256(nf), &S); if (rc) return rc; }
    if (!on_device) {
        for (int i = 0; i < n_rois; i++) if (!(rois[5 * i] >= 0 && rois[5 * i] < B)) return vido_set_error(ctx, VIDO_E_INVALID, "roi_align: roi %d has batch index %g", i, rois[5 * i]);
        memcpy(S->h, feat, nf); memcpy(S->h + al256(nf), rois, nr);
        HIP_TRY(ctx, hipMemcpyAsync(S->d, S->h, al256(nf) + al256(nr), hipMemcpyHostToDevice, st));
        df = (float*)S->d; dr = (float*)(S->d + al256(nf)); dout = (float*)(S->d + al256(nf) + al256(nr));
    }
    float* nhwc = (float*)(S->d + base);
    hipLaunchKernelGGL(k_nchw_to_nhwc, dim3((H * W + 63) / 64, (C + 63) / 64, B), dim3(256), 0, st, df, C, H * W, nhwc);
    RoiLevels L{}; L.feat[0] = nhwc; L.H[0] = H; L.W[0] = W; L.scale[0] = spatial_scale;
    const size_t lds = (size_t)64 * ((pooled_h * pooled_w) | 1) * sizeof(float);
    if (lds > 150 * 1024) return vido_set_error(ctx, VIDO_E_CAPACITY, "roi_align: pooled size %d x %d too large", pooled_h, pooled_w);
    { int rc = roi_lds_limit(ctx, lds); if (rc) return rc; }
    if (sampling_ratio == 2) hipLaunchKernelGGL(k_roi_align_nhwc<2>, dim3(n_rois, (C + 63) / 64), dim3(256), lds, st, L, C, dr, 5, (const int*)nullptr, pooled_h, pooled_w, sampling_ratio, dout);
    else hipLaunchKernelGGL(k_roi_align_nhwc<0>, dim3(n_rois, (C + 63) / 64), dim3(256), lds, st, L, C, dr, 5, (const int*)nullptr, pooled_h, pooled_w, sampling_ratio, dout);
    HIP_TRY(ctx, hipGetLastError());
    if (!on_device) {
        HIP_TRY(ctx, hipStreamSynchronize(st));
        HIP_TRY(ctx, hipMemcpyAsync(S->h, dout, nout, hipMemcpyDeviceToHost, st)); HIP_TRY(ctx, hipStreamSynchronize(st));
        memcpy(out, S->h, nout);
    }
    return VIDO_OK;
}

/* maskrcnn_benchmark.layers.nms(boxes, scores, thresh): kept ORIGINAL indices, ascending.
 * on_device: boxes_xyxy must already be sorted by descending score (scores ignored), keep_out/n_keep are device
 * pointers receiving the kept POSITIONS in that order (ascending); only enqueues on the ctx stream. */
static int nms_impl(vido_ctx* ctx, const float* boxes_xyxy, const float* scores, const int32_t* groups, int n, float thresh, int32_t* keep_out, int32_t* n_keep, int on_device)
{
    if (!ctx) return VIDO_E_INVALID;
    if (n < 0 || !n_keep || (n && (!boxes_xyxy || !keep_out)) || (!on_device && n && !scores)) return vido_set_error(ctx, VIDO_E_INVALID, "nms: bad arguments");
    if (n > 65536) return vido_set_error(ctx, VIDO_E_CAPACITY, "nms: %d boxes > 65536", n);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (on_device && ctx->has_ext_stream) ? ctx->ext_stream : ctx->stream;
    if (n == 0) { if (on_device) HIP_TRY(ctx, hipMemsetAsync(n_keep, 0, 4, st)); else *n_keep = 0; return VIDO_OK; }
    const int cb = (n + 63) / 64;
    const size_t nb = (size_t)n * 16, nm = (size_t)n * cb * 8, nk = (size_t)n * 4 + 256, ng = (size_t)n * 4;
    NetState* S = nullptr;
    int rc = net_scratch(ctx, al256(nb) + al256(nm) + al256(nk) + al256(ng), &S); if (rc) return rc;
    const int* dgroups = groups;
    const float* dboxes = boxes_xyxy; unsigned long long* dmask = (unsigned long long*)(S->d + al256(nb));
    int* dkeep = keep_out; int* dn = n_keep;
    std::vector<int> order;
    if (!on_device) {
        order.resize(n); std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return scores[a] > scores[b]; });
        float* hb = (float*)S->h;
        for (int i = 0; i < n; i++) memcpy(hb + 4 * (size_t)i, boxes_xyxy + 4 * (size_t)order[i], 16);
        HIP_TRY(ctx, hipMemcpyAsync(S->d, S->h, nb, hipMemcpyHostToDevice, st));
        dboxes = (float*)S->d; dkeep = (int*)(S->d + al256(nb) + al256(nm)); dn = dkeep + n;
        if (groups) {
            int* hg = (int*)(S->h + al256(nb)); for (int i = 0; i < n; i++) hg[i] = groups[order[i]];
            int* dg = (int*)(S->d + al256(nb) + al256(nm) + al256(nk));
            HIP_TRY(ctx, hipMemcpyAsync(dg, hg, ng, hipMemcpyHostToDevice, st)); dgroups = dg;
        }
    }
    hipLaunchKernelGGL(k_nms_mask, dim3((cb * (cb + 1) / 2 + 3) / 4), dim3(256), 0, st, dboxes, dgroups, n, thresh, dmask, cb);
    hipLaunchKernelGGL(k_nms_sweep, dim3(1), dim3(64), 0, st, dmask, n, cb, dkeep, dn);
    HIP_TRY(ctx, hipGetLastError());
    if (!on_device) {
        HIP_TRY(ctx, hipStreamSynchronize(st));
        HIP_TRY(ctx, hipMemcpyAsync(S->h, dkeep, nk, hipMemcpyDeviceToHost, st)); HIP_TRY(ctx, hipStreamSynchronize(st));
        const int* hk = (const int*)S->h; const int m = hk[n];
        for (int i = 0; i < m; i++) keep_out[i] = order[hk[i]];
        std::sort(keep_out, keep_out + m);
        *n_keep = m;
    }
    return VIDO_OK;
}

int vido_nms(vido_ctx* ctx, const float* boxes_xyxy, const float* scores, int n, float thresh, int32_t* keep_out, int32_t* n_keep, int on_device)
{
    return nms_impl(ctx, boxes_xyxy, scores, nullptr, n, thresh, keep_out, n_keep, on_device);
}
/* Per-class NMS in one pass (box_head/inference.py:96-118 runs layers.nms once per class): boxes of different `groups` never
 * suppress each other.  Same conventions as vido_nms; on_device: boxes AND groups sorted by descending score. */
int vido_nms_grouped(vido_ctx* ctx, const float* boxes_xyxy, const float* scores, const int32_t* groups, int n, float thresh,
                     int32_t* keep_out, int32_t* n_keep, int on_device)
{
    if (n > 0 && !groups) return ctx ? vido_set_error(ctx, VIDO_E_INVALID, "nms_grouped: null groups") : VIDO_E_INVALID;
    return nms_impl(ctx, boxes_xyxy, scores, groups, n, thresh, keep_out, n_keep, on_device);
}

/* Device-only batched NMS: n_seg independent problems laid out back to back in `boxes` (each segment sorted by descending score); seg_off / seg_n are DEVICE
 * int arrays, max_n >= every segment length (host-known bound), total = rows of `boxes`.  keep_out [n_seg, max_n] receives the kept positions relative to the
 * segment start, ascending, padded with -1; n_keep [n_seg].  Nothing is synchronised. */
int vido_nms_segments(vido_ctx* ctx, const float* boxes_xyxy, const int32_t* groups, const int32_t* seg_off, const int32_t* seg_n, int n_seg, int max_n, int total,
                      float thresh, int32_t* keep_out, int32_t* n_keep)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!boxes_xyxy || !seg_off || !seg_n || !keep_out || !n_keep || n_seg < 1 || max_n < 1 || total < 1) return vido_set_error(ctx, VIDO_E_INVALID, "nms_segments: bad arguments");
    if (max_n > 65536) return vido_set_error(ctx, VIDO_E_CAPACITY, "nms_segments: %d boxes per segment > 65536", max_n);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    const int cb = (max_n + 63) / 64;
    NetState* S = nullptr;
    int rc = net_scratch(ctx, al256((size_t)total * cb * 8), &S); if (rc) return rc;
    unsigned long long* dmask = (unsigned long long*)S->d;
    hipLaunchKernelGGL(k_nms_mask_seg, dim3((cb * (cb + 1) / 2 + 3) / 4, n_seg), dim3(256), 0, st, boxes_xyxy, groups, seg_off, seg_n, thresh, dmask, cb);
    hipLaunchKernelGGL(k_nms_sweep_seg, dim3(n_seg), dim3(64), 0, st, dmask, seg_off, seg_n, cb, keep_out, n_keep, max_n);
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

/* Pooler.forward (modeling/poolers.py:97-121) in one launch: feat[l] DEVICE tensors [1, C, H[l], W[l]] f32 of the 4 FPN levels, boxes [n,4] (x1,y1,x2,y2),
 * level [n] in 0..3 (the LevelMapper's result), out [n, C, pooled, pooled].  Device pointers only; enqueues on the adopted stream. */
int vido_roi_align_fpn(vido_ctx* ctx, const float* const feat[4], const int H[4], const int W[4], const float scale[4], int C, const float* boxes, const int32_t* level,
                       int n, int pooled_h, int pooled_w, int sampling_ratio, float* out)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!feat || !H || !W || !scale || C < 1 || n < 0 || pooled_h < 1 || pooled_w < 1 || (n && (!boxes || !level || !out))) return vido_set_error(ctx, VIDO_E_INVALID, "roi_align_fpn: bad arguments");
    if (n == 0) return VIDO_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    // channel-planar maps: channels-last copies in the scratch first (callers that pool twice from the same maps use vido_nchw_to_nhwc + vido_roi_align_fpn_nhwc)
    size_t off[5] = {0, 0, 0, 0, 0};
    for (int l = 0; l < 4; l++) off[l + 1] = off[l] + al256((size_t)C * H[l] * W[l] * 4);
    NetState* S = nullptr;
    { int rc = net_scratch(ctx, off[4], &S); if (rc) return rc; }
    const float* nh[4];
    for (int l = 0; l < 4; l++) {
        hipLaunchKernelGGL(k_nchw_to_nhwc, dim3((H[l] * W[l] + 63) / 64, (C + 63) / 64, 1), dim3(256), 0, st, feat[l], C, H[l] * W[l], (float*)(S->d + off[l]));
        nh[l] = (const float*)(S->d + off[l]);
    }
    return vido_roi_align_fpn_nhwc(ctx, nh, H, W, scale, C, boxes, level, n, pooled_h, pooled_w, sampling_ratio, out);
}

/* The same with CHANNELS-LAST maps feat[l] = [H[l]][W[l]][C] (vido_nchw_to_nhwc): what the detector calls — one copy per frame serves the box and the mask pooler. */
int vido_roi_align_fpn_nhwc(vido_ctx* ctx, const float* const feat[4], const int H[4], const int W[4], const float scale[4], int C, const float* boxes, const int32_t* level,
                            int n, int pooled_h, int pooled_w, int sampling_ratio, float* out)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!feat || !H || !W || !scale || C < 1 || n < 0 || pooled_h < 1 || pooled_w < 1 || (n && (!boxes || !level || !out))) return vido_set_error(ctx, VIDO_E_INVALID, "roi_align_fpn_nhwc: bad arguments");
    if (n == 0) return VIDO_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    RoiLevels L;
    for (int l = 0; l < 4; l++) { L.feat[l] = feat[l]; L.H[l] = H[l]; L.W[l] = W[l]; L.scale[l] = scale[l]; }
    const size_t lds = (size_t)64 * ((pooled_h * pooled_w) | 1) * sizeof(float);
    if (lds > 150 * 1024) return vido_set_error(ctx, VIDO_E_CAPACITY, "roi_align_fpn: pooled size %d x %d too large", pooled_h, pooled_w);
    { int rc = roi_lds_limit(ctx, lds); if (rc) return rc; }
    if (sampling_ratio == 2) hipLaunchKernelGGL(k_roi_align_nhwc<2>, dim3(n, (C + 63) / 64), dim3(256), lds, st, L, C, boxes, 4, level, pooled_h, pooled_w, sampling_ratio, out);
    else hipLaunchKernelGGL(k_roi_align_nhwc<0>, dim3(n, (C + 63) / 64), dim3(256), lds, st, L, C, boxes, 4, level, pooled_h, pooled_w, sampling_ratio, out);
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

/* Masker(threshold, padding).forward + the node's label image (mask_head/inference.py:87-160, run_mask_rcnn.py:112-118): masks [n,1,M,M] f32 probabilities,
 * boxes [n,4] f32 in the output image, labels [n] i64 (DEVICE pointers, detections in the order the node adds them) -> out [H,W] u8 = sum over detections of
 * (pasted mask) * class index, modulo 256.  n may be 0 (out is cleared). */
int vido_mask_label_image(vido_ctx* ctx, const float* masks, const float* boxes, const int64_t* labels, int n, int M, int padding, float thresh, int H, int W, uint8_t* out)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!out || H < 1 || W < 1 || n < 0 || M < 1 || padding < 0 || (n && (!masks || !boxes || !labels))) return vido_set_error(ctx, VIDO_E_INVALID, "mask_label_image: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    if (n == 0) { HIP_TRY(ctx, hipMemsetAsync(out, 0, (size_t)H * W, st)); return VIDO_OK; }
    NetState* S = nullptr;
    int rc = net_scratch(ctx, al256((size_t)n * sizeof(PasteDet)), &S); if (rc) return rc;
    PasteDet* det = (PasteDet*)S->d;
    hipLaunchKernelGGL(k_paste_prepare, dim3((n + 255) / 256), dim3(256), 0, st, boxes, labels, n, M, padding, det);
    hipLaunchKernelGGL(k_paste_label, dim3((W + 31) / 32, (H + 7) / 8), dim3(256), 0, st, masks, det, n, M, padding, thresh, H, W, out);
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

int vido_box_decode(vido_ctx* ctx, const float* deltas, const float* boxes, int n, int k, const float weights[4], float* out, int on_device)
{
    if (!ctx) return VIDO_E_INVALID;
    if (n < 0 || k < 1 || !weights || (n && (!deltas || !boxes || !out))) return vido_set_error(ctx, VIDO_E_INVALID, "box_decode: bad arguments");
    if (n == 0) return VIDO_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (on_device && ctx->has_ext_stream) ? ctx->ext_stream : ctx->stream;
    const size_t nd = (size_t)n * k * 16, nb = (size_t)n * 16;
    const float *dd = deltas, *db = boxes; float* dout = out; NetState* S = nullptr;
    if (!on_device) {
        int rc = net_scratch(ctx, 2 * al256(nd) + al256(nb), &S); if (rc) return rc;
        memcpy(S->h, deltas, nd); memcpy(S->h + al256(nd), boxes, nb);
        HIP_TRY(ctx, hipMemcpyAsync(S->d, S->h, al256(nd) + al256(nb), hipMemcpyHostToDevice, st));
        dd = (float*)S->d; db = (float*)(S->d + al256(nd)); dout = (float*)(S->d + al256(nd) + al256(nb));
    }
    hipLaunchKernelGGL(k_box_decode, dim3((n * k + 255) / 256), dim3(256), 0, st, dd, db, n, k, weights[0], weights[1], weights[2], weights[3], dout);
    HIP_TRY(ctx, hipGetLastError());
    if (!on_device) {
        HIP_TRY(ctx, hipStreamSynchronize(st));
        HIP_TRY(ctx, hipMemcpyAsync(S->h, dout, nd, hipMemcpyDeviceToHost, st)); HIP_TRY(ctx, hipStreamSynchronize(st));
        memcpy(out, S->h, nd);
    }
    return VIDO_OK;
}

}  // extern "C"
