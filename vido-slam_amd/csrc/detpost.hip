// detpost.hip — the detector's selection logic between its convolutions, on the device, as a handful of launches.
// Replaces (reference src/thirdparty/mask_rcnn/maskrcnn_benchmark):
//   RPNPostProcessor.forward_for_single_feature_map + select_over_all_levels  (modeling/rpn/inference.py:73-159): per FPN level sigmoid, top-k of the objectness,
//       decode + clip of the selected anchors; after the per-level NMS the best fpn_post_nms_top_n boxes over all levels;
//   PostProcessor.filter_results (modeling/roi_heads/box_head/inference.py:96-137): per class score threshold, per-class NMS, the detections_per_img cut through the
//       k-th largest score (kthvalue on the CPU in the reference), results in (class, proposal) order.
// The reference walks these with torch ops that return data-dependent shapes (nonzero, kthvalue + host round trips); round 3's first static form kept torch ops with
// fixed shapes (~140 small launches: sorts of 163 200 and 80 000 elements, scatters, cumulative sums).  Here the same results come from ordered selection kernels:
//   * keys are 64-bit composites  (score bits << 32) | ~index  — scores are non-negative floats, so unsigned order == float order, and "descending key" IS the reference's
//     stable descending sort (ties -> lower index first);
//   * the k best of N are found by a 4-pass radix select over the score bits (+ 3 passes over the index among ties at the cut), then ONE bitonic sort of the k selected
//     keys in LDS — never a sort of all N;
//   * ordered compaction (class-major, proposal-minor output) through block prefix sums.
// Every kernel takes and leaves fixed-size arrays (counts live in device ints): nothing synchronises, the whole detector stays one hipGraph.
#include "common.hpp"
#include <cfloat>

namespace {
typedef unsigned long long u64;

// descending bitonic sort of n (power of two) 64-bit keys in LDS by the whole workgroup
__device__ void block_sort_desc(u64* keys, int n)
{
    for (int k = 2; k <= n; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < n; t += blockDim.x) {
                const int x = t ^ j;
                if (x > t) {
                    const u64 a = keys[t], b = keys[x];
                    const bool up = (t & k) != 0;              // ascending run in the upper halves -> overall descending
                    if (up ? a > b : a < b) { keys[t] = b; keys[x] = a; }
                }
            }
            __syncthreads();
        }
}

// exclusive prefix sum of `v` over the workgroup (blockDim multiple of 64, <= 1024); *total = sum over the workgroup
__device__ int block_excl_scan(int v, int* wsum /*[17]*/, int* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    __syncthreads();
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) { int run = 0; for (int w = 0; w < nw; w++) { const int t = wsum[w]; wsum[w] = run; run += t; } wsum[16] = run; }
    __syncthreads();
    *total = wsum[16];
    return wsum[wave] + incl - v;
}

// One radix-select digit step shared by the selection kernels: hist[256] holds the digit counts of the elements matching the prefix so far; walking from the top (largest
// digit) finds the digit that holds the kk-th largest element.  Called by thread 0.
__device__ inline void pick_digit_desc(const unsigned* hist, unsigned& kk, unsigned& digit)
{
    for (int d = 255; d >= 0; d--) { if (kk <= hist[d]) { digit = (unsigned)d; return; } kk -= hist[d]; }
    digit = 0;
}
__device__ inline void pick_digit_asc(const unsigned* hist, unsigned& kk, unsigned& digit)
{
    for (int d = 0; d < 256; d++) { if (kk <= hist[d]) { digit = (unsigned)d; return; } kk -= hist[d]; }
    digit = 255;
}

// BoxCoder.decode for one box (modeling/box_coder.py:52-95; the same expressions as k_box_decode in nets.hip) followed by clip_to_image (bounding_box.py:214-224)
__device__ inline void decode_clip(const float* b, float d0, float d1, float d2, float d3, float wx, float wy, float ww, float wh, float img_w, float img_h, float* o)
{
    const float clip = (float)log(1000. / 16);
    const float w = b[2] - b[0] + 1, h = b[3] - b[1] + 1, cx = b[0] + 0.5f * w, cy = b[1] + 0.5f * h;
    float dx = d0 / wx, dy = d1 / wy, dw = d2 / ww, dh = d3 / wh;
    if (dw > clip) dw = clip;
    if (dh > clip) dh = clip;
    const float pcx = dx * w + cx, pcy = dy * h + cy, pw = expf(dw) * w, phh = expf(dh) * h;
    const float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * phh, x2 = pcx + 0.5f * pw - 1, y2 = pcy + 0.5f * phh - 1;
    o[0] = fminf(fmaxf(x1, 0.f), img_w - 1); o[1] = fminf(fmaxf(y1, 0.f), img_h - 1); o[2] = fminf(fmaxf(x2, 0.f), img_w - 1); o[3] = fminf(fmaxf(y2, 0.f), img_h - 1);
}

struct RpnLevels { const float* logits[8]; const float* deltas[8]; int h[8], w[8]; float stride[8]; float anchors[8][4][4]; };

// ---- RPN, one workgroup per FPN level: the pre_nms_top_n best anchors by sigmoid(objectness) in descending order (ties: lower anchor index, anchors counted (y, x, a)),
// decoded and clipped.  out rows [level * K, level * K + n): boxes + scores; the rest of the level's K rows: zero box, score -1 (never read by the NMS).
__global__ __launch_bounds__(1024) void k_rpn_level_topk(RpnLevels L, int A, int K, float img_w, float img_h, unsigned* __restrict__ ubuf, int ubuf_stride,
                                                         float* __restrict__ boxes, float* __restrict__ scores, int* __restrict__ n_out)
{
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_kk, s_cnt;
    __shared__ u64 keys[1024];
    const int lv = blockIdx.x, tid = threadIdx.x;
    const int h = L.h[lv], w = L.w[lv], hw = h * w, N = A * hw;
    const float* lg = L.logits[lv]; const float* dl = L.deltas[lv];
    unsigned* ub = ubuf + (size_t)lv * ubuf_stride;
    // score keys in (y, x, a) order: rpn/inference.py:88-92 permutes the [A, h, w] map to (h, w, A) before flattening; sigmoid like torch's kernel: 1 / (1 + exp(-x))
    for (int i = tid; i < N; i += 1024) { const int pix = i / A, a = i - pix * A; const float x = lg[(size_t)a * hw + pix]; ub[i] = __float_as_uint(1.0f / (1.0f + expf(-x))); }
    __syncthreads();
    unsigned T = 0, istar = 0xffffffffu; int n = min(N, K);
    if (N > K) {
        unsigned prefix = 0, mask = 0;
        if (tid == 0) s_kk = (unsigned)K;
        for (int shift = 24; shift >= 0; shift -= 8) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            for (int i = tid; i < N; i += 1024) { const unsigned u = ub[i]; if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u); }
            __syncthreads();
            if (tid == 0) { unsigned kk = s_kk, d; pick_digit_desc(hist, kk, d); s_kk = kk; s_prefix = prefix | (d << shift); }
            __syncthreads();
            prefix = s_prefix; mask |= 0xffu << shift;
        }
        T = prefix;
        const unsigned need_eq = s_kk;                       // how many of the elements equal to T belong to the top K (>= 1): the ones with the lowest indices
        // among the elements equal to T: the need_eq-th smallest index (radix select on the index, ascending)
        unsigned ip = 0, im = 0;
        __syncthreads();
        if (tid == 0) s_kk = need_eq;
        for (int shift = 16; shift >= 0; shift -= 8) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            for (int i = tid; i < N; i += 1024) if (ub[i] == T && ((unsigned)i & im) == ip) atomicAdd(&hist[((unsigned)i >> shift) & 255u], 1u);
            __syncthreads();
            if (tid == 0) { unsigned kk = s_kk, d; pick_digit_asc(hist, kk, d); s_kk = kk; s_prefix = ip | (d << shift); }
            __syncthreads();
            ip = s_prefix; im |= 0xffu << shift;
        }
        istar = ip;
    }
    keys[tid] = 0;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    for (int i = tid; i < N; i += 1024) {
        const unsigned u = ub[i];
        if (N <= K || u > T || (u == T && (unsigned)i <= istar)) { const unsigned pos = atomicAdd(&s_cnt, 1u); if (pos < 1024u) keys[pos] = ((u64)u << 32) | (u64)(0xffffffffu - (unsigned)i); }
    }
    __syncthreads();
    block_sort_desc(keys, 1024);
    float* ob = boxes + (size_t)lv * K * 4; float* os = scores + (size_t)lv * K;
    for (int r = tid; r < K; r += 1024) {
        if (r < n) {
            const u64 key = keys[r]; const unsigned u = (unsigned)(key >> 32); const int i = (int)(0xffffffffu - (unsigned)key);
            const int pix = i / A, a = i - pix * A, y = pix / w, x = pix - y * w;
            const float sx = (float)(x * (int)L.stride[lv]), sy = (float)(y * (int)L.stride[lv]);      // torch.arange(0, w * stride, step = stride, dtype = float32)
            const float an[4] = {sx + L.anchors[lv][a][0], sy + L.anchors[lv][a][1], sx + L.anchors[lv][a][2], sy + L.anchors[lv][a][3]};
            const size_t dq = (size_t)(a * 4) * hw + pix;
            decode_clip(an, dl[dq], dl[dq + hw], dl[dq + 2 * (size_t)hw], dl[dq + 3 * (size_t)hw], 1.f, 1.f, 1.f, 1.f, img_w, img_h, ob + 4 * r);
            os[r] = __uint_as_float(u);
        } else { ob[4 * r] = ob[4 * r + 1] = ob[4 * r + 2] = ob[4 * r + 3] = 0.f; os[r] = -1.f; }
    }
    if (tid == 0) n_out[lv] = n;
}

// ---- RPN, after the per-level NMS: the n_final best kept boxes over all levels (select_over_all_levels, test branch), descending objectness, ties by (level, position).
// keep [L, K]: kept positions per level ascending, cnt [L]; post: per-level cap (post_nms_top_n).  One workgroup, up to 8192 candidates.
__global__ __launch_bounds__(1024) void k_rpn_merge(const float* __restrict__ boxes, const float* __restrict__ scores, const int* __restrict__ keep, const int* __restrict__ cnt,
                                                    int Lv, int K, int post, int n_final, float* __restrict__ out_boxes, float* __restrict__ out_scores, int* __restrict__ n_valid)
{
    extern __shared__ __attribute__((aligned(16))) u64 mkeys[];
    const int tid = threadIdx.x, cap = Lv * K;
    int P2 = 1; while (P2 < cap) P2 <<= 1;
    for (int t = tid; t < P2; t += 1024) {
        u64 key = 0;
        if (t < cap) { const int l = t / K, q = t - l * K; if (q < min(cnt[l], post)) { const int flat = l * K + keep[(size_t)l * K + q]; key = ((u64)__float_as_uint(scores[flat]) << 32) | (u64)(0xffffffffu - (unsigned)flat); } }
        mkeys[t] = key;
    }
    __syncthreads();
    block_sort_desc(mkeys, P2);
    int nv = 0;
    for (int r = tid; r < n_final; r += 1024) {
        const u64 key = r < P2 ? mkeys[r] : 0;
        if (key != 0) { const int flat = (int)(0xffffffffu - (unsigned)key);
                        for (int c = 0; c < 4; c++) out_boxes[4 * r + c] = boxes[4 * (size_t)flat + c];
                        out_scores[r] = __uint_as_float((unsigned)(key >> 32)); nv++; }
        else { for (int c = 0; c < 4; c++) out_boxes[4 * r + c] = 0.f; out_scores[r] = -1.f; }
    }
    // number of real proposals: keys are sorted, so it is the position of the first zero key
    __shared__ int s_nv;
    if (tid == 0) s_nv = 0;
    __syncthreads();
    if (nv) atomicAdd(&s_nv, nv);
    __syncthreads();
    if (tid == 0) *n_valid = s_nv;
}

// ---- box head, one workgroup per foreground class j: the proposals whose class score exceeds the threshold, by descending score (ties: lower proposal index), their
// decoded + clipped class boxes gathered into the class's segment for the NMS.  prob [N, nc], deltas [N, nc * 4], proposals [N, 4], objectness [N] (rows with
// objectness < 0 are padding: no detections from them), N <= 1024.
__global__ __launch_bounds__(1024) void k_det_class_sort(const float* __restrict__ prob, const float* __restrict__ deltas, const float* __restrict__ proposals, const float* __restrict__ objectness,
                                                         int N, int nc, float thresh, float wx, float wy, float ww, float wh, float img_w, float img_h,
                                                         float* __restrict__ seg_boxes /*[(nc-1) * N, 4]*/, int* __restrict__ order /*[(nc-1), N]*/, int* __restrict__ seg_n)
{
    __shared__ u64 keys[1024];
    __shared__ int s_n;
    const int j = blockIdx.x + 1, tid = threadIdx.x;
    if (tid == 0) s_n = 0;
    __syncthreads();
    u64 key = 0;
    if (tid < N) {
        float s = prob[(size_t)tid * nc + j];
        if (objectness && objectness[tid] < 0.f) s = 0.f;
        if (s > thresh) { key = ((u64)__float_as_uint(s) << 32) | (u64)(0xffffffffu - (unsigned)tid); atomicAdd(&s_n, 1); }
    }
    keys[tid] = key;
    __syncthreads();
    block_sort_desc(keys, 1024);
    const int n = s_n;
    if (tid < N) {
        float* ob = seg_boxes + ((size_t)(j - 1) * N + tid) * 4; int* oo = order + (size_t)(j - 1) * N + tid;
        if (tid < n) {
            const int i = (int)(0xffffffffu - (unsigned)keys[tid]);
            const float* d = deltas + ((size_t)i * nc + j) * 4;
            decode_clip(proposals + 4 * (size_t)i, d[0], d[1], d[2], d[3], wx, wy, ww, wh, img_w, img_h, ob);
            *oo = i;
        } else { ob[0] = ob[1] = ob[2] = ob[3] = 0.f; *oo = 0; }
    }
    if (tid == 0) seg_n[j - 1] = n;
}

// ---- box head, the detections_per_img rule: threshold = the k-th largest score among all kept (class, proposal) pairs when there are more than k of them (0 otherwise).
// keep [(nc-1), N] kept positions (into the class's sorted segment) ascending, cnt [(nc-1)].  One workgroup.  thr_out[0] = threshold score bits, thr_out[1] = total kept.
__global__ __launch_bounds__(1024) void k_det_thresh(const float* __restrict__ prob, const int* __restrict__ order, const int* __restrict__ keep, const int* __restrict__ cnt,
                                                     int N, int nc, int kth, unsigned* __restrict__ thr_out)
{
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_kk, s_total;
    const int tid = threadIdx.x, ncls = nc - 1;
    if (tid == 0) s_total = 0;
    __syncthreads();
    { unsigned t = 0; for (int j = tid; j < ncls; j += 1024) t += (unsigned)cnt[j]; if (t) atomicAdd(&s_total, t); }
    __syncthreads();
    const unsigned total = s_total;
    if (kth <= 0 || total <= (unsigned)kth) { if (tid == 0) { thr_out[0] = 0u; thr_out[1] = total; } return; }
    unsigned prefix = 0, mask = 0;
    if (tid == 0) s_kk = (unsigned)kth;
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        for (int j = 0; j < ncls; j++) {
            const int c = cnt[j];
            for (int q = tid; q < c; q += 1024) {
                const int i = order[(size_t)j * N + keep[(size_t)j * N + q]];
                const unsigned u = __float_as_uint(prob[(size_t)i * nc + j + 1]);
                if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u);
            }
        }
        __syncthreads();
        if (tid == 0) { unsigned kk = s_kk, d; pick_digit_desc(hist, kk, d); s_kk = kk; s_prefix = prefix | (d << shift); }
        __syncthreads();
        prefix = s_prefix; mask |= 0xffu << shift;
    }
    if (tid == 0) { thr_out[0] = prefix; thr_out[1] = total; }
}

// ---- box head, the surviving detections in the reference's order (class ascending, then proposal index): per class the kept pairs whose score reaches the threshold.
// Pass 0 (emit == 0): counts per class.  Pass 1: every class knows its offset (sum of the counts before it) and writes its detections into the `cap` output slots;
// slots that stay unused are zeroed by class 0's workgroup.  n_det = number of detections the reference returns (may exceed cap on score ties at the cut).
__global__ __launch_bounds__(1024) void k_det_emit(const float* __restrict__ prob, const float* __restrict__ seg_boxes, const int* __restrict__ order, const int* __restrict__ keep,
                                                   const int* __restrict__ cnt, int N, int nc, const unsigned* __restrict__ thr, int emit, int* __restrict__ class_counts, int cap,
                                                   float* __restrict__ out_boxes, float* __restrict__ out_scores, long long* __restrict__ out_labels, int* __restrict__ n_det)
{
    __shared__ int flag[1024]; __shared__ int spos[1024]; __shared__ int wsum[17];
    const int j = blockIdx.x, tid = threadIdx.x, ncls = nc - 1;
    const unsigned T = thr[0];
    flag[tid] = 0; spos[tid] = 0;
    __syncthreads();
    const int c = cnt[j];
    for (int q = tid; q < c; q += 1024) {
        const int p = keep[(size_t)j * N + q], i = order[(size_t)j * N + p];
        if (__float_as_uint(prob[(size_t)i * nc + j + 1]) >= T) { flag[i] = 1; spos[i] = p; }
    }
    __syncthreads();
    int total = 0;
    const int rank = block_excl_scan(tid < N ? flag[tid] : 0, wsum, &total);
    if (!emit) { if (tid == 0) class_counts[j] = total; return; }
    int off = 0, all = 0;
    for (int jj = 0; jj < ncls; jj++) { const int t = class_counts[jj]; if (jj < j) off += t; all += t; }
    if (tid < N && flag[tid]) {
        const int slot = off + rank;
        if (slot < cap) {
            const float* b = seg_boxes + ((size_t)j * N + spos[tid]) * 4;
            for (int k = 0; k < 4; k++) out_boxes[4 * slot + k] = b[k];
            out_scores[slot] = prob[(size_t)tid * nc + j + 1]; out_labels[slot] = (long long)(j + 1);
        }
    }
    if (j == 0) {
        for (int s = min(all, cap) + tid; s < cap; s += 1024) { for (int k = 0; k < 4; k++) out_boxes[4 * s + k] = 0.f; out_scores[s] = 0.f; out_labels[s] = 0; }
        if (tid == 0) *n_det = all;
    }
}
}  // namespace

extern "C" {

/* RPN selection for all FPN levels in one launch (modeling/rpn/inference.py:73-123 without the NMS): logits[l] [A, h, w], deltas[l] [4A, h, w] DEVICE f32; cell_anchors
 * [n_levels][A][4] HOST (rpn/anchor_generator.py); boxes_out [n_levels * K, 4], scores_out [n_levels * K] (score -1 = padding), n_out [n_levels] DEVICE. */
int vido_rpn_select(vido_ctx* ctx, int n_levels, const float* const* logits, const float* const* deltas, const int* h, const int* w, const int* stride, const float* cell_anchors,
                    int A, int K, int img_w, int img_h, float* boxes_out, float* scores_out, int32_t* n_out)
{
    if (!ctx) return VIDO_E_INVALID;
    if (n_levels < 1 || n_levels > 8 || A < 1 || A > 4 || K < 1 || K > 1024 || !logits || !deltas || !h || !w || !stride || !cell_anchors || !boxes_out || !scores_out || !n_out)
        return vido_set_error(ctx, VIDO_E_INVALID, "rpn_select: bad arguments (<= 8 levels, <= 4 anchors per cell, K <= 1024)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    RpnLevels L; memset(&L, 0, sizeof L); int maxn = 0;
    for (int l = 0; l < n_levels; l++) {
        if (!logits[l] || !deltas[l] || h[l] < 1 || w[l] < 1 || (long long)A * h[l] * w[l] > (1 << 24)) return vido_set_error(ctx, VIDO_E_INVALID, "rpn_select: level %d: bad map", l);
        L.logits[l] = logits[l]; L.deltas[l] = deltas[l]; L.h[l] = h[l]; L.w[l] = w[l]; L.stride[l] = (float)stride[l]; maxn = std::max(maxn, A * h[l] * w[l]);
        for (int a = 0; a < A; a++) for (int c = 0; c < 4; c++) L.anchors[l][a][c] = cell_anchors[((size_t)l * A + a) * 4 + c];
    }
    if (!ctx->detpost_buf || ctx->detpost_cap < (size_t)n_levels * maxn * 4) {      // score-bit scratch; sized once (the warm-up calls happen outside any graph capture)
        HIP_TRY(ctx, hipStreamSynchronize(st));
        if (ctx->detpost_buf) hipFree(ctx->detpost_buf);
        ctx->detpost_cap = (size_t)n_levels * maxn * 4 + 4096; HIP_TRY(ctx, hipMalloc(&ctx->detpost_buf, ctx->detpost_cap));
    }
    hipLaunchKernelGGL(k_rpn_level_topk, dim3(n_levels), dim3(1024), 0, st, L, A, K, (float)img_w, (float)img_h, (unsigned*)ctx->detpost_buf, maxn, boxes_out, scores_out, n_out);
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

/* select_over_all_levels (test branch, rpn/inference.py:125-159) after the per-level NMS: boxes / scores as vido_rpn_select wrote them, keep [n_levels, K] + cnt [n_levels] from
 * vido_nms_segments; the n_final best kept boxes, descending objectness; rows beyond the kept count: zero box, objectness -1.  n_valid: DEVICE int. */
int vido_rpn_merge(vido_ctx* ctx, const float* boxes, const float* scores, const int32_t* keep, const int32_t* cnt, int n_levels, int K, int post_nms_top_n, int n_final,
                   float* out_boxes, float* out_scores, int32_t* n_valid)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!boxes || !scores || !keep || !cnt || !out_boxes || !out_scores || !n_valid || n_levels < 1 || K < 1 || n_final < 1 || (long long)n_levels * K > 8192 || n_final > n_levels * K)
        return vido_set_error(ctx, VIDO_E_INVALID, "rpn_merge: bad arguments (n_levels * K <= 8192)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    int P2 = 1; while (P2 < n_levels * K) P2 <<= 1;
    static bool attr = false;
    if (!attr) { HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_rpn_merge, hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 8)); attr = true; }
    hipLaunchKernelGGL(k_rpn_merge, dim3(1), dim3(1024), (size_t)P2 * 8, st, boxes, scores, keep, cnt, n_levels, K, post_nms_top_n, n_final, out_boxes, out_scores, n_valid);
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

/* PostProcessor.filter_results up to the NMS (box_head/inference.py:96-118): per foreground class the proposals with prob > thresh by descending score, their decoded and clipped
 * boxes in the class's segment seg_boxes[(j-1) * N ..), order[(j-1), N] = proposal index of every sorted position, seg_n[j-1] = count.  All DEVICE; N <= 1024. */
int vido_det_class_sort(vido_ctx* ctx, const float* prob, const float* deltas, const float* proposals, const float* objectness, int N, int nc, float thresh, const float weights[4],
                        int img_w, int img_h, float* seg_boxes, int32_t* order, int32_t* seg_n)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!prob || !deltas || !proposals || !weights || !seg_boxes || !order || !seg_n || N < 1 || N > 1024 || nc < 2 || nc > 65535) return vido_set_error(ctx, VIDO_E_INVALID, "det_class_sort: bad arguments (N <= 1024)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    hipLaunchKernelGGL(k_det_class_sort, dim3(nc - 1), dim3(1024), 0, st, prob, deltas, proposals, objectness, N, nc, thresh, weights[0], weights[1], weights[2], weights[3], (float)img_w, (float)img_h,
                       seg_boxes, order, seg_n);
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

/* PostProcessor.filter_results after the NMS (box_head/inference.py:119-137): the detections_per_img rule and the result lists in (class, proposal) order, written into `cap`
 * slots (zero box / score 0 / label 0 behind the detections); n_det = the number the reference returns (> cap only on score ties at the cut).  scratch: DEVICE, >= (nc + 1) ints. */
int vido_det_select(vido_ctx* ctx, const float* prob, const float* seg_boxes, const int32_t* order, const int32_t* keep, const int32_t* cnt, int N, int nc, int detections_per_img, int cap,
                    int32_t* scratch, float* out_boxes, float* out_scores, int64_t* out_labels, int32_t* n_det)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!prob || !seg_boxes || !order || !keep || !cnt || !scratch || !out_boxes || !out_scores || !out_labels || !n_det || N < 1 || N > 1024 || nc < 2 || cap < 1) return vido_set_error(ctx, VIDO_E_INVALID, "det_select: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    unsigned* thr = (unsigned*)scratch; int* counts = scratch + 2;
    hipLaunchKernelGGL(k_det_thresh, dim3(1), dim3(1024), 0, st, prob, order, keep, cnt, N, nc, detections_per_img, thr);
    hipLaunchKernelGGL(k_det_emit, dim3(nc - 1), dim3(1024), 0, st, prob, seg_boxes, order, keep, cnt, N, nc, (const unsigned*)thr, 0, counts, cap, out_boxes, out_scores, (long long*)out_labels, n_det);
    hipLaunchKernelGGL(k_det_emit, dim3(nc - 1), dim3(1024), 0, st, prob, seg_boxes, order, keep, cnt, N, nc, (const unsigned*)thr, 1, counts, cap, out_boxes, out_scores, (long long*)out_labels, n_det);
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

}  // extern "C"
