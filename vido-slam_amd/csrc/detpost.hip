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

// ---- RPN: the pre_nms_top_n best anchors of every FPN level by sigmoid(objectness), descending (ties: lower anchor index, anchors counted (y, x, a)), decoded and clipped.
// A level has up to 163 200 anchors (P2 at 800 x 1088): one workgroup walking them several times is a chain of ~1 300 dependent memory round trips (0.9 ms, measured), so the
// selection is a short sequence of wide launches over chunks of SEL_CHUNK anchors:
//   k_rpn_keys      score bits -> ubuf, histogram of bits 31..20                      (chunks x levels)
//   k_sel_resolve   digit that holds the K-th largest -> prefix, remaining k          (1 x levels)          x 3, interleaved with
//   k_sel_hist      histogram of the next digit among the keys matching the prefix    (chunks x levels)      x 2; the last one also counts, per chunk, the keys of every
//                   low-byte value (a tie group at the cut is cut in INDEX order: the chunks' counts give every chunk the rank of its first equal key)
//   k_sel_collect   keys above the cut + the first need_eq keys equal to it -> the level's list (unordered)
//   k_rpn_emit      bitonic sort of the <= 1024 selected composite keys, decode + clip, padding rows
#define SEL_CHUNK 2048
#define SEL_BINS 4096
struct SelState { unsigned prefix, mask, kk, T, need_eq, done, n_sel, pad; };      // per level

__global__ __launch_bounds__(1024) void k_rpn_keys(RpnLevels L, int A, unsigned* __restrict__ ubuf, int ubuf_stride, unsigned* __restrict__ hist /*[levels][SEL_BINS]*/)
{
    __shared__ unsigned lh[SEL_BINS];
    const int lv = blockIdx.y, tid = threadIdx.x;
    const int hw = L.h[lv] * L.w[lv], N = A * hw, i0 = blockIdx.x * SEL_CHUNK;
    if (i0 >= N) return;
    for (int t = tid; t < SEL_BINS; t += 1024) lh[t] = 0;
    __syncthreads();
    const float* lg = L.logits[lv]; unsigned* ub = ubuf + (size_t)lv * ubuf_stride;
    // score keys in (y, x, a) order: rpn/inference.py:88-92 permutes the [A, h, w] map to (h, w, A) before flattening; sigmoid like torch's kernel: 1 / (1 + exp(-x))
#pragma unroll
    for (int r = 0; r < SEL_CHUNK / 1024; r++) {
        const int i = i0 + r * 1024 + tid;
        if (i < N) { const int pix = i / A, a = i - pix * A; const float x = lg[(size_t)a * hw + pix]; const unsigned u = __float_as_uint(1.0f / (1.0f + expf(-x))); ub[i] = u; atomicAdd(&lh[u >> 20], 1u); }
    }
    __syncthreads();
    for (int t = tid; t < SEL_BINS; t += 1024) if (lh[t]) atomicAdd(&hist[(size_t)lv * SEL_BINS + t], lh[t]);
}
// one workgroup per level: the bin (counted from the TOP) that holds the kk-th largest key; step 0 / 1 / 2 = bits 31..20 / 19..8 / 7..0.  After step 2: T, need_eq and — from
// the chunks' low-byte counts — the rank of every chunk's first key equal to T (chunk_eq[chunk][256] -> eq_off[chunk]).
__global__ __launch_bounds__(1024) void k_sel_resolve(const int* __restrict__ n_of /*[levels]*/, int K, int step, unsigned* __restrict__ hist, SelState* __restrict__ state,
                                                      const unsigned* __restrict__ chunk_eq, unsigned* __restrict__ eq_off, int max_chunks)
{
    __shared__ int wsum[17];
    __shared__ unsigned s_digit, s_kk;
    const int lv = blockIdx.x, tid = threadIdx.x, N = n_of[lv];
    SelState st = state[lv];
    if (step == 0) { st.prefix = 0; st.mask = 0; st.kk = (unsigned)K; st.done = N <= K ? 1u : 0u; st.T = 0; st.need_eq = 0; st.n_sel = 0; }
    unsigned* h = hist + (size_t)lv * SEL_BINS;
    if (!st.done) {
        // thread t owns bins [4t, 4t+4) counted from the top: bin index SEL_BINS - 1 - (4t + q)
        unsigned c[4]; int mine = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) { c[q] = h[SEL_BINS - 1 - (4 * tid + q)]; mine += (int)c[q]; }
        int total; const int before = block_excl_scan(mine, wsum, &total);
        if ((unsigned)before < st.kk && st.kk <= (unsigned)(before + mine)) {          // exactly one thread
            unsigned kk = st.kk - (unsigned)before;
#pragma unroll
            for (int q = 0; q < 4; q++) { if (kk <= c[q]) { s_digit = (unsigned)(SEL_BINS - 1 - (4 * tid + q)); s_kk = kk; break; } kk -= c[q]; }
        }
        __syncthreads();
        const int shift = step == 0 ? 20 : (step == 1 ? 8 : 0);
        st.prefix |= s_digit << shift; st.mask |= (step == 2 ? 0xffu : 0xfffu) << shift; st.kk = s_kk;
        if (step == 2) { st.T = st.prefix; st.need_eq = st.kk; }
    }
    __syncthreads();
    for (int t = tid; t < SEL_BINS; t += 1024) h[t] = 0;                              // ready for the next histogram pass (and for the next frame)
    if (step == 2 && !st.done) {                                                      // exclusive scan over the chunks of their counts of keys equal to T
        const unsigned low = st.T & 0xffu; const int nch = (N + SEL_CHUNK - 1) / SEL_CHUNK;
        unsigned run = 0;
        for (int c0 = 0; c0 < nch; c0 += 1024) {
            const int ch = c0 + tid; const int v = ch < nch ? (int)chunk_eq[((size_t)lv * max_chunks + ch) * 256 + low] : 0;
            int tot; const int ex = block_excl_scan(v, wsum, &tot);
            if (ch < nch) eq_off[(size_t)lv * max_chunks + ch] = run + (unsigned)ex;
            run += (unsigned)tot;
        }
    }
    if (tid == 0) state[lv] = st;
}
__global__ __launch_bounds__(1024) void k_sel_hist(const int* __restrict__ n_of, const unsigned* __restrict__ ubuf, int ubuf_stride, int step /*1: bits 19..8, 2: bits 7..0*/,
                                                   const SelState* __restrict__ state, unsigned* __restrict__ hist, unsigned* __restrict__ chunk_eq, int max_chunks)
{
    __shared__ unsigned lh[SEL_BINS];
    const int lv = blockIdx.y, tid = threadIdx.x, N = n_of[lv], i0 = blockIdx.x * SEL_CHUNK;
    const SelState st = state[lv];
    if (i0 >= N || st.done) return;
    const int nb = step == 1 ? SEL_BINS : 256, shift = step == 1 ? 8 : 0;
    for (int t = tid; t < nb; t += 1024) lh[t] = 0;
    __syncthreads();
    const unsigned* ub = ubuf + (size_t)lv * ubuf_stride;
#pragma unroll
    for (int r = 0; r < SEL_CHUNK / 1024; r++) {
        const int i = i0 + r * 1024 + tid;
        if (i < N) { const unsigned u = ub[i]; if ((u & st.mask) == st.prefix) atomicAdd(&lh[(u >> shift) & (unsigned)(nb - 1)], 1u); }
    }
    __syncthreads();
    for (int t = tid; t < nb; t += 1024) { const unsigned v = lh[t]; if (v) atomicAdd(&hist[(size_t)lv * SEL_BINS + t], v); if (step == 2) chunk_eq[((size_t)lv * max_chunks + blockIdx.x) * 256 + t] = v; }
}
__global__ __launch_bounds__(1024) void k_sel_collect(const int* __restrict__ n_of, const unsigned* __restrict__ ubuf, int ubuf_stride, SelState* __restrict__ state,
                                                      const unsigned* __restrict__ eq_off, int max_chunks, int K, u64* __restrict__ sel /*[levels][1024]*/)
{
    __shared__ int wsum[17];
    const int lv = blockIdx.y, tid = threadIdx.x, N = n_of[lv], i0 = blockIdx.x * SEL_CHUNK;
    if (i0 >= N) return;
    const SelState st = state[lv];
    const unsigned* ub = ubuf + (size_t)lv * ubuf_stride;
    unsigned eq_run = st.done ? 0u : eq_off[(size_t)lv * max_chunks + blockIdx.x];
    for (int r = 0; r < SEL_CHUNK / 1024; r++) {                                     // (uniform trip count: the block scan below has barriers)
        const int i = i0 + r * 1024 + tid;
        const unsigned u = i < N ? ub[i] : 0u;
        const bool gt = i < N && (st.done || u > st.T), eq = i < N && !st.done && u == st.T;
        int tot; const int ex = block_excl_scan(eq ? 1 : 0, wsum, &tot);
        if (gt || (eq && eq_run + (unsigned)ex < st.need_eq)) {
            const unsigned pos = atomicAdd(&state[lv].n_sel, 1u);
            if (pos < 1024u) sel[(size_t)lv * 1024 + pos] = ((u64)u << 32) | (u64)(0xffffffffu - (unsigned)i);
        }
        eq_run += (unsigned)tot;
    }
}
__global__ __launch_bounds__(1024) void k_rpn_emit(RpnLevels L, int A, int K, float img_w, float img_h, const u64* __restrict__ sel, SelState* __restrict__ state,
                                                   float* __restrict__ boxes, float* __restrict__ scores, int* __restrict__ n_out)
{
    __shared__ u64 keys[1024];
    const int lv = blockIdx.x, tid = threadIdx.x;
    const int h = L.h[lv], w = L.w[lv], hw = h * w, N = A * hw, n = min(N, K);
    const int ns = (int)min(state[lv].n_sel, 1024u);
    keys[tid] = tid < ns ? sel[(size_t)lv * 1024 + tid] : 0;
    __syncthreads();
    block_sort_desc(keys, 1024);
    const float* dl = L.deltas[lv];
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
    __syncthreads();
    if (tid == 0) { n_out[lv] = n; state[lv].n_sel = 0; }
}

// ---- RPN, after the per-level NMS: the n_final best kept boxes over all levels (select_over_all_levels, test branch), descending objectness, ties by (level, position).
// keep [L, K]: kept positions per level ascending — i.e. every level's kept list is already in descending key order: the output position of an entry is its RANK in the
// union, the number of entries of the other lists with a larger key (binary searches in LDS) plus its position in its own list.  One workgroup; no sort.
__global__ __launch_bounds__(1024) void k_rpn_merge(const float* __restrict__ boxes, const float* __restrict__ scores, const int* __restrict__ keep, const int* __restrict__ cnt,
                                                    int Lv, int K, int post, int n_final, float* __restrict__ out_boxes, float* __restrict__ out_scores, int* __restrict__ n_valid)
{
    extern __shared__ __attribute__((aligned(16))) u64 mkeys[];      // [Lv][K]
    __shared__ int s_cnt[8];
    const int tid = threadIdx.x;
    if (tid < Lv) s_cnt[tid] = min(cnt[tid], post);
    __syncthreads();
    int total = 0;
    for (int l = 0; l < Lv; l++) total += s_cnt[l];
    for (int t = tid; t < Lv * K; t += 1024) {
        const int l = t / K, q = t - l * K;
        if (q < s_cnt[l]) { const int flat = l * K + keep[(size_t)l * K + q]; mkeys[t] = ((u64)__float_as_uint(scores[flat]) << 32) | (u64)(0xffffffffu - (unsigned)flat); }
    }
    for (int r = total + tid; r < n_final; r += 1024) { for (int c = 0; c < 4; c++) out_boxes[4 * r + c] = 0.f; out_scores[r] = -1.f; }
    __syncthreads();
    for (int t = tid; t < Lv * K; t += 1024) {
        const int l = t / K, q = t - l * K;
        if (q >= s_cnt[l]) continue;
        const u64 key = mkeys[t];
        int rank = q;
        for (int m = 0; m < Lv; m++) {
            if (m == l) continue;
            const u64* lst = mkeys + (size_t)m * K; int lo = 0, hi = s_cnt[m];           // number of keys of list m larger than `key` (lists are descending, keys are unique)
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (lst[mid] > key) lo = mid + 1; else hi = mid; }
            rank += lo;
        }
        if (rank < n_final) {
            const int flat = (int)(0xffffffffu - (unsigned)key);
            for (int c = 0; c < 4; c++) out_boxes[4 * rank + c] = boxes[4 * (size_t)flat + c];
            out_scores[rank] = __uint_as_float((unsigned)(key >> 32));
        }
    }
    if (tid == 0) *n_valid = min(total, n_final);
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
// keep [(nc-1), N] kept positions (into the class's sorted segment) ascending, cnt [(nc-1)].  One workgroup; its 16 waves take the classes round-robin (lanes over a class's
// kept list), the kept scores are staged in LDS once (up to DT_STAGE of them; beyond that the radix passes re-read them from memory).  thr_out[0] = threshold bits, [1] = total.
#define DT_STAGE 8192
__global__ __launch_bounds__(1024) void k_det_thresh(const float* __restrict__ prob, const int* __restrict__ order, const int* __restrict__ keep, const int* __restrict__ cnt,
                                                     int N, int nc, int kth, unsigned* __restrict__ thr_out)
{
    __shared__ unsigned hist[256];
    __shared__ unsigned stage[DT_STAGE];
    __shared__ int coff[1025];
    __shared__ unsigned s_prefix, s_kk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ncls = nc - 1;
    if (tid == 0) { int run = 0; for (int j = 0; j < ncls && j < 1024; j++) { coff[j] = run; run += cnt[j]; } coff[min(ncls, 1024)] = run; }
    __syncthreads();
    const unsigned total = (unsigned)coff[min(ncls, 1024)];
    if (kth <= 0 || total <= (unsigned)kth || ncls > 1024) { if (tid == 0) { thr_out[0] = 0u; thr_out[1] = total; } return; }
    const bool staged = total <= DT_STAGE;
    auto score_bits = [&](int j, int q) { const int i = order[(size_t)j * N + keep[(size_t)j * N + q]]; return __float_as_uint(prob[(size_t)i * nc + j + 1]); };
    if (staged) {
        for (int j = wave; j < ncls; j += 16) { const int c = coff[j + 1] - coff[j]; for (int q = lane; q < c; q += 64) stage[coff[j] + q] = score_bits(j, q); }
        __syncthreads();
    }
    unsigned prefix = 0, mask = 0;
    if (tid == 0) s_kk = (unsigned)kth;
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        if (staged) { for (unsigned t = tid; t < total; t += 1024) { const unsigned u = stage[t]; if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u); } }
        else for (int j = wave; j < ncls; j += 16) { const int c = coff[j + 1] - coff[j]; for (int q = lane; q < c; q += 64) { const unsigned u = score_bits(j, q); if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u); } }
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
    RpnLevels L; memset(&L, 0, sizeof L); int maxn = 0; int n_of[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int l = 0; l < n_levels; l++) {
        if (!logits[l] || !deltas[l] || h[l] < 1 || w[l] < 1 || (long long)A * h[l] * w[l] > (1 << 24)) return vido_set_error(ctx, VIDO_E_INVALID, "rpn_select: level %d: bad map", l);
        L.logits[l] = logits[l]; L.deltas[l] = deltas[l]; L.h[l] = h[l]; L.w[l] = w[l]; L.stride[l] = (float)stride[l]; n_of[l] = A * h[l] * w[l]; maxn = std::max(maxn, n_of[l]);
        for (int a = 0; a < A; a++) for (int c = 0; c < 4; c++) L.anchors[l][a][c] = cell_anchors[((size_t)l * A + a) * 4 + c];
    }
    const int max_chunks = (maxn + SEL_CHUNK - 1) / SEL_CHUNK;
    // scratch: [ubuf levels x maxn u32][hist levels x SEL_BINS][state levels][n_of 8 ints][chunk_eq levels x chunks x 256][eq_off levels x chunks][sel levels x 1024 u64]
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_ub = 0, o_hist = o_ub + al((size_t)n_levels * maxn * 4), o_state = o_hist + al((size_t)n_levels * SEL_BINS * 4), o_n = o_state + al((size_t)n_levels * sizeof(SelState)),
                 o_ceq = o_n + al(8 * 4), o_eoff = o_ceq + al((size_t)n_levels * max_chunks * 256 * 4), o_sel = o_eoff + al((size_t)n_levels * max_chunks * 4), total = o_sel + al((size_t)n_levels * 1024 * 8);
    unsigned long long sig = 1469598103934665603ull;                                  // FNV-1a over the geometry: a change re-initialises the scratch
    for (int l = 0; l < 8; l++) { sig ^= (unsigned long long)(unsigned)n_of[l]; sig *= 1099511628211ull; }
    sig ^= (unsigned long long)n_levels; sig *= 1099511628211ull; if (!sig) sig = 1;
    if (!ctx->detpost_buf || ctx->detpost_cap < total || ctx->detpost_sig != sig) {
        // sized and initialised once per geometry, by the warm-up calls outside any graph capture: the histograms are left zero by every call, n_of never changes
        HIP_TRY(ctx, hipStreamSynchronize(st));
        if (ctx->detpost_cap < total) { if (ctx->detpost_buf) hipFree(ctx->detpost_buf); ctx->detpost_cap = total + 4096; HIP_TRY(ctx, hipMalloc(&ctx->detpost_buf, ctx->detpost_cap)); }
        HIP_TRY(ctx, hipMemsetAsync(ctx->detpost_buf, 0, ctx->detpost_cap, st));
        HIP_TRY(ctx, hipMemcpyAsync((char*)ctx->detpost_buf + o_n, n_of, sizeof n_of, hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        ctx->detpost_sig = sig;
    }
    char* base = (char*)ctx->detpost_buf;
    unsigned* ub = (unsigned*)(base + o_ub); unsigned* hist = (unsigned*)(base + o_hist); SelState* state = (SelState*)(base + o_state); const int* dn = (const int*)(base + o_n);
    unsigned* ceq = (unsigned*)(base + o_ceq); unsigned* eoff = (unsigned*)(base + o_eoff); u64* sel = (u64*)(base + o_sel);
    const dim3 wide(max_chunks, n_levels);
    hipLaunchKernelGGL(k_rpn_keys, wide, dim3(1024), 0, st, L, A, ub, maxn, hist);
    hipLaunchKernelGGL(k_sel_resolve, dim3(n_levels), dim3(1024), 0, st, dn, K, 0, hist, state, (const unsigned*)ceq, eoff, max_chunks);
    hipLaunchKernelGGL(k_sel_hist, wide, dim3(1024), 0, st, dn, (const unsigned*)ub, maxn, 1, (const SelState*)state, hist, ceq, max_chunks);
    hipLaunchKernelGGL(k_sel_resolve, dim3(n_levels), dim3(1024), 0, st, dn, K, 1, hist, state, (const unsigned*)ceq, eoff, max_chunks);
    hipLaunchKernelGGL(k_sel_hist, wide, dim3(1024), 0, st, dn, (const unsigned*)ub, maxn, 2, (const SelState*)state, hist, ceq, max_chunks);
    hipLaunchKernelGGL(k_sel_resolve, dim3(n_levels), dim3(1024), 0, st, dn, K, 2, hist, state, (const unsigned*)ceq, eoff, max_chunks);
    hipLaunchKernelGGL(k_sel_collect, wide, dim3(1024), 0, st, dn, (const unsigned*)ub, maxn, state, (const unsigned*)eoff, max_chunks, K, sel);
    hipLaunchKernelGGL(k_rpn_emit, dim3(n_levels), dim3(1024), 0, st, L, A, K, (float)img_w, (float)img_h, (const u64*)sel, state, boxes_out, scores_out, n_out);
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

/* select_over_all_levels (test branch, rpn/inference.py:125-159) after the per-level NMS: boxes / scores as vido_rpn_select wrote them, keep [n_levels, K] + cnt [n_levels] from
 * vido_nms_segments; the n_final best kept boxes, descending objectness; rows beyond the kept count: zero box, objectness -1.  n_valid: DEVICE int. */
int vido_rpn_merge(vido_ctx* ctx, const float* boxes, const float* scores, const int32_t* keep, const int32_t* cnt, int n_levels, int K, int post_nms_top_n, int n_final,
                   float* out_boxes, float* out_scores, int32_t* n_valid)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!boxes || !scores || !keep || !cnt || !out_boxes || !out_scores || !n_valid || n_levels < 1 || K < 1 || n_final < 1 || (long long)n_levels * K > 8160 || n_levels > 8 || n_final > n_levels * K)
        return vido_set_error(ctx, VIDO_E_INVALID, "rpn_merge: bad arguments (n_levels * K <= 8160: 8 bytes per box of dynamic LDS + the kernel's static LDS within the 64 KB of a workgroup)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    static bool attr[64] = {};      // per device: the attribute belongs to the device's copy of the code object
    if (!attr[ctx->device & 63]) { HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_rpn_merge, hipFuncAttributeMaxDynamicSharedMemorySize, 8160 * 8)); attr[ctx->device & 63] = true; }
    hipLaunchKernelGGL(k_rpn_merge, dim3(1), dim3(1024), (size_t)n_levels * K * 8, st, boxes, scores, keep, cnt, n_levels, K, post_nms_top_n, n_final, out_boxes, out_scores, n_valid);
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
