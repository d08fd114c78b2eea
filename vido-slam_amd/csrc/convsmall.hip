// k x k convolution to TWO output channels (+ bias + residual) — the last layer of LiteFlowNet's matching / sub-pixel heads (flow_net/src/layers.py:152-160, 191-199:
// nn.Conv2d(32, 2, k, 1, k // 2) with k = 7 / 5 / 5 / 3 / 3 by level, followed by `flow + netMain(...)`).
//
// What it replaces: the library runs these as im2col + GEMM — at level 2 (240 x 320) the patch matrix is 32 * 49 * 76 800 floats = 481 MB written and read back for
// 0.24 GFLOP: 93 us per call (+ 6 us of bias / residual pass), twice per level.  The layer is a stencil with almost no arithmetic per byte: a workgroup stages a
// (16 + k - 1)^2 window of 8 input channels in LDS, a thread owns one output pixel and both channels, the chunk's weights sit in LDS as (channel 0, channel 1) pairs (broadcast reads), bias and the residual
// flow are added before the store.  fp32 FMAs in the reference's summation order per channel chunk (ascending channel, row-major taps).
#include "common.hpp"

namespace {
template <int K>
__global__ __launch_bounds__(256) void k_conv_kxk_c2(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, const float* __restrict__ res,
                                                     float* __restrict__ y, int Cin, int H, int W)
{
    constexpr int R = K / 2, TS = 16 + K - 1, CC = 8, NT = CC * TS * TS, NL = (NT + 255) / 256, NW = (CC * K * K * 2 + 255) / 256;
    __shared__ float tile[NT];                                                 // [CC][TS][TS]
    __shared__ __attribute__((aligned(8))) float wl[NW * 256];                 // the chunk's weights [CC][K * K][2]: (channel 0, channel 1) pairs, broadcast LDS reads
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4, bx = blockIdx.x * 16, by = blockIdx.y * 16;
    const size_t HW = (size_t)H * W;
    // this thread's NL elements of a chunk's window and NW of its weights: offsets are loop invariants, loads are unconditional (clamped address, then a select) and all
    // in flight together, one chunk ahead of the arithmetic (a loop of conditional loads paid one memory round trip per element: 63 us at 240 x 320 instead of ~20)
    int toff[NL]; bool tin[NL]; int tch[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) {
        const int i = min(tid + 256 * j, NT - 1), c = i / (TS * TS), r = i - c * (TS * TS), yy = r / TS, xx = r - yy * TS, gy = by + yy - R, gx = bx + xx - R;
        tch[j] = c; tin[j] = tid + 256 * j < NT && gy >= 0 && gy < H && gx >= 0 && gx < W;
        toff[j] = min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1);
    }
    float tv[NL], wv[NW];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int j = 0; j < NL; j++) { const int c = min(c0 + tch[j], Cin - 1); tv[j] = x[(size_t)c * HW + toff[j]]; }
#pragma unroll
        for (int j = 0; j < NW; j++) {
            const int i = min(tid + 256 * j, CC * K * K * 2 - 1), o = i & 1, t = (i >> 1) % (K * K), c = min(c0 + (i >> 1) / (K * K), Cin - 1);
            wv[j] = w[((size_t)o * Cin + c) * (K * K) + t];
        }
    };
    float acc0 = bias ? bias[0] : 0.f, acc1 = bias ? bias[1] : 0.f;
    fetch(0);
    for (int c0 = 0; c0 < Cin; c0 += CC) {
        __syncthreads();                                                      // the previous chunk's reads are done
#pragma unroll
        for (int j = 0; j < NL; j++) if (tid + 256 * j < NT) tile[tid + 256 * j] = (tin[j] && c0 + tch[j] < Cin) ? tv[j] : 0.f;
#pragma unroll
        for (int j = 0; j < NW; j++) { const int i = tid + 256 * j; if (i < CC * K * K * 2) wl[i] = (c0 + (i >> 1) / (K * K) < Cin) ? wv[j] : 0.f; }
        __syncthreads();
        if (c0 + CC < Cin) fetch(c0 + CC);
#pragma unroll 2
        for (int c = 0; c < CC; c++) {                                        // (channels past Cin carry zero weights)
            const float* tc = tile + c * TS * TS + ty * TS + tx;
            const float2* wc = (const float2*)wl + c * K * K;
#pragma unroll
            for (int dy = 0; dy < K; dy++)
#pragma unroll
                for (int dx = 0; dx < K; dx++) {
                    const float v = tc[dy * TS + dx];
                    const float2 ww = wc[dy * K + dx];
                    acc0 = fmaf(v, ww.x, acc0); acc1 = fmaf(v, ww.y, acc1);
                }
        }
    }
    const int gy = by + ty, gx = bx + tx;
    if (gy < H && gx < W) {
        const size_t o = (size_t)gy * W + gx;
        y[o] = acc0 + (res ? res[o] : 0.f); y[HW + o] = acc1 + (res ? res[HW + o] : 0.f);
    }
}
}  // namespace

extern "C" {

/* y = conv2d(x, w, padding k / 2) + bias + residual for one image, TWO output channels, k = 3, 5 or 7, stride 1: x [cin][h][w], w [2][cin][k][k], bias [2] or NULL,
 * residual / y [2][h][w] (residual NULL = none), f32 DEVICE tensors.  The last layer of LiteFlowNet's flow heads with the `flow + ...` behind it
 * (flow_net/src/layers.py:152-160, 191-199).  Enqueues on the adopted stream; capturable. */
int vido_conv_kxk_c2(vido_ctx* ctx, const float* x, const float* w, const float* bias, const float* residual, float* y, int cin, int k, int h, int wd)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!x || !w || !y || x == y || cin < 1 || h < 1 || wd < 1 || (k != 3 && k != 5 && k != 7)) return vido_set_error(ctx, VIDO_E_INVALID, "conv_kxk_c2: bad arguments (k = %d)", k);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    const dim3 grid((wd + 15) / 16, (h + 15) / 16), blk(256);
    if (k == 7) hipLaunchKernelGGL(k_conv_kxk_c2<7>, grid, blk, 0, st, x, w, bias, residual, y, cin, h, wd);
    else if (k == 5) hipLaunchKernelGGL(k_conv_kxk_c2<5>, grid, blk, 0, st, x, w, bias, residual, y, cin, h, wd);
    else hipLaunchKernelGGL(k_conv_kxk_c2<3>, grid, blk, 0, st, x, w, bias, residual, y, cin, h, wd);
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

}  // extern "C"
