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

// ---- 1x1 convolution with FEW input channels (<= 128) + bias + leaky ReLU: LiteFlowNet's netFeat layers (layers.py:99, 125, 140: 32 -> 64 on both feature maps of level 2,
// 32 / 64 / 96 -> 128 in the regularisation).  H*W is large (76 800 at level 2), K is tiny: the library's GEMM + our bias pass take 29-40 + 6 us for 30-50 MB of traffic.
// No LDS: a wave owns 32 positions and ALL output channels; the B operand of v_mfma_f32_32x32x2f32 (two input channels x 32 positions) IS a coalesced global load in operand
// order (lane = position & 31 + 32 * channel parity), the weights are packed in operand order [32-channel block][k-pair][64 lanes] and stay in L1 / L2; bias and the
// activation are applied in the accumulators, a register is a 128-byte line of one output channel.
typedef float f32x16s __attribute__((ext_vector_type(16)));
template <int CB, bool RES>
__global__ __launch_bounds__(256) void k_conv1x1_skinny(const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias, const float* __restrict__ res,
                                                        float* __restrict__ y, int Cin, int Cout, long long HW, float slope, unsigned xbytes, unsigned wbytes)
{
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long p0 = ((long long)blockIdx.x * 4 + wv) * 32;
    if (p0 >= HW) return;
    const long long p = min(p0 + (lane & 31), HW - 1);                        // (positions past the end repeat the last one: computed, not stored)
    const int nkp = Cin >> 1;
    f32x16s acc[CB];
#pragma unroll
    for (int cb = 0; cb < CB; cb++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[cb][r] = 0.f;
    // all addressing scalar (a vector instruction beside the matrix instructions costs 5.5 cycles of their time): per-lane byte offsets are loop invariants, the k-pair rides
    // in the buffer loads' scalar offset
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)wp, 0, wbytes, 0x00020000);
    const unsigned xvo = 4u * (unsigned)((lane >> 5) * HW + p), wvo = 4u * (unsigned)lane;
    const unsigned hw8 = 8u * (unsigned)HW;
    for (int kp0 = 0; kp0 < nkp; kp0 += 4) {                                  // four k-pairs at a time: their loads are in flight together
        float b[4], a[4][CB];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int kp = min(kp0 + u, nkp - 1);
            b[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, xvo, hw8 * (unsigned)kp, 0));
#pragma unroll
            for (int cb = 0; cb < CB; cb++) a[u][cb] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wr, wvo, 256u * (unsigned)(cb * nkp + kp), 0));
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (kp0 + u < nkp) {
#pragma unroll
                for (int cb = 0; cb < CB; cb++) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][cb], b[u], acc[cb], 0, 0, 0);
            }
    }
    if (p0 + (lane & 31) < HW) {
#pragma unroll
        for (int cb = 0; cb < CB; cb++) {
            float bv[16], rv[16];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int co = min(cb * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3), Cout - 1);
                bv[r] = bias ? bias[co] : 0.f; rv[r] = RES ? res[(size_t)co * HW + p] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int co = cb * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                const float v = acc[cb][r] + bv[r] + rv[r];
                if (co < Cout) y[(size_t)co * HW + p] = fmaxf(v, v * slope);
            }
        }
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

/* y = leaky_relu(conv2d(x, w) + bias + residual, slope) for one image, 1x1 kernel, FEW input channels (even, <= 256), cout <= 256: x [cin][hw], y / residual [cout][hw] f32
 * DEVICE tensors (residual NULL = none); w_packed: element (co, k) of the weight at [co / 32][k / 2][32 * (k & 1) + co % 32] (cout padded to a multiple of 32 with zeros;
 * vido_slam_amd/nets/ops.py::pack_conv1x1_skinny).  LiteFlowNet's netFeat layers (flow_net/src/layers.py:99, 125, 140); the detector's layer1 1x1 convolutions
 * (K = 64 / 256 on 200 x 272).  slope in [0, 1]: 0 = ReLU, 1 = none. */
int vido_conv1x1_skinny(vido_ctx* ctx, const float* x, const float* w_packed, const float* bias, const float* residual, float* y, int cin, int cout, long long hw, float slope)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!x || !w_packed || !y || x == y || cin < 2 || (cin & 1) || cin > 256 || cout < 1 || cout > 256 || hw < 1 || slope < 0.f || slope > 1.f || 4ll * cin * hw >= (1ll << 32))
        return vido_set_error(ctx, VIDO_E_INVALID, "conv1x1_skinny: no kernel for %d -> %d channels", cin, cout);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    const dim3 grid((unsigned)((hw + 127) / 128)), blk(256);
    const int cbn = (cout + 31) / 32;
    const unsigned xb = (unsigned)(4ll * cin * hw), wb = (unsigned)(4ll * cbn * 32 * cin);
#define SK_LAUNCH(CBV) { if (residual) hipLaunchKernelGGL((k_conv1x1_skinny<CBV, true>), grid, blk, 0, st, x, w_packed, bias, residual, y, cin, cout, hw, slope, xb, wb); \
                         else hipLaunchKernelGGL((k_conv1x1_skinny<CBV, false>), grid, blk, 0, st, x, w_packed, bias, residual, y, cin, cout, hw, slope, xb, wb); }
    switch (cbn) {
    case 1: SK_LAUNCH(1) break; case 2: SK_LAUNCH(2) break; case 3: SK_LAUNCH(3) break; case 4: SK_LAUNCH(4) break;
    case 5: SK_LAUNCH(5) break; case 6: SK_LAUNCH(6) break; case 7: SK_LAUNCH(7) break; default: SK_LAUNCH(8) break;
    }
#undef SK_LAUNCH
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

}  // extern "C"
