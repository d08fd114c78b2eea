// k x k convolution (any stride, zero padding) + bias + (leaky) ReLU as a DIRECT implicit GEMM on the fp32 matrix pipe (gfx950), one launch, no LDS.
//
// What it replaces: the layers of LiteFlowNet that neither the Winograd kernel (3x3, stride 1) nor the 2-channel stencil takes — the 7x7 stem (3 -> 32 at 480 x 640), the
// stride-2 3x3 convolutions of the feature pyramid (32 -> 32 ... 128 -> 192), the separable 7x1 / 1x7 and 5x1 / 1x5 distance layers of the regularisation, the 32 -> 9
// distance layers and the 1-channel 1x1 layers behind them (flow_net/src/layers.py:39-73, 217-235).  The library runs each as 2-5 launches: an im2col or a layout transpose,
// a GEMM or a Winograd kernel on the vector ALUs, transposes back, then the stand-alone bias + activation pass (profiles/r5/nets_timeline_summary.txt: ~75 launches,
// ~1.2 ms of LiteFlowNet's 3.8 ms).
//
// Formulation.  M = output channels (blocks of 32), N = output pixels (blocks of 32, numbered over image, row, column), K = (tap, input channel) with two adjacent
// CHANNELS of one tap per v_mfma_f32_32x32x2f32.  A wave owns one 32-pixel block x CBW channel blocks.  The operands of the matrix instruction are exactly one dword per
// lane each, so nothing is staged:
//   * B (the patch matrix): lane (pixel n = lane & 31, channel parity kk = lane >> 5) reads x[ci = 2 cp + kk][iy][ix] of ITS pixel with a buffer load whose per-lane offset
//     is a per-tap loop invariant (pixel origin + tap displacement, or bit 30 = outside the image -> the hardware returns the zero padding) and whose scalar offset walks the
//     channel pairs.  Consecutive lanes = consecutive pixels: a load is one or two cache lines (stride 1), the 49 taps of the stem re-read the same lines from L1.
//   * A (the weights): packed on the host in operand order [channel block][tap][channel pair][64 lanes], one coalesced 256-byte load per matrix instruction, the same
//     sequence for every wave (L1 / L2 resident).
// Loads run two steps (2 U channel pairs) ahead of the matrix instructions inside a wave (three operand register sets); several waves per SIMD (the kernel needs < 128
// registers) cover the rest of the latency on the large maps.
// Bias, activation and the stores (a lane holds 16 channels of its pixel; for one channel 32 lanes store 128 consecutive bytes) are register work.
#include "common.hpp"

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CD_OOB 0x40000000u

struct CdArgs { const float* x; const float* wp; const float* bias; float* y; int N, Cin, Cout, H, W, Ho, Wo, SH, SW, PH, PW, npix, cgroups, total, cpr, cpp; float slope; unsigned xbytes, wbytes; };

// channel pairs per inner step: 2 for the 49-tap stem (3 input channels = 2 pairs), 4 otherwise; the packed weight pads the pairs of a tap to a multiple of it
constexpr int cd_unroll(int kh, int kw) { return kh * kw >= 25 ? 2 : 4; }

template <int KH, int KW, int CBW>
__global__ __launch_bounds__(256) void k_conv_direct(CdArgs A)
{
    constexpr int T = KH * KW, U = cd_unroll(KH, KW);
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int per = gridDim.x >> 3, wg = (blockIdx.x & 7) * per + (blockIdx.x >> 3);      // an XCD walks a contiguous range of work items (channel group fastest: shared input lines)
    const int item = wg * 4 + w;
    if (item >= A.total) return;
    const int tile = item / A.cgroups, cg = item - tile * A.cgroups;
    const int n = lane & 31, kk = lane >> 5;
    const int g = tile * 32 + n, gc = min(g, A.npix - 1), hwo = A.Ho * A.Wo, hw = A.H * A.W;
    const bool pv = g < A.npix;
    const int img = gc / hwo, rem = gc - img * hwo, oy = rem / A.Wo, ox = rem - oy * A.Wo;
    const int iy0 = oy * A.SH - A.PH, ix0 = ox * A.SW - A.PW;
    const int vbase = 4 * ((img * A.Cin + kk) * hw + iy0 * A.W + ix0);
    const unsigned odd_oob = ((A.Cin & 1) && kk) ? CD_OOB : 0u;                           // the odd channel past an odd Cin
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)A.x, 0, A.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)A.wp, 0, A.wbytes, 0x00020000);
    const unsigned wl = 4u * (unsigned)lane;
    f32x16 acc[CBW];
#pragma unroll
    for (int i = 0; i < CBW; i++)
#pragma unroll
        for (int q = 0; q < 16; q++) acc[i][q] = 0.f;
    const int cpr = A.cpr, cpp = A.cpp, nmain = ((A.Cin >> 1) / U) * U;                   // real / padded channel pairs; pairs of the unmasked steps
    const unsigned cstep = 8u * (unsigned)hw;                                             // bytes between channel pairs
    const unsigned wblk = 4u * 64u * (unsigned)(T * cpp);                                 // channel block stride of the packed weight
    // The K loop is a flat sequence of steps (tap-major, U channel pairs each); the operands of a step are loaded TWO steps ahead of its matrix instructions into one of
    // three register sets, so that a wave alone on its SIMD (the small maps: fewer waves than SIMDs) still has its loads in flight under 2 U CBW matrix instructions.
    // State of the loader: tap, channel pair, the tap's per-lane offset, the scalar offsets of input and weight.
    int lt = 0, lcp = 0;
    unsigned xso = 0u, wsc = 4u * 64u * (unsigned)(cg * CBW * T * cpp);
    auto tap_off = [&](int t) -> unsigned {
        const int ky = t / KW, kx = t - ky * KW;
        const bool tv = pv && (unsigned)(iy0 + ky) < (unsigned)A.H && (unsigned)(ix0 + kx) < (unsigned)A.W;
        return tv ? (unsigned)(vbase + 4 * (ky * A.W + kx)) : CD_OOB;
    };
    unsigned vt = tap_off(0);
    auto load_step = [&](float (&b)[U], float (&a)[CBW][U]) {
        if (lcp < nmain) {                                                                // U pairs of real channels: the per-lane offset is the tap's, everything else scalar
#pragma unroll
            for (int u = 0; u < U; u++) {
                b[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, vt, xso + (unsigned)u * cstep, 0));
#pragma unroll
                for (int i = 0; i < CBW; i++)
                    a[i][u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wr, wl, wsc + (unsigned)i * wblk + 256u * (unsigned)u, 0));
            }
        } else {                                                                          // the last step of a channel count that is not a multiple of 2 U: masked
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int c = lcp + u;                                                    // (scalar)
                // pairs past the real ones (padding of the packed weight) read through voffset = out of range; the last real pair of an odd Cin masks its odd lanes
                const unsigned vo = c >= cpr ? CD_OOB : (c == cpr - 1 ? (vt | odd_oob) : vt);
                b[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, vo, cstep * (unsigned)min(c, cpr - 1), 0));
#pragma unroll
                for (int i = 0; i < CBW; i++)
                    a[i][u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wr, wl, wsc + (unsigned)i * wblk + 256u * (unsigned)u, 0));
            }
        }
        lcp += U; xso += (unsigned)U * cstep; wsc += 256u * (unsigned)U;                  // (the taps of a channel block follow each other in the packed weight)
        if (lcp >= cpp) { lcp = 0; xso = 0u; lt++; vt = tap_off(min(lt, T - 1)); }
    };
    auto mma = [&](const float (&b)[U], const float (&a)[CBW][U]) {
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int i = 0; i < CBW; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][u], b[u], acc[i], 0, 0, 0);
    };
    const int S = T * (cpp / U);
    float b0[U], a0[CBW][U], b1[U], a1[CBW][U], b2[U], a2[CBW][U];
    load_step(b0, a0);
    if (S > 1) load_step(b1, a1);
    for (int st = 0; st < S; st += 3) {
        if (st + 2 < S) load_step(b2, a2);
        mma(b0, a0);
        if (st + 1 < S) {
            if (st + 3 < S) load_step(b0, a0);
            mma(b1, a1);
            if (st + 2 < S) {
                if (st + 4 < S) load_step(b1, a1);
                mma(b2, a2);
            }
        }
    }
    if (!pv) return;
    // D: register q of a lane = output channel 8 (q / 4) + 4 (lane >> 5) + (q & 3) of the block, pixel lane & 31
    float* yb = A.y + (size_t)img * A.Cout * hwo + rem;
#pragma unroll
    for (int i = 0; i < CBW; i++)
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int co = (cg * CBW + i) * 32 + 8 * (q >> 2) + 4 * kk + (q & 3);
            if (co >= A.Cout) continue;
            float v = acc[i][q] + (A.bias ? A.bias[co] : 0.f);
            v = fmaxf(v, v * A.slope);                                                    // leaky ReLU as max(y, slope y), 0 <= slope <= 1 (1: none)
            yb[(size_t)co * hwo] = v;
        }
}

inline bool cd_has_kernel(int kh, int kw)
{
    return (kh == 7 && kw == 7) || (kh == 3 && kw == 3) || (kh == 5 && kw == 5) || (kh == 7 && kw == 1) || (kh == 1 && kw == 7) || (kh == 5 && kw == 1) || (kh == 1 && kw == 5) || (kh == 1 && kw == 1);
}
inline int cd_cpp(int cin, int kh, int kw) { const int u = cd_unroll(kh, kw), cpr = (cin + 1) / 2; return ((cpr + u - 1) / u) * u; }
inline int cd_cbp(int cout) { const int cb = (cout + 31) / 32; return cb == 1 ? 1 : ((cb + 1) / 2) * 2; }      // channel blocks in the packed weight (even, so that pairs of blocks never run past it)
}  // namespace

extern "C" {

/* 1 when vido_conv_direct_bias_act has a kernel for the layer: groups 1, dilation 1, zero padding, a k x k of {7x7, 5x5, 3x3, 7x1, 1x7, 5x1, 1x5, 1x1}, strides 1-4,
 * tensors below 1 GB.  (3x3 stride 1 with >= 8 input and >= 32 output channels is the Winograd kernel's: callers try that first.) */
int vido_conv_direct_supported(int cin, int cout, int h, int w, int kh, int kw, int sh, int sw, int ph, int pw)
{
    if (cin < 1 || cout < 1 || h < 1 || w < 1 || !cd_has_kernel(kh, kw) || sh < 1 || sh > 4 || sw < 1 || sw > 4 || ph < 0 || pw < 0 || ph >= kh || pw >= kw) return 0;
    const long long ho = (h + 2 * ph - kh) / sh + 1, wo = (w + 2 * pw - kw) / sw + 1;
    if (h + 2 * ph < kh || w + 2 * pw < kw || ho < 1 || wo < 1) return 0;
    return 4ll * cin * h * w < (1ll << 30) && 4ll * cout * ho * wo < (1ll << 30);
}

/* floats of the packed weight */
long long vido_conv_direct_packed_floats(int cin, int cout, int kh, int kw)
{
    if (cin < 1 || cout < 1 || !cd_has_kernel(kh, kw)) return 0;
    return 64ll * cd_cbp(cout) * kh * kw * cd_cpp(cin, kh, kw);
}

/* HOST: weight [cout][cin][kh][kw] f32 -> the kernel's operand order: element (output channel co, input channel c, tap t = ky * kw + kx) at
 * [co / 32][t][c / 2][32 * (c & 1) + co % 32]; padded channels / pairs / blocks are zero. */
int vido_conv_direct_pack(const float* w, int cin, int cout, int kh, int kw, float* wp)
{
    if (!w || !wp || cin < 1 || cout < 1 || !cd_has_kernel(kh, kw)) return VIDO_E_INVALID;
    const int T = kh * kw, cpp = cd_cpp(cin, kh, kw);
    std::memset(wp, 0, sizeof(float) * (size_t)vido_conv_direct_packed_floats(cin, cout, kh, kw));
    for (int co = 0; co < cout; co++)
        for (int c = 0; c < cin; c++)
            for (int t = 0; t < T; t++)
                wp[(((size_t)(co / 32) * T + t) * cpp + c / 2) * 64 + 32 * (c & 1) + co % 32] = w[((size_t)co * cin + c) * T + t];
    return VIDO_OK;
}

/* y = leaky_relu(conv2d(x, w, stride (sh, sw), padding (ph, pw)) + bias, slope): x [n][cin][h][w], y [n][cout][ho][wo] f32 DEVICE tensors (y != x), bias [cout] or NULL,
 * w_packed = vido_conv_direct_pack(w) on the device.  slope 0 = ReLU, 1 = none.  fp32 products and sums on the matrix pipe (v_mfma_f32_32x32x2f32); the order of the sum
 * over (tap, channel) differs from the library's, the result by rounding only.  Enqueues on the adopted stream; capturable. */
int vido_conv_direct_bias_act(vido_ctx* ctx, const float* x, const float* w_packed, const float* bias, float* y, int n, int cin, int cout, int h, int w,
                              int kh, int kw, int sh, int sw, int ph, int pw, float slope)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!x || !w_packed || !y || x == y || n < 1 || !vido_conv_direct_supported(cin, cout, h, w, kh, kw, sh, sw, ph, pw) || (((uintptr_t)x | (uintptr_t)y | (uintptr_t)w_packed) & 3))
        return vido_set_error(ctx, VIDO_E_INVALID, "conv_direct: no kernel for %d -> %d channels, %d x %d taps, stride %d x %d on %d x %d x %d", cin, cout, kh, kw, sh, sw, n, h, w);
    const int ho = (h + 2 * ph - kh) / sh + 1, wo = (w + 2 * pw - kw) / sw + 1;
    const long long npix = (long long)n * ho * wo;
    if (npix >= (1ll << 30) || 4ll * n * cin * h * w >= (1ll << 30) || 4ll * n * cout * ho * wo >= (1ll << 32))
        return vido_set_error(ctx, VIDO_E_INVALID, "conv_direct: a batch of %d images of %d x %d x %d is past the 1 GB the kernel addresses", n, cin, h, w);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    const int cb = (cout + 31) / 32, ntile = (int)((npix + 31) / 32);
    // two channel blocks per wave (one input load feeds two matrix instructions) when that still leaves >= 2 waves per SIMD's worth of work items
    const int cbw = (cb >= 2 && (long long)ntile * ((cb + 1) / 2) >= 2048) ? 2 : 1;
    const int cgroups = (cb + cbw - 1) / cbw;
    const long long total = (long long)ntile * cgroups;
    if (total >= (1ll << 30)) return vido_set_error(ctx, VIDO_E_INVALID, "conv_direct: too many work items");
    CdArgs A{x, w_packed, bias, y, n, cin, cout, h, w, ho, wo, sh, sw, ph, pw, (int)npix, cgroups, (int)total, (cin + 1) / 2, cd_cpp(cin, kh, kw), slope,
             (unsigned)(4ll * n * cin * h * w), (unsigned)(4ll * vido_conv_direct_packed_floats(cin, cout, kh, kw))};
    const int nwg = (int)((total + 3) / 4);
    const dim3 grid(8 * ((nwg + 7) / 8)), blk(256);
#define CD_LAUNCH(KH_, KW_) do { if (cbw == 2) hipLaunchKernelGGL((k_conv_direct<KH_, KW_, 2>), grid, blk, 0, st, A); else hipLaunchKernelGGL((k_conv_direct<KH_, KW_, 1>), grid, blk, 0, st, A); } while (0)
    if (kh == 7 && kw == 7) CD_LAUNCH(7, 7);
    else if (kh == 5 && kw == 5) CD_LAUNCH(5, 5);
    else if (kh == 3 && kw == 3) CD_LAUNCH(3, 3);
    else if (kh == 7 && kw == 1) CD_LAUNCH(7, 1);
    else if (kh == 1 && kw == 7) CD_LAUNCH(1, 7);
    else if (kh == 5 && kw == 1) CD_LAUNCH(5, 1);
    else if (kh == 1 && kw == 5) CD_LAUNCH(1, 5);
    else CD_LAUNCH(1, 1);
#undef CD_LAUNCH
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

}  // extern "C"
