// 3x3 stride-1 convolution + bias + activation as Winograd F(2x2, 3x3) whose sixteen channel contractions run on the fp32 MATRIX pipe (gfx950), one launch.
//
// What it replaces: the dense 3x3 convolutions of the three network nodes — LiteFlowNet's feature / matching / sub-pixel / regularisation chains
// (flow_net/src/layers.py:39-315: 49|130|131|.. -> 128 -> 64 -> 32 at 240x320 ... 15x20), the FPN output convolutions and the RPN / mask-head 3x3 convolutions of the
// detector (maskrcnn_benchmark/modeling/backbone/fpn.py, rpn/rpn.py:74-107, roi_heads/mask_head/roi_mask_feature_extractors.py).  The library runs them as
// miopenSp3AsmConv f2x3 — the same Winograd algorithm with its multiplications on the VECTOR ALUs (71-78 TFLOP/s "direct-equivalent") — followed by a separate bias +
// activation pass over the output.  The algorithm's multiplication stage is sixteen independent GEMMs  M_p[co][t] = sum_c U_p[co][c] V_p[c][t]  (p = position in the 4x4
// transformed tile, t = 2x2 output tile): matrix-core work.  In the transformed domain a convolution costs 16 / 36 of the direct multiply-adds.
//
// Formulation.
//   * U_p = (G g G^T)_p is computed once on the host (vido_wino3x3_pack, float64 -> f32) and stored in OPERAND order; V_p = (B^T d B)_p is formed inside the kernel.
//   * A wave owns 32 output channels x 32 tiles x ALL sixteen positions: sixteen v_mfma_f32_32x32x2f32 accumulators (256 registers — the unified 512-register file at one
//     wave per SIMD), so the inverse transform A^T M A, the bias and the activation are register arithmetic of one lane, and a lane's stores are the two output rows of its
//     tile (consecutive lanes = consecutive tiles of a tile row: full lines).
//   * Workgroup = 4 waves = CW x TW (32-channel blocks x 32-tile blocks): 2 x 2 for layers with >= 64 output channels, 1 x 4 for 32.  Per K chunk of KC input channels:
//       - every thread loads the 4x4 input windows of its (channel, tile) pairs straight from global memory (one chunk ahead, in registers; zero padding = a select),
//         transforms them (32 adds per pair) and writes V to LDS as [p][tile block][k-pair][64 operand slots] — the B operand of a matrix instruction is one ds_read_b32;
//       - U arrives by asynchronous global -> LDS copies (buffer loads with the lds bit, 16 bytes per lane, 1 KB per wave instruction) one chunk ahead; the A operands of KC / 2 consecutive matrix
//         instructions are one ds_read_b128 / b64;
//       - ONE barrier per chunk (two LDS buffers for U and V).
//   * Tiles are numbered over (image, tile row, tile column): any H x W (odd sizes: the last row / column of tiles is stored half), any batch — LiteFlowNet's image pair
//     and the mask head's per-detection 14x14 maps are just more tiles.  Input channels are padded to KC with zero weights, output channels to 32.
// Work items (tile block, channel block) are dealt so that an XCD walks a contiguous range with the channel block fastest (the blocks that share input windows run back to
// back on one L2).
#include "common.hpp"

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define WN_OOB 0x40000000u
struct WnArgs { const float* x; const float* up; const float* bias; float* y; int N, Cin, Cout, H, W, Ht, Wt, T, nchunk, cgroups, total; float slope; int vec2; unsigned xbytes, ubytes; };

template <int CW, int TW, int KC>
__global__ __launch_bounds__(256) void k_wino3x3(WnArgs A)
{
    static_assert(CW * TW == 4 && (KC == 8 || KC == 4), "workgroup = 4 waves");
    constexpr int KS = KC / 2;                        // matrix instructions (k-pairs) per position and chunk
    constexpr int U_BLK = 16 * 64 * KS;               // floats of one (32-channel block, chunk): [p][lane][KS]
    constexpr int U_BUF = CW * U_BLK, V_BUF = 16 * TW * KS * 64, V_P = TW * KS * 64;
    constexpr int NJ = TW * KS / 4;                   // (channel, tile) pairs a thread transforms per chunk
    constexpr int KSTEP = 4 / TW;
    constexpr int PPB = U_BLK * 4 / 1024;             // 1 KB copy pieces per U block
    extern __shared__ __attribute__((aligned(16))) float wn_lds[];      // [2][U_BUF] [2][V_BUF]
    float* Ul = wn_lds; float* Vl = wn_lds + 2 * U_BUF;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int per = gridDim.x >> 3, item = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (item >= A.total) return;
    const int tb = item / A.cgroups, cg = item - tb * A.cgroups, tile0 = tb * (TW * 32);
    const int hw = A.H * A.W, tpi = A.Ht * A.Wt;

    // ---- transform role: tile block tw_t, k-pairs ks0 + j * KSTEP; lane = (tile & 31) + 32 * (channel & 1) = the operand slot it fills.
    // Input windows are read with BUFFER loads: an element outside the image (zero padding), of a tile past the end or of a channel >= Cin gets a byte offset with bit 30
    // set, which is past the descriptor's range (the tensor is < 1 GB) — the hardware returns 0, no select.
    const int tw_t = w % TW, ks0 = w / TW;
    unsigned rowo[4], colo[4];
    {
        const int gt = tile0 + tw_t * 32 + (lane & 31), gtc = min(gt, A.T - 1);
        const int n = gtc / tpi, rem = gtc - n * tpi, ty = rem / A.Wt, tx = rem - ty * A.Wt;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const int iy = 2 * ty - 1 + d, ix = 2 * tx - 1 + d;
            rowo[d] = (gt < A.T && iy >= 0 && iy < A.H) ? 4u * ((unsigned)n * (unsigned)A.Cin * (unsigned)hw + (unsigned)(iy * A.W)) : WN_OOB;
            colo[d] = (ix >= 0 && ix < A.W) ? 4u * (unsigned)ix : WN_OOB;
        }
    }
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)A.x, 0, A.xbytes, 0x00020000);
    float in[NJ][16];
    // V = B^T d B of pair j in two steps the main loop spreads over its slices: rows (r = B^T d: the window registers are dead afterwards), then output row i -> LDS
    float r[16];
    auto xf_rows = [&](int j) {
        const float* d = in[j];
#pragma unroll
        for (int x = 0; x < 4; x++) { r[x] = d[x] - d[8 + x]; r[4 + x] = d[4 + x] + d[8 + x]; r[8 + x] = d[8 + x] - d[4 + x]; r[12 + x] = d[4 + x] - d[12 + x]; }
    };
    auto xf_out = [&](int j, int i, int buf) {
        float* dst = Vl + buf * V_BUF + (tw_t * KS + ks0 + j * KSTEP) * 64 + lane;
        dst[(4 * i + 0) * V_P] = r[4 * i] - r[4 * i + 2];
        dst[(4 * i + 1) * V_P] = r[4 * i + 1] + r[4 * i + 2];
        dst[(4 * i + 2) * V_P] = r[4 * i + 2] - r[4 * i + 1];
        dst[(4 * i + 3) * V_P] = r[4 * i + 1] - r[4 * i + 3];
    };
    auto load_rows = [&](int chunk, int j, int dy0) {                     // window rows dy0, dy0 + 1 of pair j of `chunk`
        const int c = chunk * KC + 2 * (ks0 + j * KSTEP) + (lane >> 5);
        const unsigned cb = c < A.Cin ? 4u * (unsigned)c * (unsigned)hw : WN_OOB;
#pragma unroll
        for (int dy = dy0; dy < dy0 + 2; dy++) {
            const unsigned rb = rowo[dy] + cb;
#pragma unroll
            for (int dx = 0; dx < 4; dx++) in[j][4 * dy + dx] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, rb + colo[dx], 0, 0));
        }
    };
    // U pieces travel global -> LDS as BUFFER loads with the lds bit (16 bytes per lane): unlike the flat-encoded global_load_lds they keep the compiler's vmcnt bookkeeping
    // in order (a pending flat LDS access makes it fall back to vmcnt(0) / lgkmcnt(0) in front of every later use of any loaded register)
    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc((void*)A.up, 0, A.ubytes, 0x00020000);
    auto issue_u = [&](int chunk, int buf) {                             // wave w moves pieces w, w + 4, ... of the CW blocks of this chunk
#pragma unroll
        for (int i0 = 0; i0 < CW * PPB; i0 += 4) {
            const int i = i0 + w, cwi = i / PPB, pi = i - cwi * PPB;
            const unsigned src = 4u * (unsigned)(((cg * CW + cwi) * A.nchunk + chunk) * U_BLK + pi * 256 + lane * 4);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ur, (__attribute__((address_space(3))) void*)(Ul + buf * U_BUF + cwi * U_BLK + pi * 256), 16, src, 0, 0, 0);
        }
    };

    // ---- matrix role: channel block cw, tile block tw
    const int cw = w % CW, tw = w / CW;
    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; p++)
#pragma unroll
        for (int q = 0; q < 16; q++) acc[p][q] = 0.f;

    static_assert(NJ == 2, "the slices below are written for two pairs per thread");
    load_rows(0, 0, 0); load_rows(0, 0, 2); load_rows(0, 1, 0); load_rows(0, 1, 2);
    issue_u(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 2; j++) {
        xf_rows(j);
#pragma unroll
        for (int i = 0; i < 4; i++) xf_out(j, i, 0);
    }
    load_rows(1, 0, 0); load_rows(1, 0, 2); load_rows(1, 1, 0); load_rows(1, 1, 2);
    typedef float uvec __attribute__((ext_vector_type(KS)));
    for (int c = 0; c < A.nchunk; c++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                 // U(c), V(c) are in buffer c & 1; everybody is done with buffer (c + 1) & 1
        const int nb = (c + 1) & 1;
        const float* Ub = Ul + (c & 1) * U_BUF + cw * U_BLK + lane * KS;
        const float* Vb = Vl + (c & 1) * V_BUF + tw * KS * 64 + lane;
        uvec ua[3][2]; float vb[3][2][KS];
        auto ldops = [&](int g) {                                        // operands of positions 2 g, 2 g + 1: A = KS consecutive floats, B = KS reads 64 floats apart
#pragma unroll
            for (int q = 0; q < 2; q++) {
                ua[g % 3][q] = *(const uvec*)(Ub + (2 * g + q) * 64 * KS);
#pragma unroll
                for (int ks = 0; ks < KS; ks++) vb[g % 3][q][ks] = Vb[(2 * g + q) * V_P + ks * 64];
            }
        };
        ldops(0); ldops(1);
        // Eight slices of 2 KS matrix instructions (positions 2 g, 2 g + 1), operands requested two slices ahead; the next chunk's transform and the window loads of the
        // chunk after it are dealt over the slices by hand and run in the matrix pipe's shadow (left alone the scheduler puts all ~200 vector instructions in front of
        // 64 back-to-back matrix instructions, and fetches every operand pair right before its use)
#pragma unroll
        for (int g = 0; g < 8; g++) {
            __builtin_amdgcn_sched_barrier(0);
            if (g + 2 < 8) ldops(g + 2);
            if (g == 0) { xf_rows(0); xf_out(0, 0, nb); }
            if (g == 1) { xf_out(0, 1, nb); xf_out(0, 2, nb); load_rows(c + 2, 0, 0); }
            if (g == 2) { xf_out(0, 3, nb); load_rows(c + 2, 0, 2); }
            // (the copy of the next U block goes out only after the last read of window registers loaded an iteration ago: the wait the compiler puts in front of that
            //  read is vmcnt(0) and would otherwise wait for the copy as well.  Past the last chunk the copy repeats the last block into the buffer nobody reads again
            //  and the window loads return zeros: channel >= Cin)
            if (g == 3) { xf_rows(1); xf_out(1, 0, nb); issue_u(min(c + 1, A.nchunk - 1), nb); }
            if (g == 4) { xf_out(1, 1, nb); xf_out(1, 2, nb); load_rows(c + 2, 1, 0); }
            if (g == 5) { xf_out(1, 3, nb); load_rows(c + 2, 1, 2); }
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                acc[2 * g] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[g % 3][0][ks], vb[g % 3][0][ks], acc[2 * g], 0, 0, 0);
                acc[2 * g + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[g % 3][1][ks], vb[g % 3][1][ks], acc[2 * g + 1], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 2 * KS; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // (the last, unused copy and loads)
    // ---- inverse transform Y = A^T M A, bias, activation, store.  D[i][j]: register r of a lane = output channel 8 (r / 4) + 4 (lane >> 5) + (r & 3), tile lane & 31.
    const int gt = tile0 + tw * 32 + (lane & 31);
    if (gt >= A.T) return;
    const int n = gt / tpi, rem = gt - n * tpi, ty = rem / A.Wt, tx = rem - ty * A.Wt;
    const bool row1 = 2 * ty + 1 < A.H, col1 = 2 * tx + 1 < A.W;
    const int cob = (cg * CW + cw) * 32 + 4 * (lane >> 5);
    float* yb = A.y + ((size_t)n * A.Cout * A.H + 2 * ty) * A.W + 2 * tx;
    float bv[16];
#pragma unroll
    for (int r = 0; r < 16; r++) { const int co = cob + 8 * (r >> 2) + (r & 3); bv[r] = (A.bias && co < A.Cout) ? A.bias[co] : 0.f; }
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int co = cob + 8 * (r >> 2) + (r & 3);
        float t0[4], t1[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { t0[j] = acc[j][r] + acc[4 + j][r] + acc[8 + j][r]; t1[j] = acc[4 + j][r] - acc[8 + j][r] - acc[12 + j][r]; }
        float y00 = t0[0] + t0[1] + t0[2] + bv[r], y01 = t0[1] - t0[2] - t0[3] + bv[r], y10 = t1[0] + t1[1] + t1[2] + bv[r], y11 = t1[1] - t1[2] - t1[3] + bv[r];
        y00 = y00 > 0.f ? y00 : y00 * A.slope; y01 = y01 > 0.f ? y01 : y01 * A.slope; y10 = y10 > 0.f ? y10 : y10 * A.slope; y11 = y11 > 0.f ? y11 : y11 * A.slope;
        if (co < A.Cout) {
            float* yp = yb + (size_t)co * hw;
            if (A.vec2) {                                                // W even and y 8-byte aligned: a tile row is one 8-byte store
                *(f32x2*)yp = f32x2{y00, y01};
                if (row1) *(f32x2*)(yp + A.W) = f32x2{y10, y11};
            } else {
                yp[0] = y00; if (col1) yp[1] = y01;
                if (row1) { yp[A.W] = y10; if (col1) yp[A.W + 1] = y11; }
            }
        }
    }
}

// KC of the configuration that serves `cout` output channels: 2 x 2 waves (64 channels per workgroup, 8-channel chunks) unless a 64-channel grouping would compute
// 32 or more padded channels (cout = 32, 96, ...), then 1 x 4 (32 channels per workgroup, 4-channel chunks)
inline int wn_kc(int cout) { return (((cout + 63) / 64) * 64 - cout) < 32 ? 8 : 4; }
inline int wn_cout_pad(int cout) { return wn_kc(cout) == 8 ? ((cout + 63) / 64) * 64 : ((cout + 31) / 32) * 32; }
inline int wn_cin_pad(int cin, int kc) { return ((cin + kc - 1) / kc) * kc; }
}  // namespace

extern "C" {

/* 1 when vido_wino3x3_bias_act takes the layer: at least 8 input and 32 output channels (fewer would be mostly padding), a map of at least 2 x 2, and input / output
 * tensors below 1 GB per image (the whole batch must stay below 1 GB as well: the kernel addresses the input through one buffer descriptor). */
int vido_wino3x3_supported(int cin, int cout, int h, int w)
{
    return cin >= 8 && cout >= 32 && h >= 2 && w >= 2 && 4ll * cin * h * w < (1ll << 30) && 4ll * cout * h * w < (1ll << 30);
}

/* 1 when the launch would put at least `min_wgs` workgroups on the chip (default 128: half the CUs).  A workgroup walks ALL input channels of its 64 tiles x 64 (32) output
 * channels — about 2.7 us per 8 channels — so a small map (FPN P4 / P5, the flow network's levels 4-6) is a few dozen workgroups that each run as long as a full chip's
 * worth of them would: measured, the library's kernels win there (profiles/r4/wino_microbench.txt), and the callers keep them. */
int vido_wino3x3_fills_chip(int n, int cout, int h, int w, int min_wgs)
{
    if (n < 1 || cout < 1 || h < 1 || w < 1) return 0;
    const int kc = wn_kc(cout);
    const long long T = (long long)n * ((h + 1) / 2) * ((w + 1) / 2), tb = kc == 8 ? 64 : 128, cg = kc == 8 ? (cout + 63) / 64 : (cout + 31) / 32;
    return ((T + tb - 1) / tb) * cg >= (min_wgs > 0 ? min_wgs : 128);
}

/* floats of the packed transformed weight of a cin -> cout layer */
long long vido_wino3x3_packed_floats(int cin, int cout)
{
    if (cin < 1 || cout < 1) return 0;
    return 16ll * wn_cout_pad(cout) * wn_cin_pad(cin, wn_kc(cout));
}

/* HOST: weight [cout][cin][3][3] f32 -> U = G g G^T (float64 arithmetic, rounded once) in the operand order of k_wino3x3:
 * element (position p = 4 i + j, output channel co, input channel c) at [co / 32][c / KC][p][32 * (c & 1) + co % 32][(c % KC) / 2], KC = 8 (4 when
 * cout rounds up to 64 with >= 32 padded channels); padded channels are zero. */
int vido_wino3x3_pack(const float* w, int cin, int cout, float* up)
{
    if (!w || !up || cin < 1 || cout < 1) return VIDO_E_INVALID;
    const int kc = wn_kc(cout), ks = kc / 2, cop = wn_cout_pad(cout), cip = wn_cin_pad(cin, kc), nchunk = cip / kc;
    std::memset(up, 0, sizeof(float) * (size_t)vido_wino3x3_packed_floats(cin, cout));
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    for (int co = 0; co < cout; co++)
        for (int c = 0; c < cin; c++) {
            const float* g = w + ((size_t)co * cin + c) * 9;
            double t[4][3], u[4][4];
            for (int i = 0; i < 4; i++) for (int b = 0; b < 3; b++) t[i][b] = G[i][0] * g[b] + G[i][1] * g[3 + b] + G[i][2] * g[6 + b];
            for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) u[i][j] = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
            const size_t base = ((size_t)(co / 32) * nchunk + c / kc) * (16 * 64 * ks);
            const int slot = 32 * (c & 1) + co % 32, kk = (c % kc) / 2;
            for (int p = 0; p < 16; p++) up[base + ((size_t)p * 64 + slot) * ks + kk] = (float)u[p >> 2][p & 3];
        }
    (void)cop;
    return VIDO_OK;
}

/* y = leaky_relu(conv2d(x, w, padding 1) + bias, slope) for n images, 3x3 kernel, stride 1: x [n][cin][h][w], y [n][cout][h][w] f32 DEVICE tensors (y != x),
 * bias [cout] or NULL, u_packed = vido_wino3x3_pack(w) on the device (16-byte aligned).  slope 0 = ReLU, 1 = none.  Winograd F(2x2, 3x3) in fp32: the result differs
 * from a direct fp32 convolution by rounding only (~1e-6 of the output scale, the class of the library's own Winograd kernels).  Enqueues on the adopted stream; capturable. */
int vido_wino3x3_bias_act(vido_ctx* ctx, const float* x, const float* u_packed, const float* bias, float* y, int n, int cin, int cout, int h, int w, float slope)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!x || !u_packed || !y || x == y || n < 1 || !vido_wino3x3_supported(cin, cout, h, w) || ((uintptr_t)u_packed & 15) || (((uintptr_t)x | (uintptr_t)y) & 3))
        return vido_set_error(ctx, VIDO_E_INVALID, "wino3x3: no kernel for %d -> %d channels on %d x %d x %d (or a pointer is misaligned)", cin, cout, n, h, w);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    const int kc = wn_kc(cout), ht = (h + 1) / 2, wt = (w + 1) / 2;
    const long long T = (long long)n * ht * wt;
    if (T >= (1ll << 30) || 4ll * n * cin * h * w >= (1ll << 30) || 4ll * n * cout * h * w >= (1ll << 32))
        return vido_set_error(ctx, VIDO_E_INVALID, "wino3x3: a batch of %d images of %d x %d x %d is past the 1 GB the kernel addresses", n, cin, h, w);
    const int tb = kc == 8 ? 64 : 128, cgroups = kc == 8 ? (cout + 63) / 64 : (cout + 31) / 32;
    const int nblk = (int)((T + tb - 1) / tb), total = nblk * cgroups;
    WnArgs A{x, u_packed, bias, y, n, cin, cout, h, w, ht, wt, (int)T, wn_cin_pad(cin, kc) / kc, cgroups, total, slope, (w % 2 == 0 && ((uintptr_t)y & 7) == 0) ? 1 : 0, (unsigned)(4ll * n * cin * h * w), (unsigned)(4ll * vido_wino3x3_packed_floats(cin, cout))};
    const dim3 grid(8 * ((total + 7) / 8)), blk(256);
    constexpr size_t LDS8 = (size_t)2 * (2 * 16 * 64 * 4 + 16 * 2 * 4 * 64) * 4, LDS4 = (size_t)2 * (1 * 16 * 64 * 2 + 16 * 4 * 2 * 64) * 4;
    static bool attr[64] = {};
    if (!attr[ctx->device & 63]) {
        HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_wino3x3<2, 2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS8));
        HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_wino3x3<1, 4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS4));
        attr[ctx->device & 63] = true;
    }
    if (kc == 8) hipLaunchKernelGGL((k_wino3x3<2, 2, 8>), grid, blk, LDS8, st, A);
    else hipLaunchKernelGGL((k_wino3x3<1, 4, 4>), grid, blk, LDS4, st, A);
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

}  // extern "C"
