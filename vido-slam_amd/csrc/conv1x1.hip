// 1x1 convolution + bias (+ residual) + activation of the detector's bottlenecks as ONE fp32 matrix-core GEMM launch (gfx950).
//
// What it replaces: conv1 / conv3 of BottleneckWithFixedBatchNorm and the FPN's lateral convolutions (maskrcnn_benchmark/modeling/backbone/resnet.py:300-372,
// backbone/fpn.py) at batch 1: 69 convolutions per frame of X-101-32x8d, 7.1 GFLOP each, which the library runs as rocBLAS GEMMs (MT64x64x16 ... MT128x128x32: 61-77 us
// per call, 92-116 TFLOP/s, 62-67 % matrix-pipe occupancy) followed — for conv3 — by a separate bias + residual + ReLU pass over the output (33 launches, 0.33 ms).
//
// Formulation.  NCHW at batch 1 IS the GEMM  Out[co][q] = sum_k W[co][k] In[k][q]  with the flattened position q contiguous: M = Cout, N = H*W, K = Cin.
//   * v_mfma_f32_32x32x2f32 with A = 32 output channels x 2 input channels, B = the same 2 channels x 32 positions: the D layout then keeps, in one register, one output
//     channel of 32 consecutive positions per half wave — stores (and the residual reads) are whole 128-byte lines.
//   * workgroup = 8 waves = 128 output channels x 128 positions, a wave 32 x 64 (two tiles, 32 accumulator registers): two waves per SIMD inside ONE workgroup, so the
//     waits of one wave (barrier, LDS round trips, the prologue of a K chunk) run under the matrix instructions of the other even when a layer has fewer tiles than the chip
//     has CUs (1024 -> 1024 at 50 x 68: 216 tiles).  The first version (4 waves, 64 x 64 per wave, one wave per SIMD) reached 79-93 TFLOP/s, below the library's 94-108.
//   * A never touches LDS: the weights are packed on the host in OPERAND order ([32-channel block][K chunk][lane][16]): a lane's 16 operands of a K chunk of 32 are four
//     consecutive 16-byte loads, a wave's load is 4 KB contiguous, the matrix stays in L2 / MALL (<= 16 MB per layer); requested one chunk ahead.
//   * B (the activations) goes global -> LDS with the asynchronous copy, 16 bytes per lane, two rows of the [32][128] chunk per instruction; two LDS buffers, the next
//     chunk's copies are issued right after the barrier that frees their buffer, ONE barrier per chunk of 64 matrix instructions per wave.  B operands are plain
//     ds_read_b32 of 32 consecutive floats per half wave.
//   * epilogue in registers: + bias[co], + residual[co][q], leaky-ReLU(slope), store.
// Work items (position tile, channel tile) are dealt so that an XCD walks a contiguous range with the channel tile fastest: the Cout / 128 workgroups that share a B tile
// run back to back on one L2.
#include "common.hpp"

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define C1_KC 32           // K chunk: 16 k-pairs = 64 matrix instructions per wave and barrier
#define C1_TM 128
#define C1_TN 128

struct C1Args { const float* x; const float* wp; const float* bias; const float* res; float* y; int M, N, K, mt, total; float slope; };

__device__ __forceinline__ void c1_glds16(const float* g, float* l)      // four dwords per lane: LDS address = l + lane * 16 (both sides 16-byte aligned)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// NSUB: 32-channel sub-chunks per barrier (1: 32 KB of LDS; 2: 64 KB, half the barriers — for layers with about one tile per CU, where nothing else hides them)
template <bool RES, int NSUB>
__global__ __launch_bounds__(512) void k_conv1x1(C1Args A)
{
    extern __shared__ __attribute__((aligned(16))) float c1_lds[];             // [2][NSUB * 32][128]
    constexpr int KC = NSUB * C1_KC, BUF = KC * C1_TN;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv & 3, wn = wv >> 2;      // 8 waves: row block wm of the workgroup's four, 64-position half wn
    // contiguous item range per XCD (the hardware deals consecutive workgroup ids round-robin over the 8 XCDs)
    const int per = gridDim.x >> 3, item = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (item >= A.total) return;
    const int nt = item / A.mt, mtile = item - nt * A.mt, n0 = nt * C1_TN, m0 = mtile * C1_TM;
    const int nsub = A.K / C1_KC, nchunk = (nsub + NSUB - 1) / NSUB;
    // ---- B copies: wave wv moves rows [4 wv, 4 wv + 4) of every 32-row sub-chunk, two rows per instruction (lanes 0..31: row r, lanes 32..63: row r + 1; 16 bytes = 4
    // positions each).  Positions past the end of the image are clamped to its last 16 bytes: those columns are computed and never stored.
    const int bcol = min(n0 + 4 * (lane & 31), A.N - 4);
    const float* bsrc = A.x + (size_t)(4 * wv + (lane >> 5)) * A.N + bcol;
    // ---- A operands of row block wm: 16 per lane and sub-chunk, four consecutive 16-byte loads
    const f32x4* ap = (const f32x4*)(A.wp + ((size_t)((m0 >> 5) + wm) * nsub * 64 + lane) * (C1_KC / 2));
    constexpr int A_STRIDE = 64 * (C1_KC / 2) / 4;               // f32x4 per (row block, sub-chunk)
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }
    f32x4 an[4 * NSUB];
#define C1_ISSUE(c, buf) { \
        _Pragma("unroll") for (int u = 0; u < NSUB; u++) { const int sc = (c) * NSUB + u; if (sc < nsub) { \
            float* dst = c1_lds + (buf) * BUF + (u * C1_KC + 4 * wv) * C1_TN; const float* src = bsrc + (size_t)sc * C1_KC * A.N; \
            c1_glds16(src, dst); c1_glds16(src + (size_t)2 * A.N, dst + 2 * C1_TN); \
            _Pragma("unroll") for (int i = 0; i < 4; i++) an[4 * u + i] = ap[(size_t)sc * A_STRIDE + i]; } } }
    C1_ISSUE(0, 0)
    const int bo = (lane >> 5) * C1_TN + 64 * wn + (lane & 31);
    for (int c = 0; c < nchunk; c++) {
        f32x4 a[4 * NSUB];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4 * NSUB; i++) a[i] = an[i];
        __syncthreads();                                              // chunk c has landed in buffer c & 1; everybody is done with buffer (c + 1) & 1
        if (c + 1 < nchunk) C1_ISSUE(c + 1, (c + 1) & 1)
        const float* Bc = c1_lds + (c & 1) * BUF + bo;
        // B operands are requested TWO k-pairs ahead, between the two matrix instructions of a pair: a wait then never meets a read that has just been issued (the
        // compiler's waits are lgkmcnt(0): with the request right in front of them every second pair paid a full LDS round trip)
        float b0 = Bc[0], b1 = Bc[32], c0 = Bc[2 * C1_TN], c1 = Bc[2 * C1_TN + 32];
#pragma unroll
        for (int kp = 0; kp < KC / 2; kp++) {
            const float av = a[kp >> 2][kp & 3];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            float d0 = 0.f, d1 = 0.f;
            if (kp + 2 < KC / 2) { d0 = Bc[2 * (kp + 2) * C1_TN]; d1 = Bc[2 * (kp + 2) * C1_TN + 32]; }
            __builtin_amdgcn_sched_barrier(0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc1, 0, 0, 0);
            b0 = c0; b1 = c1; c0 = d0; c1 = d1;
        }
    }
#undef C1_ISSUE
    // D[i][j]: lane = 32 * ((i / 4) & 1) + j, register = 4 * (i / 8) + (i & 3)  ->  register r of a lane = output channel 8 (r / 4) + 4 (lane >> 5) + (r & 3), position lane & 31.
    // All residual / bias loads of a tile are issued before the first store (loads behind a store to memory that may alias them are not reordered by the compiler: the first
    // version ran load -> add -> store 64 times in a chain and the epilogue cost as much as a third of the GEMM).
    const int co0 = m0 + 32 * wm + 4 * (lane >> 5);
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const f32x16& acc = t == 0 ? acc0 : acc1;
        const int q = n0 + 64 * wn + 32 * t + (lane & 31);
        if (q < A.N) {
            float rv[16], bv[16];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int co = co0 + 8 * (r >> 2) + (r & 3);
                bv[r] = A.bias ? A.bias[co] : 0.f;
                rv[r] = RES ? A.res[(size_t)co * A.N + q] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int co = co0 + 8 * (r >> 2) + (r & 3);
                float v = acc[r] + bv[r] + rv[r];
                v = v > 0.f ? v : v * A.slope;
                A.y[(size_t)co * A.N + q] = v;
            }
        }
    }
}
}  // namespace

extern "C" {

/* 1 when vido_conv1x1_bias_act takes the shape: output channels a multiple of 128, input channels a multiple of 32, H*W a multiple of 4 and >= 128 (16-byte copies). */
int vido_conv1x1_supported(int cin, int cout, int hw)
{
    return cin >= C1_KC && cin % C1_KC == 0 && cout >= C1_TM && cout % C1_TM == 0 && hw >= C1_TN && hw % 4 == 0;
}

/* y = leaky_relu(conv2d(x, w) + bias + residual, slope) for one image, 1x1 kernel, stride 1: x [cin][hw], y / residual [cout][hw] f32 DEVICE tensors (16-byte aligned,
 * y != x), bias [cout] or NULL, residual NULL when there is none.  w_packed: the weight [cout][cin] in operand order, element (co, k) at
 * [co / 32][k / 32][32 * (k & 1) + co % 32][(k % 32) / 2]  (vido_slam_amd/nets/ops.py::pack_conv1x1).  slope 0 = ReLU, 1 = none.  Enqueues on the adopted stream; capturable. */
int vido_conv1x1_bias_act(vido_ctx* ctx, const float* x, const float* w_packed, const float* bias, const float* residual, float* y, int cin, int cout, int hw, float slope)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!x || !w_packed || !y || x == y || !vido_conv1x1_supported(cin, cout, hw) || (((uintptr_t)x | (uintptr_t)y | (uintptr_t)w_packed | (uintptr_t)residual) & 15))
        return vido_set_error(ctx, VIDO_E_INVALID, "conv1x1: no kernel for %d -> %d channels at %d positions (or a pointer is not 16-byte aligned)", cin, cout, hw);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    const int mt = cout / C1_TM, ntl = (hw + C1_TN - 1) / C1_TN, total = mt * ntl;
    C1Args A{x, w_packed, bias, residual, y, cout, hw, cin, mt, total, slope};
    const dim3 grid(8 * ((total + 7) / 8)), blk(512);
    static const int force_nsub = [] { const char* e = getenv("VIDO_CONV1X1_NSUB"); return e ? atoi(e) : 0; }();
    // layers with at most ~1.5 tiles per CU (1024 -> 1024 at 50 x 68: 216 tiles) take 64-channel chunks: nothing else on the CU hides their barriers
    const int nsub = cin % (2 * C1_KC) ? 1 : (force_nsub == 1 || force_nsub == 2 ? force_nsub : ((total <= 384 && cin >= 128) ? 2 : 1));      // (the 64-channel form takes whole chunks only)
    static bool attr[64] = {};
    if (!attr[ctx->device & 63]) {
        for (const void* f : {(const void*)k_conv1x1<true, 2>, (const void*)k_conv1x1<false, 2>, (const void*)k_conv1x1<true, 1>, (const void*)k_conv1x1<false, 1>})
            HIP_TRY(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * C1_KC * C1_TN * 4));
        attr[ctx->device & 63] = true;
    }
    const size_t lds = (size_t)2 * nsub * C1_KC * C1_TN * 4;
    if (nsub == 2) { if (residual) hipLaunchKernelGGL((k_conv1x1<true, 2>), grid, blk, lds, st, A); else hipLaunchKernelGGL((k_conv1x1<false, 2>), grid, blk, lds, st, A); }
    else { if (residual) hipLaunchKernelGGL((k_conv1x1<true, 1>), grid, blk, lds, st, A); else hipLaunchKernelGGL((k_conv1x1<false, 1>), grid, blk, lds, st, A); }
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

}  // extern "C"
