// Fully connected layer  y[r][o] = act(sum_k x[r][k] w[o][k] + b[o])  in split-fp16 arithmetic on the matrix pipe, split over K (gfx950, round 6).
//
// What it replaces: the box head's fc6 (maskrcnn_benchmark/modeling/roi_heads/box_head/roi_box_feature_extractors.py:50-81: 1000 proposals x 12544 pooled features -> 1024),
// which the library runs as one fp32 GEMM at 125 TFLOP/s (208 us, the largest library launch left in the detector).  The arithmetic is csrc/conv1x1.hip's split-fp16 form
// (two fp16 planes per fp32 operand, three products on v_mfma_f32_32x32x16_f16, fp32 accumulators in two sets, per-output power-of-two weight scales, range flag); the
// weights use the same packing (pack_conv1x1 layout 3 of w viewed as [O][K][1][1]).  What differs from the convolution kernels is the activation operand: x is K-CONTIGUOUS
// ([r][k]), which is the matrix instruction's own B order — a lane's eight k of one row are 32 contiguous bytes — so a k-step of 128 rows is eight 1 KB copy pieces of
// 16 rows x 64 bytes and a lane's operand two 16-byte LDS reads (the convolutions read eight single dwords).
// A 128 x 128 output tile per workgroup would give 64 workgroups; the contraction is therefore SPLIT: workgroup (s, row tile, output tile) walks K / S input features and
// leaves its scaled partial sums in part[s][r][o]; k_fc_h_reduce adds the S partials in a fixed order, the bias and the activation.
#include "common.hpp"

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define FC_SLOT 16384        // one k-step: 8 pieces of weight planes (4 row blocks x 2 planes), 8 pieces of activations (16 rows x 64 bytes each)
#define FC_RB 5              // ring slots
#define FC_PW 4              // copy pieces per wave and step
#define FC_OOB 0x40000000u

struct FcArgs { const float* x; const void* wp; float* part; int R, K, O, mt, ntiles, splitk, nb, ns, total; unsigned xbytes, wbytes; unsigned* range_flag; };

__global__ __launch_bounds__(256, 2) void k_fc_h(FcArgs A)
{
    extern __shared__ __attribute__((aligned(16))) char fc_lds[];
    char* L = fc_lds;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int per = gridDim.x >> 3, item = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (item >= A.total) return;
    const int mtile = item % A.mt, rest = item / A.mt, nt = rest % A.ntiles, s = rest / A.ntiles, m0 = mtile * 128, r0 = nt * 128, t0 = s * A.nb, nb = A.nb;
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)A.wp, 0, A.wbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)A.x, 0, A.xbytes, 0x00020000);
    // copy pieces of a k-step: wave w moves A pieces w, w + 4 (row block i / 2, plane i % 2) and B pieces w, w + 4 (rows 16 i .. 16 i + 15 of the tile: lane = (row, 16-byte chunk))
    const unsigned avo = 16u * (unsigned)lane;
    unsigned abase[2], bvo[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int i = w + 4 * q, rb = i >> 1, pl = i & 1;
        abase[q] = 1024u * (unsigned)(((m0 >> 5) + rb) * A.ns * 2 + pl);
        const int row = r0 + 16 * i + (lane >> 2);
        bvo[q] = row < A.R ? 4u * (unsigned)row * (unsigned)A.K + 16u * (unsigned)(lane & 3) : FC_OOB;      // (a row past R: an offset past the descriptor's range -> zeros)
    }
    auto issue = [&](int T, int slot) {                                   // the four pieces of step t0 + T
        char* S = L + slot * FC_SLOT + w * 1024;
        const unsigned t = (unsigned)(t0 + T);
#pragma unroll
        for (int q = 0; q < 2; q++) __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (__attribute__((address_space(3))) void*)(S + q * 4096), 16, avo, abase[q] + 2048u * t, 0, 0);
#pragma unroll
        for (int q = 0; q < 2; q++) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(S + 8192 + q * 4096), 16, bvo[q], 64u * t, 0, 0);
    };
    f32x16 acc[4], acl[4];
#pragma unroll
    for (int rb = 0; rb < 4; rb++)
#pragma unroll
        for (int r = 0; r < 16; r++) { acc[rb][r] = 0.f; acl[rb][r] = 0.f; }
    for (int T = 0; T < FC_RB - 1; T++) issue(min(T, nb - 1), T);
    u32x4 a[2][4][2], bp[2][2]; f32x4 braw[2];
    float xmax = 0.f;
    const f32x2 k2048 = {2048.f, 2048.f};
    typedef const __attribute__((address_space(3))) char* lds_c;
    const unsigned a_lane = 16u * (unsigned)lane, b_lane = 8192u + 64u * (unsigned)(32 * w + (lane & 31)) + 32u * (unsigned)(lane >> 5);
    auto lda = [&](int slot, int buf) {
        lds_c Ab = (lds_c)(L + slot * FC_SLOT) + a_lane;
#pragma unroll
        for (int rb = 0; rb < 4; rb++)
#pragma unroll
            for (int pl = 0; pl < 2; pl++) a[buf][rb][pl] = *(const __attribute__((address_space(3))) u32x4*)(Ab + (rb * 2 + pl) * 1024);
    };
    auto ldb = [&](int slot) {
        lds_c Bb = (lds_c)(L + slot * FC_SLOT) + b_lane;
        braw[0] = *(const volatile __attribute__((address_space(3))) f32x4*)(Bb); braw[1] = *(const volatile __attribute__((address_space(3))) f32x4*)(Bb + 16);
    };
    auto split = [&](int buf) {
#pragma unroll
        for (int pr = 0; pr < 4; pr++) {
            const f32x2 v = {braw[pr >> 1][2 * (pr & 1)], braw[pr >> 1][2 * (pr & 1) + 1]};
            const f16x2 h = __builtin_convertvector(v, f16x2);
            f32x2 vs, r;
            asm("v_pk_mul_f32 %0, %1, %2" : "=v"(vs) : "v"(v), "v"(k2048));
            asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r.x) : "v"(h), "s"(-2048.f), "v"(vs.x));
            asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r.y) : "v"(h), "s"(-2048.f), "v"(vs.y));
            const f16x2 l = __builtin_convertvector(r, f16x2);
            bp[buf][0][pr] = __builtin_bit_cast(unsigned, h); bp[buf][1][pr] = __builtin_bit_cast(unsigned, l);
            xmax = __builtin_fmaxf(__builtin_fmaxf(xmax, __builtin_fabsf(v.x)), __builtin_fabsf(v.y));
        }
    };
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(FC_PW * (FC_RB - 2)) : "memory");      // slot 0 has landed
    lda(0, 0); ldb(0); split(0);
    int slot = 0, ti = min(FC_RB - 2, nb - 1);
    for (int T2 = 0; T2 < nb; T2 += 2) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(FC_PW * (FC_RB - 3)) : "memory");      // slot T + 1 has landed; everybody is done reading slot T - 1
            __builtin_amdgcn_sched_barrier(0);
            const int sn = slot + 1 == FC_RB ? 0 : slot + 1, sf = slot == 0 ? FC_RB - 1 : slot - 1;
            ti = min(ti + 1, nb - 1);
            ldb(sn); lda(sn, h ^ 1);
            issue(ti, sf);
#pragma unroll
            for (int term = 0; term < 3; term++) {
                constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};
#pragma unroll
                for (int rb = 0; rb < 4; rb++) {
                    f32x16& d = term == 2 ? acc[rb] : acl[rb];
                    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[h][rb][PA[term]]), __builtin_bit_cast(f16x8, bp[h][PB[term]]), d, 0, 0, 0);
                }
            }
            split(h ^ 1);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
#pragma unroll
            for (int i = 0; i < 12; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i < 5) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                if (i >= 1 && i < 5) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if (i >= 4) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            slot = sn;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (!(xmax < 65504.f) && A.range_flag) atomicOr(A.range_flag, 1u);
    // scaled partial sums: register r of a lane = output 8 (r / 4) + 4 (lane >> 5) + (r & 3) of the row block, row lane & 31 -> four consecutive outputs per 16-byte store
    const float* wsc = (const float*)((const char*)A.wp + (size_t)4 * A.K * A.O);
    const int row = r0 + 32 * w + (lane & 31);
#pragma unroll
    for (int rb = 0; rb < 4; rb++) {
        f32x4 sv[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) sv[g4] = *(const f32x4*)(wsc + m0 + 32 * rb + 8 * g4 + 4 * (lane >> 5));
        if (row < A.R) {
            float* dst = A.part + ((size_t)s * A.R + row) * A.O + m0 + 32 * rb + 4 * (lane >> 5);
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = (acc[rb][4 * g4 + e] + acl[rb][4 * g4 + e] * 0x1p-11f) * sv[g4][e];
                *(f32x4*)(dst + 8 * g4) = v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_fc_h_reduce(const float* __restrict__ part, const float* __restrict__ bias, float* __restrict__ y, int R, int O, int S, float slope)
{
    const size_t n4 = (size_t)R * O / 4, i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 v = ((const f32x4*)part)[i];
    for (int s = 1; s < S; s++) v += ((const f32x4*)part)[(size_t)s * n4 + i];      // fixed order
    const int o = (int)((i * 4) % (size_t)O);
    if (bias) v += *(const f32x4*)(bias + o);
#pragma unroll
    for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], v[e] * slope);
    ((f32x4*)y)[i] = v;
}
}  // namespace

extern "C" {

/* The K-split vido_fc_h uses for a shape, 0 when it does not take it: outputs a multiple of 128, input features a multiple of 32 S (an even number of 16-feature steps per
 * split), x below 1 GB.  S is the smallest of 1, 2, 4, 8 that gives >= 256 workgroups (or the largest that divides). */
int vido_fc_h_splitk(int rows, int k, int outs)
{
    if (rows < 1 || outs < 128 || outs % 128 || k < 32 || k % 32 || 4ll * rows * k >= (1ll << 30) || 4ll * k * outs >= (1ll << 31)) return 0;
    const int tiles = (outs / 128) * ((rows + 127) / 128);
    int best = 1;
    for (int s = 1; s <= 8; s *= 2) { if (k % (32 * s)) break; best = s; if (tiles * s >= 256) break; }
    return best;
}

/* y[rows][outs] = leaky_relu(x[rows][k] w[outs][k]^T + bias, slope): x, y, bias f32 DEVICE (16-byte aligned); w_packed = pack_conv1x1(w viewed as [outs][k][1][1], layout 3)
 * (vido_slam_amd/nets/ops.py); part: scratch of vido_fc_h_splitk(rows, k, outs) x rows x outs floats (the caller's: a launch inside a stream capture cannot allocate).
 * Activations must stay below 65504 in magnitude (vido_conv1x1_range_flag otherwise).  Enqueues two launches on the adopted stream; capturable. */
int vido_fc_h(vido_ctx* ctx, const float* x, const void* w_packed, const float* bias, float* part, float* y, int rows, int k, int outs, float slope)
{
    if (!ctx) return VIDO_E_INVALID;
    const int S = vido_fc_h_splitk(rows, k, outs);
    if (!x || !w_packed || !part || !y || !S || slope < 0.f || slope > 1.f || (((uintptr_t)x | (uintptr_t)y | (uintptr_t)part | (uintptr_t)w_packed | (uintptr_t)bias) & 15))
        return vido_set_error(ctx, VIDO_E_INVALID, "fc_h: no kernel for %d rows x %d -> %d (or a pointer is misaligned, or slope outside [0, 1])", rows, k, outs);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    const int mt = outs / 128, ntiles = (rows + 127) / 128, ns = k / 16, nb = ns / S, total = mt * ntiles * S;
    FcArgs A{x, w_packed, part, rows, k, outs, mt, ntiles, S, nb, ns, total, (unsigned)(4ll * rows * k), (unsigned)(4ll * k * outs), ctx->c1_range_flag};
    static bool attr[64] = {};
    if (!attr[ctx->device & 63]) { HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_fc_h, hipFuncAttributeMaxDynamicSharedMemorySize, FC_RB * FC_SLOT)); attr[ctx->device & 63] = true; }
    hipLaunchKernelGGL(k_fc_h, dim3(8 * ((total + 7) / 8)), dim3(256), FC_RB * FC_SLOT, st, A);
    const size_t n4 = (size_t)rows * outs / 4;
    hipLaunchKernelGGL(k_fc_h_reduce, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, (const float*)part, bias, y, rows, outs, S, slope);
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

}  // extern "C"
