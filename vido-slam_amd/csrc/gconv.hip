// Grouped 3x3 convolution + bias + ReLU of the detector's ResNeXt bottlenecks on the fp32 matrix cores (gfx950).
//
// What it replaces: `conv2` of BottleneckWithFixedBatchNorm (maskrcnn_benchmark/modeling/backbone/resnet.py:300-372: Conv2d(width, width, 3, stride, 1, groups=32)
// + FrozenBatchNorm2d + relu_) at batch 1 — 33 convolutions per frame of X-101-32x8d, 2.0 GFLOP each, which the library runs as Winograd kernels at 107 / 50 / 30 / 29 us
// per call (8 / 16 / 32 / 64 channels per group) followed by a separate bias+ReLU pass over the output.
//
// Formulation.  With the input rows zero-padded to Wp = W + 2 columns and FLATTENED (q = y * Wp + x), the nine taps of output position q are the inputs at q + dy * Wp + dx:
// constant offsets.  A tile is therefore 32 (or 16) CONSECUTIVE flattened positions — it may run across a row end; the two pad positions per row are computed and thrown
// away (3 % at W = 68) — and one group is a GEMM  Out[co][q] = sum_k Wt[co][k] In[k][q + off(k)],  k = (input channel, tap):
//   * channels per group >= 32: v_mfma_f32_32x32x2 (A = 32 output channels x 2 input channels of one tap, B = the same 2 channels x 32 positions),
//   * 16 or 8 channels per group: v_mfma_f32_16x16x4 (8: the upper 8 rows of A are zero).
// A workgroup stages a band of input rows of 8 input channels at a time in LDS (each element is read by 9 taps x all output channels of the group from there) together with
// that chunk's weights in operand order; B operands are plain ds_read_b32 of consecutive addresses, A operands one ds_read_b32 at lane * 4.  The accumulators start at the
// folded batch-norm bias and leave through the ReLU: no epilogue pass over the output.
#include "common.hpp"

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GcArgs { const float* x; const float* w; const float* bias; const float* in_bias; float* y; int H, W, cpg_in, cpg_out, R, PS; float slope; int gx, total; unsigned xbytes, wbytes; int nj, ablate; unsigned m_wq, m_wpd, m_gx; };      // m_*: ceil(2^32 / d) for d = (W + 4) / 4, W + 4, gx (k_gconv3x3_m16d: n / d = umulhi(n, m), exact while n d < 2^32)

// Workgroup -> work item.  The hardware deals consecutive workgroup ids round-robin over the 8 XCDs, each with its own L2; neighbouring position chunks of a group share
// two halo rows and the group's weights, so the 1-D grid (8 * ceil(total / 8) workgroups) is folded such that every XCD walks a CONTIGUOUS range of (group, chunk) items:
// with the plain (chunk, group) grid each band of rows was fetched from HBM by two XCDs and the weights of a group by all eight.
__device__ __forceinline__ int gc_item(int total) { const int per = gridDim.x >> 3, L = (blockIdx.x & 7) * per + (blockIdx.x >> 3); return L < total ? L : -1; }

#define GC_KC 8          // input channels per K chunk
#define GC32_PS 512      // floats per input-channel plane in LDS (k_gconv3x3_m32)

// ---- >= 32 channels per group.  Workgroup = 4 waves = 8 tiles of 32 positions (2 per wave) x 32 output channels; work items = (position chunk, output-channel block, group), see gc_item.
// Weights packed [group][co block][chunk][tap][8 ci][32 co].  The next chunk's input rows and weights are requested into registers before the matrix phase of the current one.
__global__ __launch_bounds__(256) void k_gconv3x3_m32(GcArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float gc_lds[];
    float* Wl = gc_lds;                         // [9][8][32]
    float* In = gc_lds + 9 * GC_KC * 32;        // [8][GC32_PS]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int H = A.H, W = A.W, Wpd = W + 2, npos = H * Wpd, RW = A.R * Wpd;
    const int nchunk = A.cpg_in / GC_KC, ncob = A.cpg_out / 32;
    const int item = gc_item(A.total); if (item < 0) return;
    const int chunk = item % A.gx, cob = (item / A.gx) % ncob, g = item / (A.gx * ncob), q0 = chunk * 256, r0 = q0 / Wpd;
    const size_t HW = (size_t)H * W;
    const float* xg = A.x + (size_t)g * A.cpg_in * HW;
    const float* wg = A.w + ((size_t)(g * ncob + cob) * nchunk) * (9 * GC_KC * 32);
    int poff[8];                                // this lane's 8 elements of a channel plane of the band (offset inside the channel image, -1: zero padding)
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int p = lane + 64 * j, rr = p / Wpd, xp = p - rr * Wpd, row = r0 - 1 + rr;
        poff[j] = (p < RW && row >= 0 && row < H && xp >= 1 && xp <= W) ? row * W + xp - 1 : -1;
    }
    float pin0[8], pin1[8], pw[9], pib0 = 0.f, pib1 = 0.f;
#define GC32_PREFETCH(c) { \
        if (A.in_bias) { pib0 = A.in_bias[(size_t)g * A.cpg_in + (c) * GC_KC + 2 * wv]; pib1 = A.in_bias[(size_t)g * A.cpg_in + (c) * GC_KC + 2 * wv + 1]; } \
        const float* xc0 = xg + (size_t)((c) * GC_KC + 2 * wv) * HW; const float* xc1 = xc0 + HW; \
        _Pragma("unroll") for (int j = 0; j < 8; j++) { pin0[j] = poff[j] >= 0 ? xc0[poff[j]] : 0.f; pin1[j] = poff[j] >= 0 ? xc1[poff[j]] : 0.f; } \
        const float* wc = wg + (size_t)(c) * (9 * GC_KC * 32); \
        _Pragma("unroll") for (int i = 0; i < 9; i++) pw[i] = wc[tid + 256 * i]; }
    const int co_base = g * A.cpg_out + cob * 32;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; r++) { const float b = A.bias[co_base + 8 * (r / 4) + 4 * (lane >> 5) + (r & 3)]; acc0[r] = b; acc1[r] = b; }
    const int bbase = (lane >> 5) * GC32_PS + (q0 + 64 * wv - r0 * Wpd) + (lane & 31);      // B operand of tile 2 wv (tile 2 wv + 1: + 32), tap (0, 0), channel pair 0
    GC32_PREFETCH(0)
    for (int c = 0; c < nchunk; c++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float v0 = pin0[j], v1 = pin1[j];
            if (A.in_bias) { v0 = poff[j] >= 0 ? fmaxf(v0 + pib0, 0.f) : 0.f; v1 = poff[j] >= 0 ? fmaxf(v1 + pib1, 0.f) : 0.f; }      // the 1x1 convolution's bias + ReLU, applied on the way in
            In[(2 * wv) * GC32_PS + lane + 64 * j] = v0; In[(2 * wv + 1) * GC32_PS + lane + 64 * j] = v1;
        }
#pragma unroll
        for (int i = 0; i < 9; i++) Wl[tid + 256 * i] = pw[i];
        __syncthreads();
        if (c + 1 < nchunk) GC32_PREFETCH(c + 1)
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int toff = (tap / 3) * Wpd + (tap % 3);
#pragma unroll
            for (int kp = 0; kp < 4; kp++) {
                const float a = Wl[(tap * 8 + 2 * kp) * 32 + lane];
                const float b0 = In[bbase + 2 * kp * GC32_PS + toff], b1 = In[bbase + 32 + 2 * kp * GC32_PS + toff];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
            }
        }
        __syncthreads();
    }
#undef GC32_PREFETCH
    // D[i][j]: lane = 32 * ((i / 4) & 1) + j, register = 4 * (i / 8) + (i & 3)  ->  a register holds one output channel of 32 consecutive positions per half wave
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int q = q0 + 64 * wv + 32 * t + (lane & 31), yy = q / Wpd, xx = q - yy * Wpd;
        if (q < npos && xx < W) {
            float* yo = A.y + (size_t)co_base * HW + (size_t)yy * W + xx;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int co = 8 * (r / 4) + 4 * (lane >> 5) + (r & 3);
                float v = t ? acc1[r] : acc0[r]; v = v > 0.f ? v : v * A.slope;
                yo[(size_t)co * HW] = v;
            }
        }
    }
}

// ---- >= 32 channels per group WITHOUT an input bias (the 1x1 convolution in front applied its own bias + ReLU: csrc/conv1x1.hip), round 4.  Same tiles, items and weight
// packing as k_gconv3x3_m32; what changes is everything around the matrix instructions, because beside an fp32 matrix instruction every vector-ALU instruction costs
// ~5.5 cycles of matrix time (tools/ubench/mfma_fillers*.hip) and the kernel above spends ~150 of them per 72 matrix instructions (staging through registers with
// selects for the padding, 64-bit addresses, LDS address adds):
//   * the band and the weights go global -> LDS by buffer loads with the lds bit — one dword per lane for the band (the LDS side is the padded, flattened row layout: 64
//     consecutive positions per instruction; a pad position carries an out-of-range offset, for which the copy writes 0), 16 bytes per lane for the weights; per-lane
//     offsets are loop invariants, the channel rides in the scalar offset: no vector instruction;
//   * two LDS buffers, ONE barrier per 8-channel chunk, the next chunk's copies issued right behind it;
//   * operands at immediate offsets from three per-tap-row base registers, requested two steps ahead.
__global__ __launch_bounds__(256) void k_gconv3x3_m32d(GcArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float gc_lds[];
    constexpr int WSZ = 9 * GC_KC * 32, BUF = WSZ + GC_KC * GC32_PS;      // floats per buffer: weights [9][8][32], then the band [8][GC32_PS]
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = A.H, W = A.W, Wpd = W + 2, npos = H * Wpd, RW = A.R * Wpd;
    const int nchunk = A.cpg_in / GC_KC, ncob = A.cpg_out / 32;
    const int item = gc_item(A.total); if (item < 0) return;
    const int chunk = item % A.gx, cob = (item / A.gx) % ncob, g = item / (A.gx * ncob), q0 = chunk * 256, r0 = q0 / Wpd;
    const unsigned HW = (unsigned)H * (unsigned)W;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)A.x, 0, A.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)A.w, 0, A.wbytes, 0x00020000);
    unsigned poff[8];                           // byte offset, inside a channel image, of this lane's 8 positions of the band; bit 30: padding / outside the band
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int p = lane + 64 * j, rr = p / Wpd, xp = p - rr * Wpd, row = r0 - 1 + rr;
        poff[j] = (p < RW && row >= 0 && row < H && xp >= 1 && xp <= W) ? 4u * (unsigned)(row * W + xp - 1) : 0x40000000u;
    }
    const unsigned wvo = 16u * (unsigned)lane;
    const unsigned wbase = 4u * (unsigned)(((g * ncob + cob) * nchunk) * WSZ);
    const unsigned xbase = 4u * (unsigned)(g * A.cpg_in) * HW;
    // wave wv copies channels 2 wv, 2 wv + 1 of a chunk (8 instructions each) and pieces wv, wv + 4, wv + 8 of its 9 KB of weights
    auto issue = [&](int c, int buf) {
        float* base = gc_lds + buf * BUF;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const unsigned so = xbase + 4u * (unsigned)(c * GC_KC + 2 * wv + h) * HW;
#pragma unroll
            for (int j = 0; j < 8; j++)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(base + WSZ + (2 * wv + h) * GC32_PS + 64 * j), 4, poff[j], so, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int piece = wv + 4 * q;
            if (piece < 9) __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (__attribute__((address_space(3))) void*)(base + piece * 256), 16, wvo, wbase + 4u * (unsigned)(c * WSZ + piece * 256), 0, 0);
        }
    };
    const int co_base = g * A.cpg_out + cob * 32;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; r++) { const float b = A.bias[co_base + 8 * (r / 4) + 4 * (lane >> 5) + (r & 3)]; acc0[r] = b; acc1[r] = b; }
    typedef const volatile __attribute__((address_space(3))) float* lds_f;      // (volatile: keeps the compiler from pairing neighbouring reads into ds_read2_b32, whose 8-bit
                                                                               //  offsets cost a vector add per read; a ds_read_b32 takes any of these offsets as an immediate)
    const int bbase = WSZ + (lane >> 5) * GC32_PS + (q0 + 64 * wv - r0 * Wpd) + (lane & 31);      // B operand of tile 2 wv (tile 2 wv + 1: + 32), tap (0, 0), channel pair 0
    issue(0, 0);
    for (int c = 0; c < nchunk; c++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                              // chunk c has landed in buffer c & 1; everybody is done with buffer (c + 1) & 1
        if (c + 1 < nchunk) issue(c + 1, (c + 1) & 1);
        lds_f Wb = (lds_f)(gc_lds + (c & 1) * BUF) + lane;
        lds_f B0 = (lds_f)(gc_lds + (c & 1) * BUF) + bbase, B1 = B0 + Wpd, B2 = B1 + Wpd;      // one base per tap row: everything else is an immediate
        float a[3], b0[3], b1[3];
        auto ld = [&](int s) {                                        // step s = tap * 4 + kp
            const int tap = s >> 2, kp = s & 3; lds_f Bt = tap < 3 ? B0 : (tap < 6 ? B1 : B2);
            a[s % 3] = Wb[(tap * 8 + 2 * kp) * 32]; b0[s % 3] = Bt[2 * kp * GC32_PS + tap % 3]; b1[s % 3] = Bt[2 * kp * GC32_PS + tap % 3 + 32];
        };
        ld(0); ld(1);
#pragma unroll
        for (int s = 0; s < 36; s++) {                                // (scheduling barriers: left alone, the scheduler sinks every read to just before its matrix instruction and a lone wave runs at LDS latency)
            if (s + 2 < 36) ld(s + 2);
            __builtin_amdgcn_sched_barrier(0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s % 3], b0[s % 3], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s % 3], b1[s % 3], acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // D[i][j]: lane = 32 * ((i / 4) & 1) + j, register = 4 * (i / 8) + (i & 3)  ->  a register holds one output channel of 32 consecutive positions per half wave
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int q = q0 + 64 * wv + 32 * t + (lane & 31), yy = q / Wpd, xx = q - yy * Wpd;
        if (q < npos && xx < W) {
            float* yo = A.y + (size_t)co_base * HW + (size_t)yy * W + xx;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int co = 8 * (r / 4) + 4 * (lane >> 5) + (r & 3);
                const float v = t ? acc1[r] : acc0[r];
                yo[(size_t)co * HW] = fmaxf(v, v * A.slope);
            }
        }
    }
}

// asynchronous global -> LDS copy of one dword per lane: the LDS address is the wave-uniform `l` + lane * 4, lanes switched off by the caller's branch write nothing
__device__ __forceinline__ void glds4(const float* g, float* l)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 4, 0, 0);
}

__device__ __forceinline__ void glds16(const float* g, float* l)      // four dwords per lane: LDS address = l + lane * 16 (both sides 16-byte aligned)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// ---- 16 (or 8) channels per group.  Workgroup = 4 waves x NTW tiles of 16 positions x 16 output channels; work items = (position chunk, group).  Weights packed
// [group][chunk][tap][8 ci][16 co] (co >= cpg_out: zeros).  The band goes global -> LDS with the asynchronous copy: no staging registers, every copy of a chunk in flight
// at once (the first version copied through registers in a loop the compiler could not batch and spent 100 of its 155 us waiting on one load at a time).  A copy
// instruction costs the same ~60 issue cycles whatever it moves, so when W is a multiple of 4 the rows are laid out with FOUR pad positions on the left (row pitch W + 4:
// the four pads double as the right pad of the row above) and move as 16-byte pieces — 1 KiB per instruction instead of 256 B (V4).  The padding is written once per
// workgroup: the copies only touch positions inside the image.  With an input bias (IB) the operand is max(x + in_bias[ci], 0) — the bias + ReLU pass of the 1x1
// convolution in front, applied where the value is read — and the padding holds -3e38, which that expression turns into the zero the convolution pads with.
// Several workgroups share a CU (<= 67 KB of LDS each), one copies while another multiplies.
template <int NTW, bool IB, bool V4>
__global__ __launch_bounds__(256) void k_gconv3x3_m16(GcArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float gc_lds[];
    float* Wl = gc_lds;                         // [9][8][16]
    float* In = gc_lds + 9 * GC_KC * 16;        // [8][PS]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int L = V4 ? 4 : 1;
    const int H = A.H, W = A.W, Wpd = V4 ? W + 4 : W + 2, npos = H * Wpd, PS = A.PS, R = A.R;
    constexpr int P = NTW * 64;
    const int item = gc_item(A.total); if (item < 0) return;
    const int g = item / A.gx, q0 = (item - g * A.gx) * P, r0 = q0 / Wpd;
    const int nchunk = A.cpg_in / GC_KC;
    const size_t HW = (size_t)H * W;
    const float* xg = A.x + (size_t)g * A.cpg_in * HW;
    const float* wg = A.w + (size_t)g * nchunk * (9 * GC_KC * 16);
    const int co_base = g * A.cpg_out;
    f32x4 acc[NTW];
    {
        float b4[4];
#pragma unroll
        for (int r = 0; r < 4; r++) { const int co = 4 * (lane >> 4) + r; b4[r] = co < A.cpg_out ? A.bias[co_base + co] : 0.f; }
#pragma unroll
        for (int t = 0; t < NTW; t++) { acc[t][0] = b4[0]; acc[t][1] = b4[1]; acc[t][2] = b4[2]; acc[t][3] = b4[3]; }
    }
    const int bbase = (lane >> 4) * PS + (q0 + 16 * NTW * wv - r0 * Wpd) + (lane & 15) + (L - 1);
    {   // padding: the pad columns of every row of the band, whole rows outside the image, the pad after the last row
        const float padv = IB ? -3.0e38f : 0.f;
        for (int s2 = wv; s2 < GC_KC * R; s2 += 4) {
            const int ci = s2 / R, rr = s2 - ci * R, row = r0 - 1 + rr; float* dst = In + ci * PS + rr * Wpd;
            if (row < 0 || row >= H) { for (int xp = lane; xp < Wpd; xp += 64) dst[xp] = padv; }
            else if (lane < L) dst[lane] = padv;
            else if (lane < Wpd - W) dst[W + lane] = padv;
        }
        if (tid < GC_KC * 4) In[(tid >> 2) * PS + R * Wpd + (tid & 3)] = padv;
    }
    __syncthreads();
    for (int c = 0; c < nchunk; c++) {
#pragma unroll
        for (int ci = 0; ci < GC_KC; ci++) {
            const float* xc = xg + (size_t)(c * GC_KC + ci) * HW;
            for (int rr = wv; rr < R; rr += 4) {
                const int row = r0 - 1 + rr;
                if (row < 0 || row >= H) continue;
                const float* xr = xc + (size_t)row * W; float* dst = In + ci * PS + rr * Wpd + L;
                if (V4) { for (int x0 = 0; x0 < W; x0 += 256) { const int x = x0 + 4 * lane; if (x < W) glds16(xr + x, dst + x0); } }
                else { for (int x0 = 0; x0 < W; x0 += 64) { const int x = x0 + lane; if (x < W) glds4(xr + x, dst + x0); } }
            }
        }
        { const float* wc = wg + (size_t)c * (9 * GC_KC * 16); for (int i0 = wv * 64; i0 < 9 * GC_KC * 16; i0 += 256) glds4(wc + i0 + lane, Wl + i0); }
        float ib0 = 0.f, ib1 = 0.f;
        if (IB) { const float* ibp = A.in_bias + (size_t)g * A.cpg_in + c * GC_KC + (lane >> 4); ib0 = ibp[0]; ib1 = ibp[4]; }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int toff = (tap / 3) * Wpd + (tap % 3);
#pragma unroll
            for (int k4 = 0; k4 < 2; k4++) {
                const float a = Wl[(tap * 8 + 4 * k4) * 16 + lane];
                const float* bp = In + bbase + 4 * k4 * PS + toff;
#pragma unroll
                for (int t = 0; t < NTW; t++) {
                    float bv = bp[16 * t];
                    if (IB) bv = fmaxf(bv + (k4 ? ib1 : ib0), 0.f);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc[t], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    // D[i][j]: lane = 16 * (i / 4) + j, register = i & 3
#pragma unroll
    for (int t = 0; t < NTW; t++) {
        const int q = q0 + 16 * (NTW * wv + t) + (lane & 15), yy = q / Wpd, xx = q - yy * Wpd;
        if (q < npos && xx < W) {
            float* yo = A.y + (size_t)co_base * HW + (size_t)yy * W + xx;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int co = 4 * (lane >> 4) + r;
                float v = acc[t][r]; v = v > 0.f ? v : v * A.slope;
                if (co < A.cpg_out) yo[(size_t)co * HW] = v;
            }
        }
    }
}


// ---- 16 (or 8) channels per group WITHOUT an input bias, W a multiple of 4 (round 4).  Same tiles, items, weight packing and LDS row layout (four pad positions on the left
// of every row) as k_gconv3x3_m16<NTW, false, true>; what changes is how a chunk gets into LDS.  The kernel above walks (channel, row) pairs with 64-bit addresses and a
// branch per row, writes the padding in a pass of its own and needs a barrier between the two: ~900 vector instructions and two barriers around 288 matrix instructions —
// at layer1's 200 x 272 (ONE 8-channel chunk per workgroup, nothing inside the workgroup to overlap with) 78 us per call for 26 us of matrix time.  Here a channel plane is a
// run of 16-byte slots in flattened (row, quad) order and moves as `nj` buffer-load-to-LDS instructions of 64 slots each: a lane's global offset per instruction is a loop
// invariant computed once (pad quads, rows outside the image and slots past the band carry an out-of-range offset: the copy writes the zeros the convolution pads with),
// the channel rides in the scalar offset — no vector instruction, no padding pass, one barrier per chunk; operands at immediate offsets from six base registers, requested
// one step ahead.  A plane's pitch is >= 256 nj floats so that the last instruction's spare slots stay inside the plane.
#define GC16_MAXJ 10
#define GC16_WSZ 1280     // floats reserved for a chunk's weights: 1152, copied as five 1 KB pieces (the last one drags 128 floats of the next chunk along)
template <int NTW>
__global__ __launch_bounds__(256) void k_gconv3x3_m16d(GcArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float gc_lds[];
    float* Wl = gc_lds;                         // [9][8][16] (+ 128)
    float* In = gc_lds + GC16_WSZ;              // [8][PS]
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = A.H, W = A.W, Wpd = W + 4, Wq = Wpd >> 2, npos = H * Wpd, PS = A.PS, R = A.R, nj = A.nj;
    constexpr int P = NTW * 64;
    const int item = gc_item(A.total); if (item < 0) return;
    const int dbg = A.ablate;           // VIDO_GCONV_ABLATE (measurement only): 1 no matrix instructions, 2 no stores, 4 no copies
    // (divisions by multiplication with host-made reciprocals: beside the matrix instructions of the CU's other workgroup every vector instruction of this one is paid in full)
    const int g = A.gx == 1 ? item : (int)__umulhi((unsigned)item, A.m_gx), q0 = (item - g * A.gx) * P, r0 = (int)__umulhi((unsigned)q0, A.m_wpd);
    const int nchunk = A.cpg_in / GC_KC;
    const unsigned HW = (unsigned)H * (unsigned)W;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)A.x, 0, A.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)A.w, 0, A.wbytes, 0x00020000);
    unsigned voff[GC16_MAXJ];                   // byte offset, inside a channel image, of this lane's quad of instruction j; bit 30: padding / outside the image / past the band
#pragma unroll
    for (int j = 0; j < GC16_MAXJ; j++) {
        const int s = lane + 64 * j, rr = (int)__umulhi((unsigned)s, A.m_wq), xq = s - rr * Wq, row = r0 - 1 + rr;
        voff[j] = (rr < R && xq >= 1 && row >= 0 && row < H) ? 4u * (unsigned)(row * W + 4 * (xq - 1)) : 0x40000000u;
    }
    const unsigned wvo = 16u * (unsigned)lane;
    const unsigned wbase = 4u * (unsigned)(g * nchunk * (9 * GC_KC * 16));
    const unsigned xbase = 4u * (unsigned)(g * A.cpg_in) * HW;
    // wave wv copies channels 2 wv, 2 wv + 1 of a chunk (nj instructions each) and piece wv of its weights (wave 0: piece 4 as well)
    auto issue = [&](int c) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const unsigned so = xbase + 4u * (unsigned)(c * GC_KC + 2 * wv + h) * HW;
            float* plane = In + (2 * wv + h) * PS;
#pragma unroll
            for (int j = 0; j < GC16_MAXJ; j++)
                if (j < nj) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(plane + 256 * j), 16, voff[j], so, 0, 0);
        }
        const unsigned wso = wbase + 4u * (unsigned)(c * (9 * GC_KC * 16));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (__attribute__((address_space(3))) void*)(Wl + 256 * wv), 16, wvo, wso + 1024u * (unsigned)wv, 0, 0);
        if (wv == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (__attribute__((address_space(3))) void*)(Wl + 1024), 16, wvo, wso + 4096u, 0, 0);
    };
    const int co_base = g * A.cpg_out;
    f32x4 acc[NTW];
    {
        float b4[4];
#pragma unroll
        for (int r = 0; r < 4; r++) { const int co = 4 * (lane >> 4) + r; b4[r] = co < A.cpg_out ? A.bias[co_base + co] : 0.f; }
#pragma unroll
        for (int t = 0; t < NTW; t++) { acc[t][0] = b4[0]; acc[t][1] = b4[1]; acc[t][2] = b4[2]; acc[t][3] = b4[3]; }
    }
    typedef const volatile __attribute__((address_space(3))) float* lds_f;      // (volatile: see k_gconv3x3_m32d)
    const int bbase = (lane >> 4) * PS + (q0 + 16 * NTW * wv - r0 * Wpd) + (lane & 15) + 3;
    lds_f Wb = (lds_f)Wl + lane;
    lds_f B00 = (lds_f)In + bbase, B01 = B00 + Wpd, B02 = B01 + Wpd, B10 = B00 + 4 * PS, B11 = B10 + Wpd, B12 = B11 + Wpd;      // [k4][tap row]: everything else is an immediate
    for (int c = 0; c < nchunk; c++) {
        if (c) __syncthreads();                                       // everybody is done with the previous chunk
        if (!(dbg & 4)) issue(c);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        float a[2], b[2][NTW];
        // step s = tap * 2 + k4; the operands of step s + 1 are requested in two halves between the two halves of step s's matrix instructions (at most 15 LDS reads can
        // be outstanding).  Scheduling barriers: left alone, the scheduler sinks every read to just before its matrix instruction and the loop runs at LDS latency.
        auto ld = [&](int s, int half) {
            const int tap = s >> 1, k4 = s & 1;
            lds_f Bt = k4 ? (tap < 3 ? B10 : (tap < 6 ? B11 : B12)) : (tap < 3 ? B00 : (tap < 6 ? B01 : B02));
            if (half == 0) a[s & 1] = Wb[(tap * 8 + 4 * k4) * 16];
#pragma unroll
            for (int t = half * (NTW / 2); t < (half + 1) * (NTW / 2); t++) b[s & 1][t] = Bt[tap % 3 + 16 * t];
        };
        if (dbg & 1) continue;
        ld(0, 0); ld(0, 1);
#pragma unroll
        for (int s = 0; s < 18; s++) {
#pragma unroll
            for (int half = 0; half < 2; half++) {
                if (s + 1 < 18) ld(s + 1, half);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = half * (NTW / 2); t < (half + 1) * (NTW / 2); t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s & 1], b[s & 1][t], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // D[i][j]: lane = 16 * (i / 4) + j, register = i & 3.  Buffer stores, no branches: a lane without an output (pad position, past the image, output channel >= cpg_out)
    // carries an out-of-range offset and its store is dropped; the output channel r of a lane's four rides in the scalar offset.
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)A.y, 0, A.xbytes / (unsigned)A.cpg_in * (unsigned)A.cpg_out, 0x00020000);
    const int co0 = 4 * (lane >> 4);
    const unsigned ybase = 4u * ((unsigned)(co_base + co0) * HW);
    if (dbg & 2) return;
#pragma unroll
    for (int t = 0; t < NTW; t++) {
        const int q = q0 + 16 * (NTW * wv + t) + (lane & 15), yy = (int)__umulhi((unsigned)q, A.m_wpd), xx = q - yy * Wpd;
        const unsigned vo = (q < npos && xx < W && co0 < A.cpg_out) ? ybase + 4u * (unsigned)(yy * W + xx) : 0x40000000u;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float v = acc[t][r];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(v, v * A.slope)), yr, vo, 4u * (unsigned)r * HW, 0);
        }
    }
}


// geometry of a call: which kernel, rows of the band, plane pitch, LDS bytes; 0 = not supported
struct GcPlan { int kind; int R, PS; size_t lds; int gx; bool v4; int nj, PSd; size_t lds_d; };      // nj > 0: k_gconv3x3_m16d has a plan too (plane pitch PSd, lds_d bytes)
static GcPlan gc_plan(int H, int W, int cpg_in, int cpg_out, bool aligned16 = true)
{
    GcPlan p{};
    if (H < 1 || W < 1 || cpg_in < GC_KC || cpg_in % GC_KC) return p;
    if (cpg_out % 32 == 0) {
        const int Wpd = W + 2; const long npos = (long)H * Wpd;
        const int R = 4 + 254 / Wpd;
        if (R * Wpd > GC32_PS || 3 * Wpd + 257 > GC32_PS) return p;
        p.kind = 32; p.R = R; p.PS = GC32_PS; p.lds = (size_t)(9 * GC_KC * 32 + GC_KC * GC32_PS) * 4; p.gx = (int)((npos + 255) / 256);
        return p;
    }
    if (cpg_out != 16 && cpg_out != 8) return p;
    const bool v4 = W % 4 == 0 && aligned16; const int L = v4 ? 4 : 1, Wpd = v4 ? W + 4 : W + 2; const long npos = (long)H * Wpd;
    for (int ntw : {8, 16}) {                                   // 512 positions per workgroup when the band then covers >= 3 output rows of 7, else 1024
        const int P = ntw * 64, R = 4 + (P - 2) / Wpd;
        int PS = std::max(R * Wpd + 4, 3 * Wpd + P + 2 + L) + 16; PS = ((PS + 31) & ~31) + 16;      // = 16 (mod 32): the four 16-float runs of a B operand fall in distinct banks
        const size_t lds = (size_t)(9 * GC_KC * 16 + GC_KC * PS) * 4;
        if (lds > 78 * 1024) continue;                                                      // two workgroups per CU
        if (ntw == 8 && P < 3 * Wpd && npos > 4 * 1024) continue;                           // wide rows: the halo would triple the reads
        p.kind = ntw; p.R = R; p.PS = PS; p.lds = lds; p.gx = (int)((npos + P - 1) / P); p.v4 = v4;
        if (v4) {                                                                           // k_gconv3x3_m16d: planes of 16-byte slots, 64 per copy instruction
            const int nslots = R * (Wpd / 4) + 1, nj = (nslots + 63) / 64;
            int PSd = std::max(PS, 256 * nj + 16); PSd = ((PSd + 15) & ~31) + 16;
            const size_t lds_d = (size_t)(GC16_WSZ + GC_KC * PSd) * 4;
            if (nj <= GC16_MAXJ && lds_d <= 78 * 1024) { p.nj = nj; p.PSd = PSd; p.lds_d = lds_d; }
        }
        return p;
    }
    return p;
}
template <int NTW> static int gc_m16_limits(vido_ctx* ctx)
{
    for (const void* f : {(const void*)k_gconv3x3_m16<NTW, false, false>, (const void*)k_gconv3x3_m16<NTW, false, true>, (const void*)k_gconv3x3_m16<NTW, true, false>, (const void*)k_gconv3x3_m16<NTW, true, true>})
        HIP_TRY(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_gconv3x3_m16d<NTW>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    return VIDO_OK;
}
static int gc_lds_limit(vido_ctx* ctx)
{
    static bool done[64] = {};      // per device
    if (done[ctx->device & 63]) return VIDO_OK;
    { int rc = gc_m16_limits<8>(ctx); if (rc) return rc; }
    { int rc = gc_m16_limits<16>(ctx); if (rc) return rc; }
    done[ctx->device & 63] = true;
    return VIDO_OK;
}
template <int NTW> static void gc_m16_launch(const GcPlan& p, int groups, hipStream_t st, const GcArgs& A)
{
    const dim3 grid(8 * ((A.total + 7) / 8)), blk(256);
    if (A.in_bias) { if (p.v4) hipLaunchKernelGGL((k_gconv3x3_m16<NTW, true, true>), grid, blk, p.lds, st, A); else hipLaunchKernelGGL((k_gconv3x3_m16<NTW, true, false>), grid, blk, p.lds, st, A); }
    else { if (p.v4) hipLaunchKernelGGL((k_gconv3x3_m16<NTW, false, true>), grid, blk, p.lds, st, A); else hipLaunchKernelGGL((k_gconv3x3_m16<NTW, false, false>), grid, blk, p.lds, st, A); }
}

// ---- stride 2, >= 32 channels per group (round 4): `conv2` of the first bottleneck of layer3 / layer4 (resnet.py:300-372 with STRIDE_IN_1X1 = False), which the library runs
// as a strided Winograd kernel on the vector ALUs at 57-58 us per call + our bias pass.  The flattened-position formulation survives the stride if the band's EVEN and ODD
// input rows live in separate LDS planes of row pitch PL = W + 4 (four pad positions on the left of every row, as in k_gconv3x3_m16d) and the outputs are flattened with
// pitch PL / 2: output position q = yo * (PL / 2) + xo then reads input (2 yo - 1 + dy, 2 xo - 1 + dx) at  2 q + dx + 3  in the even plane (dy = 1), in the odd plane
// (dy = 0) or one row further down the odd plane (dy = 2) — constant offsets again, the B operand of a lane is one ds_read_b32 at twice its position.  Everything else is
// k_gconv3x3_m32d (tiles, items, weight packing, two LDS buffers, one barrier per 8-channel chunk) with the 16-byte slot copies of k_gconv3x3_m16d.
#define GS2_MAXJ 12
struct Gs2Args { const float* x; const float* w; const float* bias; float* y; int H, W, Ho, Wo, Wop, PL, NE, PS, cpg_in, cpg_out, gx, total, nj; float slope; unsigned xbytes, wbytes, m_plq, ybytes; };
__global__ __launch_bounds__(256) void k_gconv3x3_s2_m32(Gs2Args A)
{
    extern __shared__ __attribute__((aligned(16))) float gc_lds[];
    constexpr int WSZ = 9 * GC_KC * 32;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = A.H, W = A.W, Wo = A.Wo, Wop = A.Wop, PL = A.PL, PLq = PL >> 2, NE = A.NE, PS = A.PS, nj = A.nj, npos = A.Ho * Wop;
    const int BUF = WSZ + GC_KC * PS;                                 // floats per buffer: weights [9][8][32], then 8 channels of [even rows | odd rows]
    const int nchunk = A.cpg_in / GC_KC, ncob = A.cpg_out / 32;
    const int item = gc_item(A.total); if (item < 0) return;
    const int chunk = item % A.gx, cob = (item / A.gx) % ncob, g = item / (A.gx * ncob), q0 = chunk * 256, r0 = q0 / Wop;
    const unsigned HW = (unsigned)H * (unsigned)W, HWo = (unsigned)A.Ho * (unsigned)Wo;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)A.x, 0, A.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)A.w, 0, A.wbytes, 0x00020000);
    unsigned voff[GS2_MAXJ];                    // byte offset, inside a channel image, of this lane's quad of instruction j; bit 30: padding / outside the image / past the band
#pragma unroll
    for (int j = 0; j < GS2_MAXJ; j++) {
        const int s = lane + 64 * j, rr = (int)__umulhi((unsigned)s, A.m_plq), xq = s - rr * PLq;
        const int row = rr < NE ? 2 * (r0 + rr) : 2 * (r0 + rr - NE) - 1;      // even plane: output rows r0 .., odd plane: the input rows above them (2 (r0 - 1) + 1 ..)
        voff[j] = (rr < 2 * NE + 1 && xq >= 1 && row >= 0 && row < H) ? 4u * (unsigned)(row * W + 4 * (xq - 1)) : 0x40000000u;
    }
    const unsigned wvo = 16u * (unsigned)lane;
    const unsigned wbase = 4u * (unsigned)(((g * ncob + cob) * nchunk) * WSZ);
    const unsigned xbase = 4u * (unsigned)(g * A.cpg_in) * HW;
    // wave wv copies channels 2 wv, 2 wv + 1 of a chunk (nj instructions each) and pieces wv, wv + 4, wv + 8 of its 9 KB of weights
    auto issue = [&](int c, int buf) {
        float* base = gc_lds + buf * BUF;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const unsigned so = xbase + 4u * (unsigned)(c * GC_KC + 2 * wv + h) * HW;
            float* plane = base + WSZ + (2 * wv + h) * PS;
#pragma unroll
            for (int j = 0; j < GS2_MAXJ; j++)
                if (j < nj) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(plane + 256 * j), 16, voff[j], so, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int piece = wv + 4 * q;
            if (piece < 9) __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (__attribute__((address_space(3))) void*)(base + piece * 256), 16, wvo, wbase + 4u * (unsigned)(c * WSZ + piece * 256), 0, 0);
        }
    };
    const int co_base = g * A.cpg_out + cob * 32;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; r++) { const float b = A.bias[co_base + 8 * (r / 4) + 4 * (lane >> 5) + (r & 3)]; acc0[r] = b; acc1[r] = b; }
    typedef const volatile __attribute__((address_space(3))) float* lds_f;      // (volatile: see k_gconv3x3_m32d)
    const int bbase = WSZ + (lane >> 5) * PS + 2 * (q0 + 64 * wv - r0 * Wop + (lane & 31)) + 3;      // B operand of tile 2 wv (tile 2 wv + 1: + 64), even plane, dx = 0, channel pair 0
    issue(0, 0);
    for (int c = 0; c < nchunk; c++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                              // chunk c has landed in buffer c & 1; everybody is done with buffer (c + 1) & 1
        if (c + 1 < nchunk) issue(c + 1, (c + 1) & 1);
        lds_f Wb = (lds_f)(gc_lds + (c & 1) * BUF) + lane;
        lds_f BE = (lds_f)(gc_lds + (c & 1) * BUF) + bbase;
        lds_f Bk[4][3];                                               // [channel pair][tap row]: odd plane (dy = 0), even plane (dy = 1), odd plane one row down (dy = 2)
#pragma unroll
        for (int kp = 0; kp < 4; kp++) { Bk[kp][1] = BE + 2 * kp * PS; Bk[kp][0] = Bk[kp][1] + NE * PL; Bk[kp][2] = Bk[kp][0] + PL; }
        float a[3], b0[3], b1[3];
        auto ld = [&](int s) {                                        // step s = tap * 4 + kp
            const int tap = s >> 2, kp = s & 3; lds_f Bt = Bk[kp][tap / 3];
            a[s % 3] = Wb[(tap * 8 + 2 * kp) * 32]; b0[s % 3] = Bt[tap % 3]; b1[s % 3] = Bt[tap % 3 + 64];
        };
        ld(0); ld(1);
#pragma unroll
        for (int s = 0; s < 36; s++) {                                // (scheduling barriers: left alone, the scheduler sinks every read to just before its matrix instruction and a lone wave runs at LDS latency)
            if (s + 2 < 36) ld(s + 2);
            __builtin_amdgcn_sched_barrier(0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s % 3], b0[s % 3], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s % 3], b1[s % 3], acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // D[i][j]: lane = 32 * ((i / 4) & 1) + j, register = 4 * (i / 8) + (i & 3)  ->  a register holds one output channel of 32 consecutive positions per half wave
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int q = q0 + 64 * wv + 32 * t + (lane & 31), yy = q / Wop, xx = q - yy * Wop;
        if (q < npos && xx < Wo) {
            float* yo = A.y + (size_t)co_base * HWo + (size_t)yy * Wo + xx;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int co = 8 * (r / 4) + 4 * (lane >> 5) + (r & 3);
                const float v = t ? acc1[r] : acc0[r];
                yo[(size_t)co * HWo] = fmaxf(v, v * A.slope);
            }
        }
    }
}


// ---- stride 2, 16 (or 8) channels per group: `conv2` of layer2's first bottleneck.  k_gconv3x3_m16d's matrix loop (16x16x4 tiles, one 8-channel chunk in ONE LDS buffer, two
// workgroups per CU) over k_gconv3x3_s2_m32's band (even / odd input rows in separate planes of pitch W + 4, outputs flattened with half that pitch: B operand at twice the
// lane's position).  NTW tiles of 16 output positions per wave.
template <int NTW>
__global__ __launch_bounds__(256) void k_gconv3x3_s2_m16(Gs2Args A)
{
    extern __shared__ __attribute__((aligned(16))) float gc_lds[];
    float* Wl = gc_lds;                         // [9][8][16] (+ 128)
    float* In = gc_lds + GC16_WSZ;              // [8][PS]: [even rows | odd rows]
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = A.H, W = A.W, Wo = A.Wo, Wop = A.Wop, PL = A.PL, PLq = PL >> 2, NE = A.NE, PS = A.PS, nj = A.nj, npos = A.Ho * Wop;
    constexpr int P = NTW * 64;
    const int item = gc_item(A.total); if (item < 0) return;
    const int g = item / A.gx, q0 = (item - g * A.gx) * P, r0 = q0 / Wop;
    const int nchunk = A.cpg_in / GC_KC;
    const unsigned HW = (unsigned)H * (unsigned)W, HWo = (unsigned)A.Ho * (unsigned)Wo;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)A.x, 0, A.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)A.w, 0, A.wbytes, 0x00020000);
    unsigned voff[GS2_MAXJ];
#pragma unroll
    for (int j = 0; j < GS2_MAXJ; j++) {
        const int s = lane + 64 * j, rr = (int)__umulhi((unsigned)s, A.m_plq), xq = s - rr * PLq;
        const int row = rr < NE ? 2 * (r0 + rr) : 2 * (r0 + rr - NE) - 1;
        voff[j] = (rr < 2 * NE + 1 && xq >= 1 && row >= 0 && row < H) ? 4u * (unsigned)(row * W + 4 * (xq - 1)) : 0x40000000u;
    }
    const unsigned wvo = 16u * (unsigned)lane;
    const unsigned wbase = 4u * (unsigned)(g * nchunk * (9 * GC_KC * 16));
    const unsigned xbase = 4u * (unsigned)(g * A.cpg_in) * HW;
    auto issue = [&](int c) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const unsigned so = xbase + 4u * (unsigned)(c * GC_KC + 2 * wv + h) * HW;
            float* plane = In + (2 * wv + h) * PS;
#pragma unroll
            for (int j = 0; j < GS2_MAXJ; j++)
                if (j < nj) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(plane + 256 * j), 16, voff[j], so, 0, 0);
        }
        const unsigned wso = wbase + 4u * (unsigned)(c * (9 * GC_KC * 16));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (__attribute__((address_space(3))) void*)(Wl + 256 * wv), 16, wvo, wso + 1024u * (unsigned)wv, 0, 0);
        if (wv == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (__attribute__((address_space(3))) void*)(Wl + 1024), 16, wvo, wso + 4096u, 0, 0);
    };
    const int co_base = g * A.cpg_out;
    f32x4 acc[NTW];
    {
        float b4[4];
#pragma unroll
        for (int r = 0; r < 4; r++) { const int co = 4 * (lane >> 4) + r; b4[r] = co < A.cpg_out ? A.bias[co_base + co] : 0.f; }
#pragma unroll
        for (int t = 0; t < NTW; t++) { acc[t][0] = b4[0]; acc[t][1] = b4[1]; acc[t][2] = b4[2]; acc[t][3] = b4[3]; }
    }
    typedef const volatile __attribute__((address_space(3))) float* lds_f;
    const int bbase = (lane >> 4) * PS + 2 * (q0 + 16 * NTW * wv - r0 * Wop + (lane & 15)) + 3;      // even plane, dx = 0, channel quad 0; tile t: + 32 t
    lds_f Wb = (lds_f)Wl + lane;
    lds_f B01 = (lds_f)In + bbase, B00 = B01 + NE * PL, B02 = B00 + PL, B11 = B01 + 4 * PS, B10 = B11 + NE * PL, B12 = B10 + PL;      // [k4][tap row]: odd plane, even plane, odd plane one row down
    for (int c = 0; c < nchunk; c++) {
        if (c) __syncthreads();
        issue(c);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        float a[2], b[2][NTW];
        auto ld = [&](int s, int half) {
            const int tap = s >> 1, k4 = s & 1;
            lds_f Bt = k4 ? (tap < 3 ? B10 : (tap < 6 ? B11 : B12)) : (tap < 3 ? B00 : (tap < 6 ? B01 : B02));
            if (half == 0) a[s & 1] = Wb[(tap * 8 + 4 * k4) * 16];
#pragma unroll
            for (int t = half * (NTW / 2); t < (half + 1) * (NTW / 2); t++) b[s & 1][t] = Bt[tap % 3 + 32 * t];
        };
        ld(0, 0); ld(0, 1);
#pragma unroll
        for (int s = 0; s < 18; s++) {
#pragma unroll
            for (int half = 0; half < 2; half++) {
                if (s + 1 < 18) ld(s + 1, half);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = half * (NTW / 2); t < (half + 1) * (NTW / 2); t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s & 1], b[s & 1][t], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)A.y, 0, A.ybytes, 0x00020000);
    const int co0 = 4 * (lane >> 4);
    const unsigned ybase = 4u * ((unsigned)(co_base + co0) * HWo);
#pragma unroll
    for (int t = 0; t < NTW; t++) {
        const int q = q0 + 16 * (NTW * wv + t) + (lane & 15), yy = q / Wop, xx = q - yy * Wop;
        const unsigned vo = (q < npos && xx < Wo && co0 < A.cpg_out) ? ybase + 4u * (unsigned)(yy * Wo + xx) : 0x40000000u;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float v = acc[t][r];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(v, v * A.slope)), yr, vo, 4u * (unsigned)r * HWo, 0);
        }
    }
}
#define GS2_NTW16 4      // tiles of 16 output positions per wave of k_gconv3x3_s2_m16: 256 positions per workgroup (one LDS buffer <= 78 KB: two workgroups per CU)

// geometry of a stride-2 call; lds == 0: not supported
struct Gs2Plan { int Ho, Wo, Wop, PL, NE, PS, gx, nj; size_t lds; };
static Gs2Plan gs2_plan(int H, int W, int cpg_in, int cpg_out)
{
    Gs2Plan p{};
    const bool m16 = cpg_out == 16 || cpg_out == 8;
    if (H < 2 || W < 4 || W % 4 || cpg_in < GC_KC || cpg_in % GC_KC || (cpg_out % 32 && !m16)) return p;
    const int P = m16 ? GS2_NTW16 * 64 : 256;                       // output positions per workgroup
    p.Ho = (H + 1) / 2; p.Wo = W / 2; p.PL = W + 4; p.Wop = p.PL / 2;
    p.NE = 2 + (P - 1) / p.Wop;                                     // output rows P consecutive positions can touch; the odd plane holds one row more
    const int nslots = (2 * p.NE + 1) * (p.PL / 4) + 1;             // (+ the pad quad behind the last row)
    p.nj = (nslots + 63) / 64;
    p.PS = std::max((2 * p.NE + 1) * p.PL + 4, 256 * p.nj);
    if (m16) p.PS = ((p.PS + 15) & ~31) + 16;                       // = 16 (mod 32), as in gc_plan
    p.gx = (p.Ho * p.Wop + P - 1) / P;
    const size_t lds = m16 ? (size_t)(GC16_WSZ + GC_KC * p.PS) * 4 : 2 * (size_t)(9 * GC_KC * 32 + GC_KC * p.PS) * 4;
    if (p.nj > GS2_MAXJ || lds > (m16 ? 78 : 156) * (size_t)1024) return p;
    p.lds = lds;
    return p;
}
}  // namespace

extern "C" {

/* 1 when vido_gconv3x3_bias_act has a kernel for this shape (channels per group 8, 16 or a multiple of 32; the row band must fit the LDS plan). */
int vido_gconv3x3_supported(int H, int W, int cpg_in, int cpg_out)
{
    return gc_plan(H, W, cpg_in, cpg_out).kind != 0;
}

/* Floats of the packed weight tensor (per group: output channels padded to the matrix-core tile). */
int64_t vido_gconv3x3_packed_size(int groups, int cpg_in, int cpg_out)
{
    const int co_pad = cpg_out % 32 == 0 ? cpg_out : 16;
    return (int64_t)groups * co_pad * cpg_in * 9;
}

/* y = leaky_relu(conv2d(x', w, stride 1, padding 1, groups) + bias, slope) for one image, x' = x, or relu(x + in_bias[channel]) when in_bias is given (the epilogue of the
 * convolution that produced x, folded into this one's operand reads): x [groups * cpg_in][H][W], y [groups * cpg_out][H][W] f32 DEVICE tensors (x != y),
 * bias [groups * cpg_out], w_packed: the convolution weight [groups * cpg_out][cpg_in][3][3] rearranged per (group, 32-channel output block, 8-channel input chunk) as
 * [tap][ci][co] (16 / 8 channels per group: [group][chunk][tap][ci][16 co], missing output channels zero) — vido_slam_amd/nets/ops.py::pack_gconv3x3 builds it once per
 * layer.  slope 0 = ReLU, 1 = none.  Enqueues on the adopted stream; capturable. */
int vido_gconv3x3_bias_act(vido_ctx* ctx, const float* x, const float* in_bias, const float* w_packed, const float* bias, float* y, int groups, int cpg_in, int cpg_out, int H, int W, float slope)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!x || !w_packed || !bias || !y || x == y || groups < 1 || groups > 65535) return vido_set_error(ctx, VIDO_E_INVALID, "gconv3x3: bad arguments");
    const GcPlan p = gc_plan(H, W, cpg_in, cpg_out, ((uintptr_t)x & 15) == 0);      // (the 16-byte copies need an aligned image; torch allocations are)
    if (!p.kind) return vido_set_error(ctx, VIDO_E_INVALID, "gconv3x3: no kernel for %d -> %d channels per group at %d x %d", cpg_in, cpg_out, H, W);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    { int rc = gc_lds_limit(ctx); if (rc) return rc; }
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    const long long xb = 4ll * groups * cpg_in * H * W, wb = 4ll * vido_gconv3x3_packed_size(groups, cpg_in, cpg_out);
    GcArgs A{x, w_packed, bias, in_bias, y, H, W, cpg_in, cpg_out, p.R, p.PS, slope, p.gx, p.gx * groups * (p.kind == 32 ? cpg_out / 32 : 1), (unsigned)xb, (unsigned)wb, 0, 0, 0, 0, 0};
    static const bool no_dma32 = getenv("VIDO_GCONV_NO_DMA") != nullptr, no_dma16 = getenv("VIDO_GCONV_NO_DMA16") != nullptr;
    const long long yb = 4ll * groups * cpg_out * H * W, npos16 = (long long)H * (W + 4) + 1024;
    if (p.kind != 32 && p.nj && !in_bias && !no_dma16 && xb < (1ll << 30) && yb < (1ll << 30) && wb < (1ll << 32) && slope >= 0.f && slope <= 1.f && ((uintptr_t)w_packed & 15) == 0
        && npos16 * (W + 4) < (1ll << 32) && (long long)A.total * p.gx < (1ll << 32)) {      // (the last two: the kernel's reciprocal divisions are exact)
        A.PS = p.PSd; A.nj = p.nj; { static const int abl = [] { const char* e = getenv("VIDO_GCONV_ABLATE"); return e ? atoi(e) : 0; }(); A.ablate = abl; }
        auto magic = [](unsigned d) { return (unsigned)((0x100000000ull + d - 1) / d); };      // d >= 2
        A.m_wq = magic((unsigned)(W + 4) / 4); A.m_wpd = magic((unsigned)(W + 4)); A.m_gx = p.gx > 1 ? magic((unsigned)p.gx) : 0;
        const dim3 grid(8 * ((A.total + 7) / 8)), blk(256);
        if (p.kind == 8) hipLaunchKernelGGL(k_gconv3x3_m16d<8>, grid, blk, p.lds_d, st, A); else hipLaunchKernelGGL(k_gconv3x3_m16d<16>, grid, blk, p.lds_d, st, A);
        HIP_TRY(ctx, hipGetLastError());
        return VIDO_OK;
    }
    if (p.kind == 32 && !in_bias && !no_dma32 && xb < (1ll << 30) && wb < (1ll << 32) && slope >= 0.f && slope <= 1.f && ((uintptr_t)w_packed & 15) == 0) {
        static bool attr[64] = {};
        if (!attr[ctx->device & 63]) { HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_gconv3x3_m32d, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * p.lds))); attr[ctx->device & 63] = true; }
        hipLaunchKernelGGL(k_gconv3x3_m32d, dim3(8 * ((A.total + 7) / 8)), dim3(256), 2 * p.lds, st, A);
    }
    else if (p.kind == 32) hipLaunchKernelGGL(k_gconv3x3_m32, dim3(8 * ((A.total + 7) / 8)), dim3(256), p.lds, st, A);
    else if (p.kind == 8) gc_m16_launch<8>(p, groups, st, A);
    else gc_m16_launch<16>(p, groups, st, A);
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

/* test hook: the geometry of k_gconv3x3_m16d for a stride-1 call with 8 / 16 channels per group as out[0..5] = R (band rows), PS (plane pitch), nj (copy instructions), gx (position
 * chunks), positions per workgroup, Wpd (flattened row pitch); 0 = that kernel does not take the shape */
int vido_debug_gc16_plan(int H, int W, int cpg_in, int cpg_out, int* out)
{
    const GcPlan p = gc_plan(H, W, cpg_in, cpg_out);
    if (!p.kind || p.kind == 32 || !p.nj || !out) return 0;
    out[0] = p.R; out[1] = p.PSd; out[2] = p.nj; out[3] = p.gx; out[4] = p.kind * 64; out[5] = W + 4;
    return 1;
}

/* test hook (tests/test_nets_cpu.py walks the band addressing on the CPU): the geometry of a stride-2 call as
 * out[0..8] = Ho, Wo, Wop (flattened output pitch), PL (LDS row pitch), NE (even-plane rows), PS (plane pitch), gx (position chunks), nj (copy instructions), positions per workgroup; 0 = no kernel */
int vido_debug_gs2_plan(int H, int W, int cpg_in, int cpg_out, int* out)
{
    const Gs2Plan p = gs2_plan(H, W, cpg_in, cpg_out);
    if (!p.lds || !out) return 0;
    out[0] = p.Ho; out[1] = p.Wo; out[2] = p.Wop; out[3] = p.PL; out[4] = p.NE; out[5] = p.PS; out[6] = p.gx; out[7] = p.nj; out[8] = (cpg_out % 32) ? GS2_NTW16 * 64 : 256;
    return 1;
}

/* 1 when vido_gconv3x3_s2_bias_act has a kernel for this shape (stride 2, padding 1: output (H + 1) / 2 x W / 2): W a multiple of 4, output channels per group 8, 16 or a multiple of 32, input channels per group a multiple of 8. */
int vido_gconv3x3_s2_supported(int H, int W, int cpg_in, int cpg_out)
{
    return gs2_plan(H, W, cpg_in, cpg_out).lds != 0;
}

/* y = leaky_relu(conv2d(x, w, stride 2, padding 1, groups) + bias, slope) for one image: x [groups * cpg_in][H][W], y [groups * cpg_out][(H + 1) / 2][W / 2] f32 DEVICE tensors
 * (x 16-byte aligned), w_packed as for vido_gconv3x3_bias_act (>= 32 output channels per group).  Replaces the strided `conv2` + `bn2` + `relu_` of the first bottleneck of a
 * ResNeXt stage (maskrcnn_benchmark/modeling/backbone/resnet.py:300-372, STRIDE_IN_1X1 = False).  Enqueues on the adopted stream; capturable. */
int vido_gconv3x3_s2_bias_act(vido_ctx* ctx, const float* x, const float* w_packed, const float* bias, float* y, int groups, int cpg_in, int cpg_out, int H, int W, float slope)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!x || !w_packed || !bias || !y || x == y || groups < 1 || groups > 65535 || ((uintptr_t)x & 15) || ((uintptr_t)w_packed & 15) || !(slope >= 0.f && slope <= 1.f))
        return vido_set_error(ctx, VIDO_E_INVALID, "gconv3x3_s2: bad arguments");
    const Gs2Plan p = gs2_plan(H, W, cpg_in, cpg_out);
    const long long xb = 4ll * groups * cpg_in * H * W, wb = 4ll * vido_gconv3x3_packed_size(groups, cpg_in, cpg_out);
    if (!p.lds || xb >= (1ll << 30) || wb >= (1ll << 32)) return vido_set_error(ctx, VIDO_E_INVALID, "gconv3x3_s2: no kernel for %d -> %d channels per group at %d x %d", cpg_in, cpg_out, H, W);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    static bool attr[64] = {};
    if (!attr[ctx->device & 63]) {
        HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_gconv3x3_s2_m32, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
        HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_gconv3x3_s2_m16<GS2_NTW16>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        attr[ctx->device & 63] = true;
    }
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    const bool m16 = cpg_out % 32 != 0;
    const long long yb = 4ll * groups * cpg_out * p.Ho * p.Wo;
    if (yb >= (1ll << 30)) return vido_set_error(ctx, VIDO_E_INVALID, "gconv3x3_s2: output too large");
    Gs2Args A{x, w_packed, bias, y, H, W, p.Ho, p.Wo, p.Wop, p.PL, p.NE, p.PS, cpg_in, cpg_out, p.gx, p.gx * groups * (m16 ? 1 : cpg_out / 32), p.nj, slope, (unsigned)xb, (unsigned)wb,
              (unsigned)((0x100000000ull + (unsigned)(p.PL / 4) - 1) / (unsigned)(p.PL / 4)), (unsigned)yb};
    if (m16) hipLaunchKernelGGL(k_gconv3x3_s2_m16<GS2_NTW16>, dim3(8 * ((A.total + 7) / 8)), dim3(256), p.lds, st, A);
    else hipLaunchKernelGGL(k_gconv3x3_s2_m32, dim3(8 * ((A.total + 7) / 8)), dim3(256), p.lds, st, A);
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

}  // extern "C"
