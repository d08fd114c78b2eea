// Dense 3x3 stride-1 `same` convolution + bias + leaky ReLU as a DIRECT implicit GEMM in split-fp16 arithmetic (gfx950), one launch (round 6).
//
// What it replaces: the Winograd kernel (csrc/wino.hip, fp32 matrix instruction) on the detector's chip-filling 256 -> 256 layers — the FPN output convolutions and the RPN
// head on P2 / P3 (maskrcnn_benchmark/modeling/backbone/fpn.py:55-66, rpn/rpn.py:74-107) and the four mask-head convolutions over the detections' 14 x 14 maps
// (roi_heads/mask_head/roi_mask_feature_extractors.py).  The arithmetic is csrc/conv1x1.hip's split-fp16 form: every fp32 operand is h + 2^-11 l' with h = rne16(x),
// l' = rne16(2^11 (x - h)); the three products w_h x_h (-> acc), w_h x_l' and w_l' x_h (-> acl) run on v_mfma_f32_32x32x16_f16 with fp32 accumulators, the result is
// acc + 2^-11 acl times the output channel's inverse weight scale (a power of two chosen at pack time so that the channel's largest |w| sits in [2^14, 2^15)).  Activations
// are taken as they are (|x| < 65504; a workgroup that meets a larger one raises the context's range flag, vido_conv1x1_range_flag).  The 16-bit instruction does 16x the
// fp32 instruction's multiply-adds per cycle, so the DIRECT form's 9 x 3 = 27 products per output and input channel cost less matrix time than Winograd's 4 on the fp32
// instruction — and a direct tile streams 36 bytes of weight planes per (input, output) channel pair where a Winograd tile streams 64 - 96: with these kernels bound by the
// operand stream L2 -> LDS, bytes per tile are what counts.
//
// Formulation.  Workgroup = 8 waves = 128 output channels x a 16 x 16 block of positions of one image; wave w owns all 128 channels x rows 2w, 2w + 1 of the block
// (4 row blocks x 32 positions: two accumulator sets of 64 registers).  The K loop walks chunks of 16 input channels x the three filter rows:
//   * weights: packed [cout / 32][chunk][dy][dx][plane 2][64 lanes][8 fp16] (pack_conv3x3_h): a 1 KB piece is the A operand of one (32-channel block, tap, plane); a step
//     (chunk, dy) is 24 pieces, copied global -> LDS by scalar-addressed buffer loads with the lds bit two steps ahead into a ring of three 24 KB slots;
//   * activations: the chunk's 18 x 18 x 16 fp32 window arrives by the same kind of copies (4 bytes per lane from per-lane offsets computed once: the halo, the image border
//     and a block that hangs over the image edge are offsets past the descriptor's range -> zeros), one chunk ahead; during the chunk's last step every thread converts five
//     channel pairs of it into the two fp16 planes, laid out [pixel][16 channels] so that the B operand of a tap is ONE 16-byte read per plane at an immediate offset;
//   * ONE bare s_barrier per step (no fence: it would drain the copies in flight); what it orders is stated by explicit s_waitcnt counts — every wave issues the same number
//     of copies per step (3 weight pieces, + 11 window pieces on a chunk's first step), so "N still in flight" means the same thing to all of them.
// Per step and wave: 36 matrix instructions beside 30 LDS reads and 3 - 14 copies; two waves per SIMD.
#include "common.hpp"
#include <type_traits>

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define C3_OOB 0x40000000u
#define C3_WSLOT 24576                   // one step of weights: 4 row blocks x 3 taps x 2 planes x 1 KB
#define C3_RBW 3                         // ring slots
// Forms <RPW, CG>: the eight waves are CG channel groups x 8 / CG row pairs; a wave owns RPW row blocks of 32 channels x two rows of positions.  Channels per workgroup
// 32 RPW CG, block rows BR = 16 / CG:
//   <4, 1> 128 channels x 16 x 16 positions (the form above);   <2, 2> 128 channels x 8 x 16: half the K loop's length per workgroup for launches that would not fill the
//   chip with the larger block;   <2, 1> 64 channels x 16 x 16 and <1, 2> 64 channels x 8 x 16 for 64-channel layers;   <1, 1> 32 channels x 16 x 16 for 32-channel layers.
// (The weight ring always holds four row blocks per step: the blocks a form does not own are out-of-range copies — zeros, no traffic.)
#define C3_WPIX(BR) (((BR) + 2) * 18)                            // pixels of the window
#define C3_XP(BR) ((16 * C3_WPIX(BR) + 511) / 512)               // window pieces per wave: 11 / 6
#define C3_RAW(BR) (C3_XP(BR) * 8 * 256)                         // the fp32 window, padded to whole pieces per wave
#define C3_PLANE(BR) (C3_WPIX(BR) * 32)                          // one fp16 plane of the window: pixels x 16 channels x 2 bytes
#define C3_PBUF(BR) (2 * C3_PLANE(BR))
#define C3_LDS(BR) (C3_RBW * C3_WSLOT + C3_RAW(BR) + 2 * C3_PBUF(BR))

struct C3Args { const float* x; const void* wp; const float* bias; float* y; int N, Cin, Cout, H, W, nby, nbx, mt, total, nchunk; float slope; unsigned xbytes, wbytes; unsigned* range_flag; };

template <int RPW, int CG>
__global__ __launch_bounds__(512) void k_conv3x3_h(C3Args A)
{
    constexpr int BR = 16 / CG, MT = 32 * RPW * CG, WPIX = C3_WPIX(BR), XP = C3_XP(BR), RAWB = C3_RAW(BR), PLANE = C3_PLANE(BR), PBUF = C3_PBUF(BR), CK = (WPIX * 8 + 511) / 512, NMM = 9 * RPW, NRD = 2 * RPW + 2;
    extern __shared__ __attribute__((aligned(16))) char c3_lds[];
    char* L = c3_lds;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int per = gridDim.x >> 3, item = (blockIdx.x & 7) * per + (blockIdx.x >> 3);      // an XCD walks a contiguous range of items, the channel block fastest
    if (item >= A.total) return;
    const int nt = item / A.mt, mtile = item - nt * A.mt, m0 = mtile * MT;
    const int bpi = A.nby * A.nbx, n = nt / bpi, brem = nt - n * bpi, by = brem / A.nbx, bx = brem - by * A.nbx, Y0 = by * BR, X0 = bx * 16;
    const int rp = CG == 1 ? w : (w & 3), rb0 = CG == 1 ? 0 : RPW * (w >> 2);      // the wave's row pair of the block, its first row block of 32 channels
    const int hw = A.H * A.W, nsteps = A.nchunk * 3;

    // ---- copies.  Weights: piece i = w + 8 q (q = 0 .. 2) of a step = (row block i / 6, tap-and-plane i % 6); everything but the step rides in abase.
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)A.wp, 0, A.wbytes, 0x00020000);
    const unsigned avo = 16u * (unsigned)lane;
    unsigned abase[3], avq[3];                                             // (a row block the form does not own — 64- / 32-channel layers — is copied from a per-lane offset past the
#pragma unroll                                                             //  descriptor's range: zeros, no read behind the weight tensor; the range check looks at the per-lane offset only)
    for (int q = 0; q < 3; q++) { const int i = w + 8 * q, rb = i / 6, rem = i - 6 * rb; abase[q] = 1024u * (unsigned)(((m0 >> 5) + rb) * A.nchunk * 18 + rem); avq[q] = rb < RPW * CG ? avo : 0x80000000u; }
    // Window: piece 8 j + w (j = 0 .. XP - 1) holds elements e = 64 (8 j + w) + lane of [16 channels][BR + 2 rows][18 columns]
    unsigned xvo[11];                                                      // (fixed bound: see csrc/conv1x1.hip on arrays of template-dependent size captured by lambdas)
#pragma unroll
    for (int j = 0; j < XP; j++) {
        const int e = 64 * (8 * j + w) + lane, ch = e / WPIX, rem = e - ch * WPIX, row = rem / 18, col = rem - row * 18, gy = Y0 - 1 + row, gx = X0 - 1 + col;
        xvo[j] = (e < 16 * WPIX && gy >= 0 && gy < A.H && gx >= 0 && gx < A.W) ? 4u * (unsigned)(ch * hw + gy * A.W + gx) : C3_OOB;
    }
    const unsigned ximg = 4u * (unsigned)n * (unsigned)A.Cin * (unsigned)hw, xchunk = 64u * (unsigned)hw;
    char* const RAW = L + C3_RBW * C3_WSLOT;
    char* const PL = RAW + RAWB;
    auto issue_w = [&](int s, int slot) {                                  // this wave's three pieces of step s (clamped by the caller)
        char* S = L + slot * C3_WSLOT + w * 1024;
        const unsigned so = 6144u * (unsigned)s;
#pragma unroll
        for (int q = 0; q < 3; q++) __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (__attribute__((address_space(3))) void*)(S + q * 8192), 16, avq[q], abase[q] + so, 0, 0);
    };
    auto issue_x = [&](int c) {                                            // this wave's XP pieces of chunk c's window
        // The chunk rides in the DESCRIPTOR (base = first channel of the chunk, num_records = the bytes of the tensor behind it), not in the scalar offset: the hardware's
        // range check looks at the per-lane offset alone, and the channels a last chunk reads past Cin (their weights are zero) must not leave the tensor behind the last image.
        const unsigned so = ximg + xchunk * (unsigned)c;
        const __amdgpu_buffer_rsrc_t xc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)A.x + so), 0, A.xbytes - so, 0x00020000);
#pragma unroll
        for (int j = 0; j < XP; j++) __builtin_amdgcn_raw_ptr_buffer_load_lds(xc, (__attribute__((address_space(3))) void*)(RAW + (8 * j + w) * 256), 4, xvo[j], 0, 0, 0);
    };
    // ---- the window's two fp16 planes: item idx = tid + 512 k -> channel pair idx / WPIX, pixel idx % WPIX
    float xmax = 0.f;
    const f32x2 k2048 = {2048.f, 2048.f};
    auto convert = [&](int buf) {
        const float* R = (const float*)RAW;
        unsigned* P = (unsigned*)(PL + buf * PBUF);
#pragma unroll
        for (int k = 0; k < CK; k++) {
            const int idx = tid + 512 * k;
            if (k < CK - 1 || idx < WPIX * 8) {
                const int q = idx / WPIX, pix = idx - q * WPIX;
                const f32x2 v = {R[(2 * q) * WPIX + pix], R[(2 * q + 1) * WPIX + pix]};
                const f16x2 h = __builtin_convertvector(v, f16x2);
                f32x2 vs, r;
                asm("v_pk_mul_f32 %0, %1, %2" : "=v"(vs) : "v"(v), "v"(k2048));
                asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r.x) : "v"(h), "s"(-2048.f), "v"(vs.x));
                asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r.y) : "v"(h), "s"(-2048.f), "v"(vs.y));
                const f16x2 l = __builtin_convertvector(r, f16x2);
                P[pix * 8 + q] = __builtin_bit_cast(unsigned, h); P[PLANE / 4 + pix * 8 + q] = __builtin_bit_cast(unsigned, l);
                xmax = __builtin_fmaxf(__builtin_fmaxf(xmax, __builtin_fabsf(v.x)), __builtin_fabsf(v.y));
            }
        }
    };

    f32x16 acc[RPW], acl[RPW];
#pragma unroll
    for (int rb = 0; rb < RPW; rb++)
#pragma unroll
        for (int r = 0; r < 16; r++) { acc[rb][r] = 0.f; acl[rb][r] = 0.f; }
    typedef const __attribute__((address_space(3))) char* lds_c;
    const unsigned a_lane = 16u * (unsigned)lane;
    const unsigned b_lane = (unsigned)(C3_RBW * C3_WSLOT + RAWB) + 32u * (unsigned)((2 * rp + ((lane & 31) >> 4)) * 18 + (lane & 15)) + 16u * (unsigned)(lane >> 5);

    // prologue: window of chunk 0 and the first two weight steps; convert chunk 0
    issue_x(0); issue_w(0, 0); issue_w(min(1, nsteps - 1), 1);
    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the window has landed (everybody's pieces); the two weight steps may be on their way
    convert(0);
    int slot = 0;
    // step (c, dy): barrier (weights of the step landed, planes visible, everybody done with the slot the next copies overwrite) -> copies two steps ahead
    // (+ the next window on dy = 0) -> 36 matrix instructions (+ the next window's conversion on dy = 2)
    // One step = three taps.  Operands of tap t + 1 are read while tap t multiplies (two register sets); the step's copies are dealt between the matrix instructions by
    // hand — a copy piece costs ~60 cycles of the wave's issue time (csrc/conv1x1.hip), fourteen of them in front of the first matrix instruction would idle the pipe.
    u32x4 a[2][RPW][2], b[2][2];
    auto step = [&](auto dy_c, auto nbuf_c, int pbuf, auto&& copies) {
        constexpr int dy = decltype(dy_c)::value, NBUF = decltype(nbuf_c)::value;
        lds_c Ab = (lds_c)(L + slot * C3_WSLOT) + a_lane;
        lds_c Bb = (lds_c)L + b_lane + pbuf * PBUF + dy * (18 * 32);
        auto ld = [&](int dx, int set) {
#pragma unroll
            for (int rb = 0; rb < RPW; rb++)
#pragma unroll
                for (int pl = 0; pl < 2; pl++) a[set][rb][pl] = *(const __attribute__((address_space(3))) u32x4*)(Ab + (((rb0 + rb) * 3 + dx) * 2 + pl) * 1024);
#pragma unroll
            for (int pl = 0; pl < 2; pl++) b[set][pl] = *(const __attribute__((address_space(3))) u32x4*)(Bb + dx * 32 + pl * PLANE);
        };
        __builtin_amdgcn_sched_barrier(0);
        ld(0, 0);
        copies();
#pragma unroll
        for (int dx = 0; dx < 3; dx++) {
            if (dx < 2) ld(dx + 1, (dx + 1) & 1);
#pragma unroll
            for (int term = 0; term < 3; term++) {
                constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};
#pragma unroll
                for (int rb = 0; rb < RPW; rb++) {
                    f32x16& d = term == 2 ? acc[rb] : acl[rb];
                    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[dx & 1][rb][PA[term]]), __builtin_bit_cast(f16x8, b[dx & 1][PB[term]]), d, 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);               // operands of the first tap
#pragma unroll
        for (int i = 0; i < NMM; i++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < 2 * NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (i >= 1 && i < 1 + NBUF) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int c = 0; c < A.nchunk; c++) {
        const int pbuf = c & 1, cn = min(c + 1, A.nchunk - 1);
        // dy = 0.  In flight behind the weights of this step: the next step's three pieces.
        asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        step(std::integral_constant<int, 0>{}, std::integral_constant<int, 3 + XP>{}, pbuf, [&] { const int sf = slot == 0 ? C3_RBW - 1 : slot - 1; issue_w(min(3 * c + 2, nsteps - 1), sf); issue_x(cn); });
        slot = slot + 1 == C3_RBW ? 0 : slot + 1;
        // dy = 1.  Behind this step's weights (issued two steps ago): step (c, 2)'s pieces and the window's eleven.
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(3 + XP) : "memory");
        step(std::integral_constant<int, 1>{}, std::integral_constant<int, 3>{}, pbuf, [&] { const int sf = slot == 0 ? C3_RBW - 1 : slot - 1; issue_w(min(3 * c + 3, nsteps - 1), sf); });
        slot = slot + 1 == C3_RBW ? 0 : slot + 1;
        // dy = 2.  This step's weights AND the window (both older than the three pieces of the step before) have landed.
        asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        step(std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{}, pbuf, [&] { const int sf = slot == 0 ? C3_RBW - 1 : slot - 1; issue_w(min(3 * c + 4, nsteps - 1), sf); });
        convert(pbuf ^ 1);                                                  // (past the last chunk: the last window again, into the buffer nobody reads)
        slot = slot + 1 == C3_RBW ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");           // (the last, unused copies)
#pragma unroll
    for (int rb = 0; rb < RPW; rb++) acc[rb] += acl[rb] * 0x1p-11f;
    if (!(xmax < 65504.f) && A.range_flag) atomicOr(A.range_flag, 1u);     // (also a NaN)

    // ---- epilogue: register r of a lane = output channel 8 (r / 4) + 4 (lane >> 5) + (r & 3) of the row block, position lane & 31 of the wave's two rows
    const float* wsc = (const float*)((const char*)A.wp + (size_t)A.nchunk * 16 * 18 * A.Cout * 2);      // the inverse channel scales behind the planes (36 bytes per weight)
    const int p = lane & 31, Y = Y0 + 2 * rp + (p >> 4), X = X0 + (p & 15);
    const bool inside = Y < A.H && X < A.W;
    float* yb = A.y + (size_t)n * A.Cout * hw + (size_t)Y * A.W + X;
#pragma unroll
    for (int rb = 0; rb < RPW; rb++) {
        float bv[16], sv[16];                                              // (loads first: a load issued between stores waits for them)
#pragma unroll
        for (int r = 0; r < 16; r++) { const int co = m0 + 32 * (rb0 + rb) + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3); bv[r] = A.bias ? A.bias[co] : 0.f; sv[r] = wsc[co]; }
        if (inside) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int co = m0 + 32 * (rb0 + rb) + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                const float v = acc[rb][r] * sv[r] + bv[r];
                yb[(size_t)co * hw] = fmaxf(v, v * A.slope);
            }
        }
    }
}
}  // namespace

extern "C" {

/* 1 when vido_conv3x3_h_bias_act takes the shape: output channels 32, 64 or a multiple of 128, tensors below 1 GB.  Input channels are padded to a multiple of 16 with ZERO
 * WEIGHTS (pack_conv3x3_h): the last chunk's window reads past the image's channels — the next image's (finite) activations times zero, or, past the tensor, the zeros of an
 * out-of-range copy. */
int vido_conv3x3_h_supported(int n, int cin, int cout, int h, int w)
{
    const long long cp = (cin + 15) / 16 * 16;
    return n >= 1 && cin >= 1 && (cout == 32 || cout == 64 || (cout >= 128 && cout % 128 == 0)) && h >= 1 && w >= 1 && 4ll * n * cin * h * w < (1ll << 30) && 4ll * n * cout * h * w < (1ll << 30)
           && 36ll * cp * cout < (1ll << 31);
}

/* The form a launch uses: channels per workgroup = min(cout, 128); position block 16 rows x 16 columns when that gives at least C3_MIN16 workgroups, else (128- and 64-channel
 * layers) 8 x 16 — half the K loop's length per workgroup, twice the workgroups.  VIDO_CONV3X3_H_ROWS = 16 / 8 forces one. */
#define C3_MIN16 190
static int c3_block_rows(int n, int cout, int h, int w)
{
    static const int force = [] { const char* e = getenv("VIDO_CONV3X3_H_ROWS"); return e ? atoi(e) : 0; }();
    if (cout == 32) return 16;
    if (force == 16 || force == 8) return force;
    const int mt = cout >= 128 ? cout / 128 : 1;
    return n * ((h + 15) / 16) * ((w + 15) / 16) * mt >= C3_MIN16 ? 16 : 8;
}
/* workgroups of a launch (the caller keeps the Winograd kernel for launches that would leave most of the chip idle) */
int vido_conv3x3_h_workgroups(int n, int cout, int h, int w) { const int br = c3_block_rows(n, cout, h, w); return n * ((h + br - 1) / br) * ((w + 15) / 16) * (cout >= 128 ? cout / 128 : 1); }

/* y = leaky_relu(conv2d(x, w, stride 1, padding 1) + bias, slope): x [n][cin][h][w], y [n][cout][h][w] f32 DEVICE tensors (4-byte aligned, y != x), bias [cout] or NULL;
 * w_packed: the weight [cout][cin][3][3] as two fp16 planes of its output channels scaled by powers of two, plane p of element (co, ci, dy, dx) at
 * [co / 32][ci / 16][dy][dx][p][32 ((ci % 16) / 8) + co % 32][ci % 8] (input channels padded to a multiple of 16 with zeros), followed by [cout] floats: the inverse scales
 * (vido_slam_amd/nets/ops.py::pack_conv3x3_h).
 * slope: 0 = ReLU, 1 = none (0 <= slope <= 1).  Activations must stay below 65504 in magnitude (else: vido_conv1x1_range_flag).  Enqueues on the adopted stream; capturable. */
int vido_conv3x3_h_bias_act(vido_ctx* ctx, const float* x, const void* w_packed, const float* bias, float* y, int n, int cin, int cout, int h, int w, float slope)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!x || !w_packed || !y || x == y || !vido_conv3x3_h_supported(n, cin, cout, h, w) || slope < 0.f || slope > 1.f || (((uintptr_t)x | (uintptr_t)y) & 3) || ((uintptr_t)w_packed & 15))
        return vido_set_error(ctx, VIDO_E_INVALID, "conv3x3_h: no kernel for %d x %d -> %d channels at %d x %d (or a pointer is misaligned, or slope outside [0, 1])", n, cin, cout, h, w);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    const int br = c3_block_rows(n, cout, h, w);
    const int nby = (h + br - 1) / br, nbx = (w + 15) / 16, mt = cout >= 128 ? cout / 128 : 1, total = n * nby * nbx * mt;
    const int nchunk = (cin + 15) / 16;
    C3Args A{x, w_packed, bias, y, n, cin, cout, h, w, nby, nbx, mt, total, nchunk, slope, (unsigned)(4ll * n * cin * h * w), (unsigned)(36ll * 16 * nchunk * cout), ctx->c1_range_flag};
    static bool attr[64] = {};
    if (!attr[ctx->device & 63]) {
#define C3_ATTR(RPW_, CG_) HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_conv3x3_h<RPW_, CG_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C3_LDS(16 / CG_)))
        C3_ATTR(4, 1); C3_ATTR(2, 2); C3_ATTR(2, 1); C3_ATTR(1, 2); C3_ATTR(1, 1);
#undef C3_ATTR
        attr[ctx->device & 63] = true;
    }
    const dim3 grid(8 * ((total + 7) / 8));
#define C3_LAUNCH(RPW_, CG_) hipLaunchKernelGGL((k_conv3x3_h<RPW_, CG_>), grid, dim3(512), C3_LDS(16 / CG_), st, A)
    if (cout >= 128) { if (br == 16) C3_LAUNCH(4, 1); else C3_LAUNCH(2, 2); }
    else if (cout == 64) { if (br == 16) C3_LAUNCH(2, 1); else C3_LAUNCH(1, 2); }
    else C3_LAUNCH(1, 1);
#undef C3_LAUNCH
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

}  // extern "C"
