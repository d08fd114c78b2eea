// 1x1 convolution + bias (+ residual) + activation of the detector's bottlenecks as ONE fp32 matrix-core GEMM launch (gfx950).
//
// What it replaces: conv1 / conv3 of BottleneckWithFixedBatchNorm and the FPN's lateral convolutions (maskrcnn_benchmark/modeling/backbone/resnet.py:300-372,
// backbone/fpn.py) at batch 1: 69 convolutions per frame of X-101-32x8d, 7.1 GFLOP each, which the library runs as rocBLAS GEMMs (MT64x64x16 ... MT128x128x32: 61-77 us
// per call, 92-116 TFLOP/s, 62-67 % matrix-pipe occupancy) followed — for conv3 — by a separate bias + residual + ReLU pass over the output (33 launches, 0.33 ms).
//
// Formulation.  NCHW at batch 1 IS the GEMM  Out[co][q] = sum_k W[co][k] In[k][q]  with the flattened position q contiguous: M = Cout, N = H*W, K = Cin.
//   * v_mfma_f32_32x32x2f32 with A = 32 output channels x 2 input channels, B = the same 2 channels x 32 positions: the D layout then keeps, in one register, one output
//     channel of 32 consecutive positions per half wave — stores (and the residual reads) are whole 128-byte lines.
//   * workgroup = 4 waves (one per SIMD) = 128 output channels x 128 positions, a wave 64 x 64 (four tiles, 64 accumulator registers).
//   * The loop carries NO vector-ALU instruction: beside the fp32 matrix instruction each one costs ~5.5 cycles of matrix time whatever it is (tools/ubench/mfma_fillers*.hip).
//     The second version of this kernel (8 waves of 32 x 64, weights from L2 into registers) had ~58 of them per 32 matrix instructions — register copies of its double
//     buffer, LDS address adds for ds_read2_b32's short offsets, 64-bit pointer arithmetic — and ran at 82-98 TFLOP/s, below the library's 94-108.
//   * Both operands go global -> LDS by buffer loads with the lds bit (16 bytes per lane, scalar-addressed: per-lane offsets are loop invariants, chunk and piece ride in
//     the scalar offset).  The weights are packed on the host in OPERAND order ([32-channel block][group of four k-pairs][lane][4]): a 1 KB piece is one (block, group),
//     a lane's four operands are one conflict-free ds_read_b128.  The activations land as [KC][128] rows; a B operand is one ds_read_b32 at an immediate offset.
//     Two LDS buffers, ONE barrier per chunk of KC = 64 input channels (128 matrix instructions per wave), operands requested two groups of 16 instructions ahead.
//   * epilogue in registers: + bias[co], + residual[co][q], leaky ReLU as max(v, slope v), store.
// Work items (position tile, channel tile) are dealt so that an XCD walks a contiguous range with the channel tile fastest: the Cout / 128 workgroups that share a B tile
// run back to back on one L2.
#include "common.hpp"
#include <atomic>

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define C1_TM 128
#define C1_TN 128

struct C1Args { const float* x; const float* wp; const float* bias; const float* res; float* y; int M, N, K, mt, total, nchunk; float slope; unsigned xbytes, wbytes; int W; unsigned* range_flag; int hwimg, c4; };      // hwimg > 0 (k_conv1x1_b3 only): x is a BATCH [img][K][hwimg] and position p = img hwimg + hw; c4: RES 3

// RES: 1 a residual [cout][H*W] is added, 2 a residual at HALF the resolution [cout][H/2][W/2] is added nearest-upsampled (the FPN's top-down path: fpn.py
// `F.interpolate(last_inner, scale_factor=2, mode="nearest") + inner_lateral`); KC: input channels per chunk and barrier (64: 128 KB of LDS, 128 matrix instructions per
// wave between barriers; 32 for K % 64 != 0)
template <int RES, int KC>
__global__ __launch_bounds__(256) void k_conv1x1(C1Args A)
{
    constexpr int NG = KC / 8;                        // groups of four k-pairs (16 matrix instructions per wave) per chunk
    constexpr int A_BUF = 4 * NG * 256, B_BUF = KC * C1_TN;
    constexpr int NPW = KC / 4;                       // 1 KB copy pieces per wave and chunk: 4 NG of A (one (row block, group) each), KC / 2 of B (two rows each)
    extern __shared__ __attribute__((aligned(16))) float c1_lds[];             // [2][A_BUF] [2][B_BUF]
    float* Al = c1_lds; float* Bl = c1_lds + 2 * A_BUF;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w & 1, wn = w >> 1;      // 4 waves: 64-row half wm, 64-position half wn
    // contiguous item range per XCD (the hardware deals consecutive workgroup ids round-robin over the 8 XCDs)
    const int per = gridDim.x >> 3, item = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (item >= A.total) return;
    const int nt = item / A.mt, mtile = item - nt * A.mt, n0 = nt * C1_TN, m0 = mtile * C1_TM;
    // ---- copies: both operands global -> LDS with buffer loads carrying the lds bit (16 bytes per lane, 1 KB per instruction); per-lane offsets are loop invariants, the
    // chunk and the piece ride in the scalar offset.  A piece of A = one (32-row block, group of four k-pairs) of the packed weight: 64 lanes x 4 operands, already in the
    // order the matrix instruction wants.  A piece of B = two rows of the [KC][128] activation chunk (lanes 0..31 row r, 32..63 row r + 1, four positions each); positions
    // past the end of a row read into the next row; past the end of the tensor they must read 0 and touch nothing.  The hardware's range check covers the PER-LANE
    // offset (+ the instruction offset) only, never the scalar offset, so the row rides in the descriptor instead: every B copy gets base = x + row offset and
    // num_records = the bytes left behind that row (two scalar adds per copy, no vector instruction) — the last row of the last position tile of a map whose H*W is
    // not a multiple of 128 (25 x 34, 50 x 68) then reads zeros instead of up to 508 bytes behind the allocation.  Those columns are computed and never stored.
    // The global side of a copy is only 4-byte aligned when H*W is not a multiple of 4 (25 x 34): the 16-byte copy takes that.
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)A.wp, 0, A.wbytes, 0x00020000);
    const unsigned avo = 16u * (unsigned)lane;
    const unsigned bvo = 4u * ((unsigned)(lane >> 5) * (unsigned)A.N + (unsigned)(n0 + 4 * (lane & 31)));      // (checked against the bytes left behind the copy's first row)
    const int kg = A.K / 8;                           // groups per row block of the packed weight
    auto issue = [&](int chunk, int buf, int first, int count) {
#pragma unroll
        for (int q = first; q < first + count; q++) {
            if (q >= NPW) break;
            const int i = 4 * q + w;                                     // (scalar) pieces 0 .. 4 NG - 1: A, then B; a wave's pieces of one q are of one kind
            if (q < NG) {
                const int mb = i / NG, g = i - mb * NG;
                const unsigned so = 4u * (unsigned)((((m0 >> 5) + mb) * kg + chunk * NG + g) * 256);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (__attribute__((address_space(3))) void*)(Al + buf * A_BUF + i * 256), 16, avo, so, 0, 0);
            } else {
                const int p = i - 4 * NG;
                const unsigned so = 4u * (unsigned)(chunk * KC + 2 * p) * (unsigned)A.N;
                const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)A.x + so), 0, A.xbytes - so, 0x00020000);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(Bl + buf * B_BUF + 2 * p * C1_TN), 16, bvo, 0, 0, 0);
            }
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int ni = 0; ni < 2; ni++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[mi][ni][r] = 0.f;
    issue(0, 0, 0, NPW);
    constexpr int PG = (NPW + (NG > 2 ? NG - 3 : 0)) / (NG > 2 ? NG - 2 : 1);      // pieces a wave sends per group: all of them gone two groups before the chunk ends
    for (int c = 0; c < A.nchunk; c++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                              // chunk c has landed in buffer c & 1; everybody is done with buffer (c + 1) & 1
        const int nb = (c + 1) & 1, cn = min(c + 1, A.nchunk - 1);    // (past the last chunk the copies repeat it into the buffer nobody reads again)
        const float* Ab = Al + (c & 1) * A_BUF + (2 * wm * NG) * 256 + lane * 4;
        const float* Bb = Bl + (c & 1) * B_BUF + (lane >> 5) * C1_TN + 64 * wn + (lane & 31);
        typedef const __attribute__((address_space(3))) float* lds_f;
        lds_f Bb1 = (lds_f)(Bb + 32);                                 // the second 32-position half through its own base register: the compiler would pair the two reads
        asm("" : "+v"(Bb1));                                          // into ds_read2_b32, whose 8-bit offsets need a vector add per row; ds_read_b32 takes any row as an immediate
        lds_f Bb0 = (lds_f)Bb;
        // Beside the fp32 matrix instruction every vector-ALU instruction of the wave costs ~5.5 cycles of matrix time (tools/ubench/mfma_fillers*.hip): the loop has none —
        // operands are LDS reads at immediate offsets (A: one 16-byte read per four k-pairs, B: one 4-byte read per k-pair and 32-position half), requested two groups
        // ahead, the copies are scalar-addressed.
        f32x4 a[3][2]; float b[3][4][2];
        auto ldops = [&](int g) {
#pragma unroll
            for (int mi = 0; mi < 2; mi++) a[g % 3][mi] = *(const f32x4*)(Ab + (mi * NG + g) * 256);
#pragma unroll
            for (int kk = 0; kk < 4; kk++)
#pragma unroll
                for (int ni = 0; ni < 2; ni++) b[g % 3][kk][ni] = (ni ? Bb1 : Bb0)[(8 * g + 2 * kk) * C1_TN];
        };
        ldops(0); ldops(1);
#pragma unroll
        for (int g = 0; g < NG; g++) {
            __builtin_amdgcn_sched_barrier(0);
            if (g + 2 < NG) ldops(g + 2);
            issue(cn, nb, g * PG, PG);
#pragma unroll
            for (int kk = 0; kk < 4; kk++)
#pragma unroll
                for (int mi = 0; mi < 2; mi++)
#pragma unroll
                    for (int ni = 0; ni < 2; ni++) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g % 3][mi][kk], b[g % 3][kk][ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // (the last, unused copies)
    // D[i][j]: lane = 32 * ((i / 4) & 1) + j, register = 4 * (i / 8) + (i & 3)  ->  register r of a lane = output channel 8 (r / 4) + 4 (lane >> 5) + (r & 3), position lane & 31.
    // All residual / bias loads of a 32 x 32 tile are issued before its first store.  (Starting the accumulators at bias + residual instead — the loads beside the first
    // chunk's copies — measured no better at 1024 -> 1024 and 5 us worse at 256 -> 256 on 200 x 272: with four rounds of workgroups the longer prologue costs more than the
    // shorter epilogue saves.)
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int ni = 0; ni < 2; ni++) {
            const int co0 = m0 + 64 * wm + 32 * mi + 4 * (lane >> 5), q = n0 + 64 * wn + 32 * ni + (lane & 31);
            if (q < A.N) {
                float rv[16], bv[16], sv[16];                                 // (all loads of the block first: a load issued between the stores below would wait for them — one counter)
                size_t rq = (size_t)q, rn = (size_t)A.N;                  // residual position and plane size
                if (RES == 2) { const int yy = q / A.W, xx = q - yy * A.W; rq = (size_t)(yy >> 1) * (A.W >> 1) + (xx >> 1); rn = (size_t)(A.N / A.W >> 1) * (A.W >> 1); }
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int co = co0 + 8 * (r >> 2) + (r & 3);
                    bv[r] = A.bias ? A.bias[co] : 0.f;
                    rv[r] = RES ? A.res[(size_t)co * rn + rq] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int co = co0 + 8 * (r >> 2) + (r & 3);
                    const float v = acc[mi][ni][r] + bv[r] + rv[r];
                    A.y[(size_t)co * A.N + q] = fmaxf(v, v * A.slope);          // leaky ReLU for 0 <= slope <= 1
                }
            }
        }
}

// ---- 128 x 112 tiles on the 16 x 16 x 4 matrix instruction (round 5) ------------------------------------------------------------------------------------------------------
// Every bottleneck shape of X-101-32x8d has M * N = 13.9 M outputs = 850 tiles of 128 x 128 = 3.32 rounds of 256 CUs (layer3: 216 tiles, 0.84 rounds): a sixth of the
// chip-time is CUs waiting for the last round.  128 x 112 tiles make it 972 tiles = 3.8 rounds (layer3: 248 tiles, one round on 97 % of the CUs), each 7 / 8 of the work.
// 112 is not a multiple of 32, so the tile runs on v_mfma_f32_16x16x4_f32 (same 64 FLOP / clk / SIMD as 32x32x2): a wave owns 32 output channels x 112 positions =
// 2 x 7 fragments (56 accumulator registers), a k-step (4 input channels) is 14 matrix instructions on 14 different accumulators (the 40-cycle dependent latency of this
// instruction never shows).  What carries over from the kernel above: both operands global -> LDS by scalar-addressed 1 KB buffer copies, two LDS buffers and one barrier
// per chunk, operands read at immediate offsets, no vector-ALU instruction in the loop.  What differs:
//   * weights packed [co / 16][k / 16][lane][4] (pack_conv1x1 layout 1): a lane's 16-byte read = its A operands of four k-steps of one 16-row fragment;
//   * a B operand of k-step s is rows 4 s .. 4 s + 3 x 16 positions: a copy piece carries rows (r, r + 2) and the two pieces of a k-step sit 272 floats apart, so that the
//     four 16-lane groups of a ds_read_b32 fall into disjoint banks (rows at the plain pitch of 128 floats would collide two by two);
//   * the copy's last four lanes of a row (positions 112 .. 127 of the 128 a piece could carry) carry an out-of-range offset: no fetch.
#define C2_TN 112
template <int RES, int KC>
__global__ __launch_bounds__(256) void k_conv1x1_n112(C1Args A)
{
    constexpr int NS = KC / 4, NG = KC / 16;          // k-steps / groups of four k-steps per chunk
    constexpr int A_BUF = 8 * NG * 256;               // floats: 8 fragments of 16 rows x NG pieces of 1 KB
    constexpr int B_PIECE = 272, B_BUF = 2 * NS * B_PIECE;
    constexpr int NPA = 2 * NG, NPB = NS / 2, NPW = NPA + NPB;      // copy pieces per wave and chunk
    extern __shared__ __attribute__((aligned(16))) float c1_lds[];  // [2][A_BUF] [2][B_BUF]
    float* Al = c1_lds; float* Bl = c1_lds + 2 * A_BUF;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int per = gridDim.x >> 3, item = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (item >= A.total) return;
    const int nt = item / A.mt, mtile = item - nt * A.mt, n0 = nt * C2_TN, m0 = mtile * C1_TM;
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)A.wp, 0, A.wbytes, 0x00020000);
    const unsigned avo = 16u * (unsigned)lane;
    const unsigned bvo = (lane & 31) < C2_TN / 4 ? 4u * ((unsigned)(lane >> 5) * 2u * (unsigned)A.N + (unsigned)(n0 + 4 * (lane & 31))) : 0xfffffff0u;      // lanes 0..31 row r, 32..63 row r + 2; the last four lanes of a row: out of range, no fetch
    const int kg = A.K / 16;                          // pieces per 16-row fragment of the packed weight
    auto issue = [&](int chunk, int buf, int first, int count) {
#pragma unroll
        for (int q = first; q < first + count; q++) {
            if (q >= NPW) break;
            const int i = 4 * q + w;                                      // (scalar) a wave's pieces of one q are of one kind
            if (q < NPA) {
                const int mb = i / NG, g = i - mb * NG;                   // fragment 0..7 of the tile, group
                const unsigned so = 4u * (unsigned)((((m0 >> 4) + mb) * kg + chunk * NG + g) * 256);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (__attribute__((address_space(3))) void*)(Al + buf * A_BUF + i * 256), 16, avo, so, 0, 0);
            } else {
                const int p = i - 4 * NPA, s = p >> 1, h = p & 1;         // piece h of k-step s: rows 4 s + h and 4 s + h + 2
                const unsigned so = 4u * (unsigned)(chunk * KC + 4 * s + h) * (unsigned)A.N;
                const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)A.x + so), 0, A.xbytes - so, 0x00020000);      // (exact range check, see above)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(Bl + buf * B_BUF + p * B_PIECE), 16, bvo, 0, 0, 0);
            }
        }
    };
    f32x4 acc[2][7];
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int j = 0; j < 7; j++) acc[mi][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    issue(0, 0, 0, NPW);
    constexpr int PS = (NPW + NS - 3) / (NS - 2);     // pieces a wave sends per k-step: all of them gone two k-steps before the chunk ends
    typedef const volatile __attribute__((address_space(3))) float* lds_f;      // (volatile: keeps the compiler from pairing the reads into ds_read2_b32, whose 8-bit offsets cost a vector add per pair)
    for (int c = 0; c < A.nchunk; c++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                              // chunk c has landed in buffer c & 1; everybody is done with buffer (c + 1) & 1
        const int nb = (c + 1) & 1, cn = min(c + 1, A.nchunk - 1);
        const float* Ab = Al + (c & 1) * A_BUF + (2 * w * NG) * 256 + lane * 4;       // fragment 2 w + mi, group g: + (mi NG + g) 256
        lds_f Bb = (lds_f)(Bl + (c & 1) * B_BUF + ((lane >> 4) & 1) * B_PIECE + (lane >> 5) * 128 + (lane & 15));      // k-step s, fragment j: + s 2 B_PIECE + 16 j
        asm("" : "+v"(Bb));
        f32x4 a[2][2]; float b[3][7];
        auto lda = [&](int g) {
#pragma unroll
            for (int mi = 0; mi < 2; mi++) a[g & 1][mi] = *(const f32x4*)(Ab + (mi * NG + g) * 256);
        };
        auto ldb = [&](int s) {
#pragma unroll
            for (int j = 0; j < 7; j++) b[s % 3][j] = Bb[s * 2 * B_PIECE + 16 * j];
        };
        lda(0); ldb(0); ldb(1);
#pragma unroll
        for (int s = 0; s < NS; s++) {
            __builtin_amdgcn_sched_barrier(0);
            if (s + 2 < NS) ldb(s + 2);
            if ((s & 3) == 1 && s / 4 + 1 < NG) lda(s / 4 + 1);
            issue(cn, nb, s * PS, PS);
#pragma unroll
            for (int mi = 0; mi < 2; mi++)
#pragma unroll
                for (int j = 0; j < 7; j++) acc[mi][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(s >> 2) & 1][mi][s & 3], b[s % 3][j], acc[mi][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 14; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // (the last, unused copies)
    // D of a 16 x 16 fragment: lane l, register r = output channel 4 (l >> 4) + r, position l & 15
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int j = 0; j < 7; j++) {
            const int co0 = m0 + 32 * w + 16 * mi + 4 * (lane >> 4), q = n0 + 16 * j + (lane & 15);
            if (q < A.N) {
                float rv[4], bv[4];
                size_t rq = (size_t)q, rn = (size_t)A.N;
                if (RES == 2) { const int yy = q / A.W, xx = q - yy * A.W; rq = (size_t)(yy >> 1) * (A.W >> 1) + (xx >> 1); rn = (size_t)(A.N / A.W >> 1) * (A.W >> 1); }
#pragma unroll
                for (int r = 0; r < 4; r++) { bv[r] = A.bias ? A.bias[co0 + r] : 0.f; rv[r] = RES ? A.res[(size_t)(co0 + r) * rn + rq] : 0.f; }
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float v = acc[mi][j][r] + bv[r] + rv[r];
                    A.y[(size_t)(co0 + r) * A.N + q] = fmaxf(v, v * A.slope);
                }
            }
        }
}

// ---- fp32-equivalent arithmetic on the bf16 matrix instruction: 3-plane split, 6 products (round 6) ----------------------------------------------------------------------
// The fp32 matrix instruction runs at the fp32 VECTOR rate (157 TFLOP/s); v_mfma_f32_32x32x16_bf16 does 16x the multiply-adds per cycle.  Every fp32 number is EXACTLY
// the sum of three bf16 numbers  x = x0 + x1 + x2  (x0 = rne(x), x1 = rne(x - x0), x2 = x - x0 - x1: 3 x (8 significant bits + sign) cover the 24-bit significand),
// a product of two bf16 numbers is exact in fp32, and the accumulators are fp32 as before.  Of the nine plane products the six with i + j <= 2 are kept
// (w0 x0, w0 x1, w1 x0, w0 x2, w1 x1, w2 x0); the dropped three are <= 2^-24 |w x| each, with either sign — two orders below what the fp32 accumulation itself loses over
// K >= 64 terms (tests/test_maskrcnn_gpu.py measures both kernels against float64).  6 instructions of 16x the rate: 2.67x the fp32 matrix peak.
//   * weights: split on the host at pack time into [co / 32][k / 16][plane][lane][8] (pack_conv1x1 layout 2): a 1 KB copy piece is one A operand of one plane,
//     a lane's operand one ds_read_b128.
//   * activations stay fp32 in HBM and in LDS ([16][128] rows per k-step, by the same scalar-addressed buffer copies as above); a wave owns ALL 128 output channels x 32
//     positions, so that every activation is split exactly once per workgroup — by the wave that multiplies with it: 8 ds_read_b32 per lane and k-step (rows 8 (lane >> 5) + i,
//     position lane & 31 = the B operand's own order), 11 vector-ALU instructions per PAIR (3 v_cvt_pk_bf16_f32, 2 shifts, 2 ands, 4 exact subtractions) beside 24 matrix
//     instructions — the bf16 instruction leaves ~5 issue slots per instruction to the wave (MI355X_MICROARCH.md), the loop needs ~3.
//   * a ring of B3_R LDS slots of one k-step (16 input channels: 12 KB of weight planes + 8 KB of activations), copies B3_R - 1 steps ahead, ONE barrier per k-step; the
//     operands of step t + 1 are read (and split) from LDS during step t, so the matrix pipe does not wait behind a barrier.
//
// ---- the same on the fp16 instruction with TWO planes and THREE products (round 6, NP = 2; the default) ----------------------------------------------------------------
// fp16 carries 11 significant bits: h = rne16(x), l = rne16(x - h) leave |x - h - l| <= 2^-22 |x| (two roundings to 11 bits; 2^-24.5 |x| rms — the fp32 instruction's
// operands are exact, its error is all accumulation: one rounding per TWO products where this form has one per sixteen) wherever both halves are NORMAL fp16 numbers, and
// the products w_h x_h, w_h x_l, w_l x_h are exact in fp32; the dropped w_l x_l is <= 2^-22 |w x|.  Measured against float64 the sum's error is 0.23 - 0.97x the fp32
// instruction's (K = 2048 .. 32; the fewer the terms, the closer).  Half the matrix instructions of the bf16 form (3.2 ms -> see DESIGN.md),
// five vector instructions per pair instead of eleven, two thirds of the weight bytes.  What fp16 lacks is RANGE (2^-14 .. 65504), handled without a pass over the data:
//   * the low planes are stored SCALED by 2^11 (l' = rne16(2^11 (x - h)): as large as h's last bits, never subnormal where h is normal); their two products meet in the
//     correction accumulators, which enter the sum as 2^-11 acl — a power of two: exact;
//   * weights: every output channel is scaled at pack time by the power of two that puts its largest |w| into [2^14, 2^15) (pack_conv1x1 layout 3; the inverse scales, one
//     float per channel, follow the planes); the epilogue multiplies the sum by it — exact;
//   * activations are taken as they are: full precision for 2.5e-4 <= |x| < 65504 (28 binades; a smaller |x| keeps an ABSOLUTE error <= 2^-36 = 1.5e-11, far below the
//     rounding of the sum it enters), and a value past the range would turn into an infinity — so the split tracks max |x| beside its conversions (one v_max3 per pair) and a wave that met
//     |x| >= 65504 raises the context's range flag (vido_conv1x1_range_flag: the caller of the network checks it where it reads the detections back and can repeat the frame
//     with vido_conv1x1_set_arith(2), the bf16 form, which has fp32's range).  The detector's activations stay below a few hundred.
// Error against float64, both forms and the fp32 instruction: tests/test_maskrcnn_gpu.py (the bar: <= 1.5x the fp32 instruction's; measured 0.4 - 1.0x).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define B3_SLOTB(NP) ((NP) * 4096 + 8192)       // bytes per k-step in the ring: 4 NP pieces of A (4 row blocks x NP planes), 8 pieces of B (two rows each)
#define B3_PWN(NP) ((NP) + 2)                   // copy pieces per wave and step

__device__ __forceinline__ unsigned b3_cvt_pk(float lo, float hi) { unsigned r; asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi)); return r; }
// two fp32 values -> their three bf16 planes, packed (low half = the first value)
__device__ __forceinline__ void b3_split2(float x0, float x1, unsigned& p0, unsigned& p1, unsigned& p2)
{
    p0 = b3_cvt_pk(x0, x1);
    const float r0 = x0 - __uint_as_float(p0 << 16), r1 = x1 - __uint_as_float(p0 & 0xffff0000u);
    p1 = b3_cvt_pk(r0, r1);
    const float s0 = r0 - __uint_as_float(p1 << 16), s1 = r1 - __uint_as_float(p1 & 0xffff0000u);
    p2 = b3_cvt_pk(s0, s1);
}

// G wave groups of four waves (256 G threads) on one 128 x 128 tile.  G = 2: the two groups take the even / the odd k-steps with accumulators of their own and meet in LDS at
// the end — two waves per SIMD: measured with one (profiles/r6/conv1x1_b3_counters.txt) a wave issues 5.2 other instructions per matrix instruction, each 4 cycles of issue,
// against the 8 slots a 32-cycle matrix instruction leaves: the matrix pipe was 61 % busy over a wave's life.  A partner wave on the same SIMD issues into those gaps.
// A ring slot holds G k-steps (one per group), RB slots; copies run RB - 1 slots ahead; one workgroup barrier per slot.
// Three forms (c1_launch picks by shape): <G 2, RB 4> one workgroup of 8 waves per CU (160 KB) for long contractions on launches of about one workgroup per CU;
// <G 1, RB 4> 80 KB and <= 256 registers, so that TWO workgroups share a CU — the prologue / epilogue of one tile runs beside the other tile's loop (layer1: 16 k-steps of
// ~0.4 us against ~6 us of fixed cost per tile, 3.3 rounds of tiles); <G 1, RB 6> the first form (one workgroup of 4 waves per CU, 120 KB), kept for comparison.
template <int RES, int G, int RB, int NP>
__global__ __launch_bounds__(256 * G, (G == 1 && RB != 6) ? 2 : G) void k_conv1x1_b3(C1Args A)
{
    constexpr int B3_SLOT = B3_SLOTB(NP), B3_PW = B3_PWN(NP);
    constexpr int SLOTB = G * B3_SLOT;
    extern __shared__ __attribute__((aligned(16))) float c1_lds[];
    char* L = (char*)c1_lds;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), w = wv & 3, g = wv >> 2;
    const int per = gridDim.x >> 3, item = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (item >= A.total) return;
    const int nt = item / A.mt, mtile = item - nt * A.mt, n0 = nt * C1_TN, m0 = mtile * C1_TM;
    const int ns = A.nchunk, nb = ns / G;                                 // k-steps of 16 input channels (even: K % 32 == 0), slots
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)A.wp, 0, A.wbytes, 0x00020000);
    // copy pieces of one k-step: piece i = 4 q + w of wave w of the group that owns the step, q = 0 .. 4; i < 12: the A operand (row block i / 3, plane i % 3), else rows
    // 2 p, 2 p + 1 (p = i - 12) of the step's 16 activation rows.  Everything that does not depend on the step is computed here: the A pieces' scalar offsets at step 0, the
    // B pieces' per-lane offsets (the piece's rows ride in the per-lane offset, the STEP in the descriptor: base = x + 64 N t bytes, num_records = the bytes left behind it —
    // the exact range check of k_conv1x1 with one descriptor per step).
    const unsigned avo = 16u * (unsigned)lane;
    unsigned abase[3], bvo[2];                                            // (abase[NP]: an array of template-dependent size captured by a lambda makes the HOST pass drop the kernel's stub — silently)
#pragma unroll
    for (int q = 0; q < NP; q++) { const int i = 4 * q + w, rb = i / NP, pl = i - NP * rb; abase[q] = 1024u * (unsigned)(((m0 >> 5) + rb) * ns * NP + pl); }
    // (a batch: the four positions of a lane stay inside one image — hwimg % 4 == 0 —, a channel row is hwimg positions long and an image K rows; a position past the last
    //  image lands past the descriptor's range like one past N does without a batch)
    unsigned pbase = 4u * (unsigned)(n0 + 4 * (lane & 31)), rowb = 4u * (unsigned)A.N;
    if (A.hwimg) { const int p = n0 + 4 * (lane & 31), img = p / A.hwimg; pbase = 4u * ((unsigned)img * (unsigned)A.K * (unsigned)A.hwimg + (unsigned)(p - img * A.hwimg)); rowb = 4u * (unsigned)A.hwimg; }
#pragma unroll
    for (int q = 0; q < 2; q++) bvo[q] = (unsigned)((lane >> 5) + 2 * (4 * q + w)) * rowb + pbase;
    const unsigned bstep = 16u * rowb;
    const int wofs = g * B3_SLOT + w * 1024;
    auto issue = [&](int T, int slot) {                                   // this wave's five pieces of its group's step of slot T
        char* S = L + slot * SLOTB + wofs;
        const int t = G * T + g;
        const unsigned aoff = (unsigned)(NP * 1024) * (unsigned)t, boff = bstep * (unsigned)t;
#pragma unroll
        for (int q = 0; q < NP; q++) __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (__attribute__((address_space(3))) void*)(S + q * 4096), 16, avo, abase[q] + aoff, 0, 0);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)A.x + boff), 0, A.xbytes - boff, 0x00020000);
#pragma unroll
        for (int q = 0; q < 2; q++) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(S + NP * 4096 + q * 4096), 16, bvo[q], 0, 0, 0);
    };
    // two accumulator sets: the leading product w0 x0 (one rounding per 16 input channels) and the five corrections (2^-8 of it and below: their roundings do not count)
    f32x16 acc[4], acl[4];
#pragma unroll
    for (int rb = 0; rb < 4; rb++)
#pragma unroll
        for (int r = 0; r < 16; r++) { acc[rb][r] = 0.f; acl[rb][r] = 0.f; }
    for (int T = 0; T < RB - 1; T++) issue(min(T, nb - 1), T);
    u32x4 a[2][4][NP], bp[2][NP]; float braw[8];
    float xmax = 0.f;                                                     // (NP == 2) largest |x| this wave has split
    const f32x2 k2048 = {2048.f, 2048.f};
    typedef const __attribute__((address_space(3))) char* lds_c;
    const unsigned a_lane = (unsigned)(g * B3_SLOT) + 16u * (unsigned)lane, b_lane = (unsigned)(g * B3_SLOT) + (unsigned)(NP * 4096) + 4u * (unsigned)((lane >> 5) * 8 * C1_TN + 32 * w + (lane & 31));
    auto lda = [&](int slot, int buf) {
        lds_c Ab = (lds_c)(L + slot * SLOTB) + a_lane;
#pragma unroll
        for (int rb = 0; rb < 4; rb++)
#pragma unroll
            for (int pl = 0; pl < NP; pl++) a[buf][rb][pl] = *(const __attribute__((address_space(3))) u32x4*)(Ab + (rb * NP + pl) * 1024);
    };
    auto ldb = [&](int slot) {
#ifdef H2_PRESPLIT_EMU
        if (NP == 2) return;
#endif
        const __attribute__((address_space(3))) float* Bb = (const __attribute__((address_space(3))) float*)((lds_c)(L + slot * SLOTB) + b_lane);      // (not volatile: the compiler pairs the eight reads, 512 bytes apart, into four ds_read2st64_b32)
#pragma unroll
        for (int i = 0; i < 8; i++) braw[i] = Bb[i * C1_TN];
    };
    auto split = [&](int buf) {
        if constexpr (NP == 2) {
#ifdef H2_PRESPLIT_EMU      // (timing experiment: what the k-step would cost if the activations arrived already split — two wide LDS reads, no vector instruction; results are garbage)
            { lds_c Bq = (lds_c)(L + (buf ? SLOTB : 0)) + 16u * (unsigned)lane; bp[buf][0] = *(const __attribute__((address_space(3))) u32x4*)(Bq + 8192); bp[buf][1] = *(const __attribute__((address_space(3))) u32x4*)(Bq + 9216); }
            return;
#endif
#pragma unroll
            for (int pr = 0; pr < 4; pr++) {
                const f32x2 v = {braw[2 * pr], braw[2 * pr + 1]};
                const f16x2 h = __builtin_convertvector(v, f16x2);                                    // v_cvt_pk_f16_f32 (round to nearest even)
                // 2^11 (x - h) = fma(h, -2^11, 2^11 x), exact (the difference has <= 13 significant bits): one packed multiply + two v_fma_mix_f32, which read h's halves as they
                // are (left to itself the compiler converts h back with two more instructions and does not pack: 9 instead of 5 vector instructions per pair)
                f32x2 vs, r;
                asm("v_pk_mul_f32 %0, %1, %2" : "=v"(vs) : "v"(v), "v"(k2048));
                asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r.x) : "v"(h), "s"(-2048.f), "v"(vs.x));
                asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r.y) : "v"(h), "s"(-2048.f), "v"(vs.y));
                const f16x2 l = __builtin_convertvector(r, f16x2);
                bp[buf][0][pr] = __builtin_bit_cast(unsigned, h); bp[buf][1][pr] = __builtin_bit_cast(unsigned, l);
                xmax = __builtin_fmaxf(__builtin_fmaxf(xmax, __builtin_fabsf(v.x)), __builtin_fabsf(v.y));      // one v_max3_f32 with |.| operand modifiers
            }
        } else {
#pragma unroll
#ifdef B3_NOSPLIT
            for (int pr = 0; pr < 4; pr++) { bp[buf][0][pr] = __float_as_uint(braw[2 * pr]); bp[buf][1][pr] = __float_as_uint(braw[2 * pr + 1]); bp[buf][2][pr] = __float_as_uint(braw[2 * pr]) ^ 1; }
#else
            for (int pr = 0; pr < 4; pr++) { unsigned p0, p1, p2; b3_split2(braw[2 * pr], braw[2 * pr + 1], p0, p1, p2); bp[buf][0][pr] = p0; bp[buf][1][pr] = p1; bp[buf][2][pr] = p2; }
#endif
        }
    };
    // The barrier is the bare instruction (no fence): __syncthreads() would drain the copies in flight (vmcnt(0)), which is the ring's whole point.  What it must order is
    // stated explicitly: this wave's copies of the slot about to be read have landed (vmcnt), its own LDS reads are back (lgkmcnt(0): they were issued a slot ago).
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(B3_PW * (RB - 2)) : "memory");      // slot 0 has landed
    lda(0, 0); ldb(0); split(0);
    int slot = 0, ti = min(RB - 2, nb - 1);                               // ring slot of T; the last slot the copies have been issued for
    for (int T2 = 0; T2 < nb; T2 += 2) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(B3_PW * (RB - 3)) : "memory");      // slot T + 1 has landed; everybody is done reading slot T - 1
            __builtin_amdgcn_sched_barrier(0);
            const int sn = slot + 1 == RB ? 0 : slot + 1, sf = slot == 0 ? RB - 1 : slot - 1;      // ring positions of T + 1 and of T + RB - 1 (= the one T - 1 left)
            ti = min(ti + 1, nb - 1);                                     // (past the last slot the copies repeat it into a position nobody reads again: the count per slot stays 5)
            ldb(sn); lda(sn, h ^ 1);                                      // (past the last slot: read, never used)
            issue(ti, sf);
            // the products, small ones first; consecutive matrix instructions go to different accumulators
            if constexpr (NP == 2) {
#pragma unroll
                for (int term = 0; term < 3; term++) {
                    constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};
#pragma unroll
                    for (int rb = 0; rb < 4; rb++) {
                        f32x16& d = term == 2 ? acc[rb] : acl[rb];
#ifdef B3_NOMFMA                  // (timing experiment: the operand stream and the barriers alone; one matrix instruction per step keeps the operands alive)
                        if (term != 2 || rb != 0) continue;
#endif
                        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[h][rb][PA[term]]), __builtin_bit_cast(f16x8, bp[h][PB[term]]), d, 0, 0, 0);
                    }
                }
                split(h ^ 1);
                // 12 matrix instructions: the 16 LDS reads and the 4 copies beside the first six, the split's ~28 vector instructions beside the rest
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
#pragma unroll
                for (int i = 0; i < 12; i++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i < 4) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0); else if (i < 6) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    if (i >= 1 && i < 5) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    if (i >= 4) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                }
            } else {
#pragma unroll
            for (int term = 0; term < 6; term++) {
                constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
                for (int rb = 0; rb < 4; rb++) {
                    f32x16& d = term == 5 ? acc[rb] : acl[rb];
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[h][rb][PA[term]]), __builtin_bit_cast(bf16x8, bp[h][PB[term]]), d, 0, 0, 0);
                }
            }
            split(h ^ 1);
            // 24 matrix instructions of 32 cycles each: the 20 LDS reads and the 5 copies go beside the first twelve, the split's 44 vector-ALU instructions beside the rest
            // (their inputs are back from LDS by then)
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);            // the two LDS base addresses of the next slot
#pragma unroll
            for (int i = 0; i < 24; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i < 8) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); else if (i < 12) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (i >= 3 && i < 8) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if (i >= 6) __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
            }
            __builtin_amdgcn_sched_barrier(0);
            slot = sn;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");           // (the last, unused copies)
#pragma unroll
    for (int rb = 0; rb < 4; rb++) acc[rb] += NP == 2 ? acl[rb] * 0x1p-11f : acl[rb];
    if (NP == 2 && !(xmax < 65504.f) && A.range_flag) atomicOr(A.range_flag, 1u);      // (also a NaN)
    const float* wsc = (const float*)((const char*)A.wp + (size_t)4 * A.K * A.M);      // (NP == 2) the inverse channel scales behind the planes
    // D[i][j] as above: register r of a lane = output channel 8 (r / 4) + 4 (lane >> 5) + (r & 3) of the row block, position lane & 31
    // Epilogue in two phases per row block — the loads (bias, channel scale, residual), then sum + activation + stores — with the loads of the NEXT block issued before
    // the stores of this one: loads and stores share one in-order counter (vmcnt), so a load issued behind a store waits for that store's round trip; measured on the
    // 256 -> 256 layer at 200 x 272: 52.7 -> 40.7 us with the scale loads moved out from between the stores.
    struct Ep { float rv[16], bv[16], sv[16]; };
    const int q_out = n0 + 32 * w + (lane & 31);
    size_t rq = (size_t)q_out, rn = (size_t)A.N;
    if (RES == 2) { const int yy = q_out / A.W, xx = q_out - yy * A.W; rq = (size_t)(yy >> 1) * (A.W >> 1) + (xx >> 1); rn = (size_t)(A.N / A.W >> 1) * (A.W >> 1); }
    // RES 3: a 2 x 2 stride-2 TRANSPOSED convolution as this GEMM — rows (tap a b, channel co) = 4 c4, positions (image, i, j) —: row tile -> one tap (c4 % 128 == 0), the output
    // is y[img][co][2 i + a][2 j + b]; bias per co
    size_t oq = (size_t)q_out, on = (size_t)A.N; int co_sub = 0;
    if (RES == 3) {
        const int tap = m0 / A.c4, img = q_out / A.hwimg, rem = q_out - img * A.hwimg, ii = rem / A.W, jj = rem - ii * A.W;
        co_sub = tap * A.c4; on = (size_t)4 * A.hwimg;
        oq = (size_t)img * A.c4 * on + (size_t)(2 * ii + (tap >> 1)) * (2 * A.W) + 2 * jj + (tap & 1);
    }
    auto ep_load = [&](int rb, Ep& E) {
        const int co0 = m0 + 32 * rb + 4 * (lane >> 5);
        if (q_out < A.N) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int co = co0 + 8 * (r >> 2) + (r & 3);
                E.bv[r] = A.bias ? A.bias[co - co_sub] : 0.f;
                E.rv[r] = (RES == 1 || RES == 2) ? A.res[(size_t)co * rn + rq] : 0.f;
                E.sv[r] = NP == 2 ? wsc[co] : 1.f;
            }
        }
    };
    auto ep_store = [&](int rb, const f32x16& v0, const float* other, const Ep& E) {
        const int co0 = m0 + 32 * rb + 4 * (lane >> 5);
        float ov[16];
#pragma unroll
        for (int r = 0; r < 16; r++) ov[r] = other ? other[r * 64] : 0.f;
        if (q_out < A.N) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int co = co0 + 8 * (r >> 2) + (r & 3);
                const float v = (NP == 2 ? (v0[r] + ov[r]) * E.sv[r] : v0[r] + ov[r]) + E.bv[r] + E.rv[r];
                A.y[(size_t)(co - co_sub) * on + oq] = fmaxf(v, v * A.slope);
            }
        }
    };
    Ep E0, E1;
    if (G == 1) {
        ep_load(0, E0);
        ep_load(1, E1); ep_store(0, acc[0], nullptr, E0);
        ep_load(2, E0); ep_store(1, acc[1], nullptr, E1);
        ep_load(3, E1); ep_store(2, acc[2], nullptr, E0);
        ep_store(3, acc[3], nullptr, E1);
    } else {
        // the two groups' sums meet in LDS (the ring is free): group 1 hands over its row blocks 0, 1 and finishes 2, 3, group 0 the other way round
        float* X = (float*)L;                                             // [which 2][w 4][rbl 2][r 16][lane 64]
        if (g == 0) { ep_load(0, E0); ep_load(1, E1); } else { ep_load(2, E0); ep_load(3, E1); }      // (on their way during the exchange)
        __builtin_amdgcn_s_barrier();                                     // every wave is past its last ring read (lgkmcnt(0) above) and its last copy has landed
        auto put = [&](int which, const f32x16& v0, const f32x16& v1) {
            float* Pw = X + (((which * 4 + w) * 2) * 16) * 64 + lane;
#pragma unroll
            for (int r = 0; r < 16; r++) { Pw[r * 64] = v0[r]; Pw[(16 + r) * 64] = v1[r]; }
        };
        if (g == 1) put(0, acc[0], acc[1]); else put(1, acc[2], acc[3]);
        __syncthreads();
        const float* Pr = X + (((g * 4 + w) * 2) * 16) * 64 + lane;
        if (g == 0) { ep_store(0, acc[0], Pr, E0); ep_store(1, acc[1], Pr + 16 * 64, E1); }
        else { ep_store(2, acc[2], Pr, E0); ep_store(3, acc[3], Pr + 16 * 64, E1); }
    }
}
}  // namespace

extern "C" {

/* 1 when vido_conv1x1_bias_act takes the shape: output channels a multiple of 128, input channels a multiple of 32, H*W >= 128, activations below 4 GB. */
int vido_conv1x1_supported(int cin, int cout, int hw)
{
    return cin >= 32 && cin % 32 == 0 && cout >= C1_TM && cout % C1_TM == 0 && hw >= C1_TN && 4ll * cin * hw < (1ll << 32) && 4ll * cin * cout < (1ll << 32);
}

/* Which tile form (= which weight packing) the library uses for a shape: 0 = 128 x 128 tiles on 32x32x2 (pack layout 0), 1 = 128 x 112 tiles on 16x16x4 (pack layout 1).
 * The form with fewer (rounds of 256 CUs) x (tile width) wins; VIDO_CONV1X1_TN = 128 / 112 forces one.  The packer (vido_slam_amd/nets/ops.py::pack_conv1x1) asks this
 * function, vido_conv1x1_bias_act asks it again with the same shape. */
static std::atomic<int>& c1_arith()          // 0 = split-fp16 (two planes, three products: the default), 1 = the fp32 instruction, 2 = split-bf16 (three planes, six products)
{
    static std::atomic<int> v{[] { const char* e = getenv("VIDO_CONV1X1_ARITH");
                                   return (e && !strcmp(e, "f32")) || (!e && getenv("VIDO_CONV1X1_TN")) ? 1 : (e && (!strcmp(e, "bf16x3") || !strcmp(e, "bf16"))) ? 2 : 0; }()};
    return v;
}

int vido_conv1x1_set_arith(int arith) { return c1_arith().exchange(arith == 1 ? 1 : arith == 2 ? 2 : 0); }

/* The range flag of the split-fp16 form: non-zero when a launch since the last reset met an activation with |x| >= 65504 (or a NaN) — its outputs are then not valid.
 * Host-visible memory written by the kernels: read it after the stream has been waited for.  reset != 0 clears it. */
int vido_conv1x1_range_flag(vido_ctx* ctx, int reset)
{
    if (!ctx || !ctx->c1_range_flag) return 0;
    const int v = (int)__atomic_load_n(ctx->c1_range_flag, __ATOMIC_ACQUIRE);
    if (reset) __atomic_store_n(ctx->c1_range_flag, 0u, __ATOMIC_RELEASE);
    return v;
}

int vido_conv1x1_layout(int cin, int cout, int hw)
{
    // Default 128: measured on the pipelined headline (two A/B pairs of 100 steps, profiles/r5/conv1x1_tile_form_ab.txt) the 112-wide form costs 2.5 % (88.3 -> 86.1 frames/s)
    // although the detector ALONE gets faster (8.87 -> 8.65 ms): beside two other streams the CUs a 216-tile launch leaves idle are not idle — they run LiteFlowNet and the
    // tracker — and the 112-wide form needs the same matrix time on 248 CUs plus more LDS reads per matrix instruction.  VIDO_CONV1X1_TN=0 lets the rounds rule below decide
    // (a detector running alone), 112 / 128 force a form.
    // round 6: the split-bf16 form (k_conv1x1_b3: fp32-equivalent arithmetic on the bf16 matrix instruction) is the default; VIDO_CONV1X1_ARITH=f32 (or a forced
    // VIDO_CONV1X1_TN) brings the fp32-instruction forms back.
    { const int ar = c1_arith().load(std::memory_order_relaxed); if (ar != 1) return ar == 2 ? 2 : 3; }
    static const int force = [] { const char* e = getenv("VIDO_CONV1X1_TN"); return e ? atoi(e) : 128; }();
    if (force == 128 || cin % 64 != 0) return 0;
    if (force == 112) return 1;
    const long long mt = cout / C1_TM, t128 = mt * ((hw + 127) / 128), t112 = mt * ((hw + C2_TN - 1) / C2_TN);
    const long long c128 = ((t128 + 255) / 256) * 128, c112 = ((t112 + 255) / 256) * C2_TN;
    return c112 < c128 ? 1 : 0;
}

/* y = leaky_relu(conv2d(x, w) + bias + residual, slope) for one image, 1x1 kernel, stride 1: x [cin][hw], y / residual [cout][hw] f32 DEVICE tensors (16-byte aligned,
 * y != x), bias [cout] or NULL, residual NULL when there is none.  w_packed: the weight [cout][cin] in operand order — layout vido_conv1x1_layout(cin, cout, hw):
 *   0: element (co, k) at [co / 32][k / 8][32 * (k & 1) + co % 32][(k % 8) / 2];   1: at [co / 16][k / 16][16 * (k & 3) + co % 16][(k % 16) / 4]
 * (vido_slam_amd/nets/ops.py::pack_conv1x1).  slope 0 = ReLU, 1 = none (0 <= slope <= 1).  Enqueues on the adopted stream; capturable. */
static int c1_launch(vido_ctx* ctx, const float* x, const float* w_packed, const float* bias, const float* residual, int res_mode, float* y, int cin, int cout, int hw, int w, float slope,
                     int hwimg = 0, int c4 = 0)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!x || !w_packed || !y || x == y || !vido_conv1x1_supported(cin, cout, hw) || slope < 0.f || slope > 1.f || (((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual) & 3) || ((uintptr_t)w_packed & 15))
        return vido_set_error(ctx, VIDO_E_INVALID, "conv1x1: no kernel for %d -> %d channels at %d positions (or a pointer is misaligned, or slope outside [0, 1])", cin, cout, hw);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    const int rm = res_mode == 3 ? 3 : (residual ? res_mode : 0);
    const int layout = vido_conv1x1_layout(cin, cout, hw);
    if ((hwimg || rm == 3) && layout != 3) return vido_set_error(ctx, VIDO_E_INVALID, "conv1x1: the batch / transposed-convolution forms exist in the split-fp16 arithmetic only");
    if (layout >= 2) {
        const int np = layout == 3 ? 2 : 3;
        const int mt = cout / C1_TM, ntl = (hw + C1_TN - 1) / C1_TN, total = mt * ntl;
        C1Args A{x, w_packed, bias, residual, y, cout, hw, cin, mt, total, cin / 16, slope, (unsigned)(4ll * cin * hw), (unsigned)(2ll * np * cin * cout), w, ctx->c1_range_flag, hwimg, c4};
        // form: VIDO_CONV1X1_B3_FORM = 0 (default: by shape), 1 = <1, 6>, 2 = <2, 4>, 3 = <1, 4> two workgroups per CU
        static const int force_form = [] { const char* e = getenv("VIDO_CONV1X1_B3_FORM"); return e ? atoi(e) : 0; }();
        int form = force_form ? force_form : (cin >= 512 && total <= 320 ? 2 : 3);
        if (form == 2 && cin % 64) form = 3;                                // (two groups: an even number of slots of two k-steps)
        // ring depths: the bf16 form 4 slots of 20 KB (x 2 groups = 160 KB; one group: 80 KB, two workgroups per CU); the fp16 form's k-step is half as long and its slot
        // 16 KB: 5 slots (x 2 groups = 160 KB; one group: 80 KB) keep the copies as far ahead in time
        static bool attr3[64] = {};
        if (!attr3[ctx->device & 63]) {
#define B3_ATTR(G_, RB_, NP_) for (const void* f : {(const void*)k_conv1x1_b3<0, G_, RB_, NP_>, (const void*)k_conv1x1_b3<1, G_, RB_, NP_>, (const void*)k_conv1x1_b3<2, G_, RB_, NP_>}) \
            HIP_TRY(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)RB_ * G_ * B3_SLOTB(NP_))))
            B3_ATTR(1, 6, 3); B3_ATTR(2, 4, 3); B3_ATTR(1, 4, 3); B3_ATTR(2, 5, 2); B3_ATTR(1, 5, 2); B3_ATTR(1, 4, 2); B3_ATTR(1, 3, 2);
#undef B3_ATTR
            HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_conv1x1_b3<3, 2, 5, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)5 * 2 * B3_SLOTB(2))));
            HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_conv1x1_b3<3, 1, 5, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)5 * 1 * B3_SLOTB(2))));
            attr3[ctx->device & 63] = true;
        }
        const dim3 grid(8 * ((total + 7) / 8));
#define B3_LAUNCH(G_, RB_, NP_) do { const dim3 blk(256 * G_); const size_t lds = (size_t)RB_ * G_ * B3_SLOTB(NP_); \
            if (rm == 2) hipLaunchKernelGGL((k_conv1x1_b3<2, G_, RB_, NP_>), grid, blk, lds, st, A); else if (rm) hipLaunchKernelGGL((k_conv1x1_b3<1, G_, RB_, NP_>), grid, blk, lds, st, A); \
            else hipLaunchKernelGGL((k_conv1x1_b3<0, G_, RB_, NP_>), grid, blk, lds, st, A); } while (0)
        static const int rb_short = [] { const char* e = getenv("VIDO_CONV1X1_H2_RB"); return e ? atoi(e) : 5; }();      // (experiment: ring depth of the one-group fp16 form)
        if (rm == 3) {      // (the transposed convolution: the fp16 form's two launch forms)
            if (form == 2) hipLaunchKernelGGL((k_conv1x1_b3<3, 2, 5, 2>), grid, dim3(512), (size_t)5 * 2 * B3_SLOTB(2), st, A);
            else hipLaunchKernelGGL((k_conv1x1_b3<3, 1, 5, 2>), grid, dim3(256), (size_t)5 * B3_SLOTB(2), st, A);
        }
        else if (np == 2) { if (form == 2) B3_LAUNCH(2, 5, 2); else if (rb_short == 4) B3_LAUNCH(1, 4, 2); else if (rb_short == 3) B3_LAUNCH(1, 3, 2); else B3_LAUNCH(1, 5, 2); }
        else if (form == 2) B3_LAUNCH(2, 4, 3); else if (form == 1) B3_LAUNCH(1, 6, 3); else B3_LAUNCH(1, 4, 3);
#undef B3_LAUNCH
        HIP_TRY(ctx, hipGetLastError());
        return VIDO_OK;
    }
    if (layout == 1) {
        const int mt = cout / C1_TM, ntl = (hw + C2_TN - 1) / C2_TN, total = mt * ntl;
        C1Args A{x, w_packed, bias, residual, y, cout, hw, cin, mt, total, cin / 64, slope, (unsigned)(4ll * cin * hw), (unsigned)(4ll * cin * cout), w};
        constexpr size_t LDS112 = (size_t)2 * (8 * 4 * 256 + 2 * 16 * 272) * 4;
        static bool attr2[64] = {};
        if (!attr2[ctx->device & 63]) {
            for (const void* f : {(const void*)k_conv1x1_n112<0, 64>, (const void*)k_conv1x1_n112<1, 64>, (const void*)k_conv1x1_n112<2, 64>}) HIP_TRY(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS112));
            attr2[ctx->device & 63] = true;
        }
        const dim3 grid(8 * ((total + 7) / 8)), blk(256);
        if (rm == 2) hipLaunchKernelGGL((k_conv1x1_n112<2, 64>), grid, blk, LDS112, st, A); else if (rm) hipLaunchKernelGGL((k_conv1x1_n112<1, 64>), grid, blk, LDS112, st, A); else hipLaunchKernelGGL((k_conv1x1_n112<0, 64>), grid, blk, LDS112, st, A);
        HIP_TRY(ctx, hipGetLastError());
        return VIDO_OK;
    }
    const int mt = cout / C1_TM, ntl = (hw + C1_TN - 1) / C1_TN, total = mt * ntl;
    static const int force_kc = [] { const char* e = getenv("VIDO_CONV1X1_KC"); return e ? atoi(e) : 0; }();
    const int kc = (cin % 64 == 0 && force_kc != 32) ? 64 : 32;
    C1Args A{x, w_packed, bias, residual, y, cout, hw, cin, mt, total, cin / kc, slope, (unsigned)(4ll * cin * hw), (unsigned)(4ll * cin * cout), w};
    const dim3 grid(8 * ((total + 7) / 8)), blk(256);
    constexpr size_t LDS64 = (size_t)2 * (4 * 8 * 256 + 64 * C1_TN) * 4, LDS32 = (size_t)2 * (4 * 4 * 256 + 32 * C1_TN) * 4;
    static bool attr[64] = {};
    if (!attr[ctx->device & 63]) {
        for (const void* f : {(const void*)k_conv1x1<0, 64>, (const void*)k_conv1x1<1, 64>, (const void*)k_conv1x1<2, 64>}) HIP_TRY(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS64));
        for (const void* f : {(const void*)k_conv1x1<0, 32>, (const void*)k_conv1x1<1, 32>, (const void*)k_conv1x1<2, 32>}) HIP_TRY(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS32));
        attr[ctx->device & 63] = true;
    }
    if (kc == 64) { if (rm == 2) hipLaunchKernelGGL((k_conv1x1<2, 64>), grid, blk, LDS64, st, A); else if (rm) hipLaunchKernelGGL((k_conv1x1<1, 64>), grid, blk, LDS64, st, A); else hipLaunchKernelGGL((k_conv1x1<0, 64>), grid, blk, LDS64, st, A); }
    else { if (rm == 2) hipLaunchKernelGGL((k_conv1x1<2, 32>), grid, blk, LDS32, st, A); else if (rm) hipLaunchKernelGGL((k_conv1x1<1, 32>), grid, blk, LDS32, st, A); else hipLaunchKernelGGL((k_conv1x1<0, 32>), grid, blk, LDS32, st, A); }
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

/* 2 x 2 stride-2 transposed convolution + bias + leaky ReLU of a BATCH as one split-fp16 GEMM (round 6): the mask head's conv5_mask over the detections' 14 x 14 maps
 * (maskrcnn_benchmark/modeling/roi_heads/mask_head/roi_mask_predictors.py:17-31).  y[n][co][2 i + a][2 j + b] = act(sum_ci x[n][ci][i][j] w[ci][co][a][b] + bias[co]):
 * GEMM rows (tap a b, co) = 4 cout, columns (image, i, j); x [n][cin][h][w], y [n][cout][2 h][2 w] f32 DEVICE.  w_packed: pack_conv1x1 layout 3 of the matrix
 * [(2 a + b) cout + co][ci] (vido_slam_amd/nets/ops.py::pack_deconv2x2).  Shapes: cout % 128 == 0, cin % 32 == 0, (h w) % 4 == 0, n h w >= 128. */
int vido_deconv2x2_supported(int n, int cin, int cout, int h, int w)
{
    return n >= 1 && h >= 1 && w >= 1 && cout >= 128 && cout % 128 == 0 && (h * w) % 4 == 0 && (long long)n * h * w < (1ll << 28) && vido_conv1x1_supported(cin, 4 * cout, n * h * w)
           && 16ll * n * cout * h * w < (1ll << 32) && vido_conv1x1_layout(cin, 4 * cout, n * h * w) == 3;
}
int vido_deconv2x2_bias_act(vido_ctx* ctx, const float* x, const float* w_packed, const float* bias, float* y, int n, int cin, int cout, int h, int w, float slope)
{
    if (ctx && !vido_deconv2x2_supported(n, cin, cout, h, w)) return vido_set_error(ctx, VIDO_E_INVALID, "deconv2x2: no kernel for %d x %d -> %d channels at %d x %d", n, cin, cout, h, w);
    return c1_launch(ctx, x, w_packed, bias, nullptr, 3, y, cin, 4 * cout, n * h * w, w, slope, h * w, cout);
}

int vido_conv1x1_bias_act(vido_ctx* ctx, const float* x, const float* w_packed, const float* bias, const float* residual, float* y, int cin, int cout, int hw, float slope)
{
    return c1_launch(ctx, x, w_packed, bias, residual, 1, y, cin, cout, hw, hw, slope);
}

/* The same with the residual at HALF the resolution, added nearest-upsampled: y[co][yy][xx] = act(conv + bias[co] + residual_half[co][yy / 2][xx / 2]) — the FPN's
 * lateral convolution + top-down sum (maskrcnn_benchmark/modeling/backbone/fpn.py:55-66) as one launch.  h, w even; residual_half [cout][h / 2][w / 2]. */
int vido_conv1x1_bias_up2_act(vido_ctx* ctx, const float* x, const float* w_packed, const float* bias, const float* residual_half, float* y, int cin, int cout, int h, int w, float slope)
{
    if (ctx && (h < 2 || w < 2 || (h & 1) || (w & 1) || !residual_half)) return vido_set_error(ctx, VIDO_E_INVALID, "conv1x1 (upsampled residual): %d x %d must be even and the residual given", h, w);
    return c1_launch(ctx, x, w_packed, bias, residual_half, 2, y, cin, cout, h * w, w, slope);
}

}  // extern "C"
