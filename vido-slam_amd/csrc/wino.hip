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
//
// K-split form (KSPL; round 5) for launches whose tile form would leave most of the chip idle (fewer than 128 workgroups: FPN P4-P6, LiteFlowNet's levels 3-6).  A wave of the
// tile form walks ALL input channels of its 32 x 32 outputs at one wave per SIMD, so a 16-workgroup launch runs as long as a chip-filling one.  Here a workgroup is 32
// channels x 32 tiles and its four waves take the 4-channel chunks w, w + 4, ... with a full set of sixteen accumulators each (the 1 x 4 form with its four tile blocks
// re-read as four channel slices, KC = 4 packing for every cout); the inverse transform is linear, so every wave transforms its own partial M, waves 1-3 leave 64 floats a
// lane in LDS and wave 0 adds them in a fixed order before the activation.  Four times the waves, a quarter of the chain each: FPN P4 73 -> 39 us, flow level 4 57 -> 34 us
// (library + its bias pass: 53 / 42; profiles/r5/wino_ksplit_microbench.txt).
// What bounds it: every wave pulls its OWN U block — 8 KB per 32 matrix instructions (0.98 us) per wave, 8.6 TB/s over 1024 waves: the L2's rate, where the tile forms share
// a U block among 2-4 tile blocks through LDS.  Tried and dropped (same file, round 5): a barrier-free variant (the lane that transforms a window IS the lane that holds that
// B operand, so V can stay in registers and U be read straight into operand registers: no LDS, no barrier in the loop) — slower on every shape (P4 53 us, level 5 50 vs 46):
// the barrier was never the bound, the U traffic is, and register loads of U are slower than the asynchronous LDS copies.  Two compiler traps met on the way, kept here for
// the next kernel: (1) __builtin_amdgcn_raw_buffer_load_b64 / _b128 whose result is bit-cast ELEMENT by element is shrunk to a 4-byte load (bit-cast the whole vector);
// (2) a vector register written by INLINE-ASSEMBLY v_pk_add_f32 and read as the B operand of the next matrix instruction gets no wait states from the hazard recognizer
// (it does not look inside inline assembly): wrong sums in one template instance, right in the other — an explicit s_nop 1 in the asm string fixes it.
#include "common.hpp"

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define WN_OOB 0x40000000u
#ifndef WN_ABLATE
#define WN_ABLATE 0          // timing experiments (tools/ubench/wino_ablate.hip): 1 no transform, 2 no window loads, 4 no U copies, 8 no barrier, 16 no operand reads
#endif
struct WnArgs { const float* x; const float* up; const float* bias; float* y; int N, Cin, Cout, H, W, Ht, Wt, T, nchunk, cgroups, total; float slope; int vec2; unsigned xbytes, ubytes; unsigned long long* prof; };

// packed fp32 add / subtract of register pairs (asm: the compiler splits a packed subtract into two single ones, and a vector instruction costs matrix time — see below)
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }

template <int CW, int TW, int KC, bool ODD, int KSPL = 0>      // KSPL: 0 = tile form; 4 / 2 = K-split form with that many channel slices (4 / KSPL tile blocks per workgroup)
__global__ __launch_bounds__(256) void k_wino3x3(WnArgs A)
{
    static_assert(CW * TW == 4 && (KC == 8 || KC == 4), "workgroup = 4 waves");
    static_assert(!KSPL || ((KSPL == 4 || KSPL == 2) && CW == 1 && TW == 4 && KC == 4), "the K-split form is the 1 x 4 form with its four tile blocks re-read as (tile block, channel slice) pairs");
    constexpr int KSL = KSPL ? KSPL : 1, KTB = KSPL ? 4 / KSPL : 1;      // channel slices, tile blocks of a K-split workgroup: wave w = KSL * (tile block) + slice
    constexpr int KS = KC / 2;                        // matrix instructions (k-pairs) per position and chunk
    constexpr int U_BLK = 16 * 64 * KS;               // floats of one (32-channel block, chunk): [p][lane][KS]
    constexpr int UB = KSPL ? KSL : CW;               // U blocks per chunk in LDS (K-split: one per channel slice)
    constexpr int U_BUF = UB * U_BLK, V_BUF = 16 * TW * KS * 64, V_P = TW * KS * 64;
    constexpr int NJ = TW * KS / 4;                   // (channel, tile) pairs a thread transforms per chunk
    constexpr int KSTEP = 4 / TW;
    constexpr int PPB = U_BLK * 4 / 1024;             // 1 KB copy pieces per U block
    static_assert(NJ == 2, "the slices below are written for two pairs per thread");
    extern __shared__ __attribute__((aligned(16))) float wn_lds[];      // [2][U_BUF] [2][V_BUF]
    float* Ul = wn_lds; float* Vl = wn_lds + 2 * U_BUF;
    // Measured on this chip (tools/ubench/mfma_fillers*.hip): beside v_mfma_f32_32x32x2f32 EVERY vector-ALU instruction of the same wave costs ~5.5 cycles of matrix time
    // (add, move, integer, packed alike — nothing hides), a buffer load with a 32-bit offset ~1.5, a 1 KB global -> LDS copy ~4, LDS reads < 1.  So the loop below keeps
    // vector instructions to the transform itself (packed: two adds per instruction) and does all addressing with SCALAR registers: the wave index is made scalar here,
    // per-lane byte offsets are loop invariants, everything that changes with the chunk rides in the loads' scalar offset.
#ifdef WN_PROF
    unsigned long long tk[4] = {0, 0, 0, 0};                             // (debug build: kernel entry, loop entry, loop exit, after the stores)
    asm volatile("s_memtime %0" : "=s"(tk[0]));
#endif
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int per = gridDim.x >> 3, item = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (item >= A.total) return;
    const int tb = item / A.cgroups, cg = item - tb * A.cgroups, tile0 = tb * (KSPL ? KTB * 32 : TW * 32);
    const int hw = A.H * A.W, tpi = A.Ht * A.Wt;

    // ---- transform role: tile block tw_t, k-pairs ks0 + j * KSTEP; lane = (tile & 31) + 32 * (channel & 1) = the operand slot it fills.
    // Input windows are read with BUFFER loads: voff[e] = byte offset of window element e of channel (lane >> 5) of the tile's image; the k-pair's channel base is the
    // scalar offset.  An element outside the image (zero padding) or of a tile past the end has bit 30 set in voff — past the descriptor's range (the tensor is < 1 GB):
    // the hardware returns 0, no select.  A k-pair past Cin reads through a descriptor of size 0 (all zeros); the odd channel past an odd Cin (ODD) gets bit 30 per lane.
    const int tw_t = w % TW, ks0 = w / TW;
    unsigned voff[16];
    {
        const int gt = tile0 + (KSPL ? (w / KSL) * 32 : tw_t * 32) + (lane & 31), gtc = min(gt, A.T - 1);
        const int n = gtc / tpi, rem = gtc - n * tpi, ty = rem / A.Wt, tx = rem - ty * A.Wt;
        unsigned rowo[4], colo[4];
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const int iy = 2 * ty - 1 + d, ix = 2 * tx - 1 + d;
            rowo[d] = (gt < A.T && iy >= 0 && iy < A.H) ? 4u * (((unsigned)n * (unsigned)A.Cin + (unsigned)(lane >> 5)) * (unsigned)hw + (unsigned)(iy * A.W)) : WN_OOB;
            colo[d] = (ix >= 0 && ix < A.W) ? 4u * (unsigned)ix : WN_OOB;
        }
#pragma unroll
        for (int e = 0; e < 16; e++) voff[e] = rowo[e >> 2] + colo[e & 3];
    }
    const unsigned par_oob = (lane >> 5) ? WN_OOB : 0u;
    f32x2 inp[NJ][8];                                                    // window of pair j: inp[j][2 * row + half] = columns (2 half, 2 half + 1)
    auto load_win = [&](int chunk, int j, int e0, int e1) {              // window elements [e0, e1) of pair j of `chunk`
        const int cb = (KSPL ? KSL * chunk + w % KSL : chunk) * KC + 2 * (ks0 + j * KSTEP);      // (scalar) first channel of the k-pair; K-split: the wave's slice of the 4 KSL-channel chunk
        const bool any = cb < A.Cin;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)A.x, 0, any ? A.xbytes : 0u, 0x00020000);
        const unsigned so = any ? 4u * (unsigned)cb * (unsigned)hw : 0u;
        const unsigned extra = (ODD && cb == A.Cin - 1) ? par_oob : 0u;
#pragma unroll
        for (int e = e0; e < e1; e++)
            inp[j][e >> 1][e & 1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, ODD ? voff[e] + extra : voff[e], so, 0));
    };
    // V = B^T d B of pair j in two steps: rows (R = B^T d, packed over column pairs: the window registers are dead afterwards), then output row i -> LDS.  The column
    // step mixes the halves of a register pair: (v0, v1) = (r0 - r2, r1 + r2), (v2, v3) = (r2 - r1, r1 - r3) are ONE packed add each through the operand selects /
    // negations of v_pk_add_f32 (the compiler turns the same expression into moves and xors).
    f32x2 R[NJ][8];
    auto xf_rows = [&](int j) {
        const f32x2* D = inp[j];
#pragma unroll
        for (int h = 0; h < 2; h++) {                                    // (asm: left to itself the compiler splits these into single adds)
            asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(R[j][h]) : "v"(D[h]), "v"(D[4 + h]));
            asm("v_pk_add_f32 %0, %1, %2" : "=v"(R[j][2 + h]) : "v"(D[2 + h]), "v"(D[4 + h]));
            asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(R[j][4 + h]) : "v"(D[4 + h]), "v"(D[2 + h]));
            asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(R[j][6 + h]) : "v"(D[2 + h]), "v"(D[6 + h]));
        }
    };
    f32x2 Vo[NJ][8];                                                     // transformed window of pair j: Vo[j][2 i] = (v[i][0], v[i][1]), Vo[j][2 i + 1] = (v[i][2], v[i][3])
    auto xf_cols = [&](int j) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(Vo[j][2 * i]) : "v"(R[j][2 * i]), "v"(R[j][2 * i + 1]));
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]" : "=v"(Vo[j][2 * i + 1]) : "v"(R[j][2 * i]), "v"(R[j][2 * i + 1]));
        }
    };
    auto xf_out = [&](int j, int i, int buf) {
        float* dst = Vl + buf * V_BUF + (tw_t * KS + ks0 + j * KSTEP) * 64 + lane;
        dst[(4 * i + 0) * V_P] = Vo[j][2 * i].x; dst[(4 * i + 1) * V_P] = Vo[j][2 * i].y; dst[(4 * i + 2) * V_P] = Vo[j][2 * i + 1].x; dst[(4 * i + 3) * V_P] = Vo[j][2 * i + 1].y;
    };
    // U pieces travel global -> LDS as BUFFER loads with the lds bit (16 bytes per lane; the flat-encoded global_load_lds makes the compiler fall back to vmcnt(0) /
    // lgkmcnt(0) in front of every later use of any loaded register): per-lane offset 16 * lane, everything else scalar
    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc((void*)A.up, 0, A.ubytes, 0x00020000);
    const unsigned uvo = 16u * (unsigned)lane;
    auto issue_u = [&](int chunk, int buf, int first, int count) {        // wave w moves pieces w, w + 4, ... of the CW blocks of this chunk: `count` of them from `first`
#pragma unroll
        for (int q = first; q < first + count; q++) {
            const int i = 4 * q + w, cwi = i / PPB, pi = i - cwi * PPB;
            // (K-split: block cwi is the slice's 4-channel chunk; a slice past the last chunk re-reads the last one — its V is zero)
            const unsigned so = KSPL ? 4u * (unsigned)((cg * A.nchunk + min(KSL * chunk + cwi, A.nchunk - 1)) * U_BLK + pi * 256)
                                     : 4u * (unsigned)(((cg * CW + cwi) * A.nchunk + chunk) * U_BLK + pi * 256);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ur, (__attribute__((address_space(3))) void*)(Ul + buf * U_BUF + cwi * U_BLK + pi * 256), 16, uvo, so, 0, 0);
        }
    };
    constexpr int NPW = UB * PPB / 4;                                    // pieces per wave and chunk

    // ---- matrix role: channel block cw, tile block tw
    const int cw = KSPL ? w % KSL : w % CW, tw = w / CW;                 // operand blocks in LDS (K-split: U block = the wave's channel slice, V block = the wave itself)
    const int cwo = KSPL ? 0 : cw, two = KSPL ? w / KSL : tw;            // output blocks
    const int ksl = KSPL ? w % KSL : 0;                                  // channel slice of the wave (0: the wave that adds the partial sums and stores)
    const int nloop = KSPL ? (A.nchunk + KSL - 1) / KSL : A.nchunk;
    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; p++)
#pragma unroll
        for (int q = 0; q < 16; q++) acc[p][q] = 0.f;
    if (A.bias && !(KSPL && ksl)) {                                        // Y = A^T M A: M[1][1] reaches all four outputs of a tile with weight 1 -> the bias starts there
#pragma unroll
        for (int q = 0; q < 16; q++) { const int co = (cg * CW + cwo) * 32 + 4 * (lane >> 5) + 8 * (q >> 2) + (q & 3); acc[5][q] = co < A.Cout ? A.bias[co] : 0.f; }
    }

    load_win(0, 0, 0, 16); load_win(0, 1, 0, 16);
    issue_u(0, 0, 0, NPW);
    xf_rows(0); xf_rows(1);
    load_win(1, 0, 0, 16); load_win(1, 1, 0, 16);
    xf_cols(0); xf_cols(1);
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int i = 0; i < 4; i++) xf_out(j, i, 0);
    typedef float uvec __attribute__((ext_vector_type(KS)));
#ifdef WN_PROF
    unsigned long long ts[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};          // (debug build: shader-clock stamps of the LAST chunk: loop top, after the barrier, after each slice)
#define WN_STAMP(k) asm volatile("s_memtime %0" : "=s"(ts[k]))
#else
#define WN_STAMP(k) do { } while (0)
#endif
#ifdef WN_PROF
    asm volatile("s_memtime %0" : "=s"(tk[1]));
#endif
    for (int c = 0; c < nloop; c++) {
        WN_STAMP(0);
        if (!(WN_ABLATE & 8)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                             // U(c), V(c) are in buffer c & 1; everybody is done with buffer (c + 1) & 1
        }
        WN_STAMP(1);
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
        ldops(0); ldops(1);                                              // (with WN_ABLATE & 16 these stay the only operand reads of the chunk)
#define WN_XF(...) do { if (!(WN_ABLATE & 1)) { __VA_ARGS__; } } while (0)
#define WN_LD(...) do { if (!(WN_ABLATE & 2)) { __VA_ARGS__; } } while (0)
        // the transform's 32 packed adds sit HERE, in front of the first matrix instruction: they run while the first operands are on their way from LDS (the matrix pipe
        // is idle then anyway); anywhere later each of them takes 5.5 cycles away from it.  They are the only reads of the window registers: every wait the compiler puts in
        // front of them finds loads that are at least three slices old.
        WN_XF(xf_rows(0); xf_rows(1); xf_cols(0); xf_cols(1));
        // Eight slices of 2 KS matrix instructions (positions 2 g, 2 g + 1), operands requested two slices ahead.  The side work of the chunk is dealt over the slices by
        // hand: slices 1-4 refill the windows (chunk c + 2; past the end they read zeros) EIGHT loads at a time — 32 gathers issued at once by four
        // waves fill the address unit's queue and the slice that issues them takes 1800 cycles instead of 512 (shader-clock stamps, tools/ubench/wino_ablate.hip) — and
        // write the transform of chunk c + 1 into the other V buffer, slices 5-6 send for the next U block (past the last chunk: the last block again, into the
        // buffer nobody reads any more).
#pragma unroll
        for (int g = 0; g < 8; g++) {
            __builtin_amdgcn_sched_barrier(0);
            if (g + 2 < 8 && !(WN_ABLATE & 16)) ldops(g + 2);
            if (g >= 1 && g <= 4) { WN_LD(load_win(c + 2, (g - 1) >> 1, 8 * ((g - 1) & 1), 8 * ((g - 1) & 1) + 8)); }      // eight window loads per slice: one per matrix instruction
            if (g == 1) { WN_XF(xf_out(0, 0, nb); xf_out(0, 1, nb)); }
            if (g == 2) { WN_XF(xf_out(0, 2, nb); xf_out(0, 3, nb)); }
            if (g == 3) { WN_XF(xf_out(1, 0, nb); xf_out(1, 1, nb)); }
            if (g == 4) { WN_XF(xf_out(1, 2, nb); xf_out(1, 3, nb)); }
            if (g == 5 && !(WN_ABLATE & 4)) issue_u(min(c + 1, nloop - 1), nb, 0, NPW / 2);
            if (g == 6 && !(WN_ABLATE & 4)) issue_u(min(c + 1, nloop - 1), nb, NPW / 2, NPW - NPW / 2);
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                acc[2 * g] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[g % 3][0][ks], vb[g % 3][0][ks], acc[2 * g], 0, 0, 0);
                acc[2 * g + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[g % 3][1][ks], vb[g % 3][1][ks], acc[2 * g + 1], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 2 * KS; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            if (g < 7) { __builtin_amdgcn_sched_barrier(0); WN_STAMP(2 + g); }
        }
        __builtin_amdgcn_sched_barrier(0);
        WN_STAMP(9);
    }
#ifdef WN_PROF
    asm volatile("s_memtime %0" : "=s"(tk[2]));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (A.prof && lane == 0 && (item == 0 || item == A.total / 2)) for (int k = 0; k < 10; k++) A.prof[((item ? 4 : 0) + w) * 10 + k] = ts[k];
#endif

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // (the last, unused copy and loads)
    // ---- inverse transform Y = A^T M A, bias, activation, store.  D[i][j]: register r of a lane = output channel 8 (r / 4) + 4 (lane >> 5) + (r & 3), tile lane & 31.
    // K-split: the four waves hold partial M of the SAME outputs; the inverse transform is linear, so waves 1-3 leave their transformed partial sums (64 floats a lane)
    // in LDS and wave 0 adds them in a fixed order (1, 2, 3) before the activation.
    f32x2* red = (f32x2*)wn_lds;                                         // [3][8][4][64] pairs of floats (48 KB; the operand buffers are dead)
    if (KSPL) __syncthreads();                                           // everybody has read its last operands
    const int gt = tile0 + two * 32 + (lane & 31);
    if (!KSPL && gt >= A.T) return;
    const int gtc = min(gt, A.T - 1);
    const int n = gtc / tpi, rem = gtc - n * tpi, ty = rem / A.Wt, tx = rem - ty * A.Wt;
    const bool row1 = 2 * ty + 1 < A.H, col1 = 2 * tx + 1 < A.W;
    const int cob = (cg * CW + cwo) * 32 + 4 * (lane >> 5);
    float* yb = A.y + ((size_t)n * A.Cout * A.H + 2 * ty) * A.W + 2 * tx;
    // (the bias is already in the accumulators: position (1, 1) of M contributes 1 to all four outputs of a tile, so acc[5] started from bias[co] instead of 0.)
    // Output channels two at a time (registers r, r + 1 of every accumulator): packed adds / multiplies; leaky ReLU as max(y, slope * y) (0 <= slope <= 1).
    const f32x2 sl = {A.slope, A.slope};
    const bool full = __all((A.vec2 != 0) & row1 & (gt < A.T)) && (cg * CW + cwo) * 32 + 32 <= A.Cout;       // whole wave: every lane stores both rows of all 32 channels as 8-byte pairs
    auto inverse = [&](int r, f32x2& y00, f32x2& y01, f32x2& y10, f32x2& y11) {
        f32x2 t0[4], t1[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const f32x2 m0 = {acc[j][r], acc[j][r + 1]}, m1 = {acc[4 + j][r], acc[4 + j][r + 1]}, m2 = {acc[8 + j][r], acc[8 + j][r + 1]}, m3 = {acc[12 + j][r], acc[12 + j][r + 1]};
            t0[j] = pk_add(pk_add(m0, m1), m2); t1[j] = pk_sub(pk_sub(m1, m2), m3);
        }
        y00 = pk_add(pk_add(t0[0], t0[1]), t0[2]); y01 = pk_sub(pk_sub(t0[1], t0[2]), t0[3]); y10 = pk_add(pk_add(t1[0], t1[1]), t1[2]); y11 = pk_sub(pk_sub(t1[1], t1[2]), t1[3]);
    };
    if (KSPL && ksl) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            f32x2 y00, y01, y10, y11; inverse(r, y00, y01, y10, y11);
            f32x2* d = red + (((two * (KSL - 1) + ksl - 1) * 8 + (r >> 1)) * 4) * 64 + lane;
            d[0] = y00; d[64] = y01; d[128] = y10; d[192] = y11;
        }
    }
    if (KSPL) { __syncthreads(); if (ksl || gt >= A.T) return; }
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        f32x2 y00, y01, y10, y11; inverse(r, y00, y01, y10, y11);
        if (KSPL) {
#pragma unroll
            for (int s = 0; s < KSL - 1; s++) {
                const f32x2* d = red + (((two * (KSL - 1) + s) * 8 + (r >> 1)) * 4) * 64 + lane;
                y00 = pk_add(y00, d[0]); y01 = pk_add(y01, d[64]); y10 = pk_add(y10, d[128]); y11 = pk_add(y11, d[192]);
            }
        }
        const f32x2 s00 = y00 * sl, s01 = y01 * sl, s10 = y10 * sl, s11 = y11 * sl;
        y00.x = fmaxf(y00.x, s00.x); y00.y = fmaxf(y00.y, s00.y); y01.x = fmaxf(y01.x, s01.x); y01.y = fmaxf(y01.y, s01.y);
        y10.x = fmaxf(y10.x, s10.x); y10.y = fmaxf(y10.y, s10.y); y11.x = fmaxf(y11.x, s11.x); y11.y = fmaxf(y11.y, s11.y);
        const int co = cob + 8 * (r >> 2) + (r & 3);                     // channel of register r; r + 1 is the next one
        float* yp = yb + (size_t)co * hw;
        if (full) {
            *(f32x2*)yp = f32x2{y00.x, y01.x}; *(f32x2*)(yp + A.W) = f32x2{y10.x, y11.x};
            *(f32x2*)(yp + hw) = f32x2{y00.y, y01.y}; *(f32x2*)(yp + hw + A.W) = f32x2{y10.y, y11.y};
        } else {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                if (co + h >= A.Cout) continue;
                float* q = yp + (size_t)h * hw;
                const float a00 = h ? y00.y : y00.x, a01 = h ? y01.y : y01.x, a10 = h ? y10.y : y10.x, a11 = h ? y11.y : y11.x;
                if (A.vec2) {                                            // W even and y 8-byte aligned: a tile row is one 8-byte store
                    *(f32x2*)q = f32x2{a00, a01};
                    if (row1) *(f32x2*)(q + A.W) = f32x2{a10, a11};
                } else {
                    q[0] = a00; if (col1) q[1] = a01;
                    if (row1) { q[A.W] = a10; if (col1) q[A.W + 1] = a11; }
                }
            }
        }
    }
#ifdef WN_PROF
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0" : "=s"(tk[3]));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (A.prof && lane == 0 && item == 0) for (int k = 0; k < 4; k++) A.prof[80 + w * 4 + k] = tk[k];
#endif
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

/* The form vido_wino3x3_bias_act_form should be given for this launch.  0: the tile form (a workgroup = 64 tiles x 64 channels or 128 x 32, each wave walks ALL input
 * channels).  1: the K-split form for launches that would leave most of the chip idle (fewer than 128 workgroups of the tile form): a workgroup = 32 tiles x 32 channels,
 * its four waves each take a quarter of the input channels and the partial sums meet in LDS — four times the waves, a quarter of the chain each.
 * 2: the same with TWO channel slices x two tile blocks per workgroup (half the U traffic per matrix instruction, twice the chain).  Under-filled launches get form 2
 * where that still gives >= 128 workgroups, else form 1; VIDO_WINO_KSPLIT=0 / 4 / 2: never / always form 1 / always form 2 for them. */
int vido_wino3x3_form(int n, int cin, int cout, int h, int w)
{
    static const int ks = [] { const char* e = getenv("VIDO_WINO_KSPLIT"); return e ? atoi(e) : -1; }();      // 0: never; 4 / 2: that form for every under-filled launch; default: by size
    (void)cin;
    if (ks == 0 || vido_wino3x3_fills_chip(n, cout, h, w, 128)) return 0;
    if (ks == 4 || ks == 1) return 1;
    if (ks == 2) return 2;
    // two slices x two tile blocks read half the U bytes per matrix instruction but walk twice the chain: they win where they still put >= 128 workgroups on the chip
    // (LiteFlowNet's level 3: 128 -> 64 at 120 x 160 41 -> 33 us, 64 -> 64 28 -> 21), four slices below that (FPN P4 39 vs 63 us; profiles/r5/wino_ksplit_microbench.txt)
    const long long T = (long long)n * ((h + 1) / 2) * ((w + 1) / 2);
    return ((T + 63) / 64) * ((cout + 31) / 32) >= 128 ? 2 : 1;
}

/* floats of the packed transformed weight of a cin -> cout layer (form 0; _form: of the given form) */
long long vido_wino3x3_packed_floats_form(int cin, int cout, int form)
{
    if (cin < 1 || cout < 1) return 0;
    if (form == 1 || form == 2) return 16ll * (((cout + 31) / 32) * 32) * wn_cin_pad(cin, 4);
    return 16ll * wn_cout_pad(cout) * wn_cin_pad(cin, wn_kc(cout));
}
long long vido_wino3x3_packed_floats(int cin, int cout) { return vido_wino3x3_packed_floats_form(cin, cout, 0); }

/* HOST: weight [cout][cin][3][3] f32 -> U = G g G^T (float64 arithmetic, rounded once) in the operand order of k_wino3x3:
 * element (position p = 4 i + j, output channel co, input channel c) at [co / 32][c / KC][p][32 * (c & 1) + co % 32][(c % KC) / 2], KC = 8 (4 when
 * cout rounds up to 64 with >= 32 padded channels, and always in form 1); padded channels are zero. */
int vido_wino3x3_pack_form(const float* w, int cin, int cout, int form, float* up)
{
    if (!w || !up || cin < 1 || cout < 1 || form < 0 || form > 2) return VIDO_E_INVALID;
    const int kc = form ? 4 : wn_kc(cout), ks = kc / 2, cip = wn_cin_pad(cin, kc), nchunk = cip / kc;
    std::memset(up, 0, sizeof(float) * (size_t)vido_wino3x3_packed_floats_form(cin, cout, form));
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
    return VIDO_OK;
}
int vido_wino3x3_pack(const float* w, int cin, int cout, float* up) { return vido_wino3x3_pack_form(w, cin, cout, 0, up); }

/* y = leaky_relu(conv2d(x, w, padding 1) + bias, slope) for n images, 3x3 kernel, stride 1: x [n][cin][h][w], y [n][cout][h][w] f32 DEVICE tensors (y != x),
 * bias [cout] or NULL, u_packed = vido_wino3x3_pack_form(w, .., form) on the device (16-byte aligned).  slope 0 = ReLU, 1 = none.  Winograd F(2x2, 3x3) in fp32: the result differs
 * from a direct fp32 convolution by rounding only (~1e-6 of the output scale, the class of the library's own Winograd kernels).  Enqueues on the adopted stream; capturable.
 * form: 0 or 1 (vido_wino3x3_form picks); the result is the same up to the order of the channel sums. */
int vido_wino3x3_bias_act_form(vido_ctx* ctx, const float* x, const float* u_packed, const float* bias, float* y, int n, int cin, int cout, int h, int w, float slope, int form)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!x || !u_packed || !y || x == y || n < 1 || form < 0 || form > 2 || !vido_wino3x3_supported(cin, cout, h, w) || ((uintptr_t)u_packed & 15) || (((uintptr_t)x | (uintptr_t)y) & 3))
        return vido_set_error(ctx, VIDO_E_INVALID, "wino3x3: no kernel for %d -> %d channels on %d x %d x %d (or a pointer is misaligned)", cin, cout, n, h, w);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->has_ext_stream ? ctx->ext_stream : ctx->stream;
    const int kc = form ? 4 : wn_kc(cout), ht = (h + 1) / 2, wt = (w + 1) / 2;
    const long long T = (long long)n * ht * wt;
    if (T >= (1ll << 30) || 4ll * n * cin * h * w >= (1ll << 30) || 4ll * n * cout * h * w >= (1ll << 32))
        return vido_set_error(ctx, VIDO_E_INVALID, "wino3x3: a batch of %d images of %d x %d x %d is past the 1 GB the kernel addresses", n, cin, h, w);
    const int tb = form == 1 ? 32 : form == 2 ? 64 : (kc == 8 ? 64 : 128), cgroups = kc == 8 ? (cout + 63) / 64 : (cout + 31) / 32;
    const long long nblk_ll = (T + tb - 1) / tb;
    if (nblk_ll * cgroups >= (1ll << 30)) return vido_set_error(ctx, VIDO_E_INVALID, "wino3x3: too many work items");
    const int nblk = (int)nblk_ll, total = nblk * cgroups;
    WnArgs A{x, u_packed, bias, y, n, cin, cout, h, w, ht, wt, (int)T, wn_cin_pad(cin, kc) / kc, cgroups, total, slope, (w % 2 == 0 && ((uintptr_t)y & 7) == 0) ? 1 : 0, (unsigned)(4ll * n * cin * h * w), (unsigned)(4ll * vido_wino3x3_packed_floats_form(cin, cout, form)),
             nullptr};
    const dim3 grid(8 * ((total + 7) / 8)), blk(256);
    constexpr size_t LDS8 = (size_t)2 * (2 * 16 * 64 * 4 + 16 * 2 * 4 * 64) * 4, LDS4 = (size_t)2 * (1 * 16 * 64 * 2 + 16 * 4 * 2 * 64) * 4, LDSK = (size_t)2 * (4 * 16 * 64 * 2 + 16 * 4 * 2 * 64) * 4, LDSK2 = (size_t)2 * (2 * 16 * 64 * 2 + 16 * 4 * 2 * 64) * 4;
    static bool attr[64] = {};
    if (!attr[ctx->device & 63]) {
        for (const void* f : {(const void*)k_wino3x3<2, 2, 8, false>, (const void*)k_wino3x3<2, 2, 8, true>}) HIP_TRY(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS8));
        for (const void* f : {(const void*)k_wino3x3<1, 4, 4, false>, (const void*)k_wino3x3<1, 4, 4, true>}) HIP_TRY(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS4));
        for (const void* f : {(const void*)k_wino3x3<1, 4, 4, false, 4>, (const void*)k_wino3x3<1, 4, 4, true, 4>}) HIP_TRY(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDSK));
        for (const void* f : {(const void*)k_wino3x3<1, 4, 4, false, 2>, (const void*)k_wino3x3<1, 4, 4, true, 2>}) HIP_TRY(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDSK2));
        attr[ctx->device & 63] = true;
    }
    const bool odd = cin & 1;                                            // (an odd channel count costs 17 vector instructions per window: its last channel pair is half padding)
    if (form == 1) { if (odd) hipLaunchKernelGGL((k_wino3x3<1, 4, 4, true, 4>), grid, blk, LDSK, st, A); else hipLaunchKernelGGL((k_wino3x3<1, 4, 4, false, 4>), grid, blk, LDSK, st, A); }
    else if (form == 2) { if (odd) hipLaunchKernelGGL((k_wino3x3<1, 4, 4, true, 2>), grid, blk, LDSK2, st, A); else hipLaunchKernelGGL((k_wino3x3<1, 4, 4, false, 2>), grid, blk, LDSK2, st, A); }
    else if (kc == 8) { if (odd) hipLaunchKernelGGL((k_wino3x3<2, 2, 8, true>), grid, blk, LDS8, st, A); else hipLaunchKernelGGL((k_wino3x3<2, 2, 8, false>), grid, blk, LDS8, st, A); }
    else { if (odd) hipLaunchKernelGGL((k_wino3x3<1, 4, 4, true>), grid, blk, LDS4, st, A); else hipLaunchKernelGGL((k_wino3x3<1, 4, 4, false>), grid, blk, LDS4, st, A); }
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}
int vido_wino3x3_bias_act(vido_ctx* ctx, const float* x, const float* u_packed, const float* bias, float* y, int n, int cin, int cout, int h, int w, float slope)
{
    return vido_wino3x3_bias_act_form(ctx, x, u_packed, bias, y, n, cin, cout, h, w, slope, 0);
}

}  // extern "C"
