// orb.hip — ORB front-end on gfx950: pyramid, per-cell FAST 9/16 + NMS, candidate compaction,
// intensity-centroid orientation, 7x7 blur, steered BRIEF.  Replaces ORBextractor::operator()
// (reference vido_slam/src/ORBextractor.cc:1034-1105) and what it calls:
//   ComputePyramid :1107-1132, ComputeKeyPointsOctTree :755-843, DistributeOctTree :529-753,
//   IC_Angle :67-94, computeOrbDescriptor :98-137.
// Design (MI355X-first, not a translation):
//   * a BATCH of frames is processed per call: every launch has (tiles x frames) workgroups so the
//     256 CUs are filled even though one 640x480 frame is < 1 MB per stage;
//   * pyramid slabs live in HBM with 64-byte row pitch, no materialised reflect border (only the
//     blur ever reads outside a level and reflects on the fly);
//   * FAST: one wave64 per reference cell (the reference's ~30x30 px cv::FAST sub-image): the cell's
//     sub-image is staged once in LDS with aligned dword loads, a threshold-free score S(p) is
//     computed per pixel ("corner at t" <=> S>=t), and the reference's two-threshold rule
//     (th=20, else th=7 if the cell came back empty) + per-sub-image 3x3 NMS run on the LDS score tile;
//     survivors are emitted in row-major order with wave ballots, so the candidate list comes out in
//     exactly the order the reference's nested loops produce;
//   * the serial quadtree (DistributeOctTree) stays on the host, one task per (frame, level), run on a
//     small thread pool while the GPU blurs the pyramids;
//   * orientation + rBRIEF: one wave per keypoint, wave-shuffle reductions for the moments and
//     shuffles to assemble the 256 descriptor bits.
// Compiled with -ffp-contract=off: the float formulas (fastAtan2 polynomial, pattern rotation) must
// round exactly like the CPU oracle.
#include "common.hpp"
#include "../../include/vido_orb_pattern.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cfloat>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <functional>

#define EDGE_THRESHOLD 19
#define HALF_PATCH 15
#define FT_PITCH 80          // LDS pitch of a cell sub-image tile (bytes)
#define FT_ROWS 72
#define FS_PITCH 72          // LDS pitch of the score tile
#define FS_ROWS 68

// ------------------------------------------------------------------------------------------------
// device tables
__constant__ int c_umax[HALF_PATCH + 1];
__constant__ signed char c_pattern[256 * 4];

// XCD-aware (tile, frame) assignment for grids of (tiles, frames): workgroup b lands on XCD b % 8 (observed
// dispatch order, MI355X_MICROARCH.md), so consecutive tiles of ONE frame would be spread over the 8 private L2s
// and every halo line would be fetched from the fabric up to 8 times.  With frames % 8 == 0 all tiles of a
// frame are steered to one XCD instead (pure performance remap, any placement stays correct).
__device__ __forceinline__ void xcd_tile_frame(int& tile, int& frame)
{
    const int nx = gridDim.x, ny = gridDim.y;
    tile = blockIdx.x; frame = blockIdx.y;
    if ((ny & 7) == 0) {
        const int lin = blockIdx.y * nx + blockIdx.x, xcd = lin & 7, j = lin >> 3;
        frame = xcd + 8 * (j / nx); tile = j % nx;
    }
}

// ------------------------------------------------------------------------------------------------
// K1: bilinear downscale of level l-1 into level l (OpenCV INTER_LINEAR u8 fixed-point semantics:
// 11-bit coefficients; (b0*(r0>>4))>>16 + (b1*(r1>>4))>>16 + 2 >> 2).  4 output pixels per thread.
__global__ __launch_bounds__(256) void k_resize(uint8_t* __restrict__ pyr, size_t slab, int sw, int sh, int spitch, int soff,
                                                int dw, int dh, int dpitch, int doff,
                                                const int2* __restrict__ xtab, const int4* __restrict__ ytab)
{
    const int x4 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (y >= dh || x4 >= dpitch) return;
    uint8_t* base = pyr + (size_t)blockIdx.z * slab;
    const int4 yt = ytab[y];
    const uint8_t* S0 = base + soff + (size_t)yt.x * spitch;
    const uint8_t* S1 = base + soff + (size_t)yt.y * spitch;
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int x = x4 + k;
        if (x < dw) {
            const int2 xt = xtab[x];
            const int sx = xt.x & 0xffff, a0 = xt.x >> 16, a1 = xt.y;
            const int sx1 = sx + 1 < sw ? sx + 1 : sx;
            const int r0 = S0[sx] * a0 + S0[sx1] * a1;
            const int r1 = S1[sx] * a0 + S1[sx1] * a1;
            int v = (((yt.z * (r0 >> 4)) >> 16) + ((yt.w * (r1 >> 4)) >> 16) + 2) >> 2;
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            out |= (uint32_t)v << (8 * k);
        }
    }
    *(uint32_t*)(base + doff + (size_t)y * dpitch + x4) = out;
}

// ------------------------------------------------------------------------------------------------
// K2: FAST 9/16 per reference cell.
__device__ __forceinline__ bool has9(uint32_t m)           // 9 contiguous set bits in a circular 16-bit mask
{
    uint32_t w = m | (m << 16);
    uint32_t x = w & (w >> 1);
    x &= x >> 2;
    x &= x >> 4;            // 8 consecutive
    x &= w >> 8;            // 9 consecutive
    return (x & 0xffffu) != 0;
}

// Byte `off` (0..11) of a 12-byte window held in three aligned dwords.
#define WIN_BYTE(W, off) ((int)(((off) < 4 ? (W)[0] >> (8 * (off)) : (off) < 8 ? (W)[1] >> (8 * ((off) - 4)) : (W)[2] >> (8 * ((off) - 8))) & 0xffu))

// Phase 1a: compass pre-test.  Every arc of 9 contiguous ring pixels contains at least two of the four compass
// pixels (ring 0, 4, 8, 12), so a corner at threshold th has >= 2 compass pixels brighter than v+th or >= 2
// darker than v-th.  Sign bits are gathered with v_alignbit, 2 ops per compass pixel and polarity.
template <int O>
__device__ __forceinline__ bool fast_compass(const uint32_t* up /*w1 of row -3*/, const uint32_t* mid /*w0..w2 of row 0*/, const uint32_t* dn, int th)
{
    const int v = WIN_BYTE(mid, O), lo = v - th, hi = v + th;
    const int r0 = (int)((dn[0] >> (8 * (O - 4))) & 0xffu), r8 = (int)((up[0] >> (8 * (O - 4))) & 0xffu);
    const int r4 = WIN_BYTE(mid, O + 3), r12 = WIN_BYTE(mid, O - 3);
    uint32_t dk = 0, br = 0;
    dk = __builtin_amdgcn_alignbit(dk, (uint32_t)(r0 - lo), 31); br = __builtin_amdgcn_alignbit(br, (uint32_t)(hi - r0), 31);
    dk = __builtin_amdgcn_alignbit(dk, (uint32_t)(r4 - lo), 31); br = __builtin_amdgcn_alignbit(br, (uint32_t)(hi - r4), 31);
    dk = __builtin_amdgcn_alignbit(dk, (uint32_t)(r8 - lo), 31); br = __builtin_amdgcn_alignbit(br, (uint32_t)(hi - r8), 31);
    dk = __builtin_amdgcn_alignbit(dk, (uint32_t)(r12 - lo), 31); br = __builtin_amdgcn_alignbit(br, (uint32_t)(hi - r12), 31);
    return (__popc(dk) >= 2) | (__popc(br) >= 2);
}

// Phase 1b: full 9/16 segment test + threshold-free corner score for one candidate pixel at LDS tile position t.
// S = (max over the 16 arcs of 9 of the arc's min one-signed |centre - ring|) - 1; 0 unless p is a corner at `th`
// (then S >= th): "corner at threshold t" <=> S >= t, so one score map serves both reference thresholds.
__device__ __forceinline__ int fast_score(const uint8_t* t, int th)
{
    const int v = t[0], lo = v - th, hi = v + th;
    int r[16];
    r[0] = t[3 * FT_PITCH]; r[1] = t[3 * FT_PITCH + 1]; r[2] = t[2 * FT_PITCH + 2]; r[3] = t[FT_PITCH + 3];
    r[4] = t[3]; r[5] = t[-FT_PITCH + 3]; r[6] = t[-2 * FT_PITCH + 2]; r[7] = t[-3 * FT_PITCH + 1];
    r[8] = t[-3 * FT_PITCH]; r[9] = t[-3 * FT_PITCH - 1]; r[10] = t[-2 * FT_PITCH - 2]; r[11] = t[-FT_PITCH - 3];
    r[12] = t[-3]; r[13] = t[FT_PITCH - 3]; r[14] = t[2 * FT_PITCH - 2]; r[15] = t[3 * FT_PITCH - 1];
    uint32_t dark = 0, bright = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) { dark = __builtin_amdgcn_alignbit(dark, (uint32_t)(r[k] - lo), 31); bright = __builtin_amdgcn_alignbit(bright, (uint32_t)(hi - r[k]), 31); }
    const bool isd = has9(dark), isb = has9(bright);       // bit order reversed w.r.t. k: irrelevant for a circular run
    if (!(isd | isb)) return 0;
    int d[16];
#pragma unroll
    for (int k = 0; k < 16; k++) d[k] = isd ? v - r[k] : r[k] - v;     // one-signed difference of the arc family
    int a2[16], a4[16], a8[16];                                        // sliding min over 9 circular neighbours: 2,4,8,+1
#pragma unroll
    for (int k = 0; k < 16; k++) a2[k] = min(d[k], d[(k + 1) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++) a4[k] = min(a2[k], a2[(k + 2) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++) a8[k] = min(a4[k], a4[(k + 4) & 15]);
    int best = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) best = max(best, min(a8[k], d[(k + 8) & 15]));
    return best - 1;
}

// dynamic LDS layout (bytes): [16 pad][tile rows*FT_PITCH][score (rows-4)*FS_PITCH][cand u16 x ncand][corner u16 x ncand][listA][listB]
__global__ __launch_bounds__(64) void k_fast_cells(const uint8_t* __restrict__ pyr, size_t slab, PyrDev P,
                                                   const CellDesc* __restrict__ cells, int n_cells,
                                                   int ini_th, int min_th, int lds_rows, int lds_ncand,
                                                   uint32_t* __restrict__ slots, int* __restrict__ counts)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    uint8_t* tile = lds_raw + 16;                                    // quads at column 0 peek one dword to the left
    uint8_t* sc = tile + lds_rows * FT_PITCH;
    uint16_t* cand = (uint16_t*)(sc + (lds_rows - 4) * FS_PITCH);
    uint16_t* corners = cand + lds_ncand;
    uint32_t* listA = (uint32_t*)(corners + lds_ncand);
    uint32_t* listB = listA + VIDO_CELL_CAP;
    int cell, f; xcd_tile_frame(cell, f);
    const int lane = threadIdx.x;
    const CellDesc c = cells[cell];
    const int pitch = P.pitch[c.level];
    const uint8_t* img = pyr + (size_t)f * slab + P.off[c.level];
    const int xa = c.x0 & ~3, shift = c.x0 - xa;
    const int nd = ((c.x0 + c.sw + 3) >> 2) - (xa >> 2);            // dwords per tile row
    for (int i = lane; i < nd * c.sh; i += 64) {
        const int row = i / nd, col = i - row * nd;
        *(uint32_t*)(tile + row * FT_PITCH + 4 * col) = *(const uint32_t*)(img + (size_t)(c.y0 + row) * pitch + xa + 4 * col);
    }
    const int iw = c.sw - 6, ih = c.sh - 6;
    for (int i = lane; i < (ih + 2) * (FS_PITCH / 4); i += 64) ((uint32_t*)sc)[i] = 0;
    __syncthreads();
    // The reference runs cv::FAST at iniThFAST and only re-runs the cell at minThFAST when that came back empty
    // (ORBextractor.cc:799-806).  Same control flow here (wave-uniform): pass 0 works on the far smaller
    // candidate set of the high threshold; the score map is threshold-free, so pass 1 only adds entries.
    const int tx0 = shift + 3;                                       // tile column of interior x = 0
    const int qc0 = tx0 >> 2, nq = ((tx0 + iw - 1) >> 2) - qc0 + 1, ntask = ih * nq;
    const unsigned long long ltmask = (1ull << lane) - 1ull;
    bool overflow = false;
    int n = 0;
    for (int pass = 0; pass < 2 && n == 0; pass++) {
        const int th = pass == 0 ? ini_th : min_th;
        // ---- phase a: compass pre-test, 4 horizontally adjacent pixels per lane (aligned quad), ordered compaction
        int ncand = 0;
        for (int t0 = 0; t0 < ntask; t0 += 64) {
            const int t = t0 + lane;
            uint32_t m4 = 0; int iy = 0, txq = 0;
            if (t < ntask) {
                iy = t / nq; const int qc = qc0 + (t - iy * nq);
                txq = 4 * qc;
                const uint8_t* base = tile + iy * FT_PITCH + txq;        // row iy = centre row - 3
                const uint32_t up = *(const uint32_t*)base, dn = *(const uint32_t*)(base + 6 * FT_PITCH);
                const uint32_t* q = (const uint32_t*)(base + 3 * FT_PITCH - 4);
                const uint32_t mid[3] = {q[0], q[1], q[2]};
                const int ixq = txq - tx0;                               // interior x of the quad's first pixel (may be < 0)
                if (ixq + 0 >= 0 && ixq + 0 < iw) m4 |= (uint32_t)fast_compass<4>(&up, mid, &dn, th) << 0;
                if (ixq + 1 >= 0 && ixq + 1 < iw) m4 |= (uint32_t)fast_compass<5>(&up, mid, &dn, th) << 1;
                if (ixq + 2 >= 0 && ixq + 2 < iw) m4 |= (uint32_t)fast_compass<6>(&up, mid, &dn, th) << 2;
                if (ixq + 3 >= 0 && ixq + 3 < iw) m4 |= (uint32_t)fast_compass<7>(&up, mid, &dn, th) << 3;
            }
            const unsigned long long b0 = __ballot(m4 & 1), b1 = __ballot(m4 & 2), b2 = __ballot(m4 & 4), b3 = __ballot(m4 & 8);
            int pos = ncand + __popcll(b0 & ltmask) + __popcll(b1 & ltmask) + __popcll(b2 & ltmask) + __popcll(b3 & ltmask);
#pragma unroll
            for (int p = 0; p < 4; p++) if (m4 & (1u << p)) { if (pos < lds_ncand) cand[pos] = (uint16_t)(iy * 128 + (txq - tx0 + p)); pos++; }
            ncand += __popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3);
        }
        overflow |= ncand > lds_ncand;                               // reported through the count (> VIDO_CELL_CAP => VIDO_E_CAPACITY)
        ncand = min(ncand, lds_ncand);
        __syncthreads();
        // ---- phase b: full segment test + score, one candidate per lane; corners keep the row-major order
        int ncorn = 0;
        for (int q0 = 0; q0 < ncand; q0 += 64) {
            const int q = q0 + lane;
            int S = 0; uint16_t code = 0;
            if (q < ncand) {
                code = cand[q];
                const int iy = code >> 7, ix = code & 127;
                S = fast_score(tile + (iy + 3) * FT_PITCH + tx0 + ix, th);
                if (S > 0) sc[(iy + 1) * FS_PITCH + ix + 1] = (uint8_t)S;
            }
            const unsigned long long bc = __ballot(S > 0);
            if (S > 0) corners[ncorn + __popcll(bc & ltmask)] = code;
            ncorn += __popcll(bc);
        }
        __syncthreads();
        // ---- phase c: per-sub-image 3x3 NMS (strictly greater than the 8 neighbours' scores at this threshold;
        // the score tile only holds scores >= th), emission in row-major order
        for (int q0 = 0; q0 < ncorn; q0 += 64) {
            const int q = q0 + lane;
            bool keep = false; uint32_t packed = 0;
            if (q < ncorn) {
                const int iy = corners[q] >> 7, ix = corners[q] & 127;
                const uint8_t* s = sc + (iy + 1) * FS_PITCH + ix + 1;
                const int S = s[0];
                const int n0 = s[-FS_PITCH - 1], n1 = s[-FS_PITCH], n2 = s[-FS_PITCH + 1], n3 = s[-1],
                          n4 = s[1], n5 = s[FS_PITCH - 1], n6 = s[FS_PITCH], n7 = s[FS_PITCH + 1];
                const int mx = max(max(max(n0, n1), max(n2, n3)), max(max(n4, n5), max(n6, n7)));
                keep = S > mx;
                packed = (uint32_t)(c.x0 + 3 + ix) | ((uint32_t)(c.y0 + 3 + iy) << 12) | ((uint32_t)S << 24);
            }
            const unsigned long long bk = __ballot(keep);
            if (keep) { const int pos = n + __popcll(bk & ltmask); if (pos < VIDO_CELL_CAP) listA[pos] = packed; }
            n += __popcll(bk);
        }
        __syncthreads();
    }
    const uint32_t* list = listA;
    const size_t ci = (size_t)f * n_cells + cell;
    if (lane == 0) counts[ci] = overflow ? VIDO_CELL_CAP + 1 : n;
    for (int i = lane; i < min(n, VIDO_CELL_CAP); i += 64) slots[ci * VIDO_CELL_CAP + i] = list[i];
}

// K3: exclusive scan of the per-cell counts (frame-major, reference cell order) -> dense offsets;
// also the per-(frame, level) start offsets the host needs.  One workgroup of 1024 threads.
__global__ __launch_bounds__(1024) void k_scan_counts(const int* __restrict__ counts, int n, int* __restrict__ offsets,
                                                      int n_cells, int n_frames, int n_levels,
                                                      const int* __restrict__ first_cell, int* __restrict__ lvloff,
                                                      int* __restrict__ overflow)
{
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int chunk = (n + 1023) / 1024;
    const int beg = tid * chunk, end = min(beg + chunk, n);
    int s = 0, ovf = 0;
    for (int i = beg; i < end; i++) { int c = counts[i]; if (c > VIDO_CELL_CAP) { ovf = 1; c = VIDO_CELL_CAP; } s += c; }
    part[tid] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        int v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s;
    for (int i = beg; i < end; i++) { offsets[i] = run; run += min(counts[i], VIDO_CELL_CAP); }
    if (tid == 1023) offsets[n] = part[1023];
    if (ovf) *overflow = 1;
    __syncthreads();
    for (int i = tid; i < n_frames * n_levels; i += 1024) {
        const int f = i / n_levels, l = i - f * n_levels;
        lvloff[i] = offsets[f * n_cells + first_cell[l]];
    }
    if (tid == 0) lvloff[n_frames * n_levels] = part[1023];
}

__global__ __launch_bounds__(64) void k_gather_cands(const uint32_t* __restrict__ slots, const int* __restrict__ counts,
                                                     const int* __restrict__ offsets, int n_cells, uint32_t* __restrict__ dense, int cap)
{
    const size_t ci = (size_t)blockIdx.y * n_cells + blockIdx.x;
    const int n = min(counts[ci], VIDO_CELL_CAP), off = offsets[ci];
    for (int i = threadIdx.x; i < n; i += 64) if (off + i < cap) dense[off + i] = slots[ci * VIDO_CELL_CAP + i];
}

// ------------------------------------------------------------------------------------------------
// K4: 7x7 sigma=2 Gaussian, reflect-101, Q0.8 taps {18,34,49,54,49,34,18}; 64x16 output tile per WG.
__device__ __forceinline__ int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) { i = i < 0 ? -i : 2 * n - 2 - i; }
    return i;
}
__global__ __launch_bounds__(256) void k_blur7(const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur, size_t slab, PyrDev P,
                                               const BlurTile* __restrict__ tiles)
{
    __shared__ uint8_t in[22][72];
    __shared__ uint16_t hb[22][64];
    int ti, fr; xcd_tile_frame(ti, fr);
    const BlurTile t = tiles[ti];
    const int w = P.w[t.level], h = P.h[t.level], pitch = P.pitch[t.level];
    const uint8_t* img = pyr + (size_t)fr * slab + P.off[t.level];
    uint8_t* out = blur + (size_t)fr * slab + P.off[t.level];
    const int tid = threadIdx.x, x0 = t.tx * 64, y0 = t.ty * 16;
    for (int i = tid; i < 22 * 70; i += 256) {
        const int r = i / 70, cx = i - r * 70;
        in[r][cx] = img[(size_t)reflect101(y0 + r - 3, h) * pitch + reflect101(x0 + cx - 3, w)];
    }
    __syncthreads();
    for (int i = tid; i < 22 * 64; i += 256) {
        const int r = i >> 6, x = i & 63;
        const int acc = 18 * (in[r][x] + in[r][x + 6]) + 34 * (in[r][x + 1] + in[r][x + 5]) + 49 * (in[r][x + 2] + in[r][x + 4]) + 54 * in[r][x + 3];
        hb[r][x] = (uint16_t)acc;
    }
    __syncthreads();
    const int row = tid >> 4, x4 = (tid & 15) * 4;
    if (y0 + row < h && x0 + x4 < pitch) {
        uint32_t o = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int x = x4 + k;
            const int acc = 18 * (hb[row][x] + hb[row + 6][x]) + 34 * (hb[row + 1][x] + hb[row + 5][x]) + 49 * (hb[row + 2][x] + hb[row + 4][x]) + 54 * hb[row + 3][x];
            o |= (uint32_t)((acc + 32768) >> 16) << (8 * k);
        }
        *(uint32_t*)(out + (size_t)(y0 + row) * pitch + x0 + x4) = o;
    }
}

// ------------------------------------------------------------------------------------------------
// K5: orientation (IC_Angle) + steered BRIEF, one wave per keypoint.
__device__ __forceinline__ float fast_atan2_deg(float y, float x)      // cv::fastAtan2 scalar polynomial
{
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) { c = ay / (ax + (float)DBL_EPSILON); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    else { c = ax / (ay + (float)DBL_EPSILON); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

__global__ __launch_bounds__(256) void k_orient_brief(const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur, size_t slab, PyrDev P,
                                                      const uint2* __restrict__ kps, int n_kp, int with_desc,
                                                      float* __restrict__ angle_out, uint8_t* __restrict__ desc_out)
{
    // XCD-aware: each of the 8 XCDs (block b -> XCD b % 8) walks one contiguous eighth of the keypoint list, i.e.
    // whole frames, so the patch / pattern gathers of a frame stay in one private L2
    const int chunk = gridDim.x >> 3, blk = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    const int k = blk * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k >= n_kp) return;
    const uint2 kp = kps[k];
    const int x = kp.x & 0xfff, y = (kp.x >> 12) & 0xfff, level = kp.x >> 24, f = kp.y;
    const int pitch = P.pitch[level];
    const uint8_t* c = pyr + (size_t)f * slab + P.off[level] + (size_t)y * pitch + x;
    int m10 = 0, m01 = 0;
    for (int q = lane; q < 31 * 31; q += 64) {
        const int vy = q / 31, v = vy - 15, u = q - vy * 31 - 15;
        if (abs(u) <= c_umax[abs(v)]) { const int val = c[v * pitch + u]; m10 += u * val; m01 += v * val; }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { m10 += __shfl_xor(m10, o, 64); m01 += __shfl_xor(m01, o, 64); }
    const float ang = fast_atan2_deg((float)m01, (float)m10);
    if (lane == 0) angle_out[k] = ang;
    if (!with_desc) return;
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    const float ar = ang * factorPI;
    const float a = (float)cos((double)ar), b = (float)sin((double)ar);
    const uint8_t* cb = blur + (size_t)f * slab + P.off[level] + (size_t)y * pitch + x;
    uint32_t nib = 0;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const signed char* pt = c_pattern + (lane * 4 + t) * 4;
        const float x0 = (float)pt[0], y0 = (float)pt[1], x1 = (float)pt[2], y1 = (float)pt[3];
        const int t0 = cb[__float2int_rn(x0 * b + y0 * a) * pitch + __float2int_rn(x0 * a - y0 * b)];
        const int t1 = cb[__float2int_rn(x1 * b + y1 * a) * pitch + __float2int_rn(x1 * a - y1 * b)];
        nib |= (uint32_t)(t0 < t1) << t;
    }
    // lanes 2i / 2i+1 hold the low / high nibble of descriptor byte i; fold 8 lanes into one dword
    uint32_t w = nib | (__shfl_down(nib, 1, 64) << 4);
    w |= __shfl_down(w, 2, 64) << 8;
    w |= __shfl_down(w, 4, 64) << 16;
    if ((lane & 7) == 0) ((uint32_t*)(desc_out + (size_t)k * 32))[lane >> 3] = w;
}

// ================================================================================================
// host side
// Persistent worker pool for the host quadtree stage (thread creation per call would cost more than the work).
struct WorkerPool {
    std::vector<std::thread> th; std::mutex mu; std::condition_variable cv_go, cv_done;
    std::function<void(int)> fn; std::atomic<int> next{0}; int n = 0, active = 0; uint64_t gen = 0; bool stop = false;
    explicit WorkerPool(int nthreads)
    {
        for (int t = 0; t < nthreads; t++) th.emplace_back([this] {
            uint64_t seen = 0;
            for (;;) {
                { std::unique_lock<std::mutex> lk(mu); cv_go.wait(lk, [&] { return stop || gen != seen; }); if (stop) return; seen = gen; }
                for (int i; (i = next.fetch_add(1)) < n;) fn(i);
                { std::lock_guard<std::mutex> lk(mu); if (--active == 0) cv_done.notify_all(); }
            }
        });
    }
    ~WorkerPool() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv_go.notify_all(); for (auto& t : th) t.join(); }
    void run(int count, std::function<void(int)> f)
    {
        if (th.empty() || count < 4) { for (int i = 0; i < count; i++) f(i); return; }
        { std::lock_guard<std::mutex> lk(mu); fn = std::move(f); n = count; next = 0; active = (int)th.size(); gen++; }
        cv_go.notify_all();
        for (int i; (i = next.fetch_add(1)) < n;) fn(i);          // the caller works too
        std::unique_lock<std::mutex> lk(mu); cv_done.wait(lk, [&] { return active == 0; });
    }
};

struct OrbState {
    int L = 0, W = 0, H = 0, B = 0;
    LevelInfo lv[VIDO_MAX_LEVELS];
    PyrDev P{};
    size_t slab = 0;
    int n_cells = 0, n_blur_tiles = 0;
    int umax[HALF_PATCH + 1];
    std::vector<int> first_cell;
    // device
    uint8_t *d_pyr = nullptr, *d_blur = nullptr;
    CellDesc* d_cells = nullptr; BlurTile* d_btiles = nullptr;
    int2* d_xtab = nullptr; int4* d_ytab = nullptr;
    uint32_t* d_slots = nullptr; int *d_counts = nullptr, *d_offsets = nullptr, *d_first_cell = nullptr, *d_lvloff = nullptr, *d_overflow = nullptr;
    uint32_t* d_cand = nullptr; size_t cand_cap = 0;
    uint2* d_kp = nullptr; float* d_angle = nullptr; uint8_t* d_desc = nullptr; size_t kp_cap = 0;
    // pinned host
    int* h_lvloff = nullptr; int* h_overflow = nullptr; uint32_t* h_cand = nullptr; uint2* h_kp = nullptr; float* h_angle = nullptr; uint8_t* h_desc = nullptr;
    hipEvent_t ev[8] = {};
    float timing[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int last_frames = 0;
    int n_threads = 1;
    struct WorkerPool* pool = nullptr;
    int fast_rows = 0, fast_ncand = 0; size_t fast_lds = 0;
};

static inline int cv_round_f(float v) { return (int)lrintf(v); }

static int build_tables(vido_ctx* ctx, OrbState* S)
{
    const vido_config& c = ctx->cfg;
    const int L = c.n_levels;
    S->L = L; S->W = c.width; S->H = c.height; S->B = c.max_batch;
    // scale factors and per-level budgets: ORBextractor ctor (ORBextractor.cc:400-437)
    float scale[VIDO_MAX_LEVELS], inv[VIDO_MAX_LEVELS];
    scale[0] = 1.0f;
    for (int i = 1; i < L; i++) scale[i] = scale[i - 1] * c.scale_factor;
    for (int i = 0; i < L; i++) inv[i] = 1.0f / scale[i];
    {
        float factor = 1.0f / c.scale_factor;
        float want = c.n_features * (1 - factor) / (1 - (float)pow((double)factor, (double)L));
        int sum = 0;
        for (int l = 0; l < L - 1; l++) { S->lv[l].n_budget = cv_round_f(want); sum += S->lv[l].n_budget; want *= factor; }
        S->lv[L - 1].n_budget = std::max(c.n_features - sum, 0);
    }
    {   // circular patch row extents (ORBextractor.cc:444-459)
        int v, v0, vmax = (int)floorf(HALF_PATCH * sqrtf(2.f) / 2 + 1), vmin = (int)ceilf(HALF_PATCH * sqrtf(2.f) / 2);
        for (v = 0; v <= vmax; ++v) S->umax[v] = (int)lrint(sqrt((double)HALF_PATCH * HALF_PATCH - v * v));
        for (v = HALF_PATCH, v0 = 0; v >= vmin; --v) { while (S->umax[v0] == S->umax[v0 + 1]) ++v0; S->umax[v] = v0; ++v0; }
    }
    // level geometry (ORBextractor.cc:1111-1112) and slab layout
    int off = 0;
    S->P.n_levels = L;
    for (int l = 0; l < L; l++) {
        LevelInfo& v = S->lv[l];
        v.w = cv_round_f((float)c.width * inv[l]); v.h = cv_round_f((float)c.height * inv[l]);
        if (v.w - 2 * (EDGE_THRESHOLD - 3) < 30 || v.h - 2 * (EDGE_THRESHOLD - 3) < 30)   // nCols/nRows would be 0 (the reference divides by it)
            return vido_set_error(ctx, VIDO_E_INVALID, "pyramid level %d is %dx%d: too small for the 30-px FAST cell grid", l, v.w, v.h);
        v.pitch = (v.w + 63) & ~63; v.off = off; v.scale = scale[l];
        off += v.pitch * v.h; off = (off + 255) & ~255;
        S->P.w[l] = v.w; S->P.h[l] = v.h; S->P.pitch[l] = v.pitch; S->P.off[l] = v.off;
    }
    S->slab = (size_t)off + 256;     // tail pad: dword tile loads may read up to 3 bytes past a row end
    // resize coefficient tables (cv::resize INTER_LINEAR, 11-bit)
    std::vector<int2> xtab; std::vector<int4> ytab;
    for (int l = 1; l < L; l++) {
        const LevelInfo &s = S->lv[l - 1]; LevelInfo& d = S->lv[l];
        d.xtab_off = (int)xtab.size(); d.ytab_off = (int)ytab.size();
        const double sx_ = (double)s.w / d.w, sy_ = (double)s.h / d.h;
        for (int dx = 0; dx < d.w; dx++) {
            float fx = (float)((dx + 0.5) * sx_ - 0.5);
            int sx = (int)floorf(fx); fx -= sx;
            if (sx < 0) { fx = 0; sx = 0; }
            if (sx >= s.w - 1) { fx = 0; sx = s.w - 1; }
            const int a0 = cv_round_f((1.f - fx) * 2048.f), a1 = cv_round_f(fx * 2048.f);
            xtab.push_back(make_int2(sx | (a0 << 16), a1));
        }
        for (int dy = 0; dy < d.h; dy++) {
            float fy = (float)((dy + 0.5) * sy_ - 0.5);
            int sy = (int)floorf(fy); fy -= sy;
            const int b0 = cv_round_f((1.f - fy) * 2048.f), b1 = cv_round_f(fy * 2048.f);
            const int y0 = std::min(std::max(sy, 0), s.h - 1), y1 = std::min(std::max(sy + 1, 0), s.h - 1);
            ytab.push_back(make_int4(y0, y1, b0, b1));
        }
    }
    // FAST cell table in the reference's loop order (ORBextractor.cc:759-796)
    std::vector<CellDesc> cells; std::vector<BlurTile> btiles;
    S->first_cell.assign(L, 0);
    for (int l = 0; l < L; l++) {
        LevelInfo& v = S->lv[l];
        v.first_cell = (int)cells.size(); S->first_cell[l] = v.first_cell;
        const float Wc = 30;
        const int minBX = EDGE_THRESHOLD - 3, minBY = minBX, maxBX = v.w - EDGE_THRESHOLD + 3, maxBY = v.h - EDGE_THRESHOLD + 3;
        const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
        const int nCols = (int)(width / Wc), nRows = (int)(height / Wc);
        const int wCell = (int)ceilf(width / nCols), hCell = (int)ceilf(height / nRows);
        for (int i = 0; i < nRows; i++) {
            const float iniY = (float)(minBY + i * hCell);
            float maxY = iniY + hCell + 6;
            if (iniY >= maxBY - 3) continue;
            if (maxY > maxBY) maxY = (float)maxBY;
            for (int j = 0; j < nCols; j++) {
                const float iniX = (float)(minBX + j * wCell);
                float maxX = iniX + wCell + 6;
                if (iniX >= maxBX - 6) continue;
                if (maxX > maxBX) maxX = (float)maxBX;
                CellDesc cd{}; cd.level = l; cd.x0 = (int)iniX; cd.y0 = (int)iniY; cd.sw = (int)maxX - (int)iniX; cd.sh = (int)maxY - (int)iniY;
                if (cd.sw + 3 > FT_PITCH - 4 || cd.sh > FT_ROWS || cd.sw - 6 + 2 > FS_PITCH || cd.sh - 6 + 2 > FS_ROWS)
                    return vido_set_error(ctx, VIDO_E_INVALID, "FAST cell %dx%d at level %d exceeds the LDS tile (%dx%d)", cd.sw, cd.sh, l, FT_PITCH - 8, FT_ROWS);
                if (cd.sw < 7 || cd.sh < 7) continue;     // cv::FAST finds nothing in a sub-image thinner than its ring
                cells.push_back(cd);
            }
        }
        v.n_cells = (int)cells.size() - v.first_cell;
        for (int ty = 0; ty < (v.h + 15) / 16; ty++)
            for (int tx = 0; tx < (v.w + 63) / 64; tx++) btiles.push_back(BlurTile{l, tx, ty, 0});
    }
    S->n_cells = (int)cells.size(); S->n_blur_tiles = (int)btiles.size();
    {   // dynamic LDS of k_fast_cells: sized for the largest cell of this pyramid
        int max_sh = 8, max_px = 64;
        for (const CellDesc& cd : cells) { max_sh = std::max(max_sh, cd.sh); max_px = std::max(max_px, (cd.sw - 6) * (cd.sh - 6)); }
        S->fast_rows = max_sh; S->fast_ncand = (max_px + 63) & ~63;
        S->fast_lds = 16 + (size_t)max_sh * FT_PITCH + (size_t)(max_sh - 4) * FS_PITCH + 2 * sizeof(uint16_t) * S->fast_ncand + 2 * sizeof(uint32_t) * VIDO_CELL_CAP + 16;
    }
    HIP_TRY(ctx, hipMalloc(&S->d_cells, cells.size() * sizeof(CellDesc)));
    HIP_TRY(ctx, hipMemcpy(S->d_cells, cells.data(), cells.size() * sizeof(CellDesc), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMalloc(&S->d_btiles, btiles.size() * sizeof(BlurTile)));
    HIP_TRY(ctx, hipMemcpy(S->d_btiles, btiles.data(), btiles.size() * sizeof(BlurTile), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMalloc(&S->d_xtab, std::max<size_t>(xtab.size(), 1) * sizeof(int2)));
    HIP_TRY(ctx, hipMalloc(&S->d_ytab, std::max<size_t>(ytab.size(), 1) * sizeof(int4)));
    if (!xtab.empty()) {
        HIP_TRY(ctx, hipMemcpy(S->d_xtab, xtab.data(), xtab.size() * sizeof(int2), hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(S->d_ytab, ytab.data(), ytab.size() * sizeof(int4), hipMemcpyHostToDevice));
    }
    HIP_TRY(ctx, hipMalloc(&S->d_first_cell, L * sizeof(int)));
    HIP_TRY(ctx, hipMemcpy(S->d_first_cell, S->first_cell.data(), L * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c_umax), S->umax, sizeof S->umax));
    HIP_TRY(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c_pattern), VIDO_ORB_PATTERN, 1024));
    return VIDO_OK;
}

int orb_state_create(vido_ctx* ctx)
{
    OrbState* S = new OrbState();
    ctx->orb = S;
    int rc = build_tables(ctx, S);
    if (rc != VIDO_OK) return rc;
    const size_t B = S->B;
    const size_t ncell = (size_t)S->n_cells * B;
    S->cand_cap = (size_t)VIDO_MAX_CAND_PER_FRAME * B;
    S->kp_cap = (size_t)(ctx->cfg.n_features * 2 + 256) * B;
    HIP_TRY(ctx, hipMalloc(&S->d_pyr, S->slab * B));
    HIP_TRY(ctx, hipMalloc(&S->d_blur, S->slab * B));
    HIP_TRY(ctx, hipMemset(S->d_pyr, 0, S->slab * B));
    HIP_TRY(ctx, hipMemset(S->d_blur, 0, S->slab * B));
    HIP_TRY(ctx, hipMalloc(&S->d_slots, ncell * VIDO_CELL_CAP * sizeof(uint32_t)));
    HIP_TRY(ctx, hipMalloc(&S->d_counts, ncell * sizeof(int)));
    HIP_TRY(ctx, hipMalloc(&S->d_offsets, (ncell + 1) * sizeof(int)));
    HIP_TRY(ctx, hipMalloc(&S->d_lvloff, (B * S->L + 1) * sizeof(int)));
    HIP_TRY(ctx, hipMalloc(&S->d_overflow, sizeof(int)));
    HIP_TRY(ctx, hipMemset(S->d_overflow, 0, sizeof(int)));
    HIP_TRY(ctx, hipMalloc(&S->d_cand, S->cand_cap * sizeof(uint32_t)));
    HIP_TRY(ctx, hipMalloc(&S->d_kp, S->kp_cap * sizeof(uint2)));
    HIP_TRY(ctx, hipMalloc(&S->d_angle, S->kp_cap * sizeof(float)));
    HIP_TRY(ctx, hipMalloc(&S->d_desc, S->kp_cap * 32));
    HIP_TRY(ctx, hipHostMalloc(&S->h_lvloff, (B * S->L + 1) * sizeof(int)));
    HIP_TRY(ctx, hipHostMalloc(&S->h_overflow, sizeof(int)));
    HIP_TRY(ctx, hipHostMalloc(&S->h_cand, S->cand_cap * sizeof(uint32_t)));
    HIP_TRY(ctx, hipHostMalloc(&S->h_kp, S->kp_cap * sizeof(uint2)));
    HIP_TRY(ctx, hipHostMalloc(&S->h_angle, S->kp_cap * sizeof(float)));
    HIP_TRY(ctx, hipHostMalloc(&S->h_desc, S->kp_cap * 32));
    for (auto& e : S->ev) HIP_TRY(ctx, hipEventCreate(&e));
    int nt = ctx->cfg.host_threads;
    if (nt <= 0) { nt = (int)std::thread::hardware_concurrency(); nt = std::max(1, std::min(nt, 32)); }
    S->n_threads = nt;
    S->pool = new WorkerPool(nt > 1 ? nt - 1 : 0);
    return VIDO_OK;
}

void orb_state_destroy(vido_ctx* ctx)
{
    OrbState* S = ctx->orb;
    if (!S) return;
    hipFree(S->d_pyr); hipFree(S->d_blur); hipFree(S->d_cells); hipFree(S->d_btiles); hipFree(S->d_xtab); hipFree(S->d_ytab);
    hipFree(S->d_slots); hipFree(S->d_counts); hipFree(S->d_offsets); hipFree(S->d_first_cell); hipFree(S->d_lvloff); hipFree(S->d_overflow);
    hipFree(S->d_cand); hipFree(S->d_kp); hipFree(S->d_angle); hipFree(S->d_desc);
    hipHostFree(S->h_lvloff); hipHostFree(S->h_overflow); hipHostFree(S->h_cand); hipHostFree(S->h_kp); hipHostFree(S->h_angle); hipHostFree(S->h_desc);
    for (auto& e : S->ev) if (e) hipEventDestroy(e);
    delete S->pool;
    delete S; ctx->orb = nullptr;
}

// ---- quadtree distribution (DistributeOctTree, ORBextractor.cc:529-753) -------------------------
// Host, serial per (frame, level).  Nodes are rectangles over an index array that is stably
// partitioned in place (the reference copies KeyPoint vectors); the std::list is an index-linked
// list over the node pool.  Size ties in the reference's sort of (size, node*) pairs fall back to
// node creation order here (the reference's pointer order is allocation-dependent).
namespace {
struct QNode { int x0, y0, x1, y1; int beg, cnt; int prev, next; bool no_more; };
struct QTree {
    std::vector<QNode> pool; std::vector<int> idx, tmp; int head = -1, tail = -1, size = 0;
    const float *cx, *cy;
    int add(int x0, int y0, int x1, int y1, int beg, int cnt) { pool.push_back(QNode{x0, y0, x1, y1, beg, cnt, -1, -1, cnt == 1}); return (int)pool.size() - 1; }
    void push_front(int i) { pool[i].prev = -1; pool[i].next = head; if (head >= 0) pool[head].prev = i; else tail = i; head = i; size++; }
    void push_back(int i) { pool[i].next = -1; pool[i].prev = tail; if (tail >= 0) pool[tail].next = i; else head = i; tail = i; size++; }
    int erase(int i) { int p = pool[i].prev, n = pool[i].next; if (p >= 0) pool[p].next = n; else head = n; if (n >= 0) pool[n].prev = p; else tail = p; size--; return n; }
    // split node i into up to four children; returns their pool indices in n1..n4 order (-1 = empty)
    void divide(int i, int ch[4])
    {
        const QNode nd = pool[i];
        const int hx = (int)ceilf((float)(nd.x1 - nd.x0) / 2), hy = (int)ceilf((float)(nd.y1 - nd.y0) / 2);
        const float mx = (float)(nd.x0 + hx), my = (float)(nd.y0 + hy);
        int cnt[4] = {0, 0, 0, 0};
        for (int k = nd.beg; k < nd.beg + nd.cnt; k++) {
            const int id = idx[k];
            const int q = cx[id] < mx ? (cy[id] < my ? 0 : 2) : (cy[id] < my ? 1 : 3);
            tmp[k] = q; cnt[q]++;
        }
        int start[4] = {nd.beg, nd.beg + cnt[0], nd.beg + cnt[0] + cnt[1], nd.beg + cnt[0] + cnt[1] + cnt[2]};
        int fill[4] = {start[0], start[1], start[2], start[3]};
        scratch.resize(nd.cnt);
        for (int k = 0; k < nd.cnt; k++) scratch[k] = idx[nd.beg + k];
        for (int k = 0; k < nd.cnt; k++) idx[fill[tmp[nd.beg + k]]++] = scratch[k];
        const int rx0[4] = {nd.x0, nd.x0 + hx, nd.x0, nd.x0 + hx}, ry0[4] = {nd.y0, nd.y0, nd.y0 + hy, nd.y0 + hy};
        const int rx1[4] = {nd.x0 + hx, nd.x1, nd.x0 + hx, nd.x1}, ry1[4] = {nd.y0 + hy, nd.y0 + hy, nd.y1, nd.y1};
        for (int q = 0; q < 4; q++) ch[q] = cnt[q] > 0 ? add(rx0[q], ry0[q], rx1[q], ry1[q], start[q], cnt[q]) : -1;
    }
    std::vector<int> scratch;
};
}  // namespace

static int quadtree_select(const float* cx, const float* cy, const float* resp, int n,
                           int minX, int maxX, int minY, int maxY, int N, std::vector<int>& out)
{
    out.clear();
    if (n <= 0) return 0;
    QTree T; T.cx = cx; T.cy = cy;
    T.pool.reserve(4 * (size_t)std::max(n, N) + 16); T.idx.resize(n); T.tmp.resize(n);
    int nIni = (int)roundf((float)(maxX - minX) / (maxY - minY));
    if (nIni < 1) nIni = 1;
    const float hX = (float)(maxX - minX) / nIni;
    {   // bucket keys into the initial column nodes, keeping input order inside each
        std::vector<int> bucket(n), cnt(nIni, 0), start(nIni, 0);
        for (int i = 0; i < n; i++) { int b = (int)(cx[i] / hX); if (b >= nIni) b = nIni - 1; bucket[i] = b; cnt[b]++; }
        for (int b = 1; b < nIni; b++) start[b] = start[b - 1] + cnt[b - 1];
        std::vector<int> fill = start;
        for (int i = 0; i < n; i++) T.idx[fill[bucket[i]]++] = i;
        for (int b = 0; b < nIni; b++) {
            int q = T.add((int)(hX * (float)b), 0, (int)(hX * (float)(b + 1)), maxY - minY, start[b], cnt[b]);
            if (cnt[b] > 0) T.push_back(q);
        }
    }
    struct SP { int size, node; };
    std::vector<SP> vs, vprev;
    bool finish = false;
    while (!finish) {
        const int prevSize = T.size;
        int nToExpand = 0;
        vs.clear();
        for (int it = T.head; it >= 0;) {
            if (T.pool[it].no_more) { it = T.pool[it].next; continue; }
            int ch[4]; T.divide(it, ch);
            for (int q = 0; q < 4; q++) if (ch[q] >= 0) {
                T.push_front(ch[q]);
                if (T.pool[ch[q]].cnt > 1) { nToExpand++; vs.push_back(SP{T.pool[ch[q]].cnt, ch[q]}); }
            }
            it = T.erase(it);
        }
        if (T.size >= N || T.size == prevSize) finish = true;
        else if (T.size + nToExpand * 3 > N) {
            while (!finish) {
                const int prev2 = T.size;
                vprev = vs; vs.clear();
                std::sort(vprev.begin(), vprev.end(), [](const SP& a, const SP& b) { return a.size != b.size ? a.size < b.size : a.node < b.node; });
                for (int j = (int)vprev.size() - 1; j >= 0; j--) {
                    int ch[4]; T.divide(vprev[j].node, ch);
                    for (int q = 0; q < 4; q++) if (ch[q] >= 0) {
                        T.push_front(ch[q]);
                        if (T.pool[ch[q]].cnt > 1) vs.push_back(SP{T.pool[ch[q]].cnt, ch[q]});
                    }
                    T.erase(vprev[j].node);
                    if (T.size >= N) break;
                }
                if (T.size >= N || T.size == prev2) finish = true;
            }
        }
    }
    for (int it = T.head; it >= 0; it = T.pool[it].next) {
        const QNode& q = T.pool[it];
        int best = T.idx[q.beg]; float mr = resp[best];
        for (int k = 1; k < q.cnt; k++) { const int id = T.idx[q.beg + k]; if (resp[id] > mr) { best = id; mr = resp[id]; } }
        out.push_back(best);
    }
    return (int)out.size();
}

static int orb_run(vido_ctx* ctx, const uint8_t* imgs, int on_device, int nf, size_t frame_stride, int stride, int width, int height,
                   vido_keypoint* kp_out, int max_kp, int* n_out, uint8_t* desc_out)
{
    OrbState* S = ctx->orb;
    if (!S) return vido_set_error(ctx, VIDO_E_INVALID, "orb: context has no ORB state");
    if (!imgs || !kp_out || !n_out || max_kp <= 0) return vido_set_error(ctx, VIDO_E_INVALID, "orb: null output/input");
    if (width != S->W || height != S->H) return vido_set_error(ctx, VIDO_E_INVALID, "orb: frame %dx%d but ctx was created for %dx%d", width, height, S->W, S->H);
    if (nf < 1 || nf > S->B) return vido_set_error(ctx, VIDO_E_INVALID, "orb: n_frames=%d outside [1,%d]", nf, S->B);
    if (stride < width) return vido_set_error(ctx, VIDO_E_INVALID, "orb: stride < width");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int L = S->L;
    auto t_start = std::chrono::steady_clock::now();
    HIP_TRY(ctx, hipEventRecord(S->ev[0], st));
    // level 0 <- input
    for (int f = 0; f < nf; f++)
        HIP_TRY(ctx, hipMemcpy2DAsync(S->d_pyr + (size_t)f * S->slab + S->lv[0].off, S->lv[0].pitch, imgs + (size_t)f * frame_stride, stride,
                                      width, height, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    for (int l = 1; l < L; l++) {
        const LevelInfo &s = S->lv[l - 1], &d = S->lv[l];
        dim3 grid((d.pitch / 4 + 63) / 64, (d.h + 3) / 4, nf), block(64, 4);
        hipLaunchKernelGGL(k_resize, grid, block, 0, st, S->d_pyr, S->slab, s.w, s.h, s.pitch, s.off, d.w, d.h, d.pitch, d.off,
                           S->d_xtab + d.xtab_off, S->d_ytab + d.ytab_off);
    }
    HIP_TRY(ctx, hipEventRecord(S->ev[1], st));
    hipLaunchKernelGGL(k_fast_cells, dim3(S->n_cells, nf), dim3(64), S->fast_lds, st, S->d_pyr, S->slab, S->P, S->d_cells, S->n_cells,
                       ctx->cfg.ini_th_fast, ctx->cfg.min_th_fast, S->fast_rows, S->fast_ncand, S->d_slots, S->d_counts);
    HIP_TRY(ctx, hipEventRecord(S->ev[7], st));
    hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, st, S->d_counts, S->n_cells * nf, S->d_offsets, S->n_cells, nf, L,
                       S->d_first_cell, S->d_lvloff, S->d_overflow);
    hipLaunchKernelGGL(k_gather_cands, dim3(S->n_cells, nf), dim3(64), 0, st, S->d_slots, S->d_counts, S->d_offsets, S->n_cells, S->d_cand, (int)S->cand_cap);
    HIP_TRY(ctx, hipEventRecord(S->ev[2], st));
    HIP_TRY(ctx, hipMemcpyAsync(S->h_lvloff, S->d_lvloff, ((size_t)nf * L + 1) * sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(S->h_overflow, S->d_overflow, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (*S->h_overflow) { hipMemsetAsync(S->d_overflow, 0, sizeof(int), st); return vido_set_error(ctx, VIDO_E_CAPACITY, "orb: a FAST cell produced more than %d corners", VIDO_CELL_CAP); }
    const int total = S->h_lvloff[nf * L];
    if ((size_t)total > S->cand_cap) return vido_set_error(ctx, VIDO_E_CAPACITY, "orb: %d FAST candidates exceed the %zu-entry buffer", total, S->cand_cap);
    if (total > 0) HIP_TRY(ctx, hipMemcpyAsync(S->h_cand, S->d_cand, (size_t)total * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipEventRecord(S->ev[3], st));
    // blur runs on the GPU while the host distributes keypoints
    if (ctx->cfg.compute_descriptors)
        hipLaunchKernelGGL(k_blur7, dim3(S->n_blur_tiles, nf), dim3(256), 0, st, S->d_pyr, S->d_blur, S->slab, S->P, S->d_btiles);
    HIP_TRY(ctx, hipEventRecord(S->ev[4], st));
    HIP_TRY(ctx, hipEventSynchronize(S->ev[3]));
    auto t_q0 = std::chrono::steady_clock::now();
    // ---- host quadtree per (frame, level)
    std::vector<std::vector<int>> sel((size_t)nf * L);
    std::vector<std::vector<float>> fx((size_t)nf * L), fy((size_t)nf * L), fr((size_t)nf * L);
    S->pool->run(nf * L, [&](int task) {
        const int l = task % L;
        const int beg = S->h_lvloff[task], end = S->h_lvloff[task + 1];
        const int n = end - beg;
        const LevelInfo& v = S->lv[l];
        const int minBX = EDGE_THRESHOLD - 3, minBY = minBX, maxBX = v.w - EDGE_THRESHOLD + 3, maxBY = v.h - EDGE_THRESHOLD + 3;
        auto &X = fx[task], &Y = fy[task], &R = fr[task];
        X.resize(n); Y.resize(n); R.resize(n);
        for (int i = 0; i < n; i++) {
            const uint32_t p = S->h_cand[beg + i];
            X[i] = (float)((int)(p & 0xfff) - minBX); Y[i] = (float)((int)((p >> 12) & 0xfff) - minBY); R[i] = (float)(p >> 24);
        }
        quadtree_select(X.data(), Y.data(), R.data(), n, minBX, maxBX, minBY, maxBY, v.n_budget, sel[task]);
    });
    // flatten the selection: frame-major, level-major, list order (== ORBextractor::operator() output order)
    std::vector<int> frame_beg(nf + 1, 0);
    size_t nk = 0;
    for (int f = 0; f < nf; f++) {
        frame_beg[f] = (int)nk;
        for (int l = 0; l < L; l++) {
            const int task = f * L + l;
            for (int id : sel[task]) {
                if (nk >= S->kp_cap) return vido_set_error(ctx, VIDO_E_CAPACITY, "orb: keypoint buffer overflow");
                const uint32_t p = S->h_cand[S->h_lvloff[task] + id];
                S->h_kp[nk] = make_uint2((p & 0xffffff) | ((uint32_t)l << 24), (uint32_t)f);
                S->h_angle[nk] = (float)(p >> 24);      // stash response; replaced by the angle after the kernel
                nk++;
            }
        }
    }
    frame_beg[nf] = (int)nk;
    std::vector<float> response(nk);
    for (size_t i = 0; i < nk; i++) response[i] = S->h_angle[i];
    auto t_q1 = std::chrono::steady_clock::now();
    HIP_TRY(ctx, hipEventRecord(S->ev[5], st));
    const int with_desc = ctx->cfg.compute_descriptors ? 1 : 0;
    if (nk > 0) {
        HIP_TRY(ctx, hipMemcpyAsync(S->d_kp, S->h_kp, nk * sizeof(uint2), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_orient_brief, dim3((unsigned)((((nk + 3) / 4) + 7) & ~(size_t)7)), dim3(256), 0, st, S->d_pyr, S->d_blur, S->slab, S->P, S->d_kp, (int)nk,
                           with_desc, S->d_angle, S->d_desc);
        HIP_TRY(ctx, hipMemcpyAsync(S->h_angle, S->d_angle, nk * sizeof(float), hipMemcpyDeviceToHost, st));
        if (with_desc) HIP_TRY(ctx, hipMemcpyAsync(S->h_desc, S->d_desc, nk * 32, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(ctx, hipEventRecord(S->ev[6], st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    HIP_TRY(ctx, hipGetLastError());
    // assemble cv::KeyPoint-equivalent output (ORBextractor.cc:827-836, 1094-1103)
    for (int f = 0; f < nf; f++) {
        const int n = frame_beg[f + 1] - frame_beg[f];
        n_out[f] = n;
        const int m = std::min(n, max_kp);
        for (int i = 0; i < m; i++) {
            const size_t g = (size_t)frame_beg[f] + i;
            const uint32_t p = S->h_kp[g].x;
            const int l = p >> 24;
            vido_keypoint& k = kp_out[(size_t)f * max_kp + i];
            float x = (float)(p & 0xfff), y = (float)((p >> 12) & 0xfff);
            if (l != 0) { x *= S->lv[l].scale; y *= S->lv[l].scale; }
            k.x = x; k.y = y; k.size = (float)(int)(31 * S->lv[l].scale); k.angle = S->h_angle[g]; k.response = response[g]; k.octave = l;
        }
        if (desc_out) {
            if (with_desc) memcpy(desc_out + (size_t)f * max_kp * 32, S->h_desc + (size_t)frame_beg[f] * 32, (size_t)m * 32);
            else memset(desc_out + (size_t)f * max_kp * 32, 0, (size_t)m * 32);
        }
    }
    auto t_end = std::chrono::steady_clock::now();
    float ms;
    hipEventElapsedTime(&ms, S->ev[0], S->ev[1]); S->timing[0] = ms;
    hipEventElapsedTime(&ms, S->ev[1], S->ev[7]); S->timing[1] = ms;
    hipEventElapsedTime(&ms, S->ev[7], S->ev[2]); S->timing[6] = ms;
    S->timing[7] = (float)S->h_lvloff[nf * L];
    S->timing[2] = std::chrono::duration<float, std::milli>(t_q1 - t_q0).count();
    hipEventElapsedTime(&ms, S->ev[3], S->ev[4]); S->timing[3] = ms;
    hipEventElapsedTime(&ms, S->ev[5], S->ev[6]); S->timing[4] = ms;
    S->timing[5] = std::chrono::duration<float, std::milli>(t_end - t_start).count();
    S->last_frames = nf;
    for (int f = 0; f < nf; f++) if (n_out[f] > max_kp) return vido_set_error(ctx, VIDO_E_CAPACITY, "orb: frame %d has %d keypoints, max_kp=%d", f, n_out[f], max_kp);
    return VIDO_OK;
}

extern "C" {

int vido_orb_extract(vido_ctx* ctx, const uint8_t* gray, int stride, int width, int height,
                     vido_keypoint* kp_out, int max_kp, int* n_out, uint8_t* desc_out)
{
    if (!ctx) return VIDO_E_INVALID;
    return orb_run(ctx, gray, 0, 1, 0, stride, width, height, kp_out, max_kp, n_out, desc_out);
}

int vido_orb_extract_batch(vido_ctx* ctx, const uint8_t* imgs, int on_device, int n_frames, size_t frame_stride, int stride,
                           int width, int height, vido_keypoint* kp_out, int max_kp, int* n_out, uint8_t* desc_out)
{
    if (!ctx) return VIDO_E_INVALID;
    return orb_run(ctx, imgs, on_device, n_frames, frame_stride, stride, width, height, kp_out, max_kp, n_out, desc_out);
}

int vido_orb_level_size(const vido_ctx* ctx, int level, int* lw, int* lh)
{
    if (!ctx || !ctx->orb || level < 0 || level >= ctx->orb->L) return VIDO_E_INVALID;
    if (lw) *lw = ctx->orb->lv[level].w;
    if (lh) *lh = ctx->orb->lv[level].h;
    return VIDO_OK;
}

int vido_orb_read_level(vido_ctx* ctx, int frame, int level, int blurred, uint8_t* out)
{
    if (!ctx || !ctx->orb || !out) return VIDO_E_INVALID;
    OrbState* S = ctx->orb;
    if (level < 0 || level >= S->L || frame < 0 || frame >= S->B) return vido_set_error(ctx, VIDO_E_INVALID, "read_level: bad frame/level");
    const LevelInfo& v = S->lv[level];
    const uint8_t* src = (blurred ? S->d_blur : S->d_pyr) + (size_t)frame * S->slab + v.off;
    HIP_TRY(ctx, hipMemcpy2D(out, v.w, src, v.pitch, v.w, v.h, hipMemcpyDeviceToHost));
    return VIDO_OK;
}

int vido_orb_read_candidates(vido_ctx* ctx, int frame, int level, uint32_t* out, int cap)
{
    if (!ctx || !ctx->orb) return VIDO_E_INVALID;
    OrbState* S = ctx->orb;
    if (level < 0 || level >= S->L || frame < 0 || frame >= S->last_frames) return vido_set_error(ctx, VIDO_E_INVALID, "read_candidates: bad frame/level");
    const int task = frame * S->L + level;
    const int beg = S->h_lvloff[task], n = S->h_lvloff[task + 1] - beg;
    if (out) for (int i = 0; i < std::min(n, cap); i++) out[i] = S->h_cand[beg + i];
    return n;
}

int vido_orb_last_timing(const vido_ctx* ctx, float ms[8])
{
    if (!ctx || !ctx->orb || !ms) return VIDO_E_INVALID;
    memcpy(ms, ctx->orb->timing, sizeof(float) * 8);
    return VIDO_OK;
}

}  // extern "C"
