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
//   * DistributeOctTree runs on the device too (k_quadtree, one workgroup per (frame, level)): the std::list walk
//     is restated as ordered-array passes built from LDS vote histograms, block scans and a bitonic sort, so the
//     candidates never leave HBM and the whole extractor is one stream of launches with a single sync at the end;
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
// 11-bit coefficients; (b0*(r0>>4))>>16 + (b1*(r1>>4))>>16 + 2 >> 2).
// One workgroup = a 128 x 8 output tile: the source rows/columns it touches (~156 x 12 bytes at scale 1.2) are staged
// into LDS with aligned dword loads (each source byte leaves HBM/L2 once, 4 B per lane instead of 1-byte gathers),
// every thread then produces 4 adjacent pixels of one row from LDS and stores one dword.
// K0: level 0 <- caller's frames (device-resident batch): one launch instead of one 2-D copy per frame
__global__ __launch_bounds__(256) void k_ingest(const uint8_t* __restrict__ imgs, size_t frame_stride, int stride, int width, int height,
                                                uint8_t* __restrict__ pyr, size_t slab, int off, int pitch)
{
    const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= height || x4 >= pitch) return;
    const uint8_t* s = imgs + (size_t)blockIdx.z * frame_stride + (size_t)y * stride + x4;
    uint32_t v = 0;
    if (x4 + 3 < width && (((uintptr_t)s) & 3) == 0) v = *(const uint32_t*)s;
    else { for (int k = 0; k < 4; k++) if (x4 + k < width) v |= (uint32_t)s[k] << (8 * k); }
    *(uint32_t*)(pyr + (size_t)blockIdx.z * slab + off + (size_t)y * pitch + x4) = v;
}

// K0c: cvtColor(BGR/RGB[A] -> GRAY) fused into the ingest (Tracking.cc:327-340 hands ORBextractor::operator() the converted image): interleaved u8
// pixels in, level 0 of the pyramid out — OpenCV's 14-bit fixed point Y = (B*1868 + G*9617 + R*4899 + 8192) >> 14.  One thread
// converts 4 adjacent pixels: 3 (or 4) aligned dword loads when the row allows, one dword store; optional tight gray copy for the caller.
template <int CN>
__global__ __launch_bounds__(256) void k_ingest_color(const uint8_t* __restrict__ imgs, size_t frame_stride, int stride, int width, int height, int rgb,
                                                      uint8_t* __restrict__ pyr, size_t slab, int off, int pitch, uint8_t* __restrict__ gray_out)
{
    const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= height || x4 >= pitch) return;
    const uint8_t* s = imgs + (size_t)blockIdx.z * frame_stride + (size_t)y * stride + (size_t)x4 * CN;
    uint8_t px[4 * CN];
    if (x4 + 3 < width && (((uintptr_t)s) & 3) == 0) {
        const uint32_t* q = (const uint32_t*)s;
#pragma unroll
        for (int k = 0; k < CN; k++) { const uint32_t v = q[k]; px[4 * k] = (uint8_t)v; px[4 * k + 1] = (uint8_t)(v >> 8); px[4 * k + 2] = (uint8_t)(v >> 16); px[4 * k + 3] = (uint8_t)(v >> 24); }
    } else {
#pragma unroll
        for (int k = 0; k < 4 * CN; k++) px[k] = (x4 + k / CN < width) ? s[k] : 0;
    }
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int c0 = px[k * CN], c1 = px[k * CN + 1], c2 = px[k * CN + 2];
        const int b = rgb ? c2 : c0, r = rgb ? c0 : c2;
        const uint32_t g = (uint32_t)(b * 1868 + c1 * 9617 + r * 4899 + 8192) >> 14;
        if (x4 + k < width) out |= g << (8 * k);
    }
    *(uint32_t*)(pyr + (size_t)blockIdx.z * slab + off + (size_t)y * pitch + x4) = out;
    if (gray_out) {
        uint8_t* g = gray_out + ((size_t)blockIdx.z * height + y) * width + x4;
        if (x4 + 3 < width && (width & 3) == 0) *(uint32_t*)g = out;
        else for (int k = 0; k < 4; k++) if (x4 + k < width) g[k] = (uint8_t)(out >> (8 * k));
    }
}

#define RS_TW 128
#define RS_TH 8
__global__ __launch_bounds__(256) void k_resize(uint8_t* __restrict__ pyr, size_t slab, int sw, int sh, int spitch, int soff,
                                                int dw, int dh, int dpitch, int doff,
                                                const int2* __restrict__ xtab, const int4* __restrict__ ytab, int lds_ndw)
{
    extern __shared__ uint32_t rs_lds[];
    uint8_t* base = pyr + (size_t)blockIdx.z * slab;
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * RS_TW, y0 = blockIdx.y * RS_TH;
    const int xl = min(x0 + RS_TW, dw) - 1, yl = min(y0 + RS_TH, dh) - 1;
    const int rs0 = ytab[y0].x, rs1 = ytab[yl].y;                          // source rows are monotone in y
    const int c0 = (xtab[x0].x & 0xffff) & ~3;
    const int c1 = min((xtab[xl].x & 0xffff) + 1, sw - 1);
    const int ndw = ((c1 - c0) >> 2) + 1, nrows = rs1 - rs0 + 1;
    const uint8_t* src = base + soff + (size_t)rs0 * spitch + c0;
    for (int rr = tid >> 6; rr < nrows; rr += 4)
        for (int dd = tid & 63; dd < ndw; dd += 64)
            rs_lds[rr * lds_ndw + dd] = *(const uint32_t*)(src + (size_t)rr * spitch + 4 * dd);
    __syncthreads();
    const int y = y0 + (tid >> 5), x4 = x0 + (tid & 31) * 4;
    if (y >= dh || x4 >= dpitch) return;
    const int4 yt = ytab[y];
    const uint8_t* L0 = (const uint8_t*)rs_lds + (yt.x - rs0) * lds_ndw * 4 - c0;
    const uint8_t* L1 = (const uint8_t*)rs_lds + (yt.y - rs0) * lds_ndw * 4 - c0;
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int x = x4 + k;
        if (x < dw) {
            const int2 xt = xtab[x];
            const int sx = xt.x & 0xffff, a0 = xt.x >> 16, a1 = xt.y;
            const int sx1 = sx + 1 < sw ? sx + 1 : sx;
            const int r0 = (int)(__umul24(L0[sx], (unsigned)a0) + __umul24(L0[sx1], (unsigned)a1));      // (24-bit multiplies: full rate; `*` became the quarter-rate v_mul_lo_u32)
            const int r1 = (int)(__umul24(L1[sx], (unsigned)a0) + __umul24(L1[sx1], (unsigned)a1));
            int v = (int)(((__umul24((unsigned)yt.z, (unsigned)(r0 >> 4)) >> 16) + (__umul24((unsigned)yt.w, (unsigned)(r1 >> 4)) >> 16) + 2u) >> 2);
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            out |= (uint32_t)v << (8 * k);
        }
    }
    *(uint32_t*)(base + doff + (size_t)y * dpitch + x4) = out;
}

// The whole pyramid of a frame in ONE launch (round 4; the tracker's single-frame path).  The seven k_resize launches above are a dependent chain — level l + 1 is resized
// from level l (ORBextractor.cc:1107-1132) — of ~5 us each; beside the networks every link of that chain queues for a CU again.  Here a workgroup owns a horizontal BAND of
// the image through all levels: it keeps, per level, the rows it owns plus the few rows the bands of the later levels need from it (PyrBand: computed on the host from the
// monotone row tables, walking back from the last level) in LDS, computes level l + 1's rows from level l's with the same fixed-point arithmetic, and writes the rows it owns.
// Rows near a band border are computed by both neighbours (about a third more resize work), there is no exchange between workgroups.
#define PB_MAXL 8
struct PyrBand { int need_lo[PB_MAXL], need_n[PB_MAXL], own_lo[PB_MAXL], own_hi[PB_MAXL], lds_off[PB_MAXL]; };      // per level: first needed row / count, owned rows [lo, hi), byte offset of the level's rows in LDS
struct PyrBandLv { int w[PB_MAXL], h[PB_MAXL], pitch[PB_MAXL], off[PB_MAXL], xoff[PB_MAXL], yoff[PB_MAXL]; int L; };
__global__ __launch_bounds__(256) void k_pyramid_bands(uint8_t* __restrict__ pyr, size_t slab, PyrBandLv V, const PyrBand* __restrict__ bands,
                                                       const int2* __restrict__ xtab, const int4* __restrict__ ytab)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t pb_lds[];
    const PyrBand& B = bands[blockIdx.x];
    uint8_t* base = pyr + (size_t)blockIdx.y * slab;
    const int tid = threadIdx.x;
    {   // level 0: the needed rows of the ingested image, dword copies
        const int ndw = V.pitch[0] >> 2, n = B.need_n[0] * ndw;
        const uint32_t* src = (const uint32_t*)(base + V.off[0] + (size_t)B.need_lo[0] * V.pitch[0]);
        uint32_t* dst = (uint32_t*)(pb_lds + B.lds_off[0]);
        for (int i = tid; i < n; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    for (int l = 1; l < V.L; l++) {
        const int sw = V.w[l - 1], spitch = V.pitch[l - 1], dw = V.w[l], dpitch = V.pitch[l], ndw = dpitch >> 2;
        const int2* xt = xtab + V.xoff[l]; const int4* yt = ytab + V.yoff[l];
        const uint8_t* S0 = pb_lds + B.lds_off[l - 1] - (size_t)B.need_lo[l - 1] * spitch;      // source row r lives at S0 + r * spitch
        uint8_t* D0 = pb_lds + B.lds_off[l];
        const int rows = B.need_n[l], ylo = B.need_lo[l];
        for (int i = tid; i < rows * ndw; i += 256) {
            const int rr = i / ndw, x4 = (i - rr * ndw) * 4, y = ylo + rr;
            const int4 ytv = yt[y];
            const uint8_t* L0 = S0 + (size_t)ytv.x * spitch; const uint8_t* L1 = S0 + (size_t)ytv.y * spitch;
            uint32_t out = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int x = x4 + k;
                if (x < dw) {
                    const int2 xv = xt[x];
                    const int sx = xv.x & 0xffff, a0 = xv.x >> 16, a1 = xv.y;
                    const int sx1 = sx + 1 < sw ? sx + 1 : sx;
                    const int r0 = (int)(__umul24(L0[sx], (unsigned)a0) + __umul24(L0[sx1], (unsigned)a1));
                    const int r1 = (int)(__umul24(L1[sx], (unsigned)a0) + __umul24(L1[sx1], (unsigned)a1));
                    int v = (int)(((__umul24((unsigned)ytv.z, (unsigned)(r0 >> 4)) >> 16) + (__umul24((unsigned)ytv.w, (unsigned)(r1 >> 4)) >> 16) + 2u) >> 2);
                    v = v < 0 ? 0 : (v > 255 ? 255 : v);
                    out |= (uint32_t)v << (8 * k);
                }
            }
            *(uint32_t*)(D0 + (size_t)rr * dpitch + x4) = out;
            if (y >= B.own_lo[l] && y < B.own_hi[l]) *(uint32_t*)(base + V.off[l] + (size_t)y * dpitch + x4) = out;
        }
        __syncthreads();
    }
}

// The whole pyramid of a BATCH in one launch (round 5): a workgroup owns a 2-D TILE of the image through all levels.  The seven k_resize launches are a dependent chain of
// latency-bound kernels (three dependent global round trips per workgroup: table corners -> source window -> table rows; 167 us per 64 frames = 7.5 % of the HBM
// roofline); the band kernel above needs the full image width in LDS (up to 150 KB: one workgroup per CU).  Here
//   * the host walks back from the last level once (build_tables): per tile and level the OWNED rectangle (a partition of the level, x cuts on multiples of 4) and the
//     NEEDED rectangle (owned + whatever the next level's needed rectangle samples), ~1.2x recomputed pixels at 128 x 60 level-0 tiles;
//   * the workgroup stages its level-0 rectangle and its slices of the coefficient tables into LDS in ONE round trip (table entries of columns past the level's width are
//     staged as zeros: they produce the 0 the pad bytes hold, no per-pixel test), then computes level l + 1 from level l entirely out of LDS — same fixed-point
//     arithmetic as k_resize, four pixels per item, one dword store into LDS (next level's source) and, for owned pixels, into the slab;
//   * ~30 KB of LDS per workgroup: five workgroups per CU, 40 tiles x frames workgroups per launch.
// Right image edge: the table has a1 = 0 where sx = sw - 1 (build_tables), so the unclamped L[sx + 1] read only ever meets weight 0.
#define PT_MAXL 8
struct PyrTile {
    int nx0[PT_MAXL], nw[PT_MAXL], ny0[PT_MAXL], nh[PT_MAXL];          // needed rectangle: first column (multiple of 4), columns (multiple of 4), first row, rows
    int ox0[PT_MAXL], ox1[PT_MAXL], oy0[PT_MAXL], oy1[PT_MAXL];        // owned rectangle [ox0, ox1) x [oy0, oy1), ox0 / ox1 multiples of 4
    int lds[PT_MAXL], xt_lds[PT_MAXL], yt_lds[PT_MAXL];                // byte offsets in LDS: the level's pixels [nh][nw], its x / y table slices
    unsigned magic[PT_MAXL];                                          // ceil(2^32 / (nw / 4)): item -> (row, quad) without a division
};
struct PyrTileLv { int w[PT_MAXL], pitch[PT_MAXL], off[PT_MAXL], xoff[PT_MAXL], yoff[PT_MAXL]; int L; };
// imgs != nullptr: level 0 comes straight from the caller's (device-resident, 4-byte aligned, width % 4 == 0) gray frames and the tile writes its owned part of it into the slab —
// the ingest copy (k_ingest: 15 us per 64 frames, a full read + write of level 0) rides on the staging loads.
__global__ __launch_bounds__(256) void k_pyramid_tiles(uint8_t* __restrict__ pyr, size_t slab, PyrTileLv V, const PyrTile* __restrict__ tiles,
                                                       const int2* __restrict__ xtab, const int4* __restrict__ ytab,
                                                       const uint8_t* __restrict__ imgs, size_t frame_stride, int stride, int ntiles, int total, int xcd_deal)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t pt_lds[];
    // Work item = (frame, tile), tiles of a frame in row-major order.  Workgroups go to the eight XCDs round-robin by their linear index, so with item = blockIdx two
    // tiles that meet at a column cut — whose row ends share a cache line of every level, and whose needed rectangles share halo pixels — always sit under DIFFERENT L2s:
    // the shared line leaves each L2 as a partial write and the halo is fetched from memory twice.  Dealt so that an XCD walks a contiguous range of items, the neighbours
    // run side by side on one L2 (round 6).
    const int per = gridDim.x >> 3, item = xcd_deal ? (blockIdx.x & 7) * per + (blockIdx.x >> 3) : (int)blockIdx.x;
    if (item >= total) return;
    const int frame = item / ntiles, tile = item - frame * ntiles;
    const PyrTile& T = tiles[tile];
    uint8_t* base = pyr + (size_t)frame * slab;
    const int tid = threadIdx.x;
    {   // one round trip: the level-0 rectangle (dword copies) and the table slices of every level
        const int nq = T.nw[0] >> 2, n = T.nh[0] * nq;
        const int spitch = imgs ? stride : V.pitch[0];
        const uint8_t* src = (imgs ? imgs + (size_t)frame * frame_stride : base + V.off[0]) + (size_t)T.ny0[0] * spitch + T.nx0[0];
        uint32_t* dst = (uint32_t*)(pt_lds + T.lds[0]);
        uint8_t* G0 = base + V.off[0];
        const int w0 = V.w[0], ox0 = T.ox0[0], ox1 = T.ox1[0], oy0 = T.oy0[0], oy1 = T.oy1[0];
        for (int i = tid; i < n; i += 256) {
            const int rr = (int)__umulhi((unsigned)i, T.magic[0]), q = i - rr * nq;
            const int y = T.ny0[0] + rr, x4 = T.nx0[0] + 4 * q;
            const uint32_t v = (!imgs || x4 < w0) ? *(const uint32_t*)(src + (size_t)rr * spitch + 4 * q) : 0u;      // (pad columns of the slab hold 0)
            dst[i] = v;
            if (imgs && y >= oy0 && y < oy1 && x4 >= ox0 && x4 < ox1) *(uint32_t*)(G0 + (size_t)y * V.pitch[0] + x4) = v;
        }
        for (int l = 1; l < V.L; l++) {
            int2* xs = (int2*)(pt_lds + T.xt_lds[l]); int4* ys = (int4*)(pt_lds + T.yt_lds[l]);
            const int2* xt = xtab + V.xoff[l]; const int4* yt = ytab + V.yoff[l];
            const int nx0 = T.nx0[l], nw = T.nw[l], dw = V.w[l];
            for (int i = tid; i < nw; i += 256) xs[i] = nx0 + i < dw ? xt[nx0 + i] : make_int2(T.nx0[l - 1], 0);      // (a column of the source rectangle, both weights 0)
            for (int i = tid; i < T.nh[l]; i += 256) ys[i] = yt[T.ny0[l] + i];
        }
    }
    __syncthreads();
    for (int l = 1; l < V.L; l++) {
        const int snw = T.nw[l - 1], nw = T.nw[l], nq = nw >> 2, n = T.nh[l] * nq, dpitch = V.pitch[l];
        const uint8_t* S0 = pt_lds + T.lds[l - 1] - T.ny0[l - 1] * snw - T.nx0[l - 1];       // source pixel (x, y) of level l - 1 lives at S0 + y * snw + x
        const int2* xs = (const int2*)(pt_lds + T.xt_lds[l]); const int4* ys = (const int4*)(pt_lds + T.yt_lds[l]);
        uint8_t* D = pt_lds + T.lds[l];
        uint8_t* G = base + V.off[l];
        const bool keep = l + 1 < V.L;
        const int nx0 = T.nx0[l], ny0 = T.ny0[l], ox0 = T.ox0[l], ox1 = T.ox1[l], oy0 = T.oy0[l], oy1 = T.oy1[l];
        const unsigned magic = T.magic[l];
        for (int i = tid; i < n; i += 256) {
            const int rr = (int)__umulhi((unsigned)i, magic), q = i - rr * nq;
            const int4 yt = ys[rr];
            // every product below has factors < 2^24 and fits 32 bits (pixels <= 255, weights <= 2048, r >> 4 <= 32 655): v_mul_u32_u24 / v_mad_u32_u24 at full rate —
            // plain `*` on values the compiler cannot bound became v_mul_lo_u32, a quarter-rate instruction, 18 times per item (half the kernel's issue time)
            const uint8_t* L0 = S0 + __umul24((unsigned)yt.x, (unsigned)snw); const uint8_t* L1 = S0 + __umul24((unsigned)yt.y, (unsigned)snw);
            const unsigned b0 = (unsigned)yt.z, b1 = (unsigned)yt.w;
            uint32_t out = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int2 xv = xs[4 * q + k];
                const unsigned sx = (unsigned)xv.x & 0xffffu, a0 = (unsigned)xv.x >> 16, a1 = (unsigned)xv.y;
                const unsigned r0 = __umul24(L0[sx], a0) + __umul24(L0[sx + 1], a1);
                const unsigned r1 = __umul24(L1[sx], a0) + __umul24(L1[sx + 1], a1);
                const unsigned v = ((__umul24(b0, r0 >> 4) >> 16) + (__umul24(b1, r1 >> 4) >> 16) + 2u) >> 2;      // <= 255: the weights sum to <= 2049 per axis (no clamp needed)
                out |= v << (8 * k);
            }
            if (keep) *(uint32_t*)(D + rr * nw + 4 * q) = out;
            const int y = ny0 + rr, x4 = nx0 + 4 * q;
            if (y >= oy0 && y < oy1 && x4 >= ox0 && x4 < ox1) *(uint32_t*)(G + (size_t)y * dpitch + x4) = out;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
typedef short v2s __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------------
// K2s: FAST 9/16 per STRIP of up to 4 horizontally adjacent reference cells, one wave64 per strip (round 2).
// Reference semantics (per-cell cv::FAST sub-images, 20-then-7 rule per cell, per-sub-image 3x3 NMS, row-major emission per
// cell: ORBextractor.cc:779-819), restructured for lane occupancy and instruction count:
//   * a wave owns the strip end to end — image tile, score tile and lists are wave-private LDS, so there is NO workgroup barrier anywhere;
//   * phase a (compass pre-test) runs over all quads of the strip (~1000 tasks: 15 full wave iterations; round 1 ran one wave per cell: 4 full + 1 ragged iteration each);
//     "at least two of the four compass pixels brighter than v+t" is "the second largest of them > v+t": a 4-element min/max network on packed
//     16-bit lanes (v_pk_min/max_u16), 36 VALU per quad instead of 58;
//   * surviving quads go to a quad list with ONE ballot per iteration; they are expanded to per-pixel candidates 64 quads at a time, so the
//     per-pixel ordered compaction (4 ballots + 4 conditional stores) runs once per 64 ACTIVE quads;
//   * phase b (segment test + score) takes 64 candidates of the whole strip per iteration (~99 % lane fill instead of ~60 %), knows from the pre-test
//     which family (bright / dark arcs) can fire and evaluates only that one, with the 9-wide sliding minima on packed 16-bit lanes
//     (v_pk_min_i16: 49 instead of 80 instructions), no separate has9 bit test: corner at t <=> max over arcs of the arc minimum > t;
//   * scores are laid out in the score tile with one zero gutter column between cells, so the per-sub-image NMS needs no boundary tests;
//   * nothing has a capacity that an image can exceed: the lists are sized for a chunk of rows, a chunk that overflows them is redone with half the
//     rows (4 rows always fit), the NMS of a chunk runs one chunk behind (it needs the first score row of the next one), and every cell has
//     ceil(w/2) * ceil(h/2) output slots — the most 3x3-NMS survivors a w x h sub-image can have.  cv::FAST has no limit either
//     (ORBextractor.cc:771-816 grows vToDistributeKeys without bound).
#define FS_MAXC 4              // most cells a strip may have (the strip's interior must also fit FS_MAXIW)
// Strip geometry for a strip interior of at most IW pixels.  Two instantiations: IW = 37 (one ~31..37-px cell per strip: ~5 KB of LDS per wave) when every cell of the
// configuration fits, else IW = 74 (cells are < 60 px wide by construction, ORBextractor.cc:781-787).  The kernel is VALU-bound (rocprofv3: 83 % of the SIMD cycles issue
// VALU), so short strips (more waves, less ragged list tails per chunk) beat longer candidate lists; per 64 frames of 640x480, r2 final form: 0.176 ms (37), 0.20 ms (74);
// the first strip kernel: 0.264 ms (148), 0.220 ms (74), 0.213 ms (37); round 1's one-workgroup-per-cell kernel: 0.197 ms.
template <int IW> struct FsGeom {
    static constexpr int PITCH = 4 * (((IW + 12) / 4) | 1);       // image tile pitch (bytes): interior + 6 frame + 3 alignment shift, an ODD number of dwords (conflict-free column walks)
    static constexpr int SPITCH = 4 * ((IW + FS_MAXC + 5) / 4);   // score tile pitch: interior + one gutter per cell + border
    static constexpr int AQ_CAP = 10 * IW + 32;                   // quad list: holds all quads of a ~35-row strip, so a strip is normally ONE chunk (full wave iterations: 0.73 -> 0.9 lane fill)
    static constexpr int CAND_CAP = 16 * IW + 64;                 // candidate list (~half of a strip's pixels): denser chunks are redone with half the rows (>= FS_RMIN rows x 2 entries per pixel fit).
                                                                  // LDS per wave decides the occupancy: 7.5 KB (21 waves per CU) runs as fast as 6 KB, +4 KB costs 20 % of the kernel
    static constexpr int CORN_CAP = CAND_CAP / 2;                 // corner lists (two: the NMS runs one chunk behind); a chunk with more corners is redone with half the rows
#ifndef FS_LDS_PAD
#define FS_LDS_PAD 0
#endif
    static size_t lds_bytes(int max_sh) { return 16 + (size_t)max_sh * PITCH + (size_t)(max_sh - 4) * SPITCH + sizeof(uint32_t) * AQ_CAP + sizeof(uint16_t) * (CAND_CAP + 2 + 2 * CORN_CAP) + 16 + FS_LDS_PAD; }
};
#ifndef FS_NARROW
#define FS_NARROW 37
#endif
#define FS_WIDE 74
#define FS_RMIN 2              // rows of the smallest chunk: its quads and its candidate entries (two per pixel at most) always fit the lists
struct FastStrip { int level, x0, y0, sw, sh, ncell, cell0, pad; int bx[FS_MAXC + 1]; int pad2[3]; };

typedef unsigned short v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2u as_v2u(uint32_t x) { return __builtin_bit_cast(v2u, x); }
__device__ __forceinline__ v2s as_v2s(uint32_t x) { return __builtin_bit_cast(v2s, x); }
__device__ __forceinline__ uint32_t as_u32(v2u x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ uint32_t as_u32(v2s x) { return __builtin_bit_cast(uint32_t, x); }

// inclusive prefix sum over the 64 lanes of a wave (all lanes active): four row_shr steps inside the rows of 16, then row_bcast:15 / row_bcast:31
__device__ __forceinline__ int wave_incl_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);          // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);          // row_bcast:31 into rows 2 and 3
    return v;
}

// Compass pre-test of the four pixels of an aligned quad.  Returns R with, for pixel p, the pair (bright, dark) at bits (15,14) p=0, (31,30) p=1,
// (13,12) p=2, (29,28) p=3: "bright" = two adjacent ones of ring 0/4/8/12 are > v + th, "dark" = two adjacent ones are < v - th (necessary for a 9-arc).
__device__ __forceinline__ uint32_t fast_compass_quad2(uint32_t up, uint32_t m0, uint32_t m1, uint32_t m2, uint32_t dn, uint32_t T2)
{
    const uint32_t E = __builtin_amdgcn_alignbyte(m2, m1, 3), W = __builtin_amdgcn_alignbyte(m1, m0, 1);
    uint32_t R = 0;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const uint32_t sel = h == 0 ? 0x0c010c00u : 0x0c030c02u;               // bytes (2h, 2h+1) -> two u16
        const v2u v = as_v2u(__builtin_amdgcn_perm(0u, m1, sel)), T = as_v2u(T2);
        const v2u r0 = as_v2u(__builtin_amdgcn_perm(0u, dn, sel)), r4 = as_v2u(__builtin_amdgcn_perm(0u, E, sel));
        const v2u r8 = as_v2u(__builtin_amdgcn_perm(0u, up, sel)), r12 = as_v2u(__builtin_amdgcn_perm(0u, W, sel));
#ifdef FS_COMPASS_ANY2
        const v2u a = __builtin_elementwise_max(r0, r4), b = __builtin_elementwise_min(r0, r4), c = __builtin_elementwise_max(r8, r12), d = __builtin_elementwise_min(r8, r12);
        const v2u x = __builtin_elementwise_min(a, c), y = __builtin_elementwise_max(b, d);
        const v2u sl = __builtin_elementwise_max(x, y), ss = __builtin_elementwise_min(x, y);      // second largest / second smallest of the four
#else
        // a 9-arc contains two ADJACENT compass points (90 degrees apart): sl = the largest over the four adjacent pairs of the pair's minimum, ss = the mirror image.
        // On the 4-cycle 0 - 4 - 8 - 12 every adjacent pair is one of {0, 8} with one of {4, 12}, and min / max distribute over each other, so
        //   max(min(r0,r4), min(r4,r8), min(r8,r12), min(r12,r0)) = min(max(r0,r8), max(r4,r12))     (round 5: 3 + 3 packed instructions instead of 8 + 6, same bits)
        const v2u sl = __builtin_elementwise_min(__builtin_elementwise_max(r0, r8), __builtin_elementwise_max(r4, r12));
        const v2u ss = __builtin_elementwise_max(__builtin_elementwise_min(r0, r8), __builtin_elementwise_min(r4, r12));
#endif
        const uint32_t br = as_u32((v2u)(v + T - sl)), dk = as_u32((v2u)(ss - (v - T)));             // 16-bit wrap-around: sign bit <=> sl > v + th, ss < v - th (|values| < 2^15)
        const uint32_t bits = (br & 0x80008000u) | ((dk & 0x80008000u) >> 1);
        R |= h == 0 ? bits : (bits >> 2);
    }
    return R;
}

// max over the 16 arcs of 9 contiguous ring pixels of the arc's minimum of X (8 registers of two u16: X[j] = (x[2j], x[2j+1]), values 0..255)
__device__ __forceinline__ int fast_arc_minmax(const uint32_t X[8])
{
    uint32_t A[8], B[8];
#pragma unroll
    for (int j = 0; j < 8; j++) A[j] = as_u32(__builtin_elementwise_min(as_v2u(X[j]), as_v2u(__builtin_amdgcn_alignbit(X[(j + 1) & 7], X[j], 16))));     // a2[k] = min(x[k], x[k+1])
#pragma unroll
    for (int j = 0; j < 8; j++) B[j] = as_u32(__builtin_elementwise_min(as_v2u(A[j]), as_v2u(A[(j + 1) & 7])));                                          // a4[k] = min(a2[k], a2[k+2])
#pragma unroll
    for (int j = 0; j < 8; j++) A[j] = as_u32(__builtin_elementwise_min(as_v2u(B[j]), as_v2u(B[(j + 2) & 7])));                                          // a8[k] = min(a4[k], a4[k+4])
#pragma unroll
    for (int j = 0; j < 8; j++) B[j] = as_u32(__builtin_elementwise_min(as_v2u(A[j]), as_v2u(X[(j + 4) & 7])));                                          // a9[k] = min(a8[k], x[k+8])
    v2u m = __builtin_elementwise_max(__builtin_elementwise_max(__builtin_elementwise_max(as_v2u(B[0]), as_v2u(B[1])), __builtin_elementwise_max(as_v2u(B[2]), as_v2u(B[3]))),
                                      __builtin_elementwise_max(__builtin_elementwise_max(as_v2u(B[4]), as_v2u(B[5])), __builtin_elementwise_max(as_v2u(B[6]), as_v2u(B[7]))));
    return max((int)m.x, (int)m.y);
}

// One-signed arc strength of the pixel at LDS tile position t for ONE arc family (dark != 0: the darker arcs): max over the 16 arcs of the arc's minimum one-signed
// difference to the centre; the pixel is a corner at threshold th iff the larger of its two families' strengths exceeds th, and cv::FAST's score (the largest
// threshold at which it is still a corner) is that maximum - 1.
// The differences are never formed: bright: max_arcs(min_arc p) - v; dark: v - min_arcs(max_arc p) = max_arcs(min_arc (255 - p)) - (255 - v), i.e. the same
// tree on p ^ 255 (one v_xor per register pair with a per-lane mask).  Sixteen byte reads at immediate offsets from one address (the compiler pairs horizontally
// adjacent ones into u16 reads) — measured faster than 7 unaligned b32 / b64 LDS reads + perms (unaligned LDS reads are slow) and than aligned dword pairs + v_alignbyte.
template <int FS_PITCH> __device__ __forceinline__ int fast_strength_pk(const uint8_t* t, bool dark)
{
    uint32_t P[8];
    // (volatile: keeps the compiler from fusing two horizontally adjacent bytes into one u16 read — at an odd address that read is split and stalls the LDS pipe:
    //  SQ_LDS_UNALIGNED_STALL went from 0 to 2.4e7 cycles per launch when it did)
    typedef const volatile __attribute__((address_space(3))) uint8_t* lds_vbyte_ptr;
    lds_vbyte_ptr tv = (lds_vbyte_ptr)t;
    auto pr = [&](int o0, int o1) { v2u p; p.x = tv[o0]; p.y = tv[o1]; return as_u32(p); };
    P[0] = pr(3 * FS_PITCH, 3 * FS_PITCH + 1);        P[1] = pr(2 * FS_PITCH + 2, FS_PITCH + 3);
    P[2] = pr(3, -FS_PITCH + 3);                      P[3] = pr(-2 * FS_PITCH + 2, -3 * FS_PITCH + 1);
    P[4] = pr(-3 * FS_PITCH, -3 * FS_PITCH - 1);      P[5] = pr(-2 * FS_PITCH - 2, -FS_PITCH - 3);
    P[6] = pr(-3, FS_PITCH - 3);                      P[7] = pr(2 * FS_PITCH - 2, 3 * FS_PITCH - 1);
    const uint32_t v = t[0];
    const uint32_t M = dark ? 0x00ff00ffu : 0u;
    uint32_t X[8];
#pragma unroll
    for (int j = 0; j < 8; j++) X[j] = P[j] ^ M;
    return fast_arc_minmax(X) - (int)(v ^ (M & 0xffu));
}

#ifdef FS_PROF
// debug build only (VIDO_EXTRA_FLAGS=-DFS_PROF): loop-trip counters of k_fast_strips, read by tools/dbg_fast_batch.py
__device__ unsigned long long fs_prof[16];
#define FS_CNT(i, v) do { if (lane == 0) atomicAdd(&fs_prof[i], (unsigned long long)(v)); } while (0)
extern "C" int vido_debug_fs_prof(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(fs_prof), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(fs_prof), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
#else
#define FS_CNT(i, v) do { } while (0)
#endif
// dynamic LDS per wave (bytes): [16 pad][tile sh x FS_PITCH][score (ih + 2) x FS_SPITCH][aq u32 x FS_AQ_CAP][cand u16 x FS_CAND_CAP][corner u16 x 2 x FS_CAND_CAP]
template <int IW> __global__ __launch_bounds__(64) void k_fast_strips(const uint8_t* __restrict__ pyr, size_t slab, PyrDev P,
                                                    const FastStrip* __restrict__ strips, int n_cells, const int* __restrict__ slot_off, int slot_total,
                                                    int ini_th, int min_th, int lds_rows, uint32_t* __restrict__ slots, int* __restrict__ counts)
{
    constexpr int FS_PITCH = FsGeom<IW>::PITCH, FS_SPITCH = FsGeom<IW>::SPITCH, FS_AQ_CAP = FsGeom<IW>::AQ_CAP, FS_CAND_CAP = FsGeom<IW>::CAND_CAP, FS_CORN_CAP = FsGeom<IW>::CORN_CAP;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    uint8_t* tile = lds_raw + 16;
    uint8_t* sc = tile + lds_rows * FS_PITCH;
    uint32_t* aq = (uint32_t*)(sc + (lds_rows - 4) * FS_SPITCH);
    uint16_t* cand = (uint16_t*)(aq + FS_AQ_CAP);
    uint16_t* corn = cand + FS_CAND_CAP + 2;                        // (cand[FS_CAND_CAP] is the expansion's dump slot) two buffers of FS_CORN_CAP
    int sidx, f; xcd_tile_frame(sidx, f);
    const int lane = threadIdx.x;
    const FastStrip S = strips[sidx];
    const int pitch = P.pitch[S.level];
    const uint8_t* img = pyr + (size_t)f * slab + P.off[S.level];
    const int xa = S.x0 & ~3, shift = S.x0 - xa;
    const int nd = ((S.x0 + S.sw + 3) >> 2) - (xa >> 2);          // dwords per tile row
    {   // lane = (row in group, dword column): 16 columns x 4 rows per load instruction for the narrow tile (nd <= 13), 32 x 2 for the wide one; no division, the five
        // loads of a batch are in flight together (one dependent HBM / L2 round trip per 20 / 10 rows).  Rows past the strip are clamped to its last row (the lanes
        // re-store that row's own data), so only the column test remains, hoisted out of the loop.
        constexpr int CB = FS_PITCH / 4 <= 16 ? 4 : 5, RPI = 64 >> CB;
        const int col = lane & ((1 << CB) - 1), rp = lane >> CB;
        if (col < nd) {
            const uint8_t* src = img + (size_t)S.y0 * pitch + xa + 4 * col;
            uint8_t* dst = tile + 4 * col;
            for (int r0 = 0; r0 < S.sh; r0 += 5 * RPI) {
                uint32_t v[5];
#pragma unroll
                for (int k = 0; k < 5; k++) { const int row = min(r0 + RPI * k + rp, S.sh - 1); v[k] = *(const uint32_t*)(src + (size_t)__umul24((unsigned)row, (unsigned)pitch)); }      // (24-bit multiplies: full rate)
#pragma unroll
                for (int k = 0; k < 5; k++) { const int row = min(r0 + RPI * k + rp, S.sh - 1); *(uint32_t*)(dst + __umul24((unsigned)row, (unsigned)FS_PITCH)) = v[k]; }
            }
        }
    }
    const int iw = S.sw - 6, ih = S.sh - 6;
    FS_CNT(11, 1); FS_CNT(14, iw * ih);
    for (int i = lane; i < (ih + 2) * (FS_SPITCH / 4); i += 64) ((uint32_t*)sc)[i] = 0;
    __builtin_amdgcn_s_waitcnt(0);                                  // single wave: LDS is coherent within it once the accesses have completed
    __builtin_amdgcn_wave_barrier();
    const int tx0 = shift + 3;                                      // tile column of interior x = 0
    const unsigned long long ltmask = (1ull << lane) - 1ull;
    const int bx1 = S.ncell > 1 ? S.bx[1] : 0x7fff, bx2 = S.ncell > 2 ? S.bx[2] : 0x7fff, bx3 = S.ncell > 3 ? S.bx[3] : 0x7fff;
    int ncnt[FS_MAXC] = {0, 0, 0, 0};
    uint32_t* out = slots + (size_t)f * slot_total;
    int so[FS_MAXC];
#pragma unroll
    for (int c = 0; c < FS_MAXC; c++) so[c] = c < S.ncell ? slot_off[S.cell0 + c] : 0;

    // per-sub-image NMS + ordered emission of the corners in list cl[0..n): strictly greater than the 8 neighbours (gutters / never-written entries are 0)
    auto nms_emit = [&](const uint16_t* cl, int n) {
        for (int q0 = 0; q0 < n; q0 += 64) {
            FS_CNT(6, 1); FS_CNT(7, min(64, n - q0));
            const int q = q0 + lane;
            bool keep = false; uint32_t packed = 0; int c = 0;
            if (q < n) {
                const int code = cl[q]; const int iy = code >> 10, ix = (code >> 2) & 255; c = code & 3;
                const uint8_t* s = sc + (iy + 1) * FS_SPITCH + ix + c + 1;
                const int Sv = s[0];
                const int n0 = s[-FS_SPITCH - 1], n1 = s[-FS_SPITCH], n2 = s[-FS_SPITCH + 1], n3 = s[-1], n4 = s[1], n5 = s[FS_SPITCH - 1], n6 = s[FS_SPITCH], n7 = s[FS_SPITCH + 1];
                const int mx = max(max(max(n0, n1), max(n2, n3)), max(max(n4, n5), max(n6, n7)));
                keep = Sv > mx;
                packed = (uint32_t)(S.x0 + 3 + ix) | ((uint32_t)(S.y0 + 3 + iy) << 12) | ((uint32_t)Sv << 24);
            }
#pragma unroll
            for (int cc = 0; cc < FS_MAXC; cc++) {
                const unsigned long long b = __ballot(keep && c == cc);
                if (keep && c == cc) out[so[cc] + ncnt[cc] + __popcll(b & ltmask)] = packed;
                ncnt[cc] += __popcll(b);
            }
        }
    };

    // one pass over interior columns [xlo, xhi) at threshold th
    auto process = [&](int xlo, int xhi, int th) {
        const int qc0 = (tx0 + xlo) >> 2, nq = ((tx0 + xhi - 1) >> 2) - qc0 + 1;
        const uint32_t T2 = (uint32_t)th | ((uint32_t)th << 16);
        // phase-a lane layout of this pass: lane = a_lrow * nq + a_q for a_lrow < a_rpi
        const int a_rpi = 64 / nq, a_lrow = lane / nq, a_q = lane - a_lrow * nq;
        const bool a_on = a_lrow < a_rpi;
        const int a_ixq = 4 * (qc0 + a_q) - tx0;                                    // interior x of the quad's first pixel (may be < xlo)
        uint32_t a_valid = 0;                                                       // (bright, dark) bit pairs of the quad's pixels inside [xlo, xhi), in the pre-test's layout
        {
            const int plo = max(0, xlo - a_ixq), phi = min(4, xhi - a_ixq);
            if (plo <= 0 && phi > 0) a_valid |= 0x0000c000u;
            if (plo <= 1 && phi > 1) a_valid |= 0xc0000000u;
            if (plo <= 2 && phi > 2) a_valid |= 0x00003000u;
            if (plo <= 3 && phi > 3) a_valid |= 0x30000000u;
        }
        const uint32_t a_code = ((uint32_t)a_lrow << 16) | ((uint32_t)(a_ixq + 4) << 8);
        const uint32_t* a_base = (const uint32_t*)tile + a_lrow * (FS_PITCH / 4) + qc0 + a_q;
        int R = min(ih, max(FS_RMIN, FS_AQ_CAP / nq)), r0 = 0, prev_n = 0, kb = 0;      // a chunk's quads always fit the quad list
        if (R < ih) { const int nch = (ih + R - 1) / R; R = (ih + nch - 1) / nch; }        // equal chunks, not a ragged last one      // a chunk's quads always fit the quad list (a first try with all rows failed for 9 strips in 10)
        FS_CNT(12, 1); FS_CNT(13, (xhi - xlo) * ih);
        while (r0 < ih) {
            const int r1 = min(r0 + R, ih);
            FS_CNT(8, 1);
            // ---- phase a: compass pre-test, one aligned quad per lane, surviving quads -> aq (ordered).  A lane keeps its quad column and steps down the rows
            // (rpi = 64 / nq rows per wave iteration): its valid-pixel mask, list code and tile address are loop invariants / one add per iteration.
            int naq = 0;
            {
                const uint32_t* bp = a_base + r0 * (FS_PITCH / 4);                 // tile row r0 = centre row r0 - 3
                uint32_t codea = a_code + ((uint32_t)r0 << 16);
                for (int r = r0; r < r1; r += a_rpi) {
                    FS_CNT(0, 1); FS_CNT(1, min(a_rpi, r1 - r) * nq);
                    uint32_t Rb = 0;
                    if (a_on && r + a_lrow < r1)
                        Rb = fast_compass_quad2(bp[0], bp[3 * (FS_PITCH / 4) - 1], bp[3 * (FS_PITCH / 4)], bp[3 * (FS_PITCH / 4) + 1], bp[6 * (FS_PITCH / 4)], T2) & a_valid;
                    const unsigned long long b = __ballot(Rb != 0);
                    if (Rb != 0) { const int pos = naq + __popcll(b & ltmask); if (pos < FS_AQ_CAP) aq[pos] = codea | (((Rb & 0xf000f000u) >> 12 | (Rb & 0xf000f000u) >> 24) & 0xffu); }
                    naq += __popcll(b);
                    bp += a_rpi * (FS_PITCH / 4); codea += (uint32_t)a_rpi << 16;
                }
            }
            if (naq > FS_AQ_CAP) { FS_CNT(9, 1); R = max((R + 1) >> 1, FS_RMIN); continue; }        // wave-uniform: redo this chunk with half the rows (FS_RMIN rows always fit)
            __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier();
            // ---- expansion: quads -> one candidate entry per (pixel, arc family the pre-test left possible), row-major order kept, a pixel's bright entry first.
            // Entry = (row << 10) | (ix << 2) | twin << 1 | dark; twin: the pixel has both entries (adjacent in the list).  Single-family entries keep phase b free of
            // the divergent "both families" block, which some lane of nearly every wave iteration used to need.
            int ncand = 0;
            for (int a0 = 0; a0 < naq; a0 += 64) {
                FS_CNT(2, 1); FS_CNT(3, min(64, naq - a0));
                const uint32_t e = a0 + lane < naq ? aq[a0 + lane] : 0u;
                const uint32_t fpx[4] = {(e >> 2) & 3, (e >> 6) & 3, e & 3, (e >> 4) & 3};       // flags byte = [3:2] px0, [7:6] px1, [1:0] px2, [5:4] px3; 2 = bright, 1 = dark
                const int n = __popc(e & 0xffu), incl = wave_incl_scan(n), tot = __builtin_amdgcn_readlane(incl, 63);
                if (ncand + tot > FS_CAND_CAP) { ncand += tot; break; }                            // wave-uniform: the chunk is redone with half the rows, nothing was stored
                int pos = ncand + incl - n;
                const int codeq = (int)((e >> 16) << 10) + (((int)((e >> 8) & 0xff) - 4) * 4);      // (row << 10) + (ix << 2) of pixel 0; its ix may be -3..-1 (that pixel is then invalid): sums, not ORs
#pragma unroll
                for (int k = 0; k < 4; k++) {                                                      // branch-free: an absent entry goes to the dump slot behind the list
                    const int code = (codeq + 4 * k) | (fpx[k] == 3 ? 2 : 0);
                    const int hb = (fpx[k] >> 1) & 1, hd = fpx[k] & 1;
                    cand[hb ? pos : FS_CAND_CAP] = (uint16_t)code;       pos += hb;
                    cand[hd ? pos : FS_CAND_CAP] = (uint16_t)(code | 1); pos += hd;
                }
                ncand += tot;
            }
            if (ncand > FS_CAND_CAP) { FS_CNT(10, 1); R = max((R + 1) >> 1, FS_RMIN); continue; }
            __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier();
            // ---- phase b: segment test + score, 64 entries per iteration; corners -> score tile + corner list (ordered).  The second entry of a twin pair takes the
            // maximum with its left neighbour's strength (DPP wave shift; across an iteration boundary through a scalar carry) and speaks for the pixel.
            uint16_t* cl = corn + kb * FS_CORN_CAP;
            int ncorn = 0, carry = 0;
            for (int q0 = 0; q0 < ncand; q0 += 64) {
                FS_CNT(4, 1); FS_CNT(5, min(64, ncand - q0));
                const int q = q0 + lane;
                int m = 0, code = 0, c = 0, ix = 0, iy = 0;
                if (q < ncand) {
                    code = cand[q];
                    iy = code >> 10; ix = (code >> 2) & 255;
                    m = fast_strength_pk<FS_PITCH>(tile + (iy + 3) * FS_PITCH + tx0 + ix, (code & 1) != 0);
                }
                const int mprev = __builtin_amdgcn_update_dpp(carry, m, 0x138, 0xf, 0xf, false);          // wave_shr:1 — lane l gets lane l-1, lane 0 keeps the carry
                carry = __builtin_amdgcn_readlane(m, 63);
                if ((code & 3) == 3) m = max(m, mprev);
                const bool speaks = q < ncand && (code & 3) != 2;                                         // not the first entry of a twin pair
                const int Sv = speaks && m > th ? m - 1 : 0;
                if (Sv > 0) {
                    c = (ix >= bx1) + (ix >= bx2) + (ix >= bx3);
                    sc[(iy + 1) * FS_SPITCH + ix + c + 1] = (uint8_t)Sv;
                }
                const unsigned long long bc = __ballot(Sv > 0);
                if (Sv > 0) { const int cp = ncorn + __popcll(bc & ltmask); if (cp < FS_CORN_CAP) cl[cp] = (uint16_t)((code & ~3) | c); }
                ncorn += __popcll(bc);
            }
            if (ncorn > FS_CORN_CAP) { FS_CNT(10, 1); R = max((R + 1) >> 1, FS_RMIN); continue; }      // (the scores already written are the ones the redo writes again)
            __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier();
            // ---- phase c, one chunk behind: the corners of the previous chunk now have their row + 1 scores
            nms_emit(corn + (kb ^ 1) * FS_CORN_CAP, prev_n);
            prev_n = ncorn; kb ^= 1; r0 = r1;
        }
        nms_emit(corn + (kb ^ 1) * FS_CORN_CAP, prev_n);
    };

    // The reference runs cv::FAST at iniThFAST and re-runs a cell at minThFAST only when that came back empty (ORBextractor.cc:799-806)
    process(0, iw, ini_th);
#pragma unroll 1
    for (int c = 0; c < S.ncell; c++) {                              // (not unrolled: one more copy of the pass instead of four)
        const int nc = c == 0 ? ncnt[0] : (c == 1 ? ncnt[1] : (c == 2 ? ncnt[2] : ncnt[3]));
        const int lo = c == 0 ? 0 : (c == 1 ? bx1 : (c == 2 ? bx2 : bx3)), hi = c + 1 == S.ncell ? iw : (c == 0 ? bx1 : (c == 1 ? bx2 : bx3));
        if (nc == 0) process(lo, hi, min_th);
    }
    if (lane < S.ncell) {
        const int v = lane == 0 ? ncnt[0] : (lane == 1 ? ncnt[1] : (lane == 2 ? ncnt[2] : ncnt[3]));
        counts[(size_t)f * n_cells + S.cell0 + lane] = v;
    }
}

// K3: exclusive scan of the per-cell counts (frame-major, reference cell order) -> dense offsets;
// also the per-(frame, level) start offsets the host needs.  One workgroup of 1024 threads.
__global__ __launch_bounds__(1024) void k_scan_counts(const int* __restrict__ counts, int n, int* __restrict__ offsets,
                                                      int n_cells, int n_frames, int n_levels,
                                                      const int* __restrict__ first_cell, int* __restrict__ lvloff)
{
    // every wave owns a contiguous segment and walks it 64 entries at a time (coalesced loads, shuffle scan); the 16 segment
    // totals are combined through LDS
    __shared__ int wtot[16];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int seg = (((n + 15) / 16) + 63) & ~63;
    const int beg = wave * seg, end = min(beg + seg, n);
    int s = 0;
#pragma unroll 4
    for (int i = beg + lane; i < end; i += 64) s += counts[i];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) wtot[wave] = s;
    __syncthreads();
    int run = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) { if (w < wave) run += wtot[w]; total += wtot[w]; }
#pragma unroll 4
    for (int i0 = beg; i0 < end; i0 += 64) {
        const int i = i0 + lane;
        const int c = i < end ? counts[i] : 0;
        int inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64); if (lane >= o) inc += v; }
        if (i < end) offsets[i] = run + inc - c;
        run += __shfl(inc, 63, 64);
    }
    if (tid == 0) offsets[n] = total;
    __syncthreads();
    __threadfence_block();
    for (int i = tid; i < n_frames * n_levels; i += 1024) {
        const int f = i / n_levels, l = i - f * n_levels;
        lvloff[i] = offsets[f * n_cells + first_cell[l]];
    }
    if (tid == 0) lvloff[n_frames * n_levels] = total;
}

// The same scan in two short launches (round 5): k_scan_counts above is ONE workgroup walking frames x cells counts twice (28 us at 64 frames x 870 cells — as long as
// a fifth of the FAST kernel it follows).  k_scan_frames: one workgroup per frame scans that frame's cells (local offsets, per-level local starts, the frame total);
// k_scan_apply: every entry adds its frame's base = the sum of the earlier frames' totals (<= max_batch values, summed by one wave).
__global__ __launch_bounds__(256) void k_scan_frames(const int* __restrict__ counts, int n_cells, int n_levels, const int* __restrict__ first_cell,
                                                     int* __restrict__ offsets, int* __restrict__ lvloff, int* __restrict__ totals)
{
    __shared__ int wtot[4];
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (n_cells + 255) / 256, beg = min(tid * per, n_cells), end = min(beg + per, n_cells);
    const int* c = counts + (size_t)f * n_cells; int* o = offsets + (size_t)f * n_cells;
    int s = 0;
    for (int i = beg; i < end; i++) s += c[i];
    int inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(inc, d, 64); if (lane >= d) inc += v; }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    int run = inc - s;
#pragma unroll
    for (int w = 0; w < 4; w++) if (w < wave) run += wtot[w];
    for (int i = beg; i < end; i++) { const int v = c[i]; o[i] = run; run += v; }
    if (tid == 255) totals[f] = run;                                 // (the last thread's running sum ends at the frame total: its segment is the last one, possibly empty)
    __syncthreads();
    __threadfence_block();
    if (tid < n_levels) lvloff[f * n_levels + tid] = o[first_cell[tid]];
}
__global__ __launch_bounds__(256) void k_scan_apply(int* __restrict__ offsets, int* __restrict__ lvloff, const int* __restrict__ totals, int n_cells, int n_levels, int n_frames)
{
    __shared__ int sbase;
    const int f = blockIdx.y, tid = threadIdx.x;
    if (tid < 64) {
        int b = 0;
        for (int g = tid; g < f; g += 64) b += totals[g];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) b += __shfl_xor(b, d, 64);
        if (tid == 0) sbase = b;
    }
    __syncthreads();
    const int base = sbase, i = blockIdx.x * 256 + tid;
    if (i < n_cells) offsets[(size_t)f * n_cells + i] += base;
    if (blockIdx.x == 0 && tid < n_levels) lvloff[f * n_levels + tid] += base;
    if (blockIdx.x == 0 && tid == 0 && f == n_frames - 1) { const int total = base + totals[f]; offsets[(size_t)n_frames * n_cells] = total; lvloff[n_frames * n_levels] = total; }
}

__global__ __launch_bounds__(64) void k_gather_cands(const uint32_t* __restrict__ slots, const int* __restrict__ counts, const int* __restrict__ offsets,
                                                     const int* __restrict__ slot_off, int slot_total, int n_cells, uint32_t* __restrict__ dense, int cap)
{
    const size_t ci = (size_t)blockIdx.y * n_cells + blockIdx.x;
    const int n = counts[ci], off = offsets[ci];
    const uint32_t* src = slots + (size_t)blockIdx.y * slot_total + slot_off[blockIdx.x];
    for (int i = threadIdx.x; i < n; i += 64) if (off + i < cap) dense[off + i] = src[i];
}

// ------------------------------------------------------------------------------------------------
// K3b: DistributeOctTree (ORBextractor.cc:529-753) on the device, one workgroup per (frame, level).
// The reference walks a std::list of nodes; what the result depends on is (a) which nodes get divided in which pass,
// (b) the list order (children are pushed to the FRONT in n1..n4 order while the pass iterates, undivided nodes keep
// their place), (c) the (size, creation order) sort of the last passes and its early stop at N nodes, (d) the
// best-response keypoint of every final node.  All of that is data-parallel once nodes are held as an ordered array:
//   pass:   candidates vote into their node's four quadrants (LDS atomics) -> per-node child counts;
//           processing order = list order (bulk passes) or (count, creation id) descending (final passes, bitonic sort),
//           cut at the first k whose running list size reaches N;
//           new list = [children of the k-th processed node, reversed] ... [children of the 1st, reversed] ++ [undivided
//           nodes in their old order]   — exactly what the push_front / erase sequence produces; positions and creation
//           ids come from block-wide exclusive scans;  candidates then move to their child slot.
//   finish: best candidate per node = max (response, lowest input index) by 64-bit LDS atomicMax; output in list order.
// Keypoints never leave the GPU: the former D2H of all candidates and the host threads are gone.
struct QtNode { short x0, y0, x1, y1; int cnt; int cid; };

// Workgroup barrier that orders LDS traffic only.  The one global array the quadtree passes rewrite (slot[i], the node of candidate i) is read and
// written by the thread that owns i in every pass, so nothing crosses threads through HBM — and __syncthreads() would also wait for those
// stores to retire (~1.5 us each time, ~100 barriers per task).
__device__ __forceinline__ void qt_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#ifndef QT_NT
#define QT_NT 512             // (round 5: 128 / 256 / 512 / 1024 threads -> 149 / 111 / 94 / 127 us per 64 frames)
#endif                     // threads of a quadtree workgroup (the passes are short dependent loops over <= ~5000 candidates / ~2000 nodes)
__device__ int qt_block_scan(int* a, int n, int* wsum)          // exclusive scan in place (LDS), returns the total; QT_NT threads
{
    const int t = threadIdx.x, per = (n + QT_NT - 1) / QT_NT, b = t * per;
    int s = 0;
    for (int i = 0; i < per; i++) if (b + i < n) s += a[b + i];
    int inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64); if ((t & 63) >= o) inc += v; }
    if ((t & 63) == 63) wsum[t >> 6] = inc;
    qt_barrier();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < QT_NT / 64; w++) { if (w < (t >> 6)) base += wsum[w]; total += wsum[w]; }
    int run = base + inc - s;
    for (int i = 0; i < per; i++) if (b + i < n) { const int v = a[b + i]; a[b + i] = run; run += v; }
    qt_barrier();
    return total;
}
__device__ __forceinline__ int qt_quadrant(const QtNode& nd, int x, int y)
{
    const int hx = (nd.x1 - nd.x0 + 1) >> 1, hy = (nd.y1 - nd.y0 + 1) >> 1;      // ceil(./2) of a non-negative int
    return (x < nd.x0 + hx ? 0 : 1) + (y < nd.y0 + hy ? 0 : 2);
}

__global__ __launch_bounds__(QT_NT) void k_quadtree(const uint32_t* __restrict__ cand, const int* __restrict__ lvloff, PyrDev P, int L,
                                                  const int* __restrict__ budget, int qcap, uint16_t* __restrict__ slot_scratch,
                                                  int* __restrict__ sel, int* __restrict__ selcnt, int* __restrict__ overflow, int cand_cap)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char qt_lds[];     // 8-byte LDS atomics on the `best` keys need the dynamic segment aligned (the static part is 20 bytes)
    QtNode* cur = (QtNode*)qt_lds;                       // [qcap]
    QtNode* nxt = cur + qcap;                            // [qcap]
    int* childcnt = (int*)(nxt + qcap);                  // [4*qcap]  (aliased by the 64-bit `best` keys at the end)
    int* childslot = childcnt + 4 * qcap;                // [4*qcap]
    int* nch = childslot + 4 * qcap;                     // [qcap]
    int* keep = nch + qcap;                              // [qcap]
    int* proc = keep + qcap;                             // [qcap]
    int* pre = proc + qcap;                              // [qcap]
    int* newslot = pre + qcap;                           // [qcap]
    unsigned long long* keys = (unsigned long long*)(newslot + qcap);   // [qcap]
    __shared__ int wsum[QT_NT / 64];
    __shared__ int sh_nexp;
    // workgroups are issued level-major, level 0 first: a (frame, level) task owns a CU's whole LDS, so a launch is two waves of workgroups on 256
    // CUs; in frame-major order both waves contain level-0 tasks (2 x the longest task), in this order the light levels fill in behind the heavy ones
    const int nfr = gridDim.x / L, l = blockIdx.x / nfr, task = (blockIdx.x - l * nfr) * L + l, tid = threadIdx.x;
    const int beg = lvloff[task], n = lvloff[task + 1] - beg, N = budget[l];
    // a FAST cell or the candidate buffer overflowed: the host reports VIDO_E_CAPACITY; nothing downstream may touch the lists
    if (n <= 0 || *overflow != 0 || lvloff[gridDim.x] > cand_cap) { if (tid == 0) selcnt[task] = 0; return; }     // (the candidate buffer is sized for the worst case; defensive)
    const int minB = EDGE_THRESHOLD - 3, width = P.w[l] - 2 * minB, height = P.h[l] - 2 * minB;
    const uint32_t* cd = cand + beg; uint16_t* slot = slot_scratch + beg;
    // ---- initial column nodes (ORBextractor.cc:534-560)
    int nIni = (int)roundf((float)width / (float)height); if (nIni < 1) nIni = 1;
    const float hX = (float)width / (float)nIni;
    if (nIni > qcap) { if (tid == 0) { atomicExch(overflow, 2); selcnt[task] = 0; } return; }
    for (int b = tid; b < nIni; b += QT_NT) { childcnt[b] = 0; }
    qt_barrier();
    for (int i = tid; i < n; i += QT_NT) {
        const int x = (int)(cd[i] & 0xfff) - minB;
        int b = (int)((float)x / hX); if (b >= nIni) b = nIni - 1;
        slot[i] = (uint16_t)b; atomicAdd(&childcnt[b], 1);
    }
    qt_barrier();
    for (int b = tid; b < nIni; b += QT_NT) keep[b] = childcnt[b] > 0 ? 1 : 0;
    qt_barrier();
    int Lc = qt_block_scan(keep, nIni, wsum);
    for (int b = tid; b < nIni; b += QT_NT) {
        newslot[b] = keep[b];
        if (childcnt[b] > 0) { QtNode q; q.x0 = (short)(int)(hX * (float)b); q.y0 = 0; q.x1 = (short)(int)(hX * (float)(b + 1)); q.y1 = (short)height; q.cnt = childcnt[b]; q.cid = b; cur[keep[b]] = q; }
    }
    qt_barrier();
    for (int i = tid; i < n; i += QT_NT) slot[i] = (uint16_t)newslot[slot[i]];
    int counter = nIni;
    bool final_mode = false, finish = false;
    qt_barrier();
    while (!finish) {
        // ---- votes
        for (int s = tid; s < 4 * Lc; s += QT_NT) childcnt[s] = 0;
        if (tid == 0) sh_nexp = 0;
        qt_barrier();
        for (int i = tid; i < n; i += QT_NT) {
            const int s = slot[i]; const QtNode nd = cur[s];
            if (nd.cnt > 1) { const uint32_t p = cd[i]; atomicAdd(&childcnt[4 * s + qt_quadrant(nd, (int)(p & 0xfff) - minB, (int)((p >> 12) & 0xfff) - minB)], 1); }
        }
        qt_barrier();
        for (int s = tid; s < Lc; s += QT_NT) {
            const bool ex = cur[s].cnt > 1;
            nch[s] = ex ? (childcnt[4 * s] > 0) + (childcnt[4 * s + 1] > 0) + (childcnt[4 * s + 2] > 0) + (childcnt[4 * s + 3] > 0) : 0;
            keep[s] = ex ? 1 : 0;                        // reused: expandable flag -> rank
        }
        qt_barrier();
        // ---- processing order proc[0..k)
        int k;
        if (!final_mode) {
            k = qt_block_scan(keep, Lc, wsum);
            for (int s = tid; s < Lc; s += QT_NT) if (cur[s].cnt > 1) proc[keep[s]] = s;
            qt_barrier();
        } else {
            const int E = qt_block_scan(keep, Lc, wsum);
            int M = 1; while (M < E) M <<= 1;
            for (int j = tid; j < M; j += QT_NT) keys[j] = 0ull;
            qt_barrier();
            for (int s = tid; s < Lc; s += QT_NT) if (cur[s].cnt > 1)
                keys[keep[s]] = ((unsigned long long)cur[s].cnt << 42) | ((unsigned long long)(unsigned)cur[s].cid << 10) | (unsigned long long)s;
            qt_barrier();
            for (int kk = 2; kk <= M; kk <<= 1)
                for (int j = kk >> 1; j > 0; j >>= 1) {
                    for (int i = tid; i < M; i += QT_NT) {
                        const int ixj = i ^ j;
                        if (ixj > i) {
                            const unsigned long long a = keys[i], b = keys[ixj];
                            const bool desc = (i & kk) == 0;
                            if (desc ? (a < b) : (a > b)) { keys[i] = b; keys[ixj] = a; }
                        }
                    }
                    qt_barrier();
                }
            for (int j = tid; j < E; j += QT_NT) { proc[j] = (int)(keys[j] & 1023ull); pre[j] = nch[proc[j]] - 1; }
            qt_barrier();
            qt_block_scan(pre, E, wsum);                 // list growth before processing the j-th node
            if (tid == 0) sh_nexp = 0;
            qt_barrier();
            int mine = 0;
            for (int j = tid; j < E; j += QT_NT) if (Lc + pre[j] < N) mine++;
            if (mine) atomicAdd(&sh_nexp, mine);
            qt_barrier();
            k = sh_nexp;
            qt_barrier();
            if (tid == 0) sh_nexp = 0;
        }
        // ---- positions
        for (int j = tid; j < k; j += QT_NT) pre[j] = nch[proc[j]];
        for (int s = tid; s < Lc; s += QT_NT) keep[s] = 1;
        qt_barrier();
        for (int j = tid; j < k; j += QT_NT) keep[proc[j]] = 0;
        qt_barrier();
        const int C = qt_block_scan(pre, k, wsum);
        const int nkeep = qt_block_scan(keep, Lc, wsum);
        const int Lnew = C + nkeep;
        if (Lnew > qcap) { if (tid == 0) { atomicExch(overflow, 2); selcnt[task] = 0; } return; }
        // keep[] now holds exclusive ranks for EVERY slot; a divided slot is recognised through childslot >= 0 below
        for (int s = tid; s < 4 * Lc; s += QT_NT) childslot[s] = -1;
        qt_barrier();
        int nexp_local = 0;
        for (int j = tid; j < k; j += QT_NT) {
            const int s = proc[j]; const QtNode nd = cur[s];
            const int hx = (nd.x1 - nd.x0 + 1) >> 1, hy = (nd.y1 - nd.y0 + 1) >> 1;
            const int c0 = childcnt[4 * s], c1 = childcnt[4 * s + 1], c2 = childcnt[4 * s + 2], c3 = childcnt[4 * s + 3];
            const int cc[4] = {c0, c1, c2, c3};
            const int base = C - pre[j] - nch[s];
            int before = 0;
            for (int q = 0; q < 4; q++) {
                if (cc[q] <= 0) continue;
                int after = 0; for (int r = q + 1; r < 4; r++) after += cc[r] > 0;
                const int pos = base + after;
                QtNode ch;
                ch.x0 = (short)((q & 1) ? nd.x0 + hx : nd.x0); ch.x1 = (short)((q & 1) ? nd.x1 : nd.x0 + hx);
                ch.y0 = (short)((q & 2) ? nd.y0 + hy : nd.y0); ch.y1 = (short)((q & 2) ? nd.y1 : nd.y0 + hy);
                ch.cnt = cc[q]; ch.cid = counter + pre[j] + before;
                nxt[pos] = ch; childslot[4 * s + q] = pos;
                if (cc[q] > 1) nexp_local++;
                before++;
            }
        }
        if (nexp_local) atomicAdd(&sh_nexp, nexp_local);
        qt_barrier();
        for (int s = tid; s < Lc; s += QT_NT) {
            const bool divided = cur[s].cnt > 1 && (childslot[4 * s] >= 0 || childslot[4 * s + 1] >= 0 || childslot[4 * s + 2] >= 0 || childslot[4 * s + 3] >= 0);
            if (!divided) { nxt[C + keep[s]] = cur[s]; newslot[s] = C + keep[s]; } else newslot[s] = -1;
        }
        qt_barrier();
        for (int i = tid; i < n; i += QT_NT) {
            const int s = slot[i];
            if (newslot[s] >= 0) slot[i] = (uint16_t)newslot[s];
            else { const QtNode nd = cur[s]; const uint32_t p = cd[i]; slot[i] = (uint16_t)childslot[4 * s + qt_quadrant(nd, (int)(p & 0xfff) - minB, (int)((p >> 12) & 0xfff) - minB)]; }
        }
        const int nToExpand = sh_nexp;
        qt_barrier();
        { QtNode* t = cur; cur = nxt; nxt = t; }
        counter += C;
        if (Lnew >= N || Lnew == Lc) finish = true;
        else if (!final_mode && Lnew + 3 * nToExpand > N) final_mode = true;
        Lc = Lnew;
    }
    // ---- best keypoint per node, list order
    // key = response << 32 | ~input index (64-bit LDS max: a level can hold more than 65535 candidates — a noise image yields one per 2x2 pixels)
    unsigned long long* best = (unsigned long long*)childcnt;        // childcnt..childslot: 8 * qcap ints
    for (int s = tid; s < Lc; s += QT_NT) best[s] = 0ull;
    qt_barrier();
    for (int i = tid; i < n; i += QT_NT) atomicMax(&best[slot[i]], ((unsigned long long)(cd[i] >> 24) << 32) | (unsigned long long)(0xffffffffu - (unsigned)i));
    qt_barrier();
    for (int s = tid; s < Lc; s += QT_NT) sel[(size_t)task * qcap + s] = (int)(0xffffffffu - (unsigned)(best[s] & 0xffffffffull));
    if (tid == 0) selcnt[task] = Lc;
}

// keypoint list: frame-major, level-major, list order (== ORBextractor::operator() output order)
__global__ __launch_bounds__(1024) void k_kp_offsets(const int* __restrict__ selcnt, int n_tasks, int L, int* __restrict__ kpoff, int* __restrict__ frame_beg, int nf)
{
    __shared__ int part[1024];
    const int t = threadIdx.x, per = (n_tasks + 1023) / 1024, b = t * per;
    int s = 0;
    for (int i = 0; i < per; i++) if (b + i < n_tasks) s += selcnt[b + i];
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) { const int v = t >= o ? part[t - o] : 0; __syncthreads(); part[t] += v; __syncthreads(); }
    int run = part[t] - s;
    for (int i = 0; i < per; i++) if (b + i < n_tasks) { kpoff[b + i] = run; if ((b + i) % L == 0) frame_beg[(b + i) / L] = run; run += selcnt[b + i]; }
    if (t == 1023) { kpoff[n_tasks] = part[1023]; frame_beg[nf] = part[1023]; }
}
struct KpLevels { float scale[VIDO_MAX_LEVELS]; float size[VIDO_MAX_LEVELS]; };
// packed (x | y<<12 | level<<24, frame) records for k_orient_brief + the final cv::KeyPoint-equivalent rows
// (ORBextractor.cc:827-836, 1094-1103) in [frame][row_cap] layout; the angle is filled in by k_orient_brief
__global__ __launch_bounds__(256) void k_kp_write(const uint32_t* __restrict__ cand, const int* __restrict__ lvloff, const int* __restrict__ sel, const int* __restrict__ selcnt,
                                                 const int* __restrict__ kpoff, int qcap, int L, uint2* __restrict__ kps, int kp_cap,
                                                 vido_keypoint* __restrict__ kpf, int row_cap, int* __restrict__ nkp, KpLevels lv)
{
    const int task = blockIdx.x, f = task / L, l = task - f * L, beg = lvloff[task], m = selcnt[task], off = kpoff[task], fbeg = kpoff[f * L];
    if (l == 0 && threadIdx.x == 0) nkp[f] = kpoff[(f + 1) * L] - fbeg;
    for (int i = threadIdx.x; i < m; i += 256) {
        if (off + i >= kp_cap) break;
        const uint32_t p = cand[beg + sel[(size_t)task * qcap + i]];
        kps[off + i] = make_uint2((p & 0xffffffu) | ((uint32_t)l << 24), (uint32_t)f);
        const int row = off + i - fbeg;
        if (row < row_cap) {
            float x = (float)(p & 0xfff), y = (float)((p >> 12) & 0xfff);
            if (l != 0) { x *= lv.scale[l]; y *= lv.scale[l]; }
            vido_keypoint k; k.x = x; k.y = y; k.size = lv.size[l]; k.angle = -1.f; k.response = (float)(p >> 24); k.octave = l;
            kpf[(size_t)f * row_cap + row] = k;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K4: 7x7 sigma=2 Gaussian, reflect-101, Q0.8 taps {18,34,49,54,49,34,18}; 64x16 output tile per WG.
__device__ __forceinline__ int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) { i = i < 0 ? -i : 2 * n - 2 - i; }
    return i;
}
// 128 x 32 output tile per workgroup, everything in dword units: the (32+6) x (128+8)-byte input window is staged with aligned
// dword loads (only dwords that touch the image border take the per-byte reflect path), the horizontal pass turns three
// LDS dwords into four u16 sums, the vertical pass reads seven 8-byte LDS words per four outputs and stores one dword.
#define BL_TW 128
#define BL_TH 32
#define BL_IN_STRIDE 35            // dwords per staged row (34 used; odd stride spreads rows over the LDS banks)
__device__ __forceinline__ uint32_t blur_load4(const uint8_t* __restrict__ row, int c, int w)
{
    if (c >= 0 && c + 3 < w) return *(const uint32_t*)(row + c);
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) v |= (uint32_t)row[reflect101(c + k, w)] << (8 * k);
    return v;
}
__global__ __launch_bounds__(256) void k_blur7(const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur, size_t slab, PyrDev P,
                                               const BlurTile* __restrict__ tiles)
{
    __shared__ uint32_t in[(BL_TH + 6) * BL_IN_STRIDE];
    __shared__ uint2 hb[(BL_TH + 6) * (BL_TW / 4)];
    int ti, fr; xcd_tile_frame(ti, fr);
    const BlurTile t = tiles[ti];
    const int w = P.w[t.level], h = P.h[t.level], pitch = P.pitch[t.level];
    const uint8_t* img = pyr + (size_t)fr * slab + P.off[t.level];
    uint8_t* out = blur + (size_t)fr * slab + P.off[t.level];
    const int tid = threadIdx.x, x0 = t.tx * BL_TW, y0 = t.ty * BL_TH;
    for (int r = tid >> 6; r < BL_TH + 6; r += 4) {
        const int d = tid & 63;
        if (d < 34) in[r * BL_IN_STRIDE + d] = blur_load4(img + (size_t)reflect101(y0 + r - 3, h) * pitch, x0 - 4 + 4 * d, w);
    }
    __syncthreads();
    for (int i = tid; i < (BL_TH + 6) * (BL_TW / 4); i += 256) {
        const int r = i >> 5, d = i & 31;
        const uint32_t* q = in + r * BL_IN_STRIDE + d;
        const uint32_t w0 = q[0], w1 = q[1], w2 = q[2];
        // bytes 1..10 of the 12-byte window: output k uses bytes k+1 .. k+7
        const int b1 = (w0 >> 8) & 255, b2 = (w0 >> 16) & 255, b3 = w0 >> 24, b4 = w1 & 255, b5 = (w1 >> 8) & 255, b6 = (w1 >> 16) & 255, b7 = w1 >> 24,
                  b8 = w2 & 255, b9 = (w2 >> 8) & 255, b10 = (w2 >> 16) & 255;
        const uint32_t s0 = 18 * (b1 + b7) + 34 * (b2 + b6) + 49 * (b3 + b5) + 54 * b4;
        const uint32_t s1 = 18 * (b2 + b8) + 34 * (b3 + b7) + 49 * (b4 + b6) + 54 * b5;
        const uint32_t s2 = 18 * (b3 + b9) + 34 * (b4 + b8) + 49 * (b5 + b7) + 54 * b6;
        const uint32_t s3 = 18 * (b4 + b10) + 34 * (b5 + b9) + 49 * (b6 + b8) + 54 * b7;
        hb[i] = make_uint2(s0 | (s1 << 16), s2 | (s3 << 16));
    }
    __syncthreads();
    for (int i = tid; i < BL_TH * (BL_TW / 4); i += 256) {
        const int y = i >> 5, d = i & 31;
        if (y0 + y >= h || x0 + 4 * d >= pitch) continue;
        uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        const int taps[7] = {18, 34, 49, 54, 49, 34, 18};
#pragma unroll
        for (int k = 0; k < 7; k++) {
            const uint2 v = hb[(y + k) * (BL_TW / 4) + d];
            a0 += taps[k] * (v.x & 0xffff); a1 += taps[k] * (v.x >> 16); a2 += taps[k] * (v.y & 0xffff); a3 += taps[k] * (v.y >> 16);
        }
        *(uint32_t*)(out + (size_t)(y0 + y) * pitch + x0 + 4 * d) =
            ((a0 + 32768) >> 16) | (((a1 + 32768) >> 16) << 8) | (((a2 + 32768) >> 16) << 16) | (((a3 + 32768) >> 16) << 24);
    }
}

// K4w (round 5): the same blur WITHOUT LDS and without workgroup barriers — one wave per strip of BS_QUADS x 4 output columns, sliding down the rows.
// k_blur7 above spends ~33 vector instructions per pixel (byte extraction, 32-bit multiply-adds, two LDS round trips, two barriers) and runs at 13 % of the HBM roofline;
// this form needs ~11:
//   * lane l owns the aligned quad of columns c = x0 - 4 + 4 l (lanes 0 and 63 are halo only) and keeps the last seven rows of it UNPACKED as two registers of two
//     u16 (bytes 0,1 | bytes 2,3): a new row costs one dword load and two v_perm;
//   * vertical pass first, on packed 16-bit lanes: S = sum_k tap[k] * row[k] as v_pk_mul_lo_u16 / v_pk_mad_u16 (the column sums are <= 255 * 256 = 65280: exact in u16);
//   * horizontal pass on the column sums with v_dot2_u32_u16: the six pairs (s[-4],s[-3]) .. (s[6],s[7]) around the quad are the lane's own two registers and its
//     neighbours' (wave_shr / wave_shl DPP moves), each output is four dot2 with constant tap pairs, the rounding constant rides in the first accumulator; the result
//     byte is bits 23:16 of the accumulator (<= 255 * 65536 + 32768), picked out by two v_perm;
//   * reflect-101 columns (left edge, right edge, a partial last quad) are built ONCE per loaded row from real lanes' dwords: every virtual byte is some real byte of
//     at most two other lanes of the wave — two ds_bpermute + one v_perm with per-lane constants, only in waves that contain an edge; reflect-101 rows are a scalar row
//     index.  Integer arithmetic throughout, no intermediate rounding: identical to the horizontal-then-vertical order of k_blur7 and of the CPU checker.
#define BS_QUADS 62                // output quads per wave (lanes 1 .. 62)
// reflect-101 for an index at most n - 1 outside [0, n): one fold per side, no loop (the host only builds strips for levels where that holds); clamped, so that the rows
// below the image a last strip still loads (and never stores an output for) stay in bounds
__device__ __forceinline__ int reflect101_once(int i, int n) { i = i < 0 ? -i : i; i = i >= n ? 2 * n - 2 - i : i; return min(max(i, 0), n - 1); }
typedef unsigned short bs_v2u __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(64) void k_blur7_strips(const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur, size_t slab, PyrDev P, const BlurTile* __restrict__ strips)
{
    int ti, fr; xcd_tile_frame(ti, fr);
    const BlurTile t = strips[ti];                                   // {level, first output column (multiple of 4), first output row, batches of 7 input rows}
    const int w = P.w[t.level], h = P.h[t.level], pitch = P.pitch[t.level];
    const uint8_t* img = pyr + (size_t)fr * slab + P.off[t.level];
    uint8_t* out = blur + (size_t)fr * slab + P.off[t.level];
    const int lane = threadIdx.x, x0 = t.tx, y0 = t.ty, nb = t.pad;
    const int c = x0 - 4 + 4 * lane;                                 // this lane's quad of (virtual) columns
    // reflect-101 fix-up constants: source lanes A / B (byte addresses for ds_bpermute) and the v_perm selector that assembles the quad from their dwords
    int srcA = lane * 4, srcB = lane * 4; uint32_t sel = 0x03020100u; bool fix = false;
    if (c < w + 4 && (c < 0 || c + 3 >= w)) {
        int la = -1, lb = -1; sel = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int r = reflect101_once(c + k, w), ls = (r - x0 + 4) >> 2;
            if (la < 0 || ls == la) { la = ls; sel |= (uint32_t)(r & 3) << (8 * k); }
            else { lb = ls; sel |= (uint32_t)(4 + (r & 3)) << (8 * k); }
        }
        srcA = la * 4; srcB = (lb < 0 ? la : lb) * 4; fix = true;
    }
    const bool any_fix = __ballot(fix) != 0ull;
    const int cl = min(max(c, 0), pitch - 4);                        // where the lane's raw dword comes from (virtual quads: any in-bounds address)
    const uint8_t* colp = img + cl;
    auto load_row = [&](int idx) -> uint32_t {                       // input row idx of the strip = image row y0 - 3 + idx, reflected
        const int ry = reflect101_once(y0 - 3 + idx, h);
        return *(const uint32_t*)(colp + (size_t)(uint32_t)(ry * pitch));
    };
    uint32_t lo[7], hi[7];                                           // the window, slot j = input row idx with idx % 7 == j
    uint32_t cur[7], nxt[7];
#pragma unroll
    for (int j = 0; j < 7; j++) cur[j] = load_row(j);
    const bool writes = lane >= 1 && lane <= BS_QUADS && c < w;
    uint8_t* op = out + c;
    constexpr uint32_t T0 = 18, T1 = 34, T2 = 49, T3 = 54;           // taps {18, 34, 49, 54, 49, 34, 18}
    const bs_v2u t0 = {T0, T0}, t1 = {T1, T1}, t2 = {T2, T2}, t3 = {T3, T3};
    for (int b = 0; b < nb; b++) {
        if (b + 1 < nb) {
#pragma unroll
            for (int j = 0; j < 7; j++) nxt[j] = load_row(7 * (b + 1) + j);
        }
#pragma unroll
        for (int j = 0; j < 7; j++) {
            uint32_t d = cur[j];
            if (any_fix) {
                const uint32_t da = (uint32_t)__builtin_amdgcn_ds_bpermute(srcA, (int)d), db = (uint32_t)__builtin_amdgcn_ds_bpermute(srcB, (int)d);
                d = __builtin_amdgcn_perm(db, da, sel);
            }
            lo[j] = __builtin_amdgcn_perm(0u, d, 0x0c010c00u);        // (byte 0, byte 1) as two u16
            hi[j] = __builtin_amdgcn_perm(0u, d, 0x0c030c02u);        // (byte 2, byte 3)
            const int idx = 7 * b + j;
            if (idx >= 6) {                                           // (wave-uniform) the window holds rows idx - 6 .. idx: output row y = y0 + idx - 6
                // slot of input row idx - 6 + k is (j + 1 + k) % 7; the taps are symmetric
                auto vs = [&](const uint32_t* r) -> uint32_t {
                    bs_v2u a = __builtin_bit_cast(bs_v2u, r[(j + 1) % 7]) * t0;
                    a = __builtin_bit_cast(bs_v2u, r[(j + 2) % 7]) * t1 + a;
                    a = __builtin_bit_cast(bs_v2u, r[(j + 3) % 7]) * t2 + a;
                    a = __builtin_bit_cast(bs_v2u, r[(j + 4) % 7]) * t3 + a;
                    a = __builtin_bit_cast(bs_v2u, r[(j + 5) % 7]) * t2 + a;
                    a = __builtin_bit_cast(bs_v2u, r[(j + 6) % 7]) * t1 + a;
                    a = __builtin_bit_cast(bs_v2u, r[(j + 7) % 7]) * t0 + a;
                    return __builtin_bit_cast(uint32_t, a);
                };
                const uint32_t P2 = vs(lo), P3 = vs(hi);              // (s0, s1), (s2, s3)
                const uint32_t P0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)P2, 0x138, 0xf, 0xf, true);      // wave_shr:1 — the left neighbour's (s[-4], s[-3])
                const uint32_t P1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)P3, 0x138, 0xf, 0xf, true);      //                                   (s[-2], s[-1])
                const uint32_t P4 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)P2, 0x130, 0xf, 0xf, true);      // wave_shl:1 — the right neighbour's (s[4], s[5])
                const uint32_t P5 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)P3, 0x130, 0xf, 0xf, true);      //                                    (s[6], s[7])
                auto dot = [](uint32_t p, uint32_t wlo, uint32_t whi, uint32_t acc) -> uint32_t {
                    const bs_v2u wv = {(unsigned short)wlo, (unsigned short)whi};
                    return __builtin_amdgcn_udot2(__builtin_bit_cast(bs_v2u, p), wv, acc, false);
                };
                // out k = sum_i tap[i] * s[k - 3 + i]
                uint32_t a0 = dot(P0, 0, T0, 32768u);  a0 = dot(P1, T1, T2, a0); a0 = dot(P2, T3, T2, a0); a0 = dot(P3, T1, T0, a0);
                uint32_t a1 = dot(P1, T0, T1, 32768u); a1 = dot(P2, T2, T3, a1); a1 = dot(P3, T2, T1, a1); a1 = dot(P4, T0, 0, a1);
                uint32_t a2 = dot(P1, 0, T0, 32768u);  a2 = dot(P2, T1, T2, a2); a2 = dot(P3, T3, T2, a2); a2 = dot(P4, T1, T0, a2);
                uint32_t a3 = dot(P2, T0, T1, 32768u); a3 = dot(P3, T2, T3, a3); a3 = dot(P4, T2, T1, a3); a3 = dot(P5, T0, 0, a3);
                const uint32_t o01 = __builtin_amdgcn_perm(a1, a0, 0x0c0c0602u);      // byte 2 of a0, byte 2 of a1
                const uint32_t o23 = __builtin_amdgcn_perm(a3, a2, 0x06020c0cu);      // byte 2 of a2 -> byte 2, byte 2 of a3 -> byte 3
                const int y = y0 + idx - 6;
                if (writes && y < h) *(uint32_t*)(op + (size_t)(uint32_t)(y * pitch)) = o01 | o23;
            }
        }
#pragma unroll
        for (int j = 0; j < 7; j++) cur[j] = nxt[j];
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

// (float)cos((double)x), (float)sin((double)x) for x in [0, 2 pi] (the steered-BRIEF rotation, ORBextractor.cc:103): the device library's double sin + cos are ~190 fp64
// instructions per wave (argument reduction for any magnitude, done twice) — 40 % of k_orient_brief.  Here: one Cody-Waite reduction by pi/2 (k <= 4, two fused multiply-adds,
// exact to < 1 ulp for this range) and fdlibm's __kernel_sin / __kernel_cos minimax polynomials on |r| <= pi/4 (< 1 ulp in double, as the libraries'): ~25 fp64 instructions.
// The float result differs from a correctly rounded one only when the double lands within ~1e-16 relative of a float rounding boundary (2^-29 of the arguments) — the same
// caveat the host libm / device library pair always had.
__device__ __forceinline__ void sincos_0_2pi(double x, float* sn, float* cs)
{
    const double k = __builtin_rint(x * 6.36619772367581382433e-01);          // 2 / pi
    double r = __builtin_fma(-k, 1.57079632679489655800e+00, x);               // pi / 2, high part
    r = __builtin_fma(-k, 6.12323399573676603587e-17, r);                      //         low part
    const double z = r * r;
    // sin r = r + r z (S1 + z (S2 + ... z S6));  cos r = 1 - z / 2 + z z (C1 + z (C2 + ... z C6))
    double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = __builtin_fma(z, ps, 2.75573137070700676789e-06); ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
    ps = __builtin_fma(z, ps, 8.33333333332248946124e-03); ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
    const double s = __builtin_fma(r * z, ps, r);
    double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = __builtin_fma(z, pc, -2.75573143513906633035e-07); pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
    pc = __builtin_fma(z, pc, -1.38888888888741095749e-03); pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
    const double c = __builtin_fma(z * z, pc, __builtin_fma(z, -0.5, 1.0));
    const int q = (int)k & 3;
    const double so = q == 0 ? s : (q == 1 ? c : (q == 2 ? -s : -c)), co = q == 0 ? c : (q == 1 ? -s : (q == 2 ? -c : s));
    *sn = (float)so; *cs = (float)co;
}

__global__ __launch_bounds__(256) void k_orient_brief(const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur, size_t slab, PyrDev P,
                                                      const uint2* __restrict__ kps, const int* __restrict__ frame_beg, int f0, int f1, int with_desc,
                                                      vido_keypoint* __restrict__ kpf, uint8_t* __restrict__ descf, int row_cap, unsigned long long umax_packed)
{
    // XCD-aware: each of the 8 XCDs (block b -> XCD b % 8) walks one contiguous eighth of the keypoint list, i.e.
    // whole frames, so the patch / pattern gathers of a frame stay in one private L2
    // (the grid is an upper bound; the real list length lives on the device)
    // this launch covers the keypoints of frames [f0, f1)
    const int k_lo = frame_beg[f0], n_kp = frame_beg[f1] - k_lo;
    const int chunk = (((n_kp + 3) >> 2) + 7) >> 3;
    if ((int)(blockIdx.x >> 3) >= chunk) return;
    const int blk = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    const int kk = blk * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (kk >= n_kp) return;
    const int k = k_lo + kk;
    const uint2 kp = kps[k];
    const int x = kp.x & 0xfff, y = (kp.x >> 12) & 0xfff, level = kp.x >> 24, f = kp.y;
    const int pitch = P.pitch[level];
    // IC moments over the radius-15 disc: 31 rows x 9 aligned dwords (the 31-byte row span + alignment slack) = 279 dword loads spread
    // over the lanes; the disc half-widths umax[|v|] travel as 16 nibbles in a kernel argument (no table fetch)
    int m10 = 0, m01 = 0;
    const unsigned x_al = (unsigned)(x - 15) & ~3u;                        // rows are 64-byte aligned, so aligning x aligns the address
    const uint8_t* rows = pyr + (size_t)f * slab + P.off[level] + (size_t)y * pitch + x_al;
    const int u0 = (int)x_al - x;                                         // u of byte 0 of dword 0
#pragma unroll
    for (int it = 0; it < 5; it++) {
        const int q = lane + 64 * it;
        if (q < 31 * 9) {
            const int vy = (int)(__umul24((unsigned)q, 7282u) >> 16), d = q - vy * 9, v = vy - 15;      // q / 9 for q < 320 (7282 / 65536 = 1 / 9 + 3.4e-6)
            const int um = (int)((umax_packed >> (4 * abs(v))) & 15ull);
            const uint32_t w = *(const uint32_t*)(rows + __mul24(v, pitch) + 4 * d);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int u = u0 + 4 * d + k;
                if (abs(u) <= um) { const int val = (int)((w >> (8 * k)) & 255u); m10 += u * val; m01 += v * val; }
            }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { m10 += __shfl_xor(m10, o, 64); m01 += __shfl_xor(m01, o, 64); }
    const float ang = fast_atan2_deg((float)m01, (float)m10);
    const int row = k - frame_beg[f];
    if (row >= row_cap) return;
    const size_t o = (size_t)f * row_cap + row;
    if (lane == 0) kpf[o].angle = ang;
    if (!with_desc) return;
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    const float ar = ang * factorPI;
    float a, b; sincos_0_2pi((double)ar, &b, &a);             // a = (float)cos((double)ar), b = (float)sin((double)ar)
    // The 512 rotated sample points of a keypoint lie within 18 px of it (the pattern's largest radius is 18.38): the wave stages that 37-row window of the BLURRED level
    // in its own LDS slice with aligned dword loads (370 dwords, six per lane: ~40 contiguous bytes per row) and gathers the test pairs from there.  Eight scattered
    // global byte loads per lane — 64 different cache lines per instruction — were what bounded this kernel (round 5).
    constexpr int OB_R = 18, OB_DW = 10, OB_LD = 11;                        // window radius, dwords per staged row, LDS row stride in dwords (odd)
    __shared__ uint32_t ob_lds[4][(2 * OB_R + 1) * OB_LD];
    uint32_t* win = ob_lds[threadIdx.x >> 6];
    const unsigned bx_al = (unsigned)(x - OB_R) & ~3u;                      // x >= 19 (EDGE_THRESHOLD): never negative
    const uint8_t* brow = blur + (size_t)f * slab + P.off[level] + (size_t)(y - OB_R) * pitch + bx_al;
#pragma unroll
    for (int it = 0; it < 6; it++) {
        const int q = lane + 64 * it;
        if (q < (2 * OB_R + 1) * OB_DW) { const int r = (int)(__umul24((unsigned)q, 6554u) >> 16), d = q - r * OB_DW;      // q / 10 for q < 370
                                              win[r * OB_LD + d] = *(const uint32_t*)(brow + __umul24((unsigned)r, (unsigned)pitch) + 4 * d); }
    }
    __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier();          // (a wave's slice is its own: no workgroup barrier)
    const uint8_t* cb = (const uint8_t*)win + OB_R * (OB_LD * 4) + (x - (int)bx_al);      // the keypoint inside the window
    uint32_t nib = 0;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const signed char* pt = c_pattern + (lane * 4 + t) * 4;
        const float x0 = (float)pt[0], y0 = (float)pt[1], x1 = (float)pt[2], y1 = (float)pt[3];
        const int t0 = cb[__mul24(__float2int_rn(x0 * b + y0 * a), OB_LD * 4) + __float2int_rn(x0 * a - y0 * b)];
        const int t1 = cb[__mul24(__float2int_rn(x1 * b + y1 * a), OB_LD * 4) + __float2int_rn(x1 * a - y1 * b)];
        nib |= (uint32_t)(t0 < t1) << t;
    }
    // lanes 2i / 2i+1 hold the low / high nibble of descriptor byte i; fold 8 lanes into one dword
    uint32_t w = nib | (__shfl_down(nib, 1, 64) << 4);
    w |= __shfl_down(w, 2, 64) << 8;
    w |= __shfl_down(w, 4, 64) << 16;
    if ((lane & 7) == 0) ((uint32_t*)(descf + o * 32))[lane >> 3] = w;
}

// ================================================================================================
// host side
struct OrbState {
    int L = 0, W = 0, H = 0, B = 0;
    LevelInfo lv[VIDO_MAX_LEVELS];
    PyrDev P{};
    size_t slab = 0;
    int n_cells = 0, n_blur_tiles = 0;
    int umax[HALF_PATCH + 1]; unsigned long long umax_packed = 0;
    std::vector<int> first_cell;
    // device
    uint8_t *d_pyr = nullptr, *d_blur = nullptr;
    CellDesc* d_cells = nullptr; BlurTile* d_btiles = nullptr;
    BlurTile* d_bstrips = nullptr; int n_blur_strips = 0;            // k_blur7_strips: {level, first column, first row, batches of 7 input rows} per wave; 0 = the tile kernel stays
    int2* d_xtab = nullptr; int4* d_ytab = nullptr;
    PyrBand* d_bands = nullptr; int n_bands = 0; size_t bands_lds = 0; PyrBandLv band_lv{};      // single-launch pyramid (k_pyramid_bands)
    const uint8_t* prefetched_src = nullptr; int prefetched_w = 0, prefetched_h = 0;      // vido_orb_prefetch_color: the device image whose extraction is already on the stream
    PyrTile* d_ptiles = nullptr; int n_ptiles = 0; size_t ptiles_lds = 0; PyrTileLv ptile_lv{};  // single-launch pyramid for any batch (k_pyramid_tiles); 0 tiles = not built
    int* d_frame_tot = nullptr;                                       // per-frame candidate totals (k_scan_frames -> k_scan_apply)
    uint32_t* d_slots = nullptr; int *d_counts = nullptr, *d_offsets = nullptr, *d_first_cell = nullptr, *d_lvloff = nullptr, *d_overflow = nullptr;
    uint32_t* d_cand = nullptr; size_t cand_cap = 0;
    uint2* d_kp = nullptr; size_t kp_cap = 0;
    // pinned host
    int* h_lvloff = nullptr; int* h_overflow = nullptr; uint32_t* h_cand = nullptr;
    hipEvent_t ev[8] = {}; hipEvent_t ev_done = nullptr, ev_pyr = nullptr, ev_qt = nullptr, ev_half = nullptr; int half_frames = 0;
    float timing[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::chrono::steady_clock::time_point t_start;
    int last_frames = 0; bool lean_last = false;
    int fast_rows = 0, fast_iw = FS_NARROW; size_t fast_lds = 0;
    FastStrip* d_strips = nullptr; int n_strips = 0; int* d_slot_off = nullptr; int slot_total = 0;      // FAST strips (<= 4 cells each), per-cell output slot offsets, slots per frame
    // device quadtree / keypoint assembly
    uint16_t* d_qt_slot = nullptr; int *d_sel = nullptr, *d_selcnt = nullptr, *d_kpoff = nullptr, *d_frame_beg = nullptr, *d_budget = nullptr;
    int* h_frame_beg = nullptr;
    int qcap = 0; size_t qt_lds = 0;
    // final results, [frame][row_cap] rows: device + pinned host mirror (the "result view")
    int row_cap = 0; vido_keypoint *d_kpf = nullptr, *h_kpf = nullptr; uint8_t *d_descf = nullptr, *h_descf = nullptr; int *d_nkp = nullptr;
    KpLevels kpl{};
    // cvtColor fused ingest: staging for host colour frames / the gray copy handed back
    uint8_t *d_color = nullptr, *d_gray_out = nullptr; size_t color_cap = 0; int in_channels = 1, in_rgb = 0; uint8_t* gray_out = nullptr; int gray_out_on_device = 0;
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
    for (int l = 1; l < L; l++) {                           // LDS staging extent of the resize tiles
        const LevelInfo& s = S->lv[l - 1]; LevelInfo& d = S->lv[l];
        const int2* xt = xtab.data() + d.xtab_off; const int4* yt = ytab.data() + d.ytab_off;
        int mr = 1, mw = 1;
        for (int y0 = 0; y0 < d.h; y0 += RS_TH) mr = std::max(mr, yt[std::min(y0 + RS_TH, d.h) - 1].y - yt[y0].x + 1);
        for (int x0 = 0; x0 < d.w; x0 += RS_TW) {
            const int c0 = (xt[x0].x & 0xffff) & ~3, c1 = std::min((xt[std::min(x0 + RS_TW, d.w) - 1].x & 0xffff) + 1, s.w - 1);
            mw = std::max(mw, ((c1 - c0) >> 2) + 1);
        }
        d.rs_rows = mr; d.rs_ndw = mw | 1;                  // odd dword stride: the two source rows of a pixel fall into different banks
        if ((size_t)d.rs_rows * d.rs_ndw * 4 > 64 * 1024) return vido_set_error(ctx, VIDO_E_CAPACITY, "pyramid scale factor too large for the resize tile");
    }
    // bands of the single-launch pyramid (k_pyramid_bands): per band and level the owned rows and the rows later levels need, walking back from the last level
    std::vector<PyrBand> bands; S->n_bands = 0; S->bands_lds = 0;
    if (L >= 2 && L <= PB_MAXL) {
        const int NB = std::max(1, std::min(24, S->lv[L - 1].h / 4));
        PyrBandLv& V = S->band_lv; V.L = L;
        for (int l = 0; l < L; l++) { V.w[l] = S->lv[l].w; V.h[l] = S->lv[l].h; V.pitch[l] = S->lv[l].pitch; V.off[l] = S->lv[l].off; V.xoff[l] = S->lv[l].xtab_off; V.yoff[l] = S->lv[l].ytab_off; }
        for (int b = 0; b < NB; b++) {
            PyrBand pb{}; int lo[PB_MAXL], hi[PB_MAXL];
            for (int l = 0; l < L; l++) { pb.own_lo[l] = (int)((long long)b * S->lv[l].h / NB); pb.own_hi[l] = (int)((long long)(b + 1) * S->lv[l].h / NB); }
            lo[L - 1] = pb.own_lo[L - 1]; hi[L - 1] = pb.own_hi[L - 1];                      // needed rows [lo, hi) per level
            for (int l = L - 2; l >= 0; l--) {
                lo[l] = pb.own_lo[l]; hi[l] = pb.own_hi[l];
                if (hi[l + 1] > lo[l + 1]) { const int4* yt = ytab.data() + S->lv[l + 1].ytab_off; lo[l] = std::min(lo[l], yt[lo[l + 1]].x); hi[l] = std::max(hi[l], yt[hi[l + 1] - 1].y + 1); }
                if (l >= 1 && hi[l] <= lo[l]) { lo[l] = hi[l] = pb.own_lo[l]; }
            }
            size_t off = 0;
            for (int l = 0; l < L; l++) { pb.need_lo[l] = lo[l]; pb.need_n[l] = std::max(hi[l] - lo[l], 0); pb.lds_off[l] = (int)off; off += (size_t)pb.need_n[l] * S->lv[l].pitch; off = (off + 15) & ~(size_t)15; }
            S->bands_lds = std::max(S->bands_lds, off);
            bands.push_back(pb);
        }
        if (S->bands_lds <= 150 * 1024) S->n_bands = NB;      // (else the per-level kernels stay)
    }
    // tiles of the single-launch batched pyramid (k_pyramid_tiles): owned / needed rectangles per tile and level, walking back from the last level
    std::vector<PyrTile> ptiles; S->n_ptiles = 0; S->ptiles_lds = 0;
    if (L >= 2 && L <= PT_MAXL && getenv("VIDO_PYR_LEVELS") == nullptr) {
        static const int env_nx = [] { const char* e = getenv("VIDO_PYR_NX"); return e ? atoi(e) : 0; }(), env_ny = [] { const char* e = getenv("VIDO_PYR_NY"); return e ? atoi(e) : 0; }();      // (experiments)
        // ~107 x 48 level-0 tiles (640 x 480: 6 x 10 = 60 workgroups per frame, 35 KB of LDS each; measured 5 x 8 .. 12 x 20: 106 / 95 / 100 / 129 us per 64 frames incl. the ingest)
        const int NX = std::max(1, std::min(env_nx > 0 ? env_nx : (c.width + 53) / 107, S->lv[L - 1].w / 8)), NY = std::max(1, std::min(env_ny > 0 ? env_ny : (c.height + 24) / 48, S->lv[L - 1].h / 4));
        PyrTileLv& V = S->ptile_lv; V.L = L;
        for (int l = 0; l < L; l++) { V.w[l] = S->lv[l].w; V.pitch[l] = S->lv[l].pitch; V.off[l] = S->lv[l].off; V.xoff[l] = S->lv[l].xtab_off; V.yoff[l] = S->lv[l].ytab_off; }
        bool ok = true;
        for (int ty = 0; ty < NY && ok; ty++)
            for (int tx = 0; tx < NX && ok; tx++) {
                PyrTile t{}; int x0[PT_MAXL], x1[PT_MAXL], y0[PT_MAXL], y1[PT_MAXL];            // computed rectangle [x0, x1) x [y0, y1): the needed columns widened to whole quads
                int n0[PT_MAXL], n1[PT_MAXL];                                                   // the columns really needed [n0, n1)
                for (int l = 0; l < L; l++) {                       // (level 0's rectangle is only written when the ingest is fused into the launch; it is staged either way)
                    const int w = S->lv[l].w, h = S->lv[l].h, w4 = l == 0 ? S->lv[0].pitch : (w + 3) & ~3;
                    t.ox0[l] = tx == 0 ? 0 : (int)((long long)tx * w / NX) & ~3; t.ox1[l] = tx + 1 == NX ? w4 : (int)((long long)(tx + 1) * w / NX) & ~3;
                    t.oy0[l] = (int)((long long)ty * h / NY); t.oy1[l] = (int)((long long)(ty + 1) * h / NY);
                }
                x0[L - 1] = t.ox0[L - 1]; x1[L - 1] = t.ox1[L - 1]; y0[L - 1] = t.oy0[L - 1]; y1[L - 1] = t.oy1[L - 1];
                n0[L - 1] = t.ox0[L - 1]; n1[L - 1] = std::min(t.ox1[L - 1], S->lv[L - 1].w);
                for (int l = L - 2; l >= 0; l--) {
                    const int w = S->lv[l].w, w4 = (w + 3) & ~3;
                    const int2* xt = xtab.data() + S->lv[l + 1].xtab_off; const int4* yt = ytab.data() + S->lv[l + 1].ytab_off;
                    // what the next level's NEEDED pixels sample: columns sx .. sx + 1 (clamped), rows ytab.x .. ytab.y; + this level's own rectangle (level 0 owns nothing).
                    // The quad padding of the next level's rectangle is computed too, from whatever lies beside the staged columns (a few bytes of the neighbouring LDS
                    // row or buffer): nobody needs those pixels, so their sources are not staged — needed rectangles grow by ~2 px per level instead of ~2 + up to 6.
                    int lo = xt[n0[l + 1]].x & 0xffff, hi = std::min((xt[n1[l + 1] - 1].x & 0xffff) + 1, w - 1) + 1;
                    int ylo = yt[y0[l + 1]].x, yhi = yt[y1[l + 1] - 1].y + 1;
                    lo = std::min(lo, t.ox0[l]); hi = std::max(hi, t.ox1[l]); ylo = std::min(ylo, t.oy0[l]); yhi = std::max(yhi, t.oy1[l]);
                    n0[l] = lo; n1[l] = std::min(hi, w);
                    x0[l] = lo & ~3; x1[l] = std::min((hi + 3) & ~3, l == 0 ? S->lv[0].pitch : w4); y0[l] = ylo; y1[l] = std::min(yhi, S->lv[l].h);
                    if (x1[l] <= x0[l] || y1[l] <= y0[l] || n1[l] <= n0[l]) ok = false;
                }
                if (x1[L - 1] <= x0[L - 1] || y1[L - 1] <= y0[L - 1]) ok = false;
                size_t off = 16;                                      // (guard: a padding pixel may sample a few bytes in front of its source rectangle)
                for (int l = 0; l < L && ok; l++) {
                    t.nx0[l] = x0[l]; t.nw[l] = x1[l] - x0[l]; t.ny0[l] = y0[l]; t.nh[l] = y1[l] - y0[l];
                    const unsigned nq = (unsigned)t.nw[l] / 4u; t.magic[l] = (unsigned)((0x100000000ull + nq - 1) / nq);
                    for (unsigned i = 0; i < nq * (unsigned)t.nh[l]; i++) if ((unsigned)(((unsigned long long)i * t.magic[l]) >> 32) != i / nq) ok = false;      // the reciprocal is exact on the item range
                    if (l + 1 < L) { t.lds[l] = (int)off; off += (size_t)t.nh[l] * t.nw[l]; off = (off + 15) & ~(size_t)15; }
                }
                off += 16;                                            // (the unclamped sx + 1 read of a last row)
                for (int l = 1; l < L && ok; l++) { t.xt_lds[l] = (int)off; off += (size_t)t.nw[l] * sizeof(int2); t.yt_lds[l] = (int)off; off += (size_t)t.nh[l] * sizeof(int4); }
                S->ptiles_lds = std::max(S->ptiles_lds, off);
                ptiles.push_back(t);
            }
        if (ok && S->ptiles_lds <= 150 * 1024) S->n_ptiles = (int)ptiles.size();
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
                if (cd.sw < 7 || cd.sh < 7) continue;     // cv::FAST finds nothing in a sub-image thinner than its ring
                cells.push_back(cd);
            }
        }
        v.n_cells = (int)cells.size() - v.first_cell;
        for (int ty = 0; ty < (v.h + BL_TH - 1) / BL_TH; ty++)
            for (int tx = 0; tx < (v.w + BL_TW - 1) / BL_TW; tx++) btiles.push_back(BlurTile{l, tx, ty, 0});
    }
    S->n_cells = (int)cells.size(); S->n_blur_tiles = (int)btiles.size();
    std::vector<BlurTile> bstrips;
    {   // strips of the LDS-free blur: a level's quads split evenly over ceil(quads / BS_QUADS) waves per row band, bands of 7 nb - 6 output rows
        static const int nb_env = [] { const char* e = getenv("VIDO_BLUR_NB"); return e ? atoi(e) : 0; }();
        const int nb = nb_env >= 1 && nb_env <= 16 ? nb_env : 3, rows_out = 7 * nb - 6;
        bool ok = getenv("VIDO_BLUR_TILES") == nullptr;                 // (A/B switch: the round-1 tile kernel)
        for (int l = 0; l < L && ok; l++) {
            const LevelInfo& v = S->lv[l];
            const int nq = (v.w + 3) / 4, nt = (nq + BS_QUADS - 1) / BS_QUADS, qpt = (nq + nt - 1) / nt, x_last = 4 * qpt * (nt - 1);
            if (v.w < 16 || v.h < 7 * nb + 8 || v.w - 8 - x_last + 4 < 0) { ok = false; break; }      // the reflected columns of the right edge must live in lanes of the last strip
            for (int y0 = 0; y0 < v.h; y0 += rows_out)
                for (int tx = 0; tx < nt; tx++) bstrips.push_back(BlurTile{l, 4 * qpt * tx, y0, nb});
        }
        if (!ok) bstrips.clear();
    }
    S->n_blur_strips = (int)bstrips.size();
    // FAST strips: runs of up to FS_MAXC horizontally adjacent cells of one cell row (consecutive in the reference order), split evenly; per-cell output slots
    std::vector<FastStrip> strips; std::vector<int> slot_off(cells.size() + 1, 0);
    {
        int max_sh = 8, max_iw = 0;
        for (size_t c = 0; c < cells.size(); c++) {
            const CellDesc& cd = cells[c];
            slot_off[c + 1] = slot_off[c] + ((cd.sw - 6 + 1) / 2) * ((cd.sh - 6 + 1) / 2);      // most 3x3-NMS survivors of a (sw-6) x (sh-6) interior
            max_sh = std::max(max_sh, cd.sh); max_iw = std::max(max_iw, cd.sw - 6);
            if (cd.sw - 6 > FS_WIDE || cd.sh - 6 > 63)
                return vido_set_error(ctx, VIDO_E_INVALID, "FAST cell %dx%d at level %d exceeds the strip tile (%dx63 interior)", cd.sw, cd.sh, cd.level, FS_WIDE);
        }
        const int FS_MAXIW = max_iw <= FS_NARROW ? FS_NARROW : FS_WIDE;      // which instantiation of k_fast_strips this configuration runs
        S->fast_iw = FS_MAXIW;
        size_t c0 = 0;
        while (c0 < cells.size()) {
            size_t c1 = c0 + 1;                                       // [c0, c1): the cells of one cell row (same level and y0, x adjacent)
            while (c1 < cells.size() && cells[c1].level == cells[c0].level && cells[c1].y0 == cells[c0].y0 && cells[c1].sh == cells[c0].sh &&
                   cells[c1].x0 + 3 == cells[c1 - 1].x0 + cells[c1 - 1].sw - 3) c1++;
            const int ncols = (int)(c1 - c0);
            int per = FS_MAXC;
            for (;;) {                                                // widest allowed run that keeps every strip's interior <= FS_MAXIW
                bool ok = true;
                const int ns = (ncols + per - 1) / per;
                for (int k = 0, b = 0; k < ns && ok; k++) { const int len = ncols / ns + (k < ncols % ns ? 1 : 0); int w = 0; for (int j = 0; j < len; j++) w += cells[c0 + b + j].sw - 6; ok = w <= FS_MAXIW; b += len; }
                if (ok || per == 1) break;
                per--;
            }
            const int ns = (ncols + per - 1) / per;
            for (int k = 0, b = 0; k < ns; k++) {
                const int len = ncols / ns + (k < ncols % ns ? 1 : 0);
                FastStrip st{}; const CellDesc& first = cells[c0 + b]; const CellDesc& last = cells[c0 + b + len - 1];
                st.level = first.level; st.x0 = first.x0; st.y0 = first.y0; st.sw = last.x0 + last.sw - first.x0; st.sh = first.sh; st.ncell = len; st.cell0 = (int)(c0 + b);
                for (int j = 0; j <= FS_MAXC; j++) st.bx[j] = st.sw - 6;
                for (int j = 0; j < len; j++) st.bx[j] = cells[c0 + b + j].x0 - first.x0;
                st.bx[len] = st.sw - 6;
                strips.push_back(st); b += len;
            }
            c0 = c1;
        }
        S->n_strips = (int)strips.size(); S->slot_total = slot_off[cells.size()];
        S->fast_rows = max_sh;
        S->fast_lds = FS_MAXIW == FS_NARROW ? FsGeom<FS_NARROW>::lds_bytes(max_sh) : FsGeom<FS_WIDE>::lds_bytes(max_sh);
    }
    HIP_TRY(ctx, hipMalloc(&S->d_strips, strips.size() * sizeof(FastStrip)));
    HIP_TRY(ctx, hipMemcpy(S->d_strips, strips.data(), strips.size() * sizeof(FastStrip), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMalloc(&S->d_slot_off, slot_off.size() * sizeof(int)));
    HIP_TRY(ctx, hipMemcpy(S->d_slot_off, slot_off.data(), slot_off.size() * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMalloc(&S->d_cells, cells.size() * sizeof(CellDesc)));
    HIP_TRY(ctx, hipMemcpy(S->d_cells, cells.data(), cells.size() * sizeof(CellDesc), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMalloc(&S->d_btiles, btiles.size() * sizeof(BlurTile)));
    HIP_TRY(ctx, hipMemcpy(S->d_btiles, btiles.data(), btiles.size() * sizeof(BlurTile), hipMemcpyHostToDevice));
    if (!bstrips.empty()) {
        HIP_TRY(ctx, hipMalloc(&S->d_bstrips, bstrips.size() * sizeof(BlurTile)));
        HIP_TRY(ctx, hipMemcpy(S->d_bstrips, bstrips.data(), bstrips.size() * sizeof(BlurTile), hipMemcpyHostToDevice));
    }
    HIP_TRY(ctx, hipMalloc(&S->d_xtab, std::max<size_t>(xtab.size(), 1) * sizeof(int2)));
    HIP_TRY(ctx, hipMalloc(&S->d_ytab, std::max<size_t>(ytab.size(), 1) * sizeof(int4)));
    if (!xtab.empty()) {
        HIP_TRY(ctx, hipMemcpy(S->d_xtab, xtab.data(), xtab.size() * sizeof(int2), hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(S->d_ytab, ytab.data(), ytab.size() * sizeof(int4), hipMemcpyHostToDevice));
    }
    if (S->n_bands) {
        HIP_TRY(ctx, hipMalloc(&S->d_bands, bands.size() * sizeof(PyrBand)));
        HIP_TRY(ctx, hipMemcpy(S->d_bands, bands.data(), bands.size() * sizeof(PyrBand), hipMemcpyHostToDevice));
        static bool battr[64] = {};                                   // per function and device, not per ctx: a second, smaller ctx must not lower the first one's limit
        if (!battr[ctx->device & 63]) { HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_pyramid_bands, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); battr[ctx->device & 63] = true; }
    }
    if (S->n_ptiles) {
        HIP_TRY(ctx, hipMalloc(&S->d_ptiles, ptiles.size() * sizeof(PyrTile)));
        HIP_TRY(ctx, hipMemcpy(S->d_ptiles, ptiles.data(), ptiles.size() * sizeof(PyrTile), hipMemcpyHostToDevice));
        static bool attr[64] = {};                                    // (the attribute belongs to the function and the device, not to this ctx: set once, to the ceiling)
        if (!attr[ctx->device & 63]) { HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_pyramid_tiles, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); attr[ctx->device & 63] = true; }
    }
    HIP_TRY(ctx, hipMalloc(&S->d_first_cell, L * sizeof(int)));
    HIP_TRY(ctx, hipMemcpy(S->d_first_cell, S->first_cell.data(), L * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c_umax), S->umax, sizeof S->umax));
    S->umax_packed = 0; for (int v = 0; v <= HALF_PATCH; v++) S->umax_packed |= (unsigned long long)(S->umax[v] & 15) << (4 * v);
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
    S->cand_cap = (size_t)S->slot_total * B;          // every output slot of every cell filled: nothing an image contains can overflow the candidate buffers
    S->kp_cap = (size_t)(ctx->cfg.n_features * 2 + 256) * B;
    HIP_TRY(ctx, hipMalloc(&S->d_pyr, S->slab * B));
    HIP_TRY(ctx, hipMalloc(&S->d_blur, S->slab * B));
    HIP_TRY(ctx, hipMemset(S->d_pyr, 0, S->slab * B));
    HIP_TRY(ctx, hipMemset(S->d_blur, 0, S->slab * B));
    HIP_TRY(ctx, hipMalloc(&S->d_slots, (size_t)S->slot_total * B * sizeof(uint32_t)));
    {   // the attribute is per (function, device), not per ctx: set ONCE to the ceiling, so that a second, smaller ctx (the local-BA helper: 64 features, one level) can never
        // lower the limit the tracker's launches need (the same rule as k_pyramid_bands / k_pyramid_tiles above)
        static bool fattr[64] = {};
        if (S->fast_lds > 64 * 1024) return vido_set_error(ctx, VIDO_E_CAPACITY, "orb: a FAST strip of %d rows needs %zu bytes of LDS (limit 65536)", S->fast_rows, S->fast_lds);
        if (!fattr[ctx->device & 63]) {
            HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_fast_strips<FS_NARROW>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
            HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_fast_strips<FS_WIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
            fattr[ctx->device & 63] = true;
        }
    }
    HIP_TRY(ctx, hipMalloc(&S->d_counts, ncell * sizeof(int)));
    HIP_TRY(ctx, hipMalloc(&S->d_offsets, (ncell + 1) * sizeof(int)));
    HIP_TRY(ctx, hipMalloc(&S->d_frame_tot, (B + 1) * sizeof(int)));
    HIP_TRY(ctx, hipMalloc(&S->d_lvloff, (B * S->L + 1) * sizeof(int)));
    HIP_TRY(ctx, hipMalloc(&S->d_overflow, sizeof(int)));
    HIP_TRY(ctx, hipMemset(S->d_overflow, 0, sizeof(int)));
    HIP_TRY(ctx, hipMalloc(&S->d_cand, S->cand_cap * sizeof(uint32_t)));
    HIP_TRY(ctx, hipMalloc(&S->d_kp, S->kp_cap * sizeof(uint2)));
    HIP_TRY(ctx, hipHostMalloc(&S->h_lvloff, (B * S->L + 1) * sizeof(int)));
    HIP_TRY(ctx, hipHostMalloc(&S->h_overflow, sizeof(int)));
    HIP_TRY(ctx, hipHostMalloc(&S->h_cand, (size_t)S->slot_total * sizeof(uint32_t)));          // debug read-back buffer: one (frame, level) list at most
    for (auto& e : S->ev) HIP_TRY(ctx, hipEventCreate(&e));
    HIP_TRY(ctx, hipEventCreateWithFlags(&S->ev_done, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventCreateWithFlags(&S->ev_pyr, hipEventDisableTiming)); HIP_TRY(ctx, hipEventCreate(&S->ev_qt));
    HIP_TRY(ctx, hipEventCreateWithFlags(&S->ev_half, hipEventDisableTiming));
    {   // device quadtree: the node list never exceeds budget + 3 entries (a pass stops at >= budget nodes)
        int maxN = 0; std::vector<int> bud(S->L);
        for (int l = 0; l < S->L; l++) { bud[l] = S->lv[l].n_budget; maxN = std::max(maxN, bud[l]); }
        S->qcap = (maxN + 8 + 63) & ~63;
        if (S->qcap > 1024) return vido_set_error(ctx, VIDO_E_CAPACITY, "orb: per-level feature budget %d exceeds the quadtree kernel's 1016", maxN);
        { int m2 = 1; while (m2 < S->qcap) m2 <<= 1; S->qt_lds = (size_t)S->qcap * 84 + (size_t)m2 * 8; }      // the bitonic sort pads its keys to a power of two
        {   // once per device, to the ceiling (qcap = 1024: 1024 * 84 + 1024 * 8 bytes): never lowered by a later, smaller ctx
            static bool qattr[64] = {};
            if (!qattr[ctx->device & 63]) { HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_quadtree, hipFuncAttributeMaxDynamicSharedMemorySize, 1024 * 84 + 1024 * 8)); qattr[ctx->device & 63] = true; }
        }
        HIP_TRY(ctx, hipMalloc(&S->d_qt_slot, S->cand_cap * sizeof(uint16_t)));
        HIP_TRY(ctx, hipMalloc(&S->d_sel, B * S->L * (size_t)S->qcap * sizeof(int)));
        HIP_TRY(ctx, hipMalloc(&S->d_selcnt, B * S->L * sizeof(int)));
        HIP_TRY(ctx, hipMalloc(&S->d_kpoff, (B * S->L + 1) * sizeof(int)));
        HIP_TRY(ctx, hipMalloc(&S->d_frame_beg, (B + 1) * sizeof(int)));
        HIP_TRY(ctx, hipMalloc(&S->d_budget, S->L * sizeof(int)));
        HIP_TRY(ctx, hipMemcpy(S->d_budget, bud.data(), S->L * sizeof(int), hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipHostMalloc(&S->h_frame_beg, (B + 1) * sizeof(int)));
        int rows = 0; for (int l = 0; l < S->L; l++) { rows += bud[l] + 3; S->kpl.scale[l] = S->lv[l].scale; S->kpl.size[l] = (float)(int)(31 * S->lv[l].scale); }
        S->row_cap = (rows + 63) & ~63;                    // a level's list never exceeds budget + 3 nodes
        HIP_TRY(ctx, hipMalloc(&S->d_kpf, B * S->row_cap * sizeof(vido_keypoint)));
        HIP_TRY(ctx, hipMalloc(&S->d_descf, B * (size_t)S->row_cap * 32));
        HIP_TRY(ctx, hipMalloc(&S->d_nkp, B * sizeof(int)));
        HIP_TRY(ctx, hipHostMalloc(&S->h_kpf, B * S->row_cap * sizeof(vido_keypoint)));
        HIP_TRY(ctx, hipHostMalloc(&S->h_descf, B * (size_t)S->row_cap * 32));
    }
    return VIDO_OK;
}

void orb_state_destroy(vido_ctx* ctx)
{
    OrbState* S = ctx->orb;
    if (!S) return;
    hipFree(S->d_pyr); hipFree(S->d_blur); hipFree(S->d_cells); hipFree(S->d_btiles); hipFree(S->d_bstrips); hipFree(S->d_xtab); hipFree(S->d_ytab); hipFree(S->d_bands); hipFree(S->d_ptiles);
    hipFree(S->d_slots); hipFree(S->d_counts); hipFree(S->d_offsets); hipFree(S->d_frame_tot); hipFree(S->d_first_cell); hipFree(S->d_lvloff); hipFree(S->d_overflow);
    hipFree(S->d_cand); hipFree(S->d_kp); hipFree(S->d_strips); hipFree(S->d_slot_off);
    hipHostFree(S->h_lvloff); hipHostFree(S->h_overflow); hipHostFree(S->h_cand);
    for (auto& e : S->ev) if (e) hipEventDestroy(e);
    if (S->ev_done) hipEventDestroy(S->ev_done);
    if (S->ev_pyr) hipEventDestroy(S->ev_pyr);
    if (S->ev_qt) hipEventDestroy(S->ev_qt);
    if (S->ev_half) hipEventDestroy(S->ev_half);
    hipFree(S->d_qt_slot); hipFree(S->d_sel); hipFree(S->d_selcnt); hipFree(S->d_kpoff); hipFree(S->d_frame_beg); hipFree(S->d_budget); hipFree(S->d_kpf); hipFree(S->d_descf); hipFree(S->d_nkp);
    hipHostFree(S->h_frame_beg); hipHostFree(S->h_kpf); hipHostFree(S->h_descf);
    hipFree(S->d_color); hipFree(S->d_gray_out);
    delete S; ctx->orb = nullptr;
}

static inline void orb_launch_blur(OrbState* S, int nf, hipStream_t st)
{
    if (S->n_blur_strips) hipLaunchKernelGGL(k_blur7_strips, dim3(S->n_blur_strips, nf), dim3(64), 0, st, S->d_pyr, S->d_blur, S->slab, S->P, S->d_bstrips);
    else hipLaunchKernelGGL(k_blur7, dim3(S->n_blur_tiles, nf), dim3(256), 0, st, S->d_pyr, S->d_blur, S->slab, S->P, S->d_btiles);
}

// Enqueues the whole extractor for nf frames on the ctx stream; nothing is synchronised.  Results land in the device
// rows d_kpf / d_descf / d_nkp ([frame][row_cap]).
int orb_enqueue(vido_ctx* ctx, const uint8_t* imgs, int on_device, int nf, size_t frame_stride, int stride, int width, int height)
{
    OrbState* S = ctx->orb;
    if (!S) return vido_set_error(ctx, VIDO_E_INVALID, "orb: context has no ORB state");
    if (!imgs) return vido_set_error(ctx, VIDO_E_INVALID, "orb: null input");
    if (width != S->W || height != S->H) return vido_set_error(ctx, VIDO_E_INVALID, "orb: frame %dx%d but ctx was created for %dx%d", width, height, S->W, S->H);
    if (nf < 1 || nf > S->B) return vido_set_error(ctx, VIDO_E_INVALID, "orb: n_frames=%d outside [1,%d]", nf, S->B);
    VidoProfScope ps_e("orb_enqueue: host time of the launches", nullptr, false);
    if (stride < width) return vido_set_error(ctx, VIDO_E_INVALID, "orb: stride < width");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int L = S->L;
    S->t_start = std::chrono::steady_clock::now();
    // The tracker's single-frame path (nf <= 2) runs LEAN: no stage-timing events (vido_orb_last_timing then reports zeros: the stage times are only meaningful batched),
    // the pyramid as one launch, the blur on the same stream — beside the networks every stream operation of this dependent chain queues for the GPU again (3.9 ms for
    // the ~30 operations of an extraction that takes 0.35 ms alone).  VIDO_ORB_FULL_TIMING=1 keeps the batched form.
    static const bool full_timing = getenv("VIDO_ORB_FULL_TIMING") != nullptr;
    const bool lean = nf <= 2 && !full_timing;
#define ORB_EV(e, s_) do { if (!lean) HIP_TRY(ctx, hipEventRecord((e), (s_))); } while (0)
    ORB_EV(S->ev[0], st);
    static const int tiles_mode = [] { const char* e = getenv("VIDO_PYR_TILES"); return e ? atoi(e) : 1; }();        // A/B switch: 0 = the round-4 kernels (bands for one frame, per-level launches for a batch); 2 = tiles, ingest not fused
    // gray device frames that allow dword loads ride on the tile kernel's staging instead of a copy of their own
    const bool fuse_ingest = S->n_ptiles && tiles_mode == 1 && S->in_channels == 1 && on_device && L >= 2 && (((uintptr_t)imgs | (uintptr_t)frame_stride | (uintptr_t)stride | (uintptr_t)width) & 3) == 0;
    // level 0 <- input
    if (S->in_channels != 1) {                          // colour frames: cvtColor fused into the ingest (set by vido_orb_extract_color for this one call)
        const int cn = S->in_channels; const uint8_t* src = imgs; size_t fs = frame_stride; int sst = stride;
        if (stride < width * cn) return vido_set_error(ctx, VIDO_E_INVALID, "orb: stride < width * channels");
        if (!on_device) {
            const size_t need = (size_t)S->B * S->W * S->H * 4;
            if (!S->d_color) { HIP_TRY(ctx, hipMalloc(&S->d_color, need)); S->color_cap = need; }
            sst = width * cn; fs = (size_t)sst * height;
            for (int f = 0; f < nf; f++)
                HIP_TRY(ctx, hipMemcpy2DAsync(S->d_color + (size_t)f * fs, sst, imgs + (size_t)f * frame_stride, stride, (size_t)width * cn, height, hipMemcpyHostToDevice, st));
            src = S->d_color;
        }
        uint8_t* gdev = nullptr;
        if (S->gray_out) {
            if (S->gray_out_on_device) gdev = S->gray_out;
            else { if (!S->d_gray_out) HIP_TRY(ctx, hipMalloc(&S->d_gray_out, (size_t)S->B * S->W * S->H)); gdev = S->d_gray_out; }
        }
        const dim3 grid((S->lv[0].pitch / 4 + 63) / 64, (height + 3) / 4, nf);
        if (cn == 3) hipLaunchKernelGGL(k_ingest_color<3>, grid, dim3(256), 0, st, src, fs, sst, width, height, S->in_rgb, S->d_pyr, S->slab, S->lv[0].off, S->lv[0].pitch, gdev);
        else         hipLaunchKernelGGL(k_ingest_color<4>, grid, dim3(256), 0, st, src, fs, sst, width, height, S->in_rgb, S->d_pyr, S->slab, S->lv[0].off, S->lv[0].pitch, gdev);
        if (S->gray_out && !S->gray_out_on_device)
            HIP_TRY(ctx, hipMemcpyAsync(S->gray_out, gdev, (size_t)nf * width * height, hipMemcpyDeviceToHost, st));
    }
    else if (on_device && fuse_ingest) { /* level 0 is written by k_pyramid_tiles below */ }
    else if (on_device)
        hipLaunchKernelGGL(k_ingest, dim3((S->lv[0].pitch / 4 + 63) / 64, (height + 3) / 4, nf), dim3(256), 0, st, imgs, frame_stride, stride, width, height,
                           S->d_pyr, S->slab, S->lv[0].off, S->lv[0].pitch);
    else for (int f = 0; f < nf; f++)
        HIP_TRY(ctx, hipMemcpy2DAsync(S->d_pyr + (size_t)f * S->slab + S->lv[0].off, S->lv[0].pitch, imgs + (size_t)f * frame_stride, stride,
                                      width, height, hipMemcpyHostToDevice, st));
    static const int bands_mode = [] { const char* e = getenv("VIDO_ORB_BANDS"); return e ? atoi(e) : -1; }();      // experiment switch: 1 = the band pyramid for every batch size, 0 = never
    if (S->n_ptiles && tiles_mode) {
        static const int xcd_deal = [] { const char* e = getenv("VIDO_PYR_XCD"); return e ? atoi(e) : 1; }();         // (0: item = workgroup index, the round-5 order)
        const int total = S->n_ptiles * nf;
        hipLaunchKernelGGL(k_pyramid_tiles, dim3(8 * ((total + 7) / 8)), dim3(256), S->ptiles_lds, st, S->d_pyr, S->slab, S->ptile_lv, (const PyrTile*)S->d_ptiles, (const int2*)S->d_xtab, (const int4*)S->d_ytab,
                           fuse_ingest ? imgs : (const uint8_t*)nullptr, frame_stride, stride, S->n_ptiles, total, xcd_deal);
    }
    else if (S->n_bands && (bands_mode == 1 || (bands_mode != 0 && lean)))
        hipLaunchKernelGGL(k_pyramid_bands, dim3(S->n_bands, nf), dim3(256), S->bands_lds, st, S->d_pyr, S->slab, S->band_lv, (const PyrBand*)S->d_bands, (const int2*)S->d_xtab, (const int4*)S->d_ytab);
    else for (int l = 1; l < L; l++) {
        const LevelInfo &s = S->lv[l - 1], &d = S->lv[l];
        dim3 grid((d.pitch + RS_TW - 1) / RS_TW, (d.h + RS_TH - 1) / RS_TH, nf), block(256);
        hipLaunchKernelGGL(k_resize, grid, block, (size_t)d.rs_rows * d.rs_ndw * 4, st, S->d_pyr, S->slab, s.w, s.h, s.pitch, s.off, d.w, d.h, d.pitch, d.off,
                           S->d_xtab + d.xtab_off, S->d_ytab + d.ytab_off, d.rs_ndw);
    }
    ORB_EV(S->ev[1], st);
    const int with_desc = ctx->cfg.compute_descriptors ? 1 : 0;
    if (S->fast_iw == FS_NARROW)
        hipLaunchKernelGGL(k_fast_strips<FS_NARROW>, dim3(S->n_strips, nf), dim3(64), S->fast_lds, st, S->d_pyr, S->slab, S->P, S->d_strips, S->n_cells, S->d_slot_off, S->slot_total,
                           ctx->cfg.ini_th_fast, ctx->cfg.min_th_fast, S->fast_rows, S->d_slots, S->d_counts);
    else
        hipLaunchKernelGGL(k_fast_strips<FS_WIDE>, dim3(S->n_strips, nf), dim3(64), S->fast_lds, st, S->d_pyr, S->slab, S->P, S->d_strips, S->n_cells, S->d_slot_off, S->slot_total,
                           ctx->cfg.ini_th_fast, ctx->cfg.min_th_fast, S->fast_rows, S->d_slots, S->d_counts);
    ORB_EV(S->ev[7], st);
    if (getenv("VIDO_DEBUG_SYNC")) {       // debugging aid: the FAST stage alone, then its per-cell counts against the slot capacities
        fprintf(stderr, "[vido] k_fast_strips (%d strips x %d frames, lds %zu)...\n", S->n_strips, nf, S->fast_lds);
        hipError_t e = hipStreamSynchronize(st); fprintf(stderr, "[vido] k_fast_strips: %s\n", hipGetErrorString(e));
        std::vector<int> hc((size_t)S->n_cells * nf), so(S->n_cells + 1);
        hipMemcpy(hc.data(), S->d_counts, hc.size() * sizeof(int), hipMemcpyDeviceToHost); hipMemcpy(so.data(), S->d_slot_off, so.size() * sizeof(int), hipMemcpyDeviceToHost);
        long long tot = 0; int bad = 0;
        for (size_t i = 0; i < hc.size(); i++) { const int c = (int)(i % S->n_cells), capc = so[c + 1] - so[c]; tot += hc[i];
            if (hc[i] < 0 || hc[i] > capc) { if (bad++ < 8) fprintf(stderr, "[vido]   cell %d (frame %zu): count %d, capacity %d\n", c, i / S->n_cells, hc[i], capc); } }
        fprintf(stderr, "[vido] counts: total %lld, %d out of range\n", tot, bad);
    }
    static const bool scan_one = getenv("VIDO_ORB_SCAN1") != nullptr;       // A/B switch: the single-workgroup scan
    if (scan_one || nf == 1 || L > 64)
        hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, st, S->d_counts, S->n_cells * nf, S->d_offsets, S->n_cells, nf, L, S->d_first_cell, S->d_lvloff);
    else {
        hipLaunchKernelGGL(k_scan_frames, dim3(nf), dim3(256), 0, st, S->d_counts, S->n_cells, L, S->d_first_cell, S->d_offsets, S->d_lvloff, S->d_frame_tot);
        hipLaunchKernelGGL(k_scan_apply, dim3((S->n_cells + 255) / 256, nf), dim3(256), 0, st, S->d_offsets, S->d_lvloff, S->d_frame_tot, S->n_cells, L, nf);
    }
    hipLaunchKernelGGL(k_gather_cands, dim3(S->n_cells, nf), dim3(64), 0, st, S->d_slots, S->d_counts, S->d_offsets, S->d_slot_off, S->slot_total, S->n_cells, S->d_cand, (int)S->cand_cap);
    ORB_EV(S->ev[2], st);
    // the blur only needs the pyramid: it runs on the second stream, concurrently with the quadtree and the keypoint list kernels (512 latency-bound
    // workgroups that leave most CUs idle; forking before FAST just makes the two full-GPU kernels contend), and joins before orientation + rBRIEF
    if (with_desc && lean)
        orb_launch_blur(S, nf, st);      // (11 us on the way; a fork / join costs four more stream operations)
    else if (with_desc) {
        HIP_TRY(ctx, hipEventRecord(S->ev_pyr, st));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream2, S->ev_pyr, 0));
        HIP_TRY(ctx, hipEventRecord(S->ev[3], ctx->stream2));
        orb_launch_blur(S, nf, ctx->stream2);
        HIP_TRY(ctx, hipEventRecord(S->ev[4], ctx->stream2));
    }
    // ---- DistributeOctTree per (frame, level) + keypoint list, all on the device
    const int n_tasks = nf * L;
    static const bool dbg = getenv("VIDO_DEBUG_SYNC") != nullptr;
#define DBG_SYNC(name) do { if (dbg) { fprintf(stderr, "[vido] %s...\n", name); hipStreamSynchronize(st); fprintf(stderr, "[vido] %s ok\n", name); } } while (0)
    DBG_SYNC("fast+gather");
    hipLaunchKernelGGL(k_quadtree, dim3(n_tasks), dim3(QT_NT), S->qt_lds, st, S->d_cand, S->d_lvloff, S->P, L, S->d_budget, S->qcap, S->d_qt_slot, S->d_sel, S->d_selcnt, S->d_overflow, (int)S->cand_cap);
    DBG_SYNC("k_quadtree");
    hipLaunchKernelGGL(k_kp_offsets, dim3(1), dim3(1024), 0, st, S->d_selcnt, n_tasks, L, S->d_kpoff, S->d_frame_beg, nf);
    DBG_SYNC("k_kp_offsets");
    hipLaunchKernelGGL(k_kp_write, dim3(n_tasks), dim3(256), 0, st, S->d_cand, S->d_lvloff, S->d_sel, S->d_selcnt, S->d_kpoff, S->qcap, L, S->d_kp, (int)S->kp_cap,
                       S->d_kpf, S->row_cap, S->d_nkp, S->kpl);
    DBG_SYNC("k_kp_write");
    ORB_EV(S->ev_qt, st);
    if (with_desc && !lean) HIP_TRY(ctx, hipStreamWaitEvent(st, S->ev[4], 0));
    ORB_EV(S->ev[5], st);
    {   // launch bound: every (frame, level) list holds at most budget + 3 nodes; the kernel reads the real counts from d_frame_beg.
        // Two launches (first / second half of the frames): the rows of the first half are already on their way to the host (second stream)
        // while the second half is being computed.
        const int fh = nf >= 8 ? nf / 2 : nf;
        for (int part = 0; part < (fh < nf ? 2 : 1); part++) {
            const int f0 = part ? fh : 0, f1 = part ? nf : fh;
            const size_t bound = std::min((size_t)S->row_cap * (f1 - f0), S->kp_cap);
            hipLaunchKernelGGL(k_orient_brief, dim3((unsigned)((((bound + 3) / 4) + 7) & ~(size_t)7)), dim3(256), 0, st, S->d_pyr, S->d_blur, S->slab, S->P, S->d_kp,
                               (const int*)S->d_frame_beg, f0, f1, with_desc, S->d_kpf, S->d_descf, S->row_cap, S->umax_packed);
            if (part == 0 && fh < nf) HIP_TRY(ctx, hipEventRecord(S->ev_half, st));
        }
        S->half_frames = fh < nf ? fh : 0;
    }
    DBG_SYNC("k_orient_brief");
    ORB_EV(S->ev[6], st);
    S->lean_last = lean;
    S->last_frames = nf;
    return VIDO_OK;
}

// Waits for the extractor, checks the capacity flags and (copy != 0) brings the result rows into the pinned mirror.
int orb_collect(vido_ctx* ctx, int nf, int copy)
{
    OrbState* S = ctx->orb; hipStream_t st = ctx->stream; const int L = S->L;
    VidoProfScope ps_c("orb_collect: wait for the extraction + downloads", st, false);
    {
    VidoProfScope ps_a("orb_collect: (a) plain wait for the stream", st, false);
    HIP_TRY(ctx, hipStreamSynchronize(st));
    }
    {
    VidoProfScope ps_b("orb_collect: (b) three count downloads + wait", st, false);
    HIP_TRY(ctx, hipMemcpyAsync(S->h_frame_beg, S->d_frame_beg, ((size_t)nf + 1) * sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(S->h_lvloff, S->d_lvloff, ((size_t)nf * L + 1) * sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(S->h_overflow, S->d_overflow, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    }
    HIP_TRY(ctx, hipGetLastError());
    if (*S->h_overflow) {
        const int code = *S->h_overflow; hipMemsetAsync(S->d_overflow, 0, sizeof(int), st);
        return vido_set_error(ctx, VIDO_E_CAPACITY, code == 2 ? "orb: quadtree node list exceeded %d entries" : "orb: internal overflow flag %d", code == 2 ? S->qcap : code);
    }
    const int total = S->h_lvloff[nf * L];
    if ((size_t)total > S->cand_cap) return vido_set_error(ctx, VIDO_E_CAPACITY, "orb: %d FAST candidates exceed the %zu-entry buffer", total, S->cand_cap);
    if ((size_t)S->h_frame_beg[nf] > S->kp_cap) return vido_set_error(ctx, VIDO_E_CAPACITY, "orb: keypoint buffer overflow");
    int maxn = 0;
    for (int f = 0; f < nf; f++) {
        const int n = S->h_frame_beg[f + 1] - S->h_frame_beg[f];
        if (n > S->row_cap) return vido_set_error(ctx, VIDO_E_CAPACITY, "orb: frame %d has %d keypoints > row capacity %d", f, n, S->row_cap);
        maxn = std::max(maxn, n);
    }
    if (copy && maxn > 0) {
        VidoProfScope ps_d("orb_collect: (c) keypoint + descriptor rows (2-D copies) + wait", st, false);
        const size_t kp_pitch = (size_t)S->row_cap * sizeof(vido_keypoint), de_pitch = (size_t)S->row_cap * 32;
        // One frame: plain copies.  A 2-D copy runs as a rectangle kernel of the runtime, which beside saturating network kernels took 1.1 ms for these 112 KB where a plain
        // copy takes 20 us (profiles/r6/tracker_zero_copy_io.txt) — it was the tracker's "first millisecond" of every frame.
        if (nf == 1) {
            HIP_TRY(ctx, hipMemcpyAsync(S->h_kpf, S->d_kpf, (size_t)maxn * sizeof(vido_keypoint), hipMemcpyDeviceToHost, st));
            if (ctx->cfg.compute_descriptors) HIP_TRY(ctx, hipMemcpyAsync(S->h_descf, S->d_descf, (size_t)maxn * 32, hipMemcpyDeviceToHost, st));
        } else {
        HIP_TRY(ctx, hipMemcpy2DAsync(S->h_kpf, kp_pitch, S->d_kpf, kp_pitch, (size_t)maxn * sizeof(vido_keypoint), nf, hipMemcpyDeviceToHost, st));
        if (ctx->cfg.compute_descriptors) HIP_TRY(ctx, hipMemcpy2DAsync(S->h_descf, de_pitch, S->d_descf, de_pitch, (size_t)maxn * 32, nf, hipMemcpyDeviceToHost, st));
        }
        HIP_TRY(ctx, hipStreamSynchronize(st));
    }
    float ms;
    if (S->lean_last) {      // (no stage events on the lean single-frame path)
        for (int i = 0; i < 5; i++) S->timing[i] = 0; S->timing[6] = 0; S->timing[7] = (float)S->h_lvloff[nf * L];
        S->timing[5] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - S->t_start).count();
        return VIDO_OK;
    }
    hipEventElapsedTime(&ms, S->ev[0], S->ev[1]); S->timing[0] = ms;
    hipEventElapsedTime(&ms, S->ev[1], S->ev[7]); S->timing[1] = ms;
    hipEventElapsedTime(&ms, S->ev[7], S->ev[2]); S->timing[6] = ms;
    S->timing[7] = (float)S->h_lvloff[nf * L];
    hipEventElapsedTime(&ms, S->ev[2], S->ev_qt); S->timing[2] = ms;
    S->timing[3] = 0; if (ctx->cfg.compute_descriptors && hipEventElapsedTime(&ms, S->ev[3], S->ev[4]) == hipSuccess) S->timing[3] = ms;      // ran concurrently on stream2
    hipEventElapsedTime(&ms, S->ev[5], S->ev[6]); S->timing[4] = ms;
    S->timing[5] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - S->t_start).count();
    return VIDO_OK;
}

// Starts mirroring the (full-width) result rows into the pinned buffers on the ctx's second stream as soon as the extractor has
// finished, so that the copy overlaps whatever the caller enqueues next on the main stream.  Pair with orb_collect(copy = 0) and a
// hipStreamSynchronize(ctx->stream2).
int orb_mirror_async(vido_ctx* ctx, int nf)
{
    OrbState* S = ctx->orb; hipStream_t s2 = ctx->stream2;
    HIP_TRY(ctx, hipEventRecord(S->ev_done, ctx->stream));
    const bool desc = ctx->cfg.compute_descriptors != 0;
    const int fh = S->half_frames;
    if (fh > 0) {
        HIP_TRY(ctx, hipStreamWaitEvent(s2, S->ev_half, 0));
        HIP_TRY(ctx, hipMemcpyAsync(S->h_kpf, S->d_kpf, (size_t)fh * S->row_cap * sizeof(vido_keypoint), hipMemcpyDeviceToHost, s2));
        if (desc) HIP_TRY(ctx, hipMemcpyAsync(S->h_descf, S->d_descf, (size_t)fh * S->row_cap * 32, hipMemcpyDeviceToHost, s2));
    }
    HIP_TRY(ctx, hipStreamWaitEvent(s2, S->ev_done, 0));
    HIP_TRY(ctx, hipMemcpyAsync(S->h_kpf + (size_t)fh * S->row_cap, S->d_kpf + (size_t)fh * S->row_cap, (size_t)(nf - fh) * S->row_cap * sizeof(vido_keypoint), hipMemcpyDeviceToHost, s2));
    if (desc) HIP_TRY(ctx, hipMemcpyAsync(S->h_descf + (size_t)fh * S->row_cap * 32, S->d_descf + (size_t)fh * S->row_cap * 32, (size_t)(nf - fh) * S->row_cap * 32, hipMemcpyDeviceToHost, s2));
    return VIDO_OK;
}

OrbView orb_view(vido_ctx* ctx)
{
    OrbState* S = ctx->orb;
    return OrbView{S->d_kpf, S->d_nkp, S->row_cap, S->h_kpf, S->h_descf, S->h_frame_beg};
}

static int orb_run(vido_ctx* ctx, const uint8_t* imgs, int on_device, int nf, size_t frame_stride, int stride, int width, int height,
                   vido_keypoint* kp_out, int max_kp, int* n_out, uint8_t* desc_out)
{
    if (!kp_out || !n_out || max_kp <= 0) return vido_set_error(ctx, VIDO_E_INVALID, "orb: null output");
    int rc;
    // (round 6) an extraction of exactly this device image already on the stream (vido_orb_prefetch_color): only the collection is left
    const bool prefetched = ctx->orb->prefetched_src && ctx->orb->prefetched_src == imgs && on_device && nf == 1 && ctx->orb->prefetched_w == width && ctx->orb->prefetched_h == height;
    ctx->orb->prefetched_src = nullptr;
    if (!prefetched) { rc = orb_enqueue(ctx, imgs, on_device, nf, frame_stride, stride, width, height); if (rc) return rc; }
    if ((rc = orb_collect(ctx, nf, 1))) return rc;
    OrbState* S = ctx->orb;
    const int with_desc = ctx->cfg.compute_descriptors ? 1 : 0;
    for (int f = 0; f < nf; f++) {
        const int n = S->h_frame_beg[f + 1] - S->h_frame_beg[f];
        n_out[f] = n;
        const int m = std::min(n, max_kp);
        memcpy(kp_out + (size_t)f * max_kp, S->h_kpf + (size_t)f * S->row_cap, (size_t)m * sizeof(vido_keypoint));
        if (desc_out) {
            if (with_desc) memcpy(desc_out + (size_t)f * max_kp * 32, S->h_descf + (size_t)f * S->row_cap * 32, (size_t)m * 32);
            else memset(desc_out + (size_t)f * max_kp * 32, 0, (size_t)m * 32);
        }
    }
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

int vido_orb_extract_color(vido_ctx* ctx, const uint8_t* img, int channels, int rgb_order, int on_device, int n_frames, size_t frame_stride, int stride,
                           int width, int height, uint8_t* gray_out, vido_keypoint* kp_out, int max_kp, int* n_out, uint8_t* desc_out)
{
    if (!ctx || !ctx->orb) return VIDO_E_INVALID;
    if (channels != 3 && channels != 4) return vido_set_error(ctx, VIDO_E_INVALID, "orb_extract_color: channels must be 3 or 4 (got %d)", channels);
    OrbState* S = ctx->orb;
    S->in_channels = channels; S->in_rgb = rgb_order != 0; S->gray_out = gray_out; S->gray_out_on_device = on_device;
    const int rc = orb_run(ctx, img, on_device, n_frames, frame_stride, stride, width, height, kp_out, max_kp, n_out, desc_out);
    S->in_channels = 1; S->in_rgb = 0; S->gray_out = nullptr;
    return rc;
}

/* Enqueue the extraction of a DEVICE-resident colour frame on the context's stream and return at once; the next vido_orb_extract_color call for the same pointer and size only
 * collects the result.  The extraction needs nothing but the image, so a pipeline can put it on the GPU while it still waits for the frame's other inputs (the networks' maps):
 * System::PrefetchImageDevice.  ready_event (hipEvent_t or NULL): the stream waits for it first (the image's upload).  One frame in flight; the image must stay untouched
 * until it has been collected. */
int vido_orb_prefetch_color(vido_ctx* ctx, const uint8_t* img_dev, int channels, int rgb_order, int stride, int width, int height, void* ready_event)
{
    if (!ctx || !ctx->orb || !img_dev) return VIDO_E_INVALID;
    if (channels != 3 && channels != 4) return vido_set_error(ctx, VIDO_E_INVALID, "orb_prefetch_color: channels must be 3 or 4 (got %d)", channels);
    OrbState* S = ctx->orb;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ready_event) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, (hipEvent_t)ready_event, 0));
    S->in_channels = channels; S->in_rgb = rgb_order != 0; S->gray_out = nullptr; S->gray_out_on_device = 1;
    const int rc = orb_enqueue(ctx, img_dev, 1, 1, 0, stride, width, height);
    S->in_channels = 1; S->in_rgb = 0; S->gray_out = nullptr;
    S->prefetched_src = rc == VIDO_OK ? img_dev : nullptr; S->prefetched_w = width; S->prefetched_h = height;
    return rc;
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
    if (out && n > 0 && cap > 0) {       // candidates stay on the device in the normal path; this debug read copies one level back
        const int m = std::min(std::min(n, cap), S->slot_total);
        HIP_TRY(ctx, hipMemcpyAsync(S->h_cand, S->d_cand + beg, (size_t)m * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        memcpy(out, S->h_cand, (size_t)m * sizeof(uint32_t));
    }
    return n;
}

int vido_orb_last_timing(const vido_ctx* ctx, float ms[8])
{
    if (!ctx || !ctx->orb || !ms) return VIDO_E_INVALID;
    memcpy(ms, ctx->orb->timing, sizeof(float) * 8);
    return VIDO_OK;
}

}  // extern "C"
