// hamming.hip — brute-force 256-bit Hamming matcher on gfx950.
// No reference call site exists (the reference associates features by optical flow only,
// SURVEY.md fact 2); BASELINE.json north_star asks for it: for every query descriptor the closest
// train descriptor, smallest distance, lowest index on ties.
// Mapping: one wave64 owns QW=8 queries (wave-uniform, kept in scalar registers) and streams a slab
// of the train set: lane j loads train descriptor j (32 B, coalesced 2 KiB per wave load), scores it
// against the 8 queries with v_bcnt (popcount-accumulate), keeps a per-lane running minimum, then a
// wavefront-shuffle (DPP) reduction picks the wave minimum of (distance<<32 | index) — the packed key
// also implements the lowest-index tie rule; per-slab partial minima land in a [query][slab] table
// that a second tiny kernel folds (deterministic, no atomics).
#include "common.hpp"

#define QW 8

__global__ __launch_bounds__(256) void k_hamming(const uint32_t* __restrict__ A, int na, const uint32_t* __restrict__ Bd, int nb,
                                                 int slab, int nsl, unsigned long long* __restrict__ part)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q0 = __builtin_amdgcn_readfirstlane((blockIdx.x * 4 + wave) * QW);
    if (q0 >= na) return;
    uint32_t Q[QW][8];
#pragma unroll
    for (int q = 0; q < QW; q++) {
        const int qi = min(q0 + q, na - 1);
#pragma unroll
        for (int w = 0; w < 8; w++) Q[q][w] = A[(size_t)qi * 8 + w];       // wave-uniform address -> scalar loads
    }
    const int jbeg = blockIdx.y * slab, jend = min(jbeg + slab, nb);
    unsigned long long loc[QW];
#pragma unroll
    for (int q = 0; q < QW; q++) loc[q] = ~0ull;
    for (int j = jbeg + lane; j < jend; j += 64) {
        const uint4 b0 = *(const uint4*)(Bd + (size_t)j * 8), b1 = *(const uint4*)(Bd + (size_t)j * 8 + 4);
#pragma unroll
        for (int q = 0; q < QW; q++) {
            int d = __popc(b0.x ^ Q[q][0]) + __popc(b0.y ^ Q[q][1]) + __popc(b0.z ^ Q[q][2]) + __popc(b0.w ^ Q[q][3]) +
                    __popc(b1.x ^ Q[q][4]) + __popc(b1.y ^ Q[q][5]) + __popc(b1.z ^ Q[q][6]) + __popc(b1.w ^ Q[q][7]);
            const unsigned long long key = ((unsigned long long)(uint32_t)d << 32) | (uint32_t)j;
            loc[q] = key < loc[q] ? key : loc[q];
        }
    }
#pragma unroll
    for (int q = 0; q < QW; q++) {
        unsigned long long v = loc[q];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const unsigned long long other = __shfl_xor(v, o, 64);
            v = other < v ? other : v;
        }
        if (lane == 0 && q0 + q < na) part[(size_t)(q0 + q) * nsl + blockIdx.y] = v;
    }
}

__global__ void k_hamming_finish(const unsigned long long* __restrict__ part, int nsl, int na, int* __restrict__ idx, int* __restrict__ dist)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= na) return;
    unsigned long long v = ~0ull;
    for (int s = 0; s < nsl; s++) { const unsigned long long o = part[(size_t)i * nsl + s]; v = o < v ? o : v; }
    if (v == ~0ull) { idx[i] = -1; dist[i] = -1; }
    else { idx[i] = (int)(uint32_t)v; dist[i] = (int)(v >> 32); }
}

// Persistent, growable device/pinned buffers (no per-call stream-ordered allocations: inputs are staged
// through pinned memory so every transfer is ordered on the ctx stream).
struct HamState {
    uint8_t *d_a = nullptr, *d_b = nullptr; int *d_idx = nullptr, *d_dist = nullptr; unsigned long long* d_part = nullptr;
    uint8_t* h_stage = nullptr;
    size_t cap_a = 0, cap_b = 0, cap_part = 0, cap_stage = 0;
};

template <class T>
static int grow(vido_ctx* ctx, T** p, size_t* cap, size_t need, bool pinned = false)
{
    if (need <= *cap) return VIDO_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (*p) { if (pinned) HIP_TRY(ctx, hipHostFree(*p)); else HIP_TRY(ctx, hipFree(*p)); *p = nullptr; }
    const size_t n = need + need / 2 + 64;
    if (pinned) HIP_TRY(ctx, hipHostMalloc((void**)p, n * sizeof(T))); else HIP_TRY(ctx, hipMalloc((void**)p, n * sizeof(T)));
    *cap = n;
    return VIDO_OK;
}

void ham_state_destroy(vido_ctx* ctx)
{
    HamState* S = ctx->ham;
    if (!S) return;
    hipFree(S->d_a); hipFree(S->d_b); hipFree(S->d_idx); hipFree(S->d_dist); hipFree(S->d_part); hipHostFree(S->h_stage);
    delete S; ctx->ham = nullptr;
}

extern "C" int vido_hamming_match(vido_ctx* ctx, const uint8_t* a, int na, const uint8_t* b, int nb,
                                  int32_t* idx_out, int32_t* dist_out, int on_device)
{
    if (!ctx) return VIDO_E_INVALID;
    if (na < 0 || nb < 0 || (na > 0 && (!a || !idx_out || !dist_out)) || (nb > 0 && !b))
        return vido_set_error(ctx, VIDO_E_INVALID, "hamming: bad arguments");
    if (na == 0) return VIDO_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->ham) ctx->ham = new HamState();
    HamState* S = ctx->ham;
    hipStream_t st = ctx->stream;
    const int qblocks = (na + 4 * QW - 1) / (4 * QW);
    // enough train slabs that the grid has >= ~2048 workgroups, slabs a multiple of 64 descriptors
    int nsl = std::max(1, std::min((nb + 63) / 64, (2048 + qblocks - 1) / qblocks));
    int slab = ((std::max(nb, 1) + nsl - 1) / nsl + 63) & ~63;
    nsl = std::max(1, (nb + slab - 1) / slab);
    int rc;
    if ((rc = grow(ctx, &S->d_part, &S->cap_part, (size_t)na * nsl))) return rc;
    const uint8_t *da = a, *db = b; int *didx = idx_out, *ddist = dist_out;
    if (!on_device) {
        size_t capi = S->cap_a;
        if ((rc = grow(ctx, &S->d_a, &S->cap_a, (size_t)na * 32))) return rc;
        if (S->cap_a != capi || !S->d_idx) {
            if (S->d_idx) { HIP_TRY(ctx, hipFree(S->d_idx)); HIP_TRY(ctx, hipFree(S->d_dist)); }
            HIP_TRY(ctx, hipMalloc((void**)&S->d_idx, S->cap_a / 32 * 4 + 64)); HIP_TRY(ctx, hipMalloc((void**)&S->d_dist, S->cap_a / 32 * 4 + 64));
        }
        if ((rc = grow(ctx, &S->d_b, &S->cap_b, (size_t)std::max(nb, 1) * 32))) return rc;
        const size_t bytes_in = ((size_t)na + nb) * 32, bytes_out = (size_t)na * 8;
        if ((rc = grow(ctx, &S->h_stage, &S->cap_stage, std::max(bytes_in, bytes_out), true))) return rc;
        memcpy(S->h_stage, a, (size_t)na * 32);
        if (nb > 0) memcpy(S->h_stage + (size_t)na * 32, b, (size_t)nb * 32);
        HIP_TRY(ctx, hipMemcpyAsync(S->d_a, S->h_stage, (size_t)na * 32, hipMemcpyHostToDevice, st));
        if (nb > 0) HIP_TRY(ctx, hipMemcpyAsync(S->d_b, S->h_stage + (size_t)na * 32, (size_t)nb * 32, hipMemcpyHostToDevice, st));
        da = S->d_a; db = S->d_b; didx = S->d_idx; ddist = S->d_dist;
    }
    if (nb > 0)
        hipLaunchKernelGGL(k_hamming, dim3(qblocks, nsl), dim3(256), 0, st, (const uint32_t*)da, na, (const uint32_t*)db, nb, slab, nsl, S->d_part);
    else
        HIP_TRY(ctx, hipMemsetAsync(S->d_part, 0xff, (size_t)na * nsl * 8, st));
    hipLaunchKernelGGL(k_hamming_finish, dim3((na + 255) / 256), dim3(256), 0, st, S->d_part, nsl, na, didx, ddist);
    HIP_TRY(ctx, hipGetLastError());
    if (!on_device) {
        HIP_TRY(ctx, hipStreamSynchronize(st));        // staging buffer is reused for the results
        int* hs = (int*)S->h_stage;
        HIP_TRY(ctx, hipMemcpyAsync(hs, S->d_idx, (size_t)na * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(hs + na, S->d_dist, (size_t)na * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        memcpy(idx_out, hs, (size_t)na * 4); memcpy(dist_out, hs + na, (size_t)na * 4);
    }
    return VIDO_OK;
}
