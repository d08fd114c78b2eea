// xwg.hpp — exchange between the WORKGROUPS of one launch on gfx950 (8 XCDs with private L2s, a vector L1 per CU): the two forms the persistent solvers use.
//   * tagged granules (poseopt.hip): a workgroup publishes a handful of doubles, every workgroup of its cluster needs all of them — an all-reduce of 2..56 values.  Each
//     double travels as two 8-byte words {tag = epoch, 32 bits of payload}, written with agent-scope (write-through) stores and re-read with agent-scope loads until every tag
//     is the current epoch: the data is its own flag, one memory hop per exchange, no counter, no fence.
//   * counter barrier (ba.hip): phases that hand large arrays to each other (slot records, partial reduced systems) through plain stores: agent-scope release (L2 write-back)
//     by one lane per workgroup, arrival counter, relaxed polling, agent-scope acquire.
// Residency: every workgroup of the launch must be resident for either form to make progress; the callers keep their grids far below the chip (<= 64 workgroups, one per CU)
// and EVERY spin is bounded: a timeout raises the launch's abort word, all workgroups leave their loops, the host reports VIDO_E_HIP instead of hanging the queue.
#pragma once
#include <hip/hip_runtime.h>

typedef __attribute__((address_space(1))) unsigned long long xwg_gu64;
typedef __attribute__((address_space(1))) unsigned int xwg_gu32;
#define XWG_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#ifndef XWG_SPIN_MAX
#define XWG_SPIN_MAX (1u << 20)          // ~1 us per poll: about a second before a wait gives up
#endif

__device__ __forceinline__ void xwg_store64(unsigned long long* p, unsigned long long v) { __hip_atomic_store((xwg_gu64*)p, v, XWG_RLX_AGENT); }
__device__ __forceinline__ unsigned long long xwg_load64(const unsigned long long* p) { return __hip_atomic_load((xwg_gu64*)p, XWG_RLX_AGENT); }
__device__ __forceinline__ unsigned xwg_load32(const unsigned* p) { return __hip_atomic_load((xwg_gu32*)p, XWG_RLX_AGENT); }
__device__ __forceinline__ void xwg_store32(unsigned* p, unsigned v) { __hip_atomic_store((xwg_gu32*)p, v, XWG_RLX_AGENT); }

// one double as two tagged words at g[0], g[1]
__device__ __forceinline__ void xwg_publish_f64(unsigned long long* g, unsigned epoch, double x)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(x), e = (unsigned long long)epoch << 32;
    xwg_store64(g, e | (b & 0xffffffffull));
    xwg_store64(g + 1, e | (b >> 32));
}
// waits until word *g carries `epoch`, returns its payload; false on timeout / abort (the caller makes the outcome uniform)
__device__ __forceinline__ bool xwg_wait_word(const unsigned long long* g, unsigned epoch, const unsigned* abort_word, unsigned* payload)
{
    for (unsigned spins = 0;; spins++) {
        const unsigned long long x = xwg_load64(g);
        if ((unsigned)(x >> 32) == epoch) { *payload = (unsigned)x; return true; }
        if (spins > XWG_SPIN_MAX || ((spins & 255u) == 255u && xwg_load32(abort_word) != 0u)) { *payload = 0u; return false; }
        __builtin_amdgcn_s_sleep(1);
    }
}

// Grid-wide barrier over `nwg` workgroups.  `count` only ever grows: it stood at `base` when the launch began (the host keeps the sum: every launch reports how many
// barriers it ran), `phase` counts this launch's barriers from 1.  Returns false when the wait gave up.  lflag: one int of LDS.  All threads of the workgroup call it.
__device__ __forceinline__ bool xwg_grid_barrier(unsigned* count, unsigned* abort_word, unsigned nwg, unsigned base, unsigned phase, int* lflag)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every wave: its stores of the finished phase have left
    __syncthreads();
    if (threadIdx.x == 0) {
        int ok = 1;
        if (nwg > 1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // write the XCD's dirty L2 lines back: the other XCDs read memory
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add((xwg_gu32*)count, 1u, XWG_RLX_AGENT);
            const unsigned target = base + phase * nwg;
            for (unsigned spins = 0; (int)(xwg_load32(count) - target) < 0; spins++) {      // (wrap-safe comparison)
                if (spins > XWG_SPIN_MAX || ((spins & 255u) == 255u && xwg_load32(abort_word) != 0u)) { ok = 0; xwg_store32(abort_word, 1u); break; }
                __builtin_amdgcn_s_sleep(2);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // drop this CU's (and this XCD's non-local) stale lines; covers the workgroup through the barrier below
        }
        *lflag = ok;
    }
    __syncthreads();
    return *lflag != 0;
}
