// What bounds k_fast_strips: the issue rate of the packed 16-bit integer instructions it is made of, measured (round 6).
// For each instruction (v_pk_min_u16, v_pk_max_u16, v_perm_b32, v_alignbit_b32, v_pk_sub_u16 with clamp, v_and_b32 as a plain 32-bit reference, v_mfma-free) a wave issues
// 8 INDEPENDENT dependency chains, 2048 x 8 instructions between two s_memtime stamps; 1 .. 8 waves per SIMD (a workgroup of 256 w threads per CU: waves are dealt
// round-robin to the four SIMDs).  Reported: shader cycles per wave-instruction as ONE wave sees them, and per SIMD (that number / waves per SIMD = the issue cost).
// MI355X_MICROARCH.md says SIMD-32, 2 cycles for v_fma_f32; bench.py's FAST `limiter` assumed 4 cycles per wave64 instruction.  Build: hipcc --offload-arch=gfx950 -O2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define REP8(OPSTR) \
    asm volatile(OPSTR(%0) OPSTR(%1) OPSTR(%2) OPSTR(%3) OPSTR(%4) OPSTR(%5) OPSTR(%6) OPSTR(%7) \
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c))
#define OP_PKMIN(x)  "v_pk_min_u16 " #x ", " #x ", %8\n\t"
#define OP_PKMAX(x)  "v_pk_max_u16 " #x ", " #x ", %8\n\t"
#define OP_PERM(x)   "v_perm_b32 " #x ", " #x ", %8, %8\n\t"
#define OP_ALIGN(x)  "v_alignbit_b32 " #x ", " #x ", %8, 8\n\t"
#define OP_PKSUB(x)  "v_pk_sub_u16 " #x ", " #x ", %8 clamp\n\t"
#define OP_AND(x)    "v_and_b32 " #x ", " #x ", %8\n\t"
#define OP_ADD(x)    "v_add_u32 " #x ", " #x ", %8\n\t"
#define OP_FMA(x)    "v_fma_f32 " #x ", " #x ", %8, %8\n\t"
#define OP_MAX3(x)   "v_max3_u32 " #x ", " #x ", %8, %8\n\t"
#define OP_BCNT(x)   "v_bcnt_u32_b32 " #x ", " #x ", %8\n\t"
template <int OP> __global__ void k(unsigned long long* out, unsigned seed, int iters)
{
    unsigned r0 = threadIdx.x + seed, r1 = r0 * 3, r2 = r0 * 5, r3 = r0 * 7, r4 = r0 * 11, r5 = r0 * 13, r6 = r0 * 17, r7 = r0 * 19, c = seed * 0x9e3779b9u + 0x01010101u;
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {                       // 64 instructions per loop trip: the trip's three scalar instructions are 5 % of the stream
            if (OP == 0) REP8(OP_PKMIN); else if (OP == 1) REP8(OP_PKMAX); else if (OP == 2) REP8(OP_PERM); else if (OP == 3) REP8(OP_ALIGN); else if (OP == 4) REP8(OP_PKSUB);
            else if (OP == 5) REP8(OP_AND); else if (OP == 6) REP8(OP_ADD); else if (OP == 7) REP8(OP_FMA); else if (OP == 8) REP8(OP_MAX3); else REP8(OP_BCNT);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if ((threadIdx.x & 63) == 0) out[(size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 0x12345u) out[0] = 0;            // keep the chains alive
}
typedef void (*kern_t)(unsigned long long*, unsigned, int);
int main()
{
    const char* names[10] = {"v_pk_min_u16", "v_pk_max_u16", "v_perm_b32", "v_alignbit_b32", "v_pk_sub_u16 clamp", "v_and_b32", "v_add_u32", "v_fma_f32", "v_max3_u32", "v_bcnt_u32_b32"};
    kern_t ks[10] = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>, k<9>};
    unsigned long long* d; CHECK(hipMalloc(&d, 256 * 32 * 8));
    static double wall_us[10][9] = {};
    const int iters = 512; const double ninstr = 64.0 * iters;
    printf("%-20s", "waves per SIMD ->");
    for (int w = 1; w <= 8; w++) printf("      %d      ", w);
    printf("\n(each cell: shader cycles per wave-instruction seen by one wave / the same divided by the waves per SIMD = issue cycles per instruction per SIMD)\n");
    for (int op = 0; op < 10; op++) {
        printf("%-20s", names[op]);
        for (int w = 1; w <= 8; w *= 2) {
            if (w == 8) { }      // 1, 2, 4, 8 measured; the columns between stay empty
            const int threads = 256 * w; if (threads > 1024) {      // more than 1024 threads: several workgroups per CU
                const int wgs = threads / 1024;
                for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(ks[op], dim3(256 * wgs), dim3(1024), 0, 0, d, 7u + rep, iters); }
                CHECK(hipDeviceSynchronize());
                std::vector<unsigned long long> h(256 * wgs * 16); CHECK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
                std::sort(h.begin(), h.end()); const double med = (double)h[h.size() / 2];
                printf("  %5.2f /%5.2f ", med / ninstr, med / ninstr / w);
                continue;
            }
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(ks[op], dim3(256), dim3(threads), 0, 0, d, 7u, iters);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(ks[op], dim3(256), dim3(threads), 0, 0, d, 8u, iters);
            hipEventRecord(e1, 0);
            CHECK(hipDeviceSynchronize());
            float ms = 0; hipEventElapsedTime(&ms, e0, e1); wall_us[op][w] = ms * 1e3;
            std::vector<unsigned long long> h(256 * (threads / 64)); CHECK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end()); const double med = (double)h[h.size() / 2];
            printf("  %5.2f /%5.2f ", med / ninstr, med / ninstr / w);
            if (w < 8) for (int skip = w + 1; skip < 2 * w && skip <= 8; skip++) printf("              ");
        }
        printf("\n");
    }
    printf("\nwall clock of one launch (us) and the shader clock it implies (median wave cycles / wall time):\n");
    for (int op = 0; op < 10; op++) { printf("%-20s", names[op]); for (int w = 1; w <= 4; w *= 2) printf("  w=%d %7.1f us", w, wall_us[op][w]); printf("\n"); }
    return 0;
}
