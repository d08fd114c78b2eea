// Cost, in matrix-pipe time, of one instruction of each kind placed after every v_mfma_f32_32x32x2f32 of a one-wave-per-SIMD stream (see mfma_fillers.hip).
// build: hipcc -w --offload-arch=gfx950 -O3 tools/ubench/mfma_fillers2.hip -o tools/ubench/mfma_fillers2.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int KIND, int N> __global__ __launch_bounds__(256) void k(float* out, const float* src, int iters, unsigned long long* clk)
{
    f32x16 acc[16];
#pragma unroll
    for (int i = 0; i < 16; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    float v[8]; for (int i = 0; i < 8; i++) v[i] = a + i;
    f32x2 p[4]; for (int i = 0; i < 4; i++) p[i] = f32x2{a + i, b + i};
    unsigned u[8]; for (int i = 0; i < 8; i++) u[i] = threadIdx.x + i;
    float ld[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const float* gp = src + threadIdx.x;
    unsigned s0 = blockIdx.x;
    const unsigned long long c0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < N; f++) {
                if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[f & 7]) : "v"(b));
                if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[f & 3]) : "v"(p[(f + 1) & 3]));
                if (KIND == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[f & 7]) : "v"(u[(f + 1) & 7]));
                if (KIND == 3) asm volatile("v_mov_b32 %0, %1" : "=v"(u[f & 7]) : "v"(u[(f + 1) & 7]));
                if (KIND == 4) asm volatile("global_load_dword %0, %1, off" : "=v"(ld[f & 7]) : "v"(gp));
                if (KIND == 5) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0));
                if (KIND == 6) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[f & 7]) : "v"(b));
                if (KIND == 7) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]" : "+v"(p[f & 3]) : "v"(p[(f + 1) & 3]));
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long c1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    for (int i = 0; i < 8; i++) s += v[i] + (float)u[i] + ld[i];
    for (int i = 0; i < 4; i++) s += p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)s0;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = c1 - c0;
}
template <int KIND, int N> void run(const char* name)
{
    const int wgs = 256, iters = 3000;
    float *out, *src; unsigned long long* clk; (void)hipMalloc(&out, wgs * 256 * 4); (void)hipMalloc(&src, 4096); (void)hipMalloc(&clk, 16);
    hipLaunchKernelGGL((k<KIND, N>), dim3(wgs), dim3(256), 0, 0, out, src, iters, clk); (void)hipDeviceSynchronize();
    hipLaunchKernelGGL((k<KIND, N>), dim3(wgs), dim3(256), 0, 0, out, src, iters, clk); (void)hipDeviceSynchronize();
    unsigned long long h[2]; (void)hipMemcpy(h, clk, 8, hipMemcpyDeviceToHost);
    const double cyc = (double)h[0] / (16.0 * iters);
    printf("%-44s x %d per matrix instruction -> %6.1f cycles (%+.1f per instruction)\n", name, N, cyc, (cyc - 64.0) / N);
    (void)hipFree(out); (void)hipFree(src); (void)hipFree(clk);
}
int main()
{
    run<0, 4>("v_add_f32"); run<6, 4>("v_fma_f32"); run<1, 4>("v_pk_add_f32"); run<1, 2>("v_pk_add_f32"); run<7, 4>("v_pk_add_f32 op_sel / neg modifiers");
    run<2, 4>("v_add_u32"); run<3, 4>("v_mov_b32"); run<4, 1>("global_load_dword"); run<4, 2>("global_load_dword"); run<4, 4>("global_load_dword"); run<5, 8>("s_add_u32");
    return 0;
}
