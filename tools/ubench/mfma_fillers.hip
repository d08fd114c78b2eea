// How many vector / LDS instructions of the SAME wave hide behind one v_mfma_f32_32x32x2f32 (64 cycles) at one wave per SIMD: back-to-back matrix instructions on 16
// accumulators with NV v_add_f32, NR ds_read_b32, NW ds_write_b32 placed after each of them (inline asm, so the compiler neither removes nor moves them).
// build: hipcc -w --offload-arch=gfx950 -O3 tools/ubench/mfma_fillers.hip -o tools/ubench/mfma_fillers.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV, int NR, int NW> __global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* clk)
{
    __shared__ float lds[4096];
    f32x16 acc[16];
#pragma unroll
    for (int i = 0; i < 16; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    float v[8]; for (int i = 0; i < 8; i++) v[i] = a + i;
    float rd[4] = {0, 0, 0, 0};
    const unsigned la = (threadIdx.x & 1023) * 4;
    const unsigned long long c0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < NV; f++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[f & 7]) : "v"(b));
#pragma unroll
            for (int f = 0; f < NR; f++) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(rd[f & 3]) : "v"(la), "n"(256 * (f & 3)));
#pragma unroll
            for (int f = 0; f < NW; f++) asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(la), "v"(v[f & 7]), "n"(4096 + 256 * (f & 3)));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long c1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    for (int i = 0; i < 8; i++) s += v[i];
    for (int i = 0; i < 4; i++) s += rd[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + lds[threadIdx.x];
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = c1 - c0;
}
template <int NV, int NR, int NW> void run()
{
    const int wgs = 256, iters = 4000;
    float* out; unsigned long long* clk; (void)hipMalloc(&out, wgs * 256 * 4); (void)hipMalloc(&clk, 16);
    hipLaunchKernelGGL((k<NV, NR, NW>), dim3(wgs), dim3(256), 0, 0, out, iters, clk); (void)hipDeviceSynchronize();
    hipLaunchKernelGGL((k<NV, NR, NW>), dim3(wgs), dim3(256), 0, 0, out, iters, clk); (void)hipDeviceSynchronize();
    unsigned long long h[2]; (void)hipMemcpy(h, clk, 8, hipMemcpyDeviceToHost);
    printf("per matrix instruction: %2d v_add_f32 %d ds_read_b32 %d ds_write_b32 -> %.1f cycles\n", NV, NR, NW, (double)h[0] / (16.0 * iters));
    (void)hipFree(out); (void)hipFree(clk);
}
int main()
{
    run<0, 0, 0>(); run<2, 0, 0>(); run<4, 0, 0>(); run<6, 0, 0>(); run<8, 0, 0>(); run<12, 0, 0>(); run<16, 0, 0>();
    run<0, 1, 0>(); run<0, 2, 0>(); run<0, 4, 0>(); run<0, 0, 1>(); run<0, 0, 2>(); run<4, 1, 1>(); run<4, 2, 0>();
    return 0;
}
