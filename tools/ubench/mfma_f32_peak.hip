// What the fp32 matrix pipe sustains on this chip: every wave issues back-to-back v_mfma_f32_32x32x2f32 on NA independent accumulators, nothing else in the loop.
// 4 waves per workgroup (one per SIMD), `wgs` workgroups; prints TFLOP/s and the shader clock the run saw (clock64 / wall_clock64 at 100 MHz).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_f32_peak.hip -o tools/ubench/mfma_f32_peak.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NA> __global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* clk)
{
    f32x16 acc[NA];
#pragma unroll
    for (int i = 0; i < NA; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NA; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NA; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
template <int NA> void run(int wgs, int iters)
{
    float* out; unsigned long long* clk; hipMalloc(&out, wgs * 256 * 4); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NA>, dim3(wgs), dim3(256), 0, 0, out, iters, clk); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<NA>, dim3(wgs), dim3(256), 0, 0, out, iters, clk); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flop = 2.0 * 32 * 32 * 2 * (double)NA * iters * 4 * wgs;
    printf("NA %2d wgs %4d iters %d: %.3f ms  %.1f TFLOP/s   cycles per mfma %.1f  shader clock %.0f MHz\n", NA, wgs, iters, ms, flop / ms / 1e9, (double)h[0] / ((double)NA * iters), (double)h[0] / (double)h[1] * 100.0);
    hipFree(out); hipFree(clk);
}
int main()
{
    run<16>(256, 2000); run<16>(256, 20000); run<16>(128, 20000); run<16>(512, 20000); run<4>(256, 40000); run<1>(256, 100000);
    return 0;
}
