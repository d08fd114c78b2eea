// Do the vector-ALU instructions of ANOTHER wave on the same SIMD hide behind fp32 matrix instructions?  Workgroup of 8 waves = two per SIMD: waves 0-3 run back-to-back
// v_mfma_f32_32x32x2f32 (or 16x16x4) on 16 independent accumulators, waves 4-7 run independent v_add_f32 (8 chains).  Each role is timed alone and beside the other one.
// build: hipcc -w --offload-arch=gfx950 -O3 tools/ubench/mfma_two_waves.hip -o tools/ubench/mfma_two_waves.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND, int VK> __global__ __launch_bounds__(512) void k(float* out, int it_m, int it_v, unsigned long long* clk, const float* src)
{
    const int wv = threadIdx.x >> 6;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f, s = 0.f;
    if (wv < 4) {
        if (it_m == 0) return;
        f32x16 acc[8]; f32x4 acc4[16];      // (two waves per SIMD: 256 registers each)
#pragma unroll
        for (int i = 0; i < 16; i++) { for (int r = 0; r < 16; r++) acc[i & 7][r] = 0.f; for (int r = 0; r < 4; r++) acc4[i][r] = 0.f; }
        const unsigned long long c0 = clock64();
        for (int it = 0; it < it_m; it++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (KIND == 0) acc[i & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i & 7], 0, 0, 0);
                else acc4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[i], 0, 0, 0);
            }
        }
        const unsigned long long c1 = clock64();
#pragma unroll
        for (int i = 0; i < 16; i++) { for (int r = 0; r < 16; r++) s += acc[i & 7][r]; for (int r = 0; r < 4; r++) s += acc4[i][r]; }
        if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = c1 - c0;
    } else if (it_v < 0) {      // the second wave of the SIMD issues matrix instructions as well
        f32x16 acc[8];
#pragma unroll
        for (int i = 0; i < 8; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
        const unsigned long long c0 = clock64();
        for (int it = 0; it < -it_v; it++) {
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i & 7], 0, 0, 0);
        }
        const unsigned long long c1 = clock64();
#pragma unroll
        for (int i = 0; i < 8; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
        if (blockIdx.x == 0 && threadIdx.x == 256) clk[1] = c1 - c0;
    } else {
        if (it_v == 0) return;
        float v[8]; for (int i = 0; i < 8; i++) v[i] = a + i;
        __shared__ float lds[8192];
        const unsigned la = (threadIdx.x & 255) * 4;
        const float* gp = src + (threadIdx.x & 255) + (size_t)blockIdx.x * 4096;
        const unsigned long long c0 = clock64();
        for (int it = 0; it < it_v; it++) {
#pragma unroll
            for (int f = 0; f < 64; f++) {
                if (VK == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[f & 7]) : "v"(b));
                if (VK == 1) asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(la), "v"(v[f & 7]), "n"(1024 * (f & 7)) : "memory");
                if (VK == 2) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[f & 7]) : "v"(la), "n"(1024 * (f & 7)) : "memory");
                if (VK == 3) asm volatile("global_load_dword %0, %1, off offset:%2" : "=v"(v[f & 7]) : "v"(gp), "n"(1024 * (f & 3)) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        const unsigned long long c1 = clock64();
        if (threadIdx.x == 511) out[0] = lds[threadIdx.x];
        for (int i = 0; i < 8; i++) s += v[i];
        if (blockIdx.x == 0 && threadIdx.x == 256) clk[1] = c1 - c0;
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int KIND, int VK = 0> void run(int it_m, int it_v, const char* what)
{
    const int wgs = 256;
    float* out; unsigned long long* clk; float* src; (void)hipMalloc(&out, wgs * 512 * 4); (void)hipMalloc(&clk, 16); (void)hipMemset(clk, 0, 16); (void)hipMalloc(&src, (size_t)wgs * 4096 * 4 + 65536); (void)hipMemset(src, 0, (size_t)wgs * 4096 * 4 + 65536);
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL((k<KIND, VK>), dim3(wgs), dim3(512), 0, 0, out, it_m, it_v, clk, (const float*)src); (void)hipDeviceSynchronize(); }
    unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double cyc_m = KIND == 0 ? 64.0 : 32.0;
    printf("%-34s %s: ", what, KIND == 0 ? "32x32x2" : "16x16x4");
    if (it_m) printf("matrix waves %.1f cycles per instruction (alone: %.0f)  ", (double)h[0] / (16.0 * it_m), cyc_m);
    if (it_v > 0) printf("second waves %.2f cycles per %s", (double)h[1] / (64.0 * it_v), VK == 0 ? "v_add_f32" : VK == 1 ? "ds_write_b32" : VK == 2 ? "ds_read_b32" : "global_load_dword");
    if (it_v < 0) printf("second matrix waves %.1f cycles per instruction", (double)h[1] / (16.0 * -it_v));
    printf("\n");
    (void)hipFree(out); (void)hipFree(clk); (void)hipFree(src);
}
int main()
{
    run<0>(4000, 0, "matrix alone"); run<0>(0, 16000, "vector alone"); run<0>(4000, 16000, "both (equal work if hidden)"); run<0>(4000, 64000, "both, vector outlasts matrix");
    run<0>(4000, -4000, "two matrix waves per SIMD");
    run<0, 1>(0, 4000, "LDS writes alone"); run<0, 1>(4000, 4000, "LDS writes beside matrix");
    run<0, 2>(0, 4000, "LDS reads alone"); run<0, 2>(4000, 4000, "LDS reads beside matrix");
    run<0, 3>(0, 1000, "global loads alone"); run<0, 3>(4000, 1000, "global loads beside matrix");
    run<1>(8000, 0, "matrix alone"); run<1>(8000, 16000, "both"); run<1>(8000, 64000, "both, vector outlasts matrix");
    return 0;
}
