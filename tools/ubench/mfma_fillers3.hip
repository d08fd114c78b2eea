// Issue cost beside v_mfma_f32_32x32x2f32 of the memory instructions k_wino3x3 could use for its input windows and its U copies (see mfma_fillers.hip), and a check that
// a buffer_load_dwordx4 at a 4-byte-aligned (not 16-byte-aligned) offset returns the right four dwords, with per-dword range checking at the end of the buffer.
// build: hipcc -w --offload-arch=gfx950 -O3 tools/ubench/mfma_fillers3.hip -o tools/ubench/mfma_fillers3.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int KIND, int N> __global__ __launch_bounds__(256) void k(float* out, const float* src, int nsrc, int iters, unsigned long long* clk)
{
    __shared__ __attribute__((aligned(16))) float lds[8192];
    f32x16 acc[16];
#pragma unroll
    for (int i = 0; i < 16; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nsrc * 4, 0x00020000);
    const unsigned vo = (threadIdx.x & 63) * 8 + 4;                        // 4-byte aligned, not 16
    const unsigned vo16 = (threadIdx.x & 63) * 16;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float ld[4] = {0, 0, 0, 0}; f32x4 l4[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    const unsigned long long c0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < N; f++) {
                if (KIND == 0) ld[f & 3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, 1024 * f, 0));
                if (KIND == 1) l4[f & 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 1024 * f, 0));
                if (KIND == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + wv * 2048 + (f & 3) * 256), 16, vo16, 4096 * f, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long c1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    for (int i = 0; i < 4; i++) s += ld[i] + l4[0][i] + l4[1][i];
    out[blockIdx.x * 256 + threadIdx.x] = s + lds[threadIdx.x];
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = c1 - c0;
}
__global__ void check(const float* src, int nsrc, float* out)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nsrc * 4, 0x00020000);
    const unsigned off = 4u * (threadIdx.x * 2 + 1);                       // odd element index
    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    for (int i = 0; i < 4; i++) out[threadIdx.x * 4 + i] = v[i];
}
template <int KIND, int N> void run(const char* name)
{
    const int wgs = 256, iters = 3000, nsrc = 1 << 20;
    float *out, *src; unsigned long long* clk; (void)hipMalloc(&out, wgs * 256 * 4); (void)hipMalloc(&src, nsrc * 4); (void)hipMalloc(&clk, 16);
    hipLaunchKernelGGL((k<KIND, N>), dim3(wgs), dim3(256), 0, 0, out, src, nsrc, iters, clk); (void)hipDeviceSynchronize();
    hipLaunchKernelGGL((k<KIND, N>), dim3(wgs), dim3(256), 0, 0, out, src, nsrc, iters, clk); (void)hipDeviceSynchronize();
    unsigned long long h[2]; (void)hipMemcpy(h, clk, 8, hipMemcpyDeviceToHost);
    const double cyc = (double)h[0] / (16.0 * iters);
    printf("%-44s x %d per matrix instruction -> %6.1f cycles (%+.1f per instruction)\n", name, N, cyc, (cyc - 64.0) / N);
    (void)hipFree(out); (void)hipFree(src); (void)hipFree(clk);
}
int main()
{
    const int n = 64 * 2 + 2;                                              // the last lanes run past the end: per-dword zeros expected
    float h[n]; for (int i = 0; i < n; i++) h[i] = 1.0f + i;
    float *src, *out; (void)hipMalloc(&src, n * 4); (void)hipMalloc(&out, 64 * 4 * 4); (void)hipMemcpy(src, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, src, n, out); float o[256]; (void)hipMemcpy(o, out, sizeof o, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 64; t++) for (int i = 0; i < 4; i++) { const int e = 2 * t + 1 + i; const float want = e < n ? 1.0f + e : 0.f; if (o[t * 4 + i] != want) { if (bad < 5) printf("lane %d dword %d: got %g want %g\n", t, i, o[t * 4 + i], want); bad++; } }
    printf("unaligned buffer_load_dwordx4 + per-dword range check: %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
    run<0, 1>("buffer_load_dword offen"); run<0, 2>("buffer_load_dword offen"); run<1, 1>("buffer_load_dwordx4 offen (4-byte aligned)"); run<1, 2>("buffer_load_dwordx4 offen (4-byte aligned)");
    run<2, 1>("buffer_load_dwordx4 offen lds"); run<2, 2>("buffer_load_dwordx4 offen lds");
    return 0;
}
