// Micro-benchmark: LDS FP64 atomic add vs plain read-add-write vs FP32 atomic, one workgroup per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int MODE>
__global__ __launch_bounds__(1024) void k(double* out, int iters, int stride)
{
    __shared__ double s[8192];
    for (int t = threadIdx.x; t < 8192; t += blockDim.x) s[t] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int idx = (lane * stride + wave * 37) & 8191;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int a = (idx + u * 97) & 8191;
            if (MODE == 0) atomicAdd(&s[a], 1.0);                                  // ds_add_f64
            else if (MODE == 1) { s[a] += 1.0; }                                   // ds_read_b64 + add + ds_write_b64 (racy across waves; throughput only)
            else if (MODE == 2) atomicAdd((float*)&s[a], 1.0f);                    // ds_add_f32
            else if (MODE == 3) { unsigned long long* p = (unsigned long long*)&s[a]; atomicAdd(p, 1ull); }   // ds_add_u64
        }
        idx = (idx + 13) & 8191;
    }
    __syncthreads();
    if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = s[threadIdx.x];
}
template <int MODE> int run(const char* name, int threads, int stride)
{
    double* d; CHECK(hipMalloc(&d, 256 * 64 * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, 10, stride);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, iters, stride);
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double ops = (double)iters * 16 * threads;          // lane-ops per CU
    printf("%-28s threads %4d stride %2d: %.3f ms  -> %.2f lane-ops/clk/CU (at 2.4 GHz)\n", name, threads, stride, ms, ops / (ms * 1e-3 * 2.4e9));
    hipFree(d); return 0;
}
int main()
{
    for (int threads : {64, 256, 1024}) for (int stride : {1, 7}) {
        run<0>("ds_add_f64 (atomicAdd double)", threads, stride);
        run<1>("read+add+write b64", threads, stride);
        run<2>("ds_add_f32", threads, stride);
        run<3>("ds_add_u64", threads, stride);
    }
    return 0;
}
