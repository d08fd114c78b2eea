// How many workgroups of a given dynamic-LDS size share a CU?  512 workgroups (2 per CU on 256 CUs) of 256 threads sleep ~20 us each: the launch takes ~20 us when two
// are co-resident, ~40 us when the LDS request lets only one in.
// build: hipcc -w --offload-arch=gfx950 -O3 tools/ubench/lds_occupancy.hip -o tools/ubench/lds_occupancy.bin
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(float* out, int spins)
{
    extern __shared__ float lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    for (int i = 0; i < spins; i++) __builtin_amdgcn_s_sleep(100);      // 6400 cycles each
    out[blockIdx.x * 256 + threadIdx.x] = lds[255 - threadIdx.x];
}
int main()
{
    float* out; (void)hipMalloc(&out, 4096 * 256 * 4);
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int wgs : {256, 512, 1024})
        for (int kb : {8, 32, 40, 52, 53, 60, 64, 70, 72, 76, 79, 80, 81, 100}) {
            for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k, dim3(wgs), dim3(256), kb * 1024, 0, out, 8);
            (void)hipEventRecord(e0, 0);
            for (int rep = 0; rep < 5; rep++) hipLaunchKernelGGL(k, dim3(wgs), dim3(256), kb * 1024, 0, out, 8);
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%4d workgroups x %3d KB LDS: %.1f us per launch\n", wgs, kb, ms * 1000 / 5);
        }
    return 0;
}
