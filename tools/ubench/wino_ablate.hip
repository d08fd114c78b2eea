// Where a chunk of k_wino3x3<2,2,8> spends its time: the kernel compiled with parts of its loop removed (WN_ABLATE bits; results are wrong, timings are the point).
// build: for a in 0 1 2 3 4 7 8 15 31; do hipcc -w --offload-arch=gfx950 -O3 -std=c++17 -DWN_ABLATE=$a tools/ubench/wino_ablate.hip -o tools/ubench/wino_ablate_$a.bin; done
#include "../../vido-slam_amd/csrc/wino.hip"
#include <cstdarg>
int vido_set_error(vido_ctx*, int code, const char*, ...) { return code; }
int main()
{
    const int H = 256, W = 128, cout = 64;
    for (int cin : {256, 1024}) {
        float *x, *u, *y; const size_t nx = (size_t)cin * H * W, nu = (size_t)vido_wino3x3_packed_floats(cin, cout), ny = (size_t)cout * H * W;
        if (hipMalloc(&x, nx * 4) != hipSuccess || hipMalloc(&u, nu * 4) != hipSuccess || hipMalloc(&y, ny * 4) != hipSuccess) return 1;
        (void)hipMemset(x, 0, nx * 4); (void)hipMemset(u, 0, nu * 4);
        const int ht = H / 2, wt = W / 2, T = ht * wt, total = T / 64;
        WnArgs A{x, u, nullptr, y, 1, cin, cout, H, W, ht, wt, T, cin / 8, 1, total, 0.1f, 1, (unsigned)(nx * 4), (unsigned)(nu * 4), nullptr};
#ifdef WN_PROF
        unsigned long long* prof; (void)hipMalloc(&prof, 96 * 8); (void)hipMemset(prof, 0, 96 * 8); A.prof = prof;
#endif
        const size_t lds = (size_t)2 * (2 * 16 * 64 * 4 + 16 * 2 * 4 * 64) * 4;
        (void)hipFuncSetAttribute((const void*)k_wino3x3<2, 2, 8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL((k_wino3x3<2, 2, 8, false>), dim3(8 * ((total + 7) / 8)), dim3(256), lds, 0, A);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int rep = 0; rep < 10; rep++) hipLaunchKernelGGL((k_wino3x3<2, 2, 8, false>), dim3(8 * ((total + 7) / 8)), dim3(256), lds, 0, A);
        (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("ablate %2d cin %4d: %7.1f us per launch, %.3f us per chunk (%s)\n", WN_ABLATE, cin, ms * 100.f, ms * 100.f / (cin / 8), hipGetErrorString(hipGetLastError()));
#ifdef WN_PROF
        unsigned long long hp[96]; (void)hipMemcpy(hp, prof, sizeof hp, hipMemcpyDeviceToHost);
        for (int wv = 0; wv < 8; wv++) {                                   // waves 0-3 of workgroup 0, then of the middle one: last chunk, cycles (100 MHz stamps would be s_memrealtime; s_memtime = shader clock)
            printf("  wg %s wave %d: wait+barrier %5lld |", wv < 4 ? "0  " : "mid", wv & 3, (long long)(hp[wv * 10 + 1] - hp[wv * 10]));
            for (int k = 2; k < 10; k++) printf(" %5lld", (long long)(hp[wv * 10 + k] - hp[wv * 10 + k - 1]));
            printf(" | chunk %lld\n", (long long)(hp[wv * 10 + 9] - hp[wv * 10]));
        }
        for (int wv = 0; wv < 4; wv++) printf("  wg 0 wave %d: entry -> loop %lld cycles, loop %lld, loop exit -> stores done %lld\n", wv, (long long)(hp[80 + wv * 4 + 1] - hp[80 + wv * 4]), (long long)(hp[80 + wv * 4 + 2] - hp[80 + wv * 4 + 1]), (long long)(hp[80 + wv * 4 + 3] - hp[80 + wv * 4 + 2]));
#endif
        (void)hipFree(x); (void)hipFree(u); (void)hipFree(y);
    }
    return 0;
}
