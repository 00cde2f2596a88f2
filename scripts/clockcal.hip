// Build: hipcc -O2 --offload-arch=gfx950 scripts/clockcal.hip -o /tmp/clockcal
// Calibration: shader-clock ticks (s_memtime) against the constant 100 MHz counter (s_memrealtime) inside short kernels.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
__global__ void spin(unsigned long long *out, unsigned long long ticks)
{
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long t = t0;
    while (t - t0 < ticks) { __builtin_amdgcn_s_sleep(1); t = __builtin_amdgcn_s_memtime(); }
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t - t0; out[blockIdx.x * 2 + 1] = r1 - r0; }
}
__global__ void empty(int *p) { if (p && threadIdx.x == 12345) *p = 0; }
int main()
{
    unsigned long long *d; (void)hipMalloc(&d, 4096 * 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int grid : {1, 256, 1024}) for (unsigned long long ticks : {20000ull, 200000ull}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipExtLaunchKernelGGL(spin, dim3(grid), dim3(256), 0, 0, e0, e1, 0, d, ticks);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(grid * 2); hipMemcpy(h.data(), d, grid * 16, hipMemcpyDeviceToHost);
            double st = 0, rt = 0; for (int i = 0; i < grid; ++i) { st += h[2 * i]; rt += h[2 * i + 1]; }
            printf("grid %4d spin %6llu ticks: kernel %.2f us, per block: %.0f ticks in %.0f refclk -> %.1f MHz if refclk = 100 MHz\n",
                   grid, ticks, ms * 1e3, st / grid, rt / grid, st / rt * 100.0);
        }
    }
    // empty-kernel durations by grid size (dispatch cost): workgroups of 256 threads with 38 KB of LDS like nsd_step
    for (int grid : {1, 16, 256, 1024, 4096}) for (int lds : {0, 38912}) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipExtLaunchKernelGGL(empty, dim3(grid), dim3(256), lds, 0, e0, e1, 0, (int *)nullptr);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("empty kernel grid %4d x 256 threads, %5d B LDS: %.2f us\n", grid, lds, best * 1e3);
    }
    return 0;
}
