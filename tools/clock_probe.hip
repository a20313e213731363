// What the two in-kernel clocks of gfx950 count (tools/stamps.py, tools/trio_stamps.py read them): a one-wave kernel spins until
// s_memtime has advanced by N, s_memrealtime is read at both ends, HIP events time the launch.  Prints both rates in MHz.
//   hipcc --offload-arch=gfx950 -O2 tools/clock_probe.hip -o tools/clock_probe && tools/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void spin(unsigned long long n, unsigned long long *out) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    unsigned long long c = c0;
    while (c - c0 < n) c = __builtin_amdgcn_s_memtime();
    out[0] = __builtin_amdgcn_s_memrealtime() - r0;
    out[1] = c - c0;
}
int main(int argc, char **argv) {
    int wall = 0, clk = 0;
    (void)hipDeviceGetAttribute(&wall, hipDeviceAttributeWallClockRate, 0);
    (void)hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("hipDeviceAttributeWallClockRate %d kHz, hipDeviceAttributeClockRate %d kHz\n", wall, clk);
    unsigned long long *d, h[2];
    (void)hipMalloc((void **)&d, 16);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (unsigned long long n : {1000000ULL, 10000000ULL, 100000000ULL}) {
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, n, d);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            printf("spin %llu s_memtime ticks: %.3f ms by events -> s_memtime %.1f MHz, s_memrealtime %.1f MHz\n", (unsigned long long)h[1], ms,
                   h[1] / (ms * 1e3), h[0] / (ms * 1e3));
        }
    }
    // clock_probe <seconds>: keep sampling (one line per ~50 ms spin) -- run it beside another process to read the shader clock under
    // that process's load (the spin is one wave on one SIMD)
    if (argc > 1) {
        const double secs = atof(argv[1]);
        hipEvent_t s0;
        (void)hipEventCreate(&s0);
        (void)hipEventRecord(s0, 0);
        for (int k = 0;; ++k) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, 100000000ULL, d);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0, t = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            (void)hipEventElapsedTime(&t, s0, e1);
            (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            printf("t %.2f s: s_memtime %.0f MHz (by s_memrealtime: %.0f MHz)\n", t / 1e3, h[1] / (ms * 1e3), h[1] / (h[0] / 100.0));
            fflush(stdout);
            if (t / 1e3 > secs) break;
        }
    }
    return 0;
}
