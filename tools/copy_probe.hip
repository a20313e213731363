// Microbenchmark: how fast can a 33 MB read + 33 MB write pass go on this chip when the
// buffers cycle through a ~150 MB working set (the per-layer activation chain)?  Decides whether
// the per-layer kernels are bounded by HBM or can live in the 256 MiB Infinity Cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void copy16(const int4* __restrict__ in, int4* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}
int main() {
    const size_t bytes = (size_t)1080 * 1920 * 16, n = bytes / 16;
    for (int nbuf : {2, 5, 12}) {
        std::vector<int4*> b(nbuf);
        for (auto& p : b) { hipMalloc(&p, bytes); hipMemset(p, 1, bytes); }
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int grid : {2048, 8192, 32400}) {
            for (int w = 0; w < 3; ++w) for (int k = 0; k < nbuf; ++k) copy16<<<grid, 256>>>(b[k], b[(k + 1) % nbuf], n);
            hipEventRecord(e0);
            const int reps = 20;
            for (int r = 0; r < reps; ++r) for (int k = 0; k < nbuf; ++k) copy16<<<grid, 256>>>(b[k], b[(k + 1) % nbuf], n);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double us = ms * 1e3 / (reps * nbuf);
            printf("nbuf=%2d (%4.0f MB set) grid=%5d: %.2f us per 33MB->33MB pass = %.2f TB/s\n", nbuf, nbuf * bytes / 1e6, grid, us, 2 * bytes / us / 1e6);
        }
        for (auto& p : b) hipFree(p);
    }
    return 0;
}
