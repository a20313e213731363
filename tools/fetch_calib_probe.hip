// Calibrates rocprofv3's FETCH_SIZE for the access widths the kernels use (MI355X_MICROARCH.md: exactly 1/2 of the bytes for 16-B/lane
// streaming reads, other widths uncalibrated): a 256 MiB buffer (larger than the Infinity Cache is not possible to guarantee, so the buffer
// is written by the host copy first and each kernel reads it ONCE) read with 4 B per lane and with 16 B per lane.
//   hipcc --offload-arch=gfx950 -O2 tools/fetch_calib_probe.hip -o tools/fetch_calib_probe
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- tools/fetch_calib_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void read4(const unsigned *p, size_t n, unsigned *out) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void read16(const uint4 *p, size_t n, unsigned *out) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    const size_t bytes = (size_t)1 << 30;      // 1 GiB: four times the Infinity Cache
    unsigned *d, *o;
    (void)hipMalloc(&d, bytes); (void)hipMalloc(&o, 64);
    (void)hipMemset(d, 1, bytes);
    (void)hipDeviceSynchronize();
    read4<<<4096, 256>>>(d, bytes / 4, o);
    read16<<<4096, 256>>>((const uint4 *)d, bytes / 16, o);
    (void)hipDeviceSynchronize();
    printf("each kernel read %zu bytes once\n", bytes);
    return 0;
}
