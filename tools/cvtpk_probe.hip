// Probe: does v_cvt_pk_u8_f32 round to nearest even and saturate to [0,255]?  Exhaustive over all fp32 bit patterns.
//   hipcc --offload-arch=gfx950 -O2 tools/cvtpk_probe.hip -o .scratch/cvtpk_probe && .scratch/cvtpk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void probe(unsigned long long *bad, unsigned *ex, int sel) {
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long nbad = 0;
    for (unsigned long long i = tid; i < (1ull << 32); i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned bits = (unsigned)i;
        const float x = __builtin_bit_cast(float, bits);
        if (x != x) continue;
        const unsigned got = (__builtin_amdgcn_cvt_pk_u8_f32(x, sel, 0xA5A5A5A5u) >> (8 * sel)) & 0xffu;
        const float r = rintf(x);
        const unsigned want = r <= 0.f ? 0u : (r >= 255.f ? 255u : (unsigned)r);
        if (got != want) { if (nbad == 0 && atomicAdd(&ex[0], 1u) < 16) { unsigned k = atomicAdd(&ex[1], 1u); ex[2 + 3 * k] = bits; ex[3 + 3 * k] = got; ex[4 + 3 * k] = want; } ++nbad; }
    }
    atomicAdd(bad, nbad);
}
int main() {
    unsigned long long *bad; unsigned *ex;
    hipMalloc(&bad, 8); hipMalloc(&ex, 4 * 64);
    for (int sel = 0; sel < 4; ++sel) {
        hipMemset(bad, 0, 8); hipMemset(ex, 0, 4 * 64);
        probe<<<4096, 256>>>(bad, ex, sel);
        unsigned long long hb; unsigned he[64];
        hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(he, ex, 4 * 64, hipMemcpyDeviceToHost);
        printf("sel %d mismatches vs clamp(rint(x),0,255): %llu\n", sel, hb);
        for (unsigned k = 0; k < he[1] && k < 8; ++k) { float f = __builtin_bit_cast(float, he[2 + 3 * k]); printf("   x=%.9g (0x%08x) got %u want %u\n", f, he[2 + 3 * k], he[3 + 3 * k], he[4 + 3 * k]); }
    }
    return 0;
}
