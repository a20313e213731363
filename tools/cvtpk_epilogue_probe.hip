// Exhaustive check of the "convert-pack" requant tail used when the target zero point is -128:
//     reference   q  = clamp(rint(y), -128, 127),  y = fl(t + (-128))        (myQL/quan_func.py:280, 604)
//     candidate   q' = v_cvt_pk_u8_f32(y + 128) ^ 0x80                       (one add, one convert+pack, one xor per word)
// for EVERY fp32 t, and with a ReLU lower bound of -128.  y + 128 is exact whenever the result is inside (0, 255.5)
// (Sterbenz for y in [-256,-64], y = t - 128 exactly for t in [64, 256)), and both saturate alike outside.
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/cvtpk_epilogue_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void probe(unsigned long long *bad, unsigned *ex) {
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long nbad = 0;
    for (unsigned long long i = tid; i < (1ull << 32); i += (unsigned long long)gridDim.x * blockDim.x) {
        const float t = __builtin_bit_cast(float, (unsigned)i);
        if (t != t) continue;
        const float y = __fadd_rn(t, -128.f);
        const float r = rintf(y);
        const int want = r < -128.f ? -128 : (r > 127.f ? 127 : (int)r);
        const unsigned u = __builtin_amdgcn_cvt_pk_u8_f32(__fadd_rn(y, 128.f), 0, 0u) & 0xffu;
        const int got = (int)(signed char)(u ^ 0x80u);
        if (got != want) { if (nbad == 0) { unsigned k = atomicAdd(&ex[0], 1u); if (k < 8) { ex[1 + 3 * k] = (unsigned)i; ex[2 + 3 * k] = (unsigned)got; ex[3 + 3 * k] = (unsigned)want; } } ++nbad; }
    }
    if (nbad) atomicAdd(bad, nbad);
}
int main() {
    unsigned long long *bad; unsigned *ex;
    (void)hipMalloc(&bad, 8); (void)hipMalloc(&ex, 4 * 64);
    (void)hipMemset(bad, 0, 8); (void)hipMemset(ex, 0, 4 * 64);
    probe<<<4096, 256>>>(bad, ex);
    unsigned long long hb; unsigned he[64];
    (void)hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(he, ex, 4 * 64, hipMemcpyDeviceToHost);
    printf("mismatches over all fp32 t: %llu\n", hb);
    for (unsigned k = 0; k < he[0] && k < 8; ++k) printf("   t=%.9g (0x%08x) got %d want %d\n", __builtin_bit_cast(float, he[1 + 3 * k]), he[1 + 3 * k], (int)he[2 + 3 * k], (int)he[3 + 3 * k]);
    return hb != 0;
}
