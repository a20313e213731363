// Probe: what does a typed buffer store (buffer_store_format_xyzw, 8_8_8_8) do to fp32 / int32 inputs on gfx950?
// Candidate for clamp + round + pack of the requant epilogue in the memory pipeline.  Checks, exhaustively over
// every fp32 with |x| < 1024, whether SSCALED == clamp(rint(x), -128, 127); prints the first mismatches.
//   hipcc --offload-arch=gfx950 -O2 tools/fmtstore_probe.hip -o /tmp/fmtstore_probe && /tmp/fmtstore_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
template <int NFMT>
__global__ void store_f(char *out, unsigned base) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;          // 4 consecutive bit patterns per thread
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, 1 << 22, 0xFAC | (NFMT << 12) | (10 << 15));
    const unsigned b = base + 4 * t;
    v4f v = {__builtin_bit_cast(float, b), __builtin_bit_cast(float, b + 1), __builtin_bit_cast(float, b + 2), __builtin_bit_cast(float, b + 3)};
    int off = t * 4;
    asm volatile("buffer_store_format_xyzw %0, %1, %2, 0 offen" : : "v"(v), "v"(off), "s"(r) : "memory");
}
__global__ void check_f(const signed char *out, unsigned base, unsigned long long *bad, unsigned *ex) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < 4; ++i) {
        const unsigned b = base + 4 * t + i;
        const float x = __builtin_bit_cast(float, b);
        if (x != x) continue;
        const float rr = rintf(x);
        const int want = rr < -128.f ? -128 : (rr > 127.f ? 127 : (int)rr);
        const int got = out[4 * t + i];
        if (got != want) {
            atomicAdd(bad, 1ull);
            unsigned k = atomicAdd(&ex[0], 1u);
            if (k < 12) { ex[1 + 3 * k] = b; ex[2 + 3 * k] = (unsigned)got; ex[3 + 3 * k] = (unsigned)want; }
        }
    }
}
__global__ void store_i(char *out, int base) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, 1 << 22, 0xFAC | (5 << 12) | (10 << 15));
    v4i v = {base + 4 * t, base + 4 * t + 1, base + 4 * t + 2, base + 4 * t + 3};
    int off = t * 4;
    asm volatile("buffer_store_format_xyzw %0, %1, %2, 0 offen" : : "v"(v), "v"(off), "s"(r) : "memory");
}
template <int NFMT>
static void run_f(const char *name, char *out, unsigned long long *bad, unsigned *ex) {
    hipMemset(bad, 0, 8); hipMemset(ex, 0, 4 * 64);
    // positive and negative halves, exponents up to 2^10
    const unsigned top = 0x44800000u;          // 1024.0f
    for (unsigned sign = 0; sign < 2; ++sign)
        for (unsigned long long b = 0; b < top; b += (1u << 22)) {
            const unsigned base = (unsigned)b | (sign << 31);
            store_f<NFMT><<<(1 << 20) / 256, 256>>>(out, base);
            check_f<<<(1 << 20) / 256, 256>>>((const signed char *)out, base, bad, ex);
        }
    unsigned long long hb; unsigned he[64];
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(he, ex, 4 * 64, hipMemcpyDeviceToHost);
    printf("%s: mismatches vs clamp(rint(x),-128,127) over |x|<1024: %llu\n", name, hb);
    for (unsigned k = 0; k < he[0] && k < 12; ++k) printf("   x=%.9g (0x%08x) got %d want %d\n", __builtin_bit_cast(float, he[1 + 3 * k]), he[1 + 3 * k], (int)he[2 + 3 * k], (int)he[3 + 3 * k]);
}
int main() {
    char *out; unsigned long long *bad; unsigned *ex;
    hipMalloc(&out, 1 << 22); hipMalloc(&bad, 8); hipMalloc(&ex, 4 * 64);
    run_f<3>("SSCALED", out, bad, ex);
    run_f<1>("SNORM(for reference)", out, bad, ex);
    // integer inputs through SINT
    hipMemset(out, 0x55, 1 << 22);
    store_i<<<4, 256>>>(out, -2048);
    signed char h[4096];
    hipMemcpy(h, out, 4096, hipMemcpyDeviceToHost);
    int nb = 0;
    for (int i = 0; i < 4096; ++i) { int v = -2048 + i, want = v < -128 ? -128 : (v > 127 ? 127 : v); if (h[i] != want) { if (nb < 6) printf("   SINT in %d got %d want(clamp) %d\n", v, h[i], want); ++nb; } }
    printf("SINT: %d of 4096 integer inputs differ from clamp8\n", nb);
    return 0;
}
