// Probe: buffer_store_format_d16_xyzw with an 8_8_8_8 SINT descriptor -- does the memory pipeline clamp four packed
// int16 inputs to int8 and pack them?  (candidate for the clamp + pack of the requant epilogue)
//   hipcc --offload-arch=gfx950 -O2 tools/fmtstore_d16_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__global__ void store_d16(char *out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;               // 4 consecutive int16 values per thread
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, 1 << 16, 0xFAC | (5 << 12) | (10 << 15));
    const unsigned a = (unsigned)(4 * t) & 0xffffu, b = (unsigned)(4 * t + 1) & 0xffffu, c = (unsigned)(4 * t + 2) & 0xffffu, d = (unsigned)(4 * t + 3) & 0xffffu;
    v2u v = {a | (b << 16), c | (d << 16)};
    int off = t * 4;
    asm volatile("buffer_store_format_d16_xyzw %0, %1, %2, 0 offen" : : "v"(v), "v"(off), "s"(r) : "memory");
}
__global__ void store_d16_xy(char *out) {                               // 8_8 format, 2 components
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, 1 << 16, 0xFAC | (5 << 12) | (3 << 15));
    const unsigned a = (unsigned)(2 * t) & 0xffffu, b = (unsigned)(2 * t + 1) & 0xffffu;
    unsigned v = a | (b << 16);
    int off = t * 2;
    asm volatile("buffer_store_format_d16_xy %0, %1, %2, 0 offen" : : "v"(v), "v"(off), "s"(r) : "memory");
}
int main() {
    char *out; hipMalloc(&out, 1 << 16);
    static signed char h[1 << 16];
    for (int variant = 0; variant < 2; ++variant) {
        hipMemset(out, 0x55, 1 << 16);
        if (variant == 0) store_d16<<<(1 << 14) / 256, 256>>>(out); else store_d16_xy<<<(1 << 15) / 256, 256>>>(out);
        hipMemcpy(h, out, 1 << 16, hipMemcpyDeviceToHost);
        int nb = 0;
        for (int i = 0; i < (1 << 16); ++i) {
            const int v = (short)(unsigned short)i, want = v < -128 ? -128 : (v > 127 ? 127 : v);
            if (h[i] != want) { if (nb < 8) printf("   in %d got %d want %d\n", v, h[i], want); ++nb; }
        }
        printf("%s SINT: %d of 65536 int16 inputs differ from clamp8\n", variant == 0 ? "d16_xyzw 8_8_8_8" : "d16_xy 8_8", nb);
    }
    return 0;
}
