// What bounds a "3-MFMA chain + requant epilogue" row on gfx950, and what would change it?  Models one row of a fused-trio inner
// phase (sesrq_trio.hip) with operands in registers (no LDS, no memory), everything in asm volatile so nothing is re-ordered:
//   a16     : 3 dependent v_mfma_i32_16x16x64_i8 -> 13-instruction epilogue (2+2 pk_fma, 4 med3, 2 pk_add, 3 perm) on its 4 values
//   a16cvt  : same chain, 11-instruction epilogue (2+2 pk_fma, 2 pk_add, 4 cvt_pk_u8, 1 xor)
//   a16pipe : the chain of row i+1 interleaved with the epilogue of row i (two accumulators)
//   b32     : 6 dependent v_mfma_i32_32x32x32_i8 (2 rows x 32 pixels = 4 rows of a16) -> 4 epilogues
//   b32cvt  : same with the 11-instruction epilogue
//   chain / epi13 : the three MFMAs alone / the 13 instructions alone;  a16indep: three independent MFMAs + the epilogue
// Reported: ns per 16-pixel row per SIMD at 1..4 waves per SIMD (work per row is identical across variants).
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_shape_probe.hip -o /tmp/msp && /tmp/msp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define MFMA16(acc, cin) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %3" : "=v"(acc) : "v"(A), "v"(B), "v"(cin));
#define MFMA16A(acc) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(A), "v"(B));
#define MFMA32(acc, cin) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %3" : "=v"(acc) : "v"(A), "v"(B), "v"(cin));
#define MFMA32A(acc) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(A), "v"(B));
// epilogue pieces on a pair of values (x = two floats in a register pair)
#define PKFMA(p, m, c) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(m), "v"(c));
#define PKADD(p, c) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(c));
#define MED3(x) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(lo), "v"(hi));
#define PERM(d, x, y) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(sel));
#define CVT(d, x, k) asm volatile("v_cvt_pk_u8_f32 %0, %1, " #k ", %0" : "+v"(d) : "v"(x));
#define XOR(d) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(d) : "v"(sel));

// 13 instructions: values in p, q (two pairs) -> packed word w
#define EPI13(p, q, w)                                                                        \
    { PKFMA(p, m2, c2) PKFMA(q, m2, c2) PKFMA(p, s2, z2) PKFMA(q, s2, z2)                       \
      MED3(p[0]) MED3(p[1]) MED3(q[0]) MED3(q[1]) PKADD(p, g2) PKADD(q, g2)                     \
      unsigned t0_, t1_; PERM(t0_, p[1], p[0]) PERM(t1_, q[1], q[0]) PERM(w, t1_, t0_) }
#define EPI11(p, q, w)                                                                        \
    { PKFMA(p, m2, c2) PKFMA(q, m2, c2) PKFMA(p, s2, z2) PKFMA(q, s2, z2) PKADD(p, g2) PKADD(q, g2) \
      CVT(w, p[0], 0) CVT(w, p[1], 1) CVT(w, q[0], 2) CVT(w, q[1], 3) XOR(w) }

template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters, float mf, float sh, unsigned long long *clk) {
    const unsigned long long t0c = __builtin_amdgcn_s_memtime(), t0r = __builtin_amdgcn_s_memrealtime();
    const v4i A = {(int)threadIdx.x, 2, 3, 4}, B = {5, (int)threadIdx.x, 7, 8};
    const v2f m2 = {mf, mf}, c2 = {-mf, -mf}, s2 = {sh, sh}, z2 = {-128.f, -128.f}, g2 = {12582912.f, 12582912.f};
    const float lo = -128.f, hi = 127.f;
    const unsigned sel = 0x0c0c0400u;
    unsigned keep = 0;
    if constexpr (KIND == 0 || KIND == 1) {            // a16 / a16cvt: one row per iteration
        const v4i zero = {0, 0, 0, 0};
        for (int i = 0; i < iters; ++i) {
            v4i acc;
            MFMA16(acc, zero) MFMA16A(acc) MFMA16A(acc)
            v2f p = {__builtin_bit_cast(float, acc[0]), __builtin_bit_cast(float, acc[1])}, q = {__builtin_bit_cast(float, acc[2]), __builtin_bit_cast(float, acc[3])};
            unsigned w = 0;
            if constexpr (KIND == 0) EPI13(p, q, w) else EPI11(p, q, w)
            keep ^= w;
        }
    } else if constexpr (KIND == 2) {                  // a16pipe: chain of the next row between the epilogue pieces of this one
        const v4i zero = {0, 0, 0, 0};
        v4i cur;
        MFMA16(cur, zero) MFMA16A(cur) MFMA16A(cur)
        for (int i = 0; i < iters; ++i) {
            v4i nxt;
            v2f p = {__builtin_bit_cast(float, cur[0]), __builtin_bit_cast(float, cur[1])}, q = {__builtin_bit_cast(float, cur[2]), __builtin_bit_cast(float, cur[3])};
            unsigned w, t0_, t1_;
            MFMA16(nxt, zero)
            PKFMA(p, m2, c2) PKFMA(q, m2, c2) PKFMA(p, s2, z2) PKFMA(q, s2, z2)
            MFMA16A(nxt)
            MED3(p[0]) MED3(p[1]) MED3(q[0]) MED3(q[1])
            MFMA16A(nxt)
            PKADD(p, g2) PKADD(q, g2) PERM(t0_, p[1], p[0]) PERM(t1_, q[1], q[0]) PERM(w, t1_, t0_)
            keep ^= w;
            cur = nxt;
        }
    } else if constexpr (KIND == 5) {                  // the chain alone
        const v4i zero = {0, 0, 0, 0};
        for (int i = 0; i < iters; ++i) {
            v4i acc;
            MFMA16(acc, zero) MFMA16A(acc) MFMA16A(acc)
            keep ^= (unsigned)acc[0];
        }
    } else if constexpr (KIND == 6) {                  // the 13-instruction epilogue alone
        v2f p = {1.f, 2.f}, q = {3.f, 4.f};
        for (int i = 0; i < iters; ++i) {
            unsigned w = 0;
            EPI13(p, q, w)
            keep ^= w;
        }
    } else if constexpr (KIND == 7) {                  // three INDEPENDENT MFMAs + epilogue (is the chain's dependency what costs?)
        const v4i zero = {0, 0, 0, 0};
        for (int i = 0; i < iters; ++i) {
            v4i acc, acc2, acc3;
            MFMA16(acc, zero) MFMA16(acc2, zero) MFMA16(acc3, zero)
            v2f p = {__builtin_bit_cast(float, acc[0]), __builtin_bit_cast(float, acc2[1])}, q = {__builtin_bit_cast(float, acc3[2]), __builtin_bit_cast(float, acc[3])};
            unsigned w = 0;
            EPI13(p, q, w)
            keep ^= w;
        }
    } else if constexpr (KIND == 8) {                  // three dependent sparse MFMAs 16x16x128 (2:4 structured A): cycles per instruction?
        v4i acc = {0, 0, 0, 0};
        const v4i As = {1, 2, 3, 4};
        typedef int v8i __attribute__((ext_vector_type(8)));
        const v8i Bd = {5, (int)threadIdx.x, 7, 8, 9, 10, 11, 12};
        const int idx = 0x44444444;
        for (int i = 0; i < iters; ++i) {
            asm volatile("v_smfmac_i32_16x16x128_i8 %0, %1, %2, %3" : "+v"(acc) : "v"(As), "v"(Bd), "v"(idx));
            asm volatile("v_smfmac_i32_16x16x128_i8 %0, %1, %2, %3" : "+v"(acc) : "v"(As), "v"(Bd), "v"(idx));
            asm volatile("v_smfmac_i32_16x16x128_i8 %0, %1, %2, %3" : "+v"(acc) : "v"(As), "v"(Bd), "v"(idx));
        }
        keep ^= (unsigned)acc[0];
    } else if constexpr (KIND == 9) {                  // three dependent legacy K=32 MFMAs (gfx940 form, 8-byte operands): cycles per instruction?
        v4i acc = {0, 0, 0, 0};
        typedef int v2i __attribute__((ext_vector_type(2)));
        const v2i A2 = {(int)threadIdx.x, 2}, B2 = {5, (int)threadIdx.x};
        for (int i = 0; i < iters; ++i) {
            asm volatile("v_mfma_i32_16x16x32_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(A2), "v"(B2));
            asm volatile("v_mfma_i32_16x16x32_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(A2), "v"(B2));
            asm volatile("v_mfma_i32_16x16x32_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(A2), "v"(B2));
        }
        keep ^= (unsigned)acc[0];
    } else if constexpr (KIND == 10) {                 // a16cvt with the third K-chunk (one tap of 9 = 16 of 64 bytes used) on the K=32 form
        const v4i zero = {0, 0, 0, 0};
        typedef int v2i __attribute__((ext_vector_type(2)));
        const v2i A2 = {(int)threadIdx.x, 2}, B2 = {5, (int)threadIdx.x};
        for (int i = 0; i < iters; ++i) {
            v4i acc;
            MFMA16(acc, zero) MFMA16A(acc)
            asm volatile("v_mfma_i32_16x16x32_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(A2), "v"(B2));
            v2f p = {__builtin_bit_cast(float, acc[0]), __builtin_bit_cast(float, acc[1])}, q = {__builtin_bit_cast(float, acc[2]), __builtin_bit_cast(float, acc[3])};
            unsigned w = 0;
            EPI11(p, q, w)
            keep ^= w;
        }
    } else {                                            // b32 / b32cvt: 2 rows x 32 pixels = 4 a16 rows per iteration
        v16i zero;
        for (int j = 0; j < 16; ++j) zero[j] = 0;
        for (int i = 0; i < iters; i += 4) {
            v16i acc;
            MFMA32(acc, zero) MFMA32A(acc) MFMA32A(acc) MFMA32A(acc) MFMA32A(acc) MFMA32A(acc)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v2f p = {__builtin_bit_cast(float, acc[4 * j]), __builtin_bit_cast(float, acc[4 * j + 1])}, q = {__builtin_bit_cast(float, acc[4 * j + 2]), __builtin_bit_cast(float, acc[4 * j + 3])};
                unsigned w = 0;
                if constexpr (KIND == 3) EPI13(p, q, w) else EPI11(p, q, w)
                keep ^= w;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = keep;
    if (clk && threadIdx.x == 0) {                      // shader clock: s_memtime ticks per 100 MHz s_memrealtime tick (microarch guide, DVFS item 6)
        clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0c;
        clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - t0r;
    }
}
template <int KIND>
static void run(const char *name, unsigned *d) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps : {1, 2, 3, 4}) {
        dim3 grid(256 * wps);
        static unsigned long long *clk = nullptr;
        if (!clk) (void)hipMalloc(&clk, 2048 * 2 * sizeof(unsigned long long));
        k<KIND><<<grid, 256>>>(d, 400, 3.0f, 0.25f, nullptr);
        (void)hipEventRecord(e0);
        k<KIND><<<grid, 256>>>(d, iters, 3.0f, 0.25f, clk);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2];
        (void)hipMemcpy(h, clk + 2 * 100, sizeof(h), hipMemcpyDeviceToHost);
        const double ghz = (double)h[0] / (double)h[1] * 0.1;
        printf("%-10s waves/SIMD %d: %6.1f ns = %6.1f cycles per 16-pixel row per SIMD (shader clock %.2f GHz)\n", name, wps, ms * 1e6 / iters / wps,
               ms * 1e6 / iters / wps * ghz, ghz);
    }
}
int main() {
    unsigned *d; (void)hipMalloc(&d, 256 * 4 * 256 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("a16", d); run<1>("a16cvt", d); run<2>("a16pipe", d); run<3>("b32", d); run<4>("b32cvt", d);
        run<5>("chain", d); run<6>("epi13", d); run<7>("a16indep", d); run<8>("smfmac128", d); run<9>("k32x3", d); run<10>("a16cvt_k32", d);
    }
    return 0;
}
