// Round 4: does an UNPACKED fp32 requant epilogue hide behind the MFMA chain of another wave (or of the same wave) on gfx950?
// Round 3's mfma_shape_probe.hip built every epilogue from v_pk_fma_f32 / v_pk_add_f32 -- the one instruction class the microarch
// guide and tools/coissue_probe.hip (round 1) say does NOT share a SIMD with the matrix pipe -- and concluded "MFMA or VALU, never
// both".  This probe repeats the "3 dependent v_mfma_i32_16x16x64_i8 + epilogue on the 4 results" row with each epilogue in its
// packed and its unpacked form, alone and beside the chain, in program order (SER: chain, then epilogue) and interleaved (PIPE: the
// chain of row i+1 between the epilogue pieces of row i), everything in asm volatile so that nothing is re-ordered or re-packed.
//   epilogues (per 4 values):
//     pk7   2 v_pk_fma + 4 v_cvt_pk_u8 + 1 v_xor                (the one-fma requant as round 3 shipped it)
//     u9    4 v_fma    + 4 v_cvt_pk_u8 + 1 v_xor                (the same arithmetic, unpacked)
//     pk11  4 v_pk_fma + 2 v_pk_add + 4 cvt + 1 xor             (two-fma requant, cvt_pk_u8 form)
//     u17   8 v_fma    + 4 v_add    + 4 cvt + 1 xor
//     pk13  4 v_pk_fma + 4 v_med3 + 2 v_pk_add + 3 v_perm       (general zero points)
//     u19   8 v_fma    + 4 v_med3 + 4 v_add    + 3 v_perm
//     cvt5  4 v_cvt_pk_u8 + 1 v_xor          med5  4 v_med3_f32 + 1 v_xor         fma4  4 v_fma        pkfma2  2 v_pk_fma
//     fmac4 4 v_fmac_f32 (VOP2)              add8  8 v_add_f32
// Reported: shader cycles per 16-pixel row per SIMD at 1..4 waves per SIMD.  "hidden" = 1 - (with - chain) / alone.
//   hipcc --offload-arch=gfx950 -O2 tools/coissue2_probe.hip -o tools/coissue2_probe && tools/coissue2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));

#define MFMA16(acc, cin) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %3" : "=v"(acc) : "v"(A), "v"(B), "v"(cin));
#define MFMA16A(acc) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(A), "v"(B));
#define PKFMA(p, m, c) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(m), "v"(c));
#define PKADD(p, c) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(c));
#define FMA(x, m, c) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c));
#define FMAC(x, m, c) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(m), "v"(c));
#define ADD(x, c) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(c));
#define MED3(x) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(lo), "v"(hi));
#define PERM(d, x, y) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(sel));
#define CVT(d, x, k) asm volatile("v_cvt_pk_u8_f32 %0, %1, " #k ", %0" : "+v"(d) : "v"(x));
#define XOR(d) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(d) : "v"(sel));

enum { PK7, U9, PK11, U17, PK13, U19, CVT5, MED5, FMA4, PKFMA2, FMAC4, ADD8, NONE, NEPI };
static const char *epi_name[] = {"pk7", "u9", "pk11", "u17", "pk13", "u19", "cvt5", "med5", "fma4", "pkfma2", "fmac4", "add8", "none"};

// one epilogue on x0..x3 (plain floats; the packed forms view them as the pairs (x0,x1), (x2,x3)) -> word w.
// G0 / G1 / G2: three groups of roughly a third of the instructions each (PIPE puts one MFMA of the next row in front of each group)
#define CONSTS                                                                                       \
    const float m = mf, c = -mf, s = sh, z = -128.f, mg = 12582912.f, lo = -128.f, hi = 127.f;       \
    const v2f m2 = {m, m}, c2 = {c, c}, s2 = {s, s}, z2 = {z, z}, g2 = {mg, mg};                      \
    const unsigned sel = 0x0c0c0400u;

template <int E, int G>
__device__ __forceinline__ void epi_group(float &x0, float &x1, float &x2, float &x3, unsigned &w, float mf, float sh) {
    CONSTS
    v2f p = {x0, x1}, q = {x2, x3};
    unsigned t0 = 0, t1 = 0;
    if constexpr (E == PK7) {
        if constexpr (G == 0) { PKFMA(p, m2, c2) PKFMA(q, m2, c2) }
        if constexpr (G == 1) { CVT(w, p[0], 0) CVT(w, p[1], 1) }
        if constexpr (G == 2) { CVT(w, q[0], 2) CVT(w, q[1], 3) XOR(w) }
    } else if constexpr (E == U9) {
        if constexpr (G == 0) { FMA(x0, m, c) FMA(x1, m, c) FMA(x2, m, c) }
        if constexpr (G == 1) { FMA(x3, m, c) CVT(w, x0, 0) CVT(w, x1, 1) }
        if constexpr (G == 2) { CVT(w, x2, 2) CVT(w, x3, 3) XOR(w) }
    } else if constexpr (E == PK11) {
        if constexpr (G == 0) { PKFMA(p, m2, c2) PKFMA(q, m2, c2) PKFMA(p, s2, z2) PKFMA(q, s2, z2) }
        if constexpr (G == 1) { PKADD(p, g2) PKADD(q, g2) CVT(w, p[0], 0) }
        if constexpr (G == 2) { CVT(w, p[1], 1) CVT(w, q[0], 2) CVT(w, q[1], 3) XOR(w) }
    } else if constexpr (E == U17) {
        if constexpr (G == 0) { FMA(x0, m, c) FMA(x1, m, c) FMA(x2, m, c) FMA(x3, m, c) FMA(x0, s, z) FMA(x1, s, z) }
        if constexpr (G == 1) { FMA(x2, s, z) FMA(x3, s, z) ADD(x0, mg) ADD(x1, mg) ADD(x2, mg) ADD(x3, mg) }
        if constexpr (G == 2) { CVT(w, x0, 0) CVT(w, x1, 1) CVT(w, x2, 2) CVT(w, x3, 3) XOR(w) }
    } else if constexpr (E == PK13) {
        if constexpr (G == 0) { PKFMA(p, m2, c2) PKFMA(q, m2, c2) PKFMA(p, s2, z2) PKFMA(q, s2, z2) }
        if constexpr (G == 1) { MED3(p[0]) MED3(p[1]) MED3(q[0]) MED3(q[1]) }
        if constexpr (G == 2) { PKADD(p, g2) PKADD(q, g2) PERM(t0, p[1], p[0]) PERM(t1, q[1], q[0]) PERM(w, t1, t0) }
    } else if constexpr (E == U19) {
        if constexpr (G == 0) { FMA(x0, m, c) FMA(x1, m, c) FMA(x2, m, c) FMA(x3, m, c) FMA(x0, s, z) FMA(x1, s, z) }
        if constexpr (G == 1) { FMA(x2, s, z) FMA(x3, s, z) MED3(x0) MED3(x1) MED3(x2) MED3(x3) }
        if constexpr (G == 2) { ADD(x0, mg) ADD(x1, mg) ADD(x2, mg) ADD(x3, mg) PERM(t0, x1, x0) PERM(t1, x3, x2) PERM(w, t1, t0) }
    } else if constexpr (E == CVT5) {
        if constexpr (G == 0) { CVT(w, x0, 0) CVT(w, x1, 1) }
        if constexpr (G == 1) { CVT(w, x2, 2) CVT(w, x3, 3) }
        if constexpr (G == 2) { XOR(w) }
    } else if constexpr (E == MED5) {
        if constexpr (G == 0) { MED3(x0) MED3(x1) }
        if constexpr (G == 1) { MED3(x2) MED3(x3) }
        if constexpr (G == 2) { w = __builtin_bit_cast(unsigned, x0); XOR(w) }
    } else if constexpr (E == FMA4) {
        if constexpr (G == 0) { FMA(x0, m, c) FMA(x1, m, c) }
        if constexpr (G == 1) { FMA(x2, m, c) }
        if constexpr (G == 2) { FMA(x3, m, c) w = __builtin_bit_cast(unsigned, x3); }
    } else if constexpr (E == PKFMA2) {
        if constexpr (G == 0) { PKFMA(p, m2, c2) }
        if constexpr (G == 1) { PKFMA(q, m2, c2) }
        if constexpr (G == 2) { w = __builtin_bit_cast(unsigned, q[1]); }
    } else if constexpr (E == FMAC4) {
        if constexpr (G == 0) { FMAC(x0, m, c) FMAC(x1, m, c) }
        if constexpr (G == 1) { FMAC(x2, m, c) }
        if constexpr (G == 2) { FMAC(x3, m, c) w = __builtin_bit_cast(unsigned, x3); }
    } else if constexpr (E == ADD8) {
        if constexpr (G == 0) { ADD(x0, mg) ADD(x1, mg) ADD(x2, mg) }
        if constexpr (G == 1) { ADD(x3, mg) ADD(x0, z) ADD(x1, z) }
        if constexpr (G == 2) { ADD(x2, z) ADD(x3, z) w = __builtin_bit_cast(unsigned, x3); }
    }
    if constexpr (E == PK7 || E == PK11 || E == PK13 || E == PKFMA2) { x0 = p[0]; x1 = p[1]; x2 = q[0]; x3 = q[1]; }
}

// SHAPE 0: epilogue alone;  1: chain, then epilogue (program order);  2: chain of the next row interleaved with the epilogue groups
template <int E, int SHAPE>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters, float mf, float sh, unsigned long long *clk) {
    const unsigned long long t0c = __builtin_amdgcn_s_memtime(), t0r = __builtin_amdgcn_s_memrealtime();
    const v4i A = {(int)threadIdx.x, 2, 3, 4}, B = {5, (int)threadIdx.x, 7, 8};
    const v4i zero = {0, 0, 0, 0};
    unsigned keep = 0;
    if constexpr (SHAPE == 0) {
        float x0 = 1.f, x1 = 2.f, x2 = 3.f, x3 = 4.f;
        for (int i = 0; i < iters; ++i) {
            unsigned w = 0;
            epi_group<E, 0>(x0, x1, x2, x3, w, mf, sh); epi_group<E, 1>(x0, x1, x2, x3, w, mf, sh); epi_group<E, 2>(x0, x1, x2, x3, w, mf, sh);
            keep ^= w;
        }
    } else if constexpr (SHAPE == 1) {
        for (int i = 0; i < iters; ++i) {
            v4i acc;
            MFMA16(acc, zero) MFMA16A(acc) MFMA16A(acc)
            float x0 = __builtin_bit_cast(float, acc[0]), x1 = __builtin_bit_cast(float, acc[1]), x2 = __builtin_bit_cast(float, acc[2]), x3 = __builtin_bit_cast(float, acc[3]);
            unsigned w = 0;
            epi_group<E, 0>(x0, x1, x2, x3, w, mf, sh); epi_group<E, 1>(x0, x1, x2, x3, w, mf, sh); epi_group<E, 2>(x0, x1, x2, x3, w, mf, sh);
            keep ^= w;
        }
    } else {
        v4i cur;
        MFMA16(cur, zero) MFMA16A(cur) MFMA16A(cur)
        for (int i = 0; i < iters; ++i) {
            v4i nxt;
            float x0 = __builtin_bit_cast(float, cur[0]), x1 = __builtin_bit_cast(float, cur[1]), x2 = __builtin_bit_cast(float, cur[2]), x3 = __builtin_bit_cast(float, cur[3]);
            unsigned w = 0;
            MFMA16(nxt, zero)
            epi_group<E, 0>(x0, x1, x2, x3, w, mf, sh);
            MFMA16A(nxt)
            epi_group<E, 1>(x0, x1, x2, x3, w, mf, sh);
            MFMA16A(nxt)
            epi_group<E, 2>(x0, x1, x2, x3, w, mf, sh);
            keep ^= w;
            cur = nxt;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = keep;
    if (clk && threadIdx.x == 0) {
        clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0c;
        clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - t0r;
    }
}

static double res[NEPI][3][5];      // cycles per row per SIMD [epilogue][shape][waves per SIMD]
template <int E, int SHAPE>
static void run(unsigned *d) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    static unsigned long long *clk = nullptr;
    if (!clk) (void)hipMalloc(&clk, 2048 * 2 * sizeof(unsigned long long));
    for (int wps = 1; wps <= 4; ++wps) {
        dim3 grid(256 * wps);
        double best = 1e30;
        for (int rep = 0; rep < 3; ++rep) {
            k<E, SHAPE><<<grid, 256>>>(d, 400, 3.0f, 0.25f, nullptr);
            (void)hipEventRecord(e0);
            k<E, SHAPE><<<grid, 256>>>(d, iters, 3.0f, 0.25f, clk);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[2];
            (void)hipMemcpy(h, clk + 2 * 100, sizeof(h), hipMemcpyDeviceToHost);
            const double ghz = (double)h[0] / (double)h[1] * 0.1;
            best = std::min(best, ms * 1e6 / iters / wps * ghz);
        }
        res[E][SHAPE][wps] = best;
    }
}
template <int E>
static void run_all(unsigned *d) { run<E, 0>(d); run<E, 1>(d); run<E, 2>(d); }

int main() {
    unsigned *d; (void)hipMalloc(&d, 256 * 4 * 256 * 4);
    run_all<PK7>(d); run_all<U9>(d); run_all<PK11>(d); run_all<U17>(d); run_all<PK13>(d); run_all<U19>(d);
    run_all<CVT5>(d); run_all<MED5>(d); run_all<FMA4>(d); run_all<PKFMA2>(d); run_all<FMAC4>(d); run_all<ADD8>(d);
    run<NONE, 1>(d);
    printf("shader cycles per 16-pixel row per SIMD (3 dependent v_mfma_i32_16x16x64_i8 + epilogue on their 4 values), min of 3 runs\n");
    for (int wps = 1; wps <= 4; ++wps) {
        const double chain = res[NONE][1][wps];
        printf("\n-- %d wave(s) per SIMD: chain alone %.1f\n", wps, chain);
        printf("%-8s %10s %12s %8s %12s %8s\n", "epilogue", "alone", "chain;epi", "hidden", "interleaved", "hidden");
        for (int e = 0; e < NONE; ++e) {
            const double al = res[e][0][wps], ser = res[e][1][wps], pipe = res[e][2][wps];
            printf("%-8s %10.1f %12.1f %7.0f%% %12.1f %7.0f%%\n", epi_name[e], al, ser, 100.0 * (1.0 - (ser - chain) / al), pipe, 100.0 * (1.0 - (pipe - chain) / al));
        }
    }
    return 0;
}
