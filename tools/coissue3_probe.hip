// Round 4, second question: do an MFMA-only wave and a VALU-only wave that share ONE SIMD run side by side?
// (coissue2_probe.hip puts chain and epilogue into the same wave: every wave alternates, the hardware may or may not overlap wave A's
// chain with wave B's epilogue.  Here the roles are split by wave, so instruction order inside a wave cannot matter.)
// A 512-thread workgroup puts two waves on each SIMD (waves w and w + 4); 1024 threads put four (w, w + 4, w + 8, w + 12).
//   roles  M : NM dependent-by-three v_mfma_i32_16x16x64_i8 per iteration, nothing else
//          V : NV vector instructions of one kind per iteration (independent, 8 registers round robin), nothing else
// Cases per kind: every wave M ("M only"), every wave V ("V only"), waves 0-3 (+ 8-11) M and 4-7 (+ 12-15) V ("split").
// If the pipes are independent:  split ~ max(M only, V only) / 1  (each role has half the waves); if one issue port: ~ (M + V) / 2 ... see
// the printed model columns.  Reported in shader cycles per iteration per SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/coissue3_probe.hip -o tools/coissue3_probe && tools/coissue3_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));

#define MFMA16A(acc) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(A), "v"(B));
enum { K_FMA, K_PKFMA, K_CVT, K_MED3, K_ADD, K_XOR, K_PERM, NKIND };
static const char *kname[] = {"v_fma_f32", "v_pk_fma_f32", "v_cvt_pk_u8_f32", "v_med3_f32", "v_add_f32", "v_xor_b32", "v_perm_b32"};

template <int KIND>
__device__ __forceinline__ void valu8(float (&x)[8], v2f (&p)[4], unsigned (&u)[8], float a, float b) {
    const v2f a2 = {a, a}, b2 = {b, b};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if constexpr (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        if constexpr (KIND == K_PKFMA) { if (i < 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(a2), "v"(b2)); }      // 4 packed = 8 values
        if constexpr (KIND == K_CVT) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(u[i]) : "v"(x[i]));
        if constexpr (KIND == K_MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        if constexpr (KIND == K_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
        if constexpr (KIND == K_XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if constexpr (KIND == K_PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
    }
}

// role_mask bit (wave >> 2) & 3 ... simpler: mode 0 = all M, 1 = all V, 2 = waves with (wave >> 2) even are M, odd are V
template <int KIND>
__global__ __launch_bounds__(1024) void k(unsigned *out, int iters, int mode, float a, float b, unsigned long long *clk) {
    const unsigned long long t0c = __builtin_amdgcn_s_memtime(), t0r = __builtin_amdgcn_s_memrealtime();
    const int wave = threadIdx.x >> 6;
    const bool roleM = mode == 0 || (mode == 2 && ((wave >> 2) & 1) == 0);
    const v4i A = {(int)threadIdx.x, 2, 3, 4}, B = {5, (int)threadIdx.x, 7, 8};
    unsigned keep = 0;
    if (roleM) {
        v4i acc = {0, 0, 0, 0};
        for (int i = 0; i < iters; ++i) { MFMA16A(acc) MFMA16A(acc) MFMA16A(acc) }
        keep = (unsigned)acc[0];
    } else {
        float x[8]; v2f p[4]; unsigned u[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { x[i] = (float)(threadIdx.x + i); u[i] = threadIdx.x * 7 + i; }
#pragma unroll
        for (int i = 0; i < 4; ++i) p[i] = (v2f){x[2 * i], x[2 * i + 1]};
        for (int i = 0; i < iters; ++i) valu8<KIND>(x, p, u, a, b);
#pragma unroll
        for (int i = 0; i < 8; ++i) keep ^= __builtin_bit_cast(unsigned, x[i]) ^ u[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) keep ^= __builtin_bit_cast(unsigned, p[i][0]) ^ __builtin_bit_cast(unsigned, p[i][1]);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = keep;
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0c;
        clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - t0r;
    }
}

template <int KIND>
static double run(unsigned *d, unsigned long long *clk, int threads, int mode) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        k<KIND><<<256, threads>>>(d, 400, mode, 1.0001f, 0.5f, clk);
        (void)hipEventRecord(e0);
        k<KIND><<<256, threads>>>(d, iters, mode, 1.0001f, 0.5f, clk);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2];
        (void)hipMemcpy(h, clk + 2 * 100, sizeof(h), hipMemcpyDeviceToHost);
        const double ghz = (double)h[0] / (double)h[1] * 0.1;
        best = std::min(best, ms * 1e6 / iters * ghz);      // cycles per iteration of the slowest wave
    }
    return best;
}
template <int KIND>
static void report(unsigned *d, unsigned long long *clk) {
    for (int threads : {512, 1024}) {
        const int wps = threads / 256;
        const double m = run<KIND>(d, clk, threads, 0), v = run<KIND>(d, clk, threads, 1), s = run<KIND>(d, clk, threads, 2);
        // all-M: wps waves x 3 MFMA per iteration per SIMD take m cycles; split: wps/2 waves of each role.
        // independent pipes -> max(m, v) / 2 ... each role's own time with half the waves; one port -> (m + v) / 2
        printf("%-16s %d waves/SIMD: all-M %7.1f  all-V %7.1f  split %7.1f   (independent pipes: %6.1f, one issue port: %6.1f)\n", kname[KIND], wps, m, v, s,
               std::max(m, v) / 2, (m + v) / 2);
    }
}
int main() {
    unsigned *d; (void)hipMalloc(&d, 256 * 1024 * 4);
    unsigned long long *clk; (void)hipMalloc(&clk, 2048 * 2 * sizeof(unsigned long long));
    printf("cycles per loop iteration (M: 3 dependent v_mfma_i32_16x16x64_i8; V: 8 values through one instruction kind), slowest wave\n");
    report<K_FMA>(d, clk); report<K_PKFMA>(d, clk); report<K_CVT>(d, clk); report<K_MED3>(d, clk); report<K_ADD>(d, clk); report<K_XOR>(d, clk); report<K_PERM>(d, clk);
    return 0;
}
