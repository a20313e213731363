// How many cycles does a wave64 VALU instruction cost on gfx950 as a function of waves per SIMD and of
// the opcode class?  (v_fma_f32, v_pk_fma_f32, v_med3_f32, v_perm_b32, mixed with MFMA 16x16x64 i8)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ void k(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    v2f p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
    v4i acc = {0, 0, 0, 0}, A = {1, 2, 3, 4}, B = {(int)threadIdx.x, 5, 6, 7};
    const v2f a2 = {a, a}, b2 = {b, b};
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {        // 8 independent v_fma_f32
            x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
            x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
        } else if (KIND == 1) { // 4 independent v_pk_fma_f32 (+4 more)
            p0 = __builtin_elementwise_fma(p0, a2, b2); p1 = __builtin_elementwise_fma(p1, a2, b2); p2 = __builtin_elementwise_fma(p2, a2, b2); p3 = __builtin_elementwise_fma(p3, a2, b2);
            p0 = __builtin_elementwise_fma(p0, b2, a2); p1 = __builtin_elementwise_fma(p1, b2, a2); p2 = __builtin_elementwise_fma(p2, b2, a2); p3 = __builtin_elementwise_fma(p3, b2, a2);
        } else if (KIND == 2) { // 8 v_med3_f32
            x0 = __builtin_amdgcn_fmed3f(x0, a, x1); x1 = __builtin_amdgcn_fmed3f(x1, b, x2); x2 = __builtin_amdgcn_fmed3f(x2, a, x3); x3 = __builtin_amdgcn_fmed3f(x3, b, x4);
            x4 = __builtin_amdgcn_fmed3f(x4, a, x5); x5 = __builtin_amdgcn_fmed3f(x5, b, x6); x6 = __builtin_amdgcn_fmed3f(x6, a, x7); x7 = __builtin_amdgcn_fmed3f(x7, b, x0);
        } else {                // 3 MFMA + 8 fma (the hidden-layer mix)
            acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, acc, 0, 0, 0);
            x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b);
            acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, acc, 0, 0, 0);
            x3 = __builtin_fmaf(x3, a, b); x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b);
            acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, acc, 0, 0, 0);
            x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1] + acc[0];
}
template <int KIND>
static void run(const char* name, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps : {1, 2, 4, 8}) {                 // waves per SIMD: blocks of 256 threads (4 waves = 1 per SIMD)
        dim3 grid(256 * wps);
        k<KIND><<<grid, 256>>>(d, 100, 1.0001f, 0.5f);
        hipEventRecord(e0);
        k<KIND><<<grid, 256>>>(d, iters, 1.0001f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)iters * (KIND == 3 ? 11 : 8) * wps;
        printf("%-22s waves/SIMD %d: %.2f ns per wave-instruction per SIMD (= %.2f cycles @2.4GHz)\n", name, wps, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    }
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_fma_f32", d); run<1>("v_pk_fma_f32", d); run<2>("v_med3_f32", d); run<3>("3 mfma + 8 fma", d);
    return 0;
}
