// Do VALU instructions issue in the shadow of an MFMA on gfx950?  (hipcc unpacks v_pk_*_f32 next to MFMAs -- is that a win?)
// Per loop iteration: 4 independent v_mfma_i32_16x16x64_i8 (4 accumulators) + one of: nothing / 8 v_fma_f32 /
// 4 v_pk_fma_f32 (= the same 8 fmas) / 8 v_med3_f32, all written in inline asm so that nothing is rewritten.
//   hipcc --offload-arch=gfx950 -O2 tools/coissue_probe.hip -o /tmp/cp && /tmp/cp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
#define MFMA(acc) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(A), "v"(B));
#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define PKFMA(p) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(a2), "v"(b2));
#define MED3(x) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
template <int KIND, bool WITH_MFMA>
__global__ void k(float *out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    v2f p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
    v4i c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, A = {1, 2, 3, 4}, B = {(int)threadIdx.x, 5, 6, 7};
    const v2f a2 = {a, a}, b2 = {b, b};
    for (int i = 0; i < iters; ++i) {
        if (WITH_MFMA) MFMA(c0)
        if (KIND == 1) { FMA(x0) FMA(x1) } else if (KIND == 2) { PKFMA(p0) } else if (KIND == 3) { MED3(x0) MED3(x1) }
        if (WITH_MFMA) MFMA(c1)
        if (KIND == 1) { FMA(x2) FMA(x3) } else if (KIND == 2) { PKFMA(p1) } else if (KIND == 3) { MED3(x2) MED3(x3) }
        if (WITH_MFMA) MFMA(c2)
        if (KIND == 1) { FMA(x4) FMA(x5) } else if (KIND == 2) { PKFMA(p2) } else if (KIND == 3) { MED3(x4) MED3(x5) }
        if (WITH_MFMA) MFMA(c3)
        if (KIND == 1) { FMA(x6) FMA(x7) } else if (KIND == 2) { PKFMA(p3) } else if (KIND == 3) { MED3(x6) MED3(x7) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1] + c0[0] + c1[0] + c2[0] + c3[0];
}
template <int KIND, bool WITH_MFMA>
static void run(const char *name, float *d) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps : {1, 2, 3, 4}) {
        dim3 grid(256 * wps);
        k<KIND, WITH_MFMA><<<grid, 256>>>(d, 100, 1.0001f, 0.5f);
        (void)hipEventRecord(e0);
        k<KIND, WITH_MFMA><<<grid, 256>>>(d, iters, 1.0001f, 0.5f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s waves/SIMD %d: %.1f ns per iteration per wave-slot (x%d waves share a SIMD)\n", name, wps, ms * 1e6 / iters / wps, wps);
    }
}
int main() {
    float *d; (void)hipMalloc(&d, 256 * 4 * 256 * 4);
    run<0, true>("4 mfma", d);
    run<1, false>("8 fma", d); run<2, false>("4 pk_fma", d); run<3, false>("8 med3", d);
    run<1, true>("4 mfma + 8 fma", d); run<2, true>("4 mfma + 4 pk_fma", d); run<3, true>("4 mfma + 8 med3", d);
    return 0;
}
