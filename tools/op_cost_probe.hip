// Throughput of the VALU instructions the sesrq epilogues are made of: ns (and cycles at the measured clock) per
// wave64 instruction per SIMD with 4 resident waves per SIMD, 8 independent chains per wave (inline asm, nothing rewritten).
//   hipcc --offload-arch=gfx950 -O2 tools/op_cost_probe.hip -o /tmp/oc && /tmp/oc
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
#define REP8(M) M(x0) M(x1) M(x2) M(x3) M(x4) M(x5) M(x6) M(x7)
#define REP4P(M) M(p0) M(p1) M(p2) M(p3)
#define K_FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define K_FMA_S(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "s"(a), "v"(b));
#define K_ADD(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(a));
#define K_MED3(x) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define K_MED3_S(x) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "s"(b));
#define K_MED3I(x) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define K_MED3_SS(x) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x) : "s"(a), "s"(a));
#define K_MED3I_S(x) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(x) : "s"(a), "v"(b));
#define K_FMAC(x) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define K_MUL(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(a));
#define K_MULS(x) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x) : "s"(a));
#define K_XOR(x) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(a));
#define K_MOV64(p) asm volatile("v_mov_b64 %0, %1" : "=v"(p) : "v"(a2));
#define K_PKFMA_S(p) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "s"(a2s), "v"(b2));
#define K_CNDMASK64(x) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "s"(m64));
#define K_ANDS(x) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x) : "s"(a));
#define K_AND(x) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x) : "v"(a));
#define K_MAX(x) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(a));
#define K_PERM(x) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define K_PERM_S(x) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "s"(b));
#define K_ADD3(x) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define K_ADDU(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(a));
#define K_LSHLOR(x) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(x) : "v"(a));
#define K_ANDOR(x) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define K_CVTPK(x) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(x) : "v"(a));
#define K_CVTI(x) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(x));
#define K_RNDNE(x) asm volatile("v_rndne_f32 %0, %0" : "+v"(x));
#define K_SDWA(x) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(x) : "v"(a));
#define K_SWAP32(x) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y0));
#define K_SWAP16(x) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y0));
#define K_MOV(x) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(a));
#define K_CNDMASK(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(a) : "vcc");
#define K_PKFMA(p) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(a2), "v"(b2));
#define K_PKADD(p) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(a2));
#define K_PKMUL(p) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(a2));
#define K_PKMOV(p) asm volatile("v_pk_mov_b32 %0, %0, %1" : "+v"(p) : "v"(a2));
#define KERNEL(NAME, BODY, N)                                                                         \
    __global__ void NAME(float *out, int iters, float a, float b) {                                  \
        float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7, y0 = 3.f; \
        v2f p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};                               \
        const v2f a2 = {a, a}, b2 = {b, b}; const v2f a2s = {a, b}; const unsigned long long m64 = 0x5555555555555555ull + (unsigned long long)iters;                                 \
        for (int i = 0; i < iters; ++i) { BODY }                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + y0 + p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1] + a2[0] + b2[0]; \
    }                                                                                                 \
    static const int NAME##_n = N;
KERNEL(k_fma, REP8(K_FMA), 8) KERNEL(k_fma_s, REP8(K_FMA_S), 8) KERNEL(k_add, REP8(K_ADD), 8) KERNEL(k_med3, REP8(K_MED3), 8)
KERNEL(k_med3_s, REP8(K_MED3_S), 8) KERNEL(k_med3i, REP8(K_MED3I), 8) KERNEL(k_max, REP8(K_MAX), 8) KERNEL(k_perm, REP8(K_PERM), 8)
KERNEL(k_perm_s, REP8(K_PERM_S), 8) KERNEL(k_add3, REP8(K_ADD3), 8) KERNEL(k_addu, REP8(K_ADDU), 8) KERNEL(k_lshlor, REP8(K_LSHLOR), 8)
KERNEL(k_andor, REP8(K_ANDOR), 8) KERNEL(k_cvtpk, REP8(K_CVTPK), 8) KERNEL(k_cvti, REP8(K_CVTI), 8) KERNEL(k_rndne, REP8(K_RNDNE), 8)
KERNEL(k_sdwa, REP8(K_SDWA), 8) KERNEL(k_swap32, REP8(K_SWAP32), 8) KERNEL(k_swap16, REP8(K_SWAP16), 8) KERNEL(k_mov, REP8(K_MOV), 8)
KERNEL(k_cndmask, REP8(K_CNDMASK), 8) KERNEL(k_pkfma, REP4P(K_PKFMA) REP4P(K_PKFMA), 8) KERNEL(k_pkadd, REP4P(K_PKADD) REP4P(K_PKADD), 8)
KERNEL(k_pkmul, REP4P(K_PKMUL) REP4P(K_PKMUL), 8) KERNEL(k_pkmov, REP4P(K_PKMOV) REP4P(K_PKMOV), 8)
KERNEL(k_med3_ss, REP8(K_MED3_SS), 8) KERNEL(k_med3i_s, REP8(K_MED3I_S), 8) KERNEL(k_fmac, REP8(K_FMAC), 8) KERNEL(k_mul, REP8(K_MUL), 8) KERNEL(k_muls, REP8(K_MULS), 8)
KERNEL(k_cndmask64, REP8(K_CNDMASK64), 8) KERNEL(k_ands, REP8(K_ANDS), 8) KERNEL(k_and, REP8(K_AND), 8) KERNEL(k_xor, REP8(K_XOR), 8) KERNEL(k_mov64, REP4P(K_MOV64) REP4P(K_MOV64), 8) KERNEL(k_pkfma_s, REP4P(K_PKFMA_S) REP4P(K_PKFMA_S), 8)
template <typename K>
static void run(const char *name, K kern, int n, float *d, double ref_ns) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000, wps = 4;
    kern<<<256 * wps, 256>>>(d, 100, 1.0001f, 0.5f);
    (void)hipEventRecord(e0);
    kern<<<256 * wps, 256>>>(d, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / ((double)iters * n * wps);
    printf("%-22s %.2f ns / wave-instruction / SIMD  (%.2f x v_add_f32)\n", name, ns, ref_ns > 0 ? ns / ref_ns : 1.0);
}
#define RUN(NAME, LABEL) run(LABEL, NAME, NAME##_n, d, ref)
int main() {
    float *d; (void)hipMalloc(&d, 256 * 4 * 256 * 4);
    double ref = 0;
    {   // reference: v_add_f32
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        k_add<<<1024, 256>>>(d, 100, 1.0001f, 0.5f);
        (void)hipEventRecord(e0); k_add<<<1024, 256>>>(d, 20000, 1.0001f, 0.5f); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ref = ms * 1e6 / (20000.0 * 8 * 4);
    }
    RUN(k_add, "v_add_f32"); RUN(k_fma, "v_fma_f32 (3 vgpr)"); RUN(k_fma_s, "v_fma_f32 (1 sgpr)"); RUN(k_max, "v_max_f32");
    RUN(k_med3, "v_med3_f32 (3 vgpr)"); RUN(k_med3_s, "v_med3_f32 (1 sgpr)"); RUN(k_med3i, "v_med3_i32"); RUN(k_perm, "v_perm_b32 (3 vgpr)");
    RUN(k_perm_s, "v_perm_b32 (1 sgpr)"); RUN(k_add3, "v_add3_u32"); RUN(k_addu, "v_add_u32"); RUN(k_lshlor, "v_lshl_or_b32");
    RUN(k_andor, "v_and_or_b32"); RUN(k_cvtpk, "v_cvt_pk_u8_f32"); RUN(k_cvti, "v_cvt_i32_f32"); RUN(k_rndne, "v_rndne_f32");
    RUN(k_sdwa, "v_add_u32_sdwa"); RUN(k_swap32, "v_permlane32_swap"); RUN(k_swap16, "v_permlane16_swap"); RUN(k_mov, "v_mov_b32");
    RUN(k_cndmask, "v_cndmask_b32"); RUN(k_pkfma, "v_pk_fma_f32"); RUN(k_pkadd, "v_pk_add_f32"); RUN(k_pkmul, "v_pk_mul_f32"); RUN(k_pkmov, "v_pk_mov_b32");
    RUN(k_med3_ss, "v_med3_f32 (2 sgpr same)"); RUN(k_med3i_s, "v_med3_i32 (1 sgpr)"); RUN(k_fmac, "v_fmac_f32"); RUN(k_mul, "v_mul_f32"); RUN(k_muls, "v_mul_f32 (sgpr)");
    RUN(k_cndmask64, "v_cndmask_b32 (sgpr pair mask)"); RUN(k_ands, "v_and_b32 (sgpr)"); RUN(k_and, "v_and_b32"); RUN(k_xor, "v_xor_b32"); RUN(k_mov64, "v_mov_b64"); RUN(k_pkfma_s, "v_pk_fma_f32 (sgpr pair)");
    return 0;
}
