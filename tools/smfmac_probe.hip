// Operand layout of v_smfmac_i32_16x16x128_i8 (2:4 structured-sparse A, gfx950), found empirically with exact integer data:
// B holds code(k) = k - 64 at the position HYPOTHESISED for logical K index k (lane (n, g) register r byte b <-> k = 32 g + 4 r + b),
// A holds a single stored 1 (lane group ga, stored byte s of the 16 per lane, every row m), the index register the same value in
// every lane.  D[m][n] + 64 is then the logical k that stored byte selected -- printed per (ga, s) for a few index patterns.
//   hipcc --offload-arch=gfx950 -O2 tools/smfmac_probe.hip -o tools/smfmac_probe && tools/smfmac_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
__global__ void k(const int *A, const int *B, const int *idx, int *D) {
    const int l = threadIdx.x;
    v4i a = {A[l * 4], A[l * 4 + 1], A[l * 4 + 2], A[l * 4 + 3]};
    v8i b;
    for (int i = 0; i < 8; ++i) b[i] = B[l * 8 + i];
    v4i acc = {0, 0, 0, 0};
    const int ix = idx[l];
    asm volatile("s_nop 4\n\tv_smfmac_i32_16x16x128_i8 %0, %1, %2, %3\n\ts_nop 15\n\ts_nop 15" : "+v"(acc) : "v"(a), "v"(b), "v"(ix));
    for (int i = 0; i < 4; ++i) D[l * 4 + i] = acc[i];
}
int main() {
    int *dA, *dB, *dI, *dD;
    (void)hipMalloc(&dA, 64 * 16); (void)hipMalloc(&dB, 64 * 32); (void)hipMalloc(&dI, 64 * 4); (void)hipMalloc(&dD, 64 * 16);
    std::vector<int> A(64 * 4), B(64 * 8), I(64), D(64 * 4);
    signed char *Bb = reinterpret_cast<signed char *>(B.data());
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 32; ++j) Bb[l * 32 + j] = (signed char)(32 * (l >> 4) + j - 64);
    (void)hipMemcpy(dB, B.data(), 64 * 32, hipMemcpyHostToDevice);
    const unsigned pats[] = {0x00000000u, 0x44444444u, 0xeeeeeeeeu, 0x93939393u, 0x4e4e4e4eu, 0x11111111u, 0xffffffffu, 0x76543210u};
    for (unsigned pat : pats) {
        printf("idx = 0x%08x: logical k selected by stored byte s of lane group ga (rows m agree unless flagged)\n", pat);
        for (int ga = 0; ga < 4; ++ga) {
            printf("  ga %d:", ga);
            for (int s = 0; s < 16; ++s) {
                std::fill(A.begin(), A.end(), 0);
                signed char *Ab = reinterpret_cast<signed char *>(A.data());
                for (int m = 0; m < 16; ++m) Ab[(16 * ga + m) * 16 + s] = 1;
                for (int l = 0; l < 64; ++l) I[l] = (int)pat;
                (void)hipMemcpy(dA, A.data(), 64 * 16, hipMemcpyHostToDevice);
                (void)hipMemcpy(dI, I.data(), 64 * 4, hipMemcpyHostToDevice);
                k<<<1, 64>>>(dA, dB, dI, dD);
                (void)hipMemcpy(D.data(), dD, 64 * 16, hipMemcpyDeviceToHost);
                // D layout: lane (n, g) register i = row 4 g + i, column n
                int v0 = D[0];
                bool same = true;
                for (int j = 0; j < 256; ++j) same &= D[j] == v0;
                printf(" %s%d", same ? "" : "*", v0 + 64);
            }
            printf("\n");
        }
    }
    return 0;
}
