// Minimal repro: is "buffer_store_dwordx4, then a VALU write of its first data VGPR" a hazard on gfx950 when the store carries an
// SGPR soffset?  LLVM's hazard recognizer (GCNHazardRecognizer::createsVALUHazard) and the ISA manual's wait-state table say NO
// pad is needed in that case; with soffset = 0 two wait states are required on gfx940+.  Every lane stores {tag, tag, tag, tag} to
// its own 16-byte slot and overwrites the first data register with POISON after N wait states (everything inside ONE asm block,
// physical registers named, so nothing is re-scheduled).  A slot whose word 0 reads POISON = the store read the overwritten register.
//   hipcc --offload-arch=gfx950 -O2 tools/store_hazard_probe.hip -o tools/store_hazard_probe && tools/store_hazard_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define POISON 0xdeadbeefu
template <int NOPS, bool SOFF_SGPR>
__global__ __launch_bounds__(256) void k(unsigned *out, int rounds, int soff_bytes) {
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x7fffffff, 0x00020000);
    for (int r = 0; r < rounds; ++r) {
        const unsigned tag = tid * 16 + r + 1;
        const unsigned voff = (tid * (unsigned)rounds + r) * 16;
        if constexpr (SOFF_SGPR) {
            asm volatile("v_mov_b32 v40, %0\n\tv_mov_b32 v41, %0\n\tv_mov_b32 v42, %0\n\tv_mov_b32 v43, %0\n\ts_nop 4\n\t"
                         "buffer_store_dwordx4 v[40:43], %1, %2, %3 offen sc1\n\t"
                         ".rept %c4\n\ts_nop 0\n\t.endr\n\t"
                         "v_mov_b32 v40, %5"
                         :: "v"(tag), "v"(voff), "s"(rs), "s"(soff_bytes), "i"(NOPS), "v"(POISON) : "v40", "v41", "v42", "v43", "memory");
        } else {
            asm volatile("v_mov_b32 v40, %0\n\tv_mov_b32 v41, %0\n\tv_mov_b32 v42, %0\n\tv_mov_b32 v43, %0\n\ts_nop 4\n\t"
                         "buffer_store_dwordx4 v[40:43], %1, %2, 0 offen sc1\n\t"
                         ".rept %c3\n\ts_nop 0\n\t.endr\n\t"
                         "v_mov_b32 v40, %4"
                         :: "v"(tag), "v"(voff), "s"(rs), "i"(NOPS), "v"(POISON) : "v40", "v41", "v42", "v43", "memory");
        }
    }
}
template <int NOPS, bool SOFF_SGPR>
static void run(unsigned *d, std::vector<unsigned> &h, int blocks, int rounds) {
    const size_t n = (size_t)blocks * 256 * rounds * 4;
    (void)hipMemset(d, 0, n * 4);
    k<NOPS, SOFF_SGPR><<<blocks, 256>>>(d, rounds, 0);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    size_t poisoned = 0, wrong = 0;
    for (size_t s = 0; s < n / 4; ++s) {
        const unsigned tid = (unsigned)(s / rounds), r = (unsigned)(s % rounds), tag = tid * 16 + r + 1;
        if (h[4 * s] == POISON) ++poisoned;
        else if (h[4 * s] != tag) ++wrong;
        for (int j = 1; j < 4; ++j) if (h[4 * s + j] != tag) ++wrong;
    }
    printf("soffset %-8s wait states %d: %zu of %zu stores read the overwritten register (%zu other wrong words)\n",
           SOFF_SGPR ? "SGPR" : "0", NOPS, poisoned, n / 4, wrong);
}
int main() {
    const int blocks = 2048, rounds = 16;
    unsigned *d;
    (void)hipMalloc(&d, (size_t)blocks * 256 * rounds * 16);
    std::vector<unsigned> h((size_t)blocks * 256 * rounds * 4);
    run<0, false>(d, h, blocks, rounds); run<1, false>(d, h, blocks, rounds); run<2, false>(d, h, blocks, rounds);
    run<0, true>(d, h, blocks, rounds); run<1, true>(d, h, blocks, rounds); run<2, true>(d, h, blocks, rounds);
    return 0;
}
