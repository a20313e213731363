// Round 4: are dword-aligned (not 8- / 16-byte-aligned) ds_read_b64 / ds_read_b128 legal on gfx950 as ROCm 7.2 configures it, do they
// return the right bytes, and what do they cost?  (The last layer's per-PE operands are vertical pixel pairs = two adjacent dwords of a
// column-major LDS image whose alignment changes with the output row: today two ds_read2_b32 per 16 operand bytes.)
// Per case: lane l reads WIDTH bytes at byte address 4 * (l * STRIDE + off), off = 0..3 dwords; checked against the fill pattern,
// timed over a loop of 16 independent reads per iteration (4 waves per SIMD, 256 CUs).
//   hipcc --offload-arch=gfx950 -O2 tools/lds_unaligned_probe.hip -o tools/lds_unaligned_probe && tools/lds_unaligned_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int W>   // W = 1: ds_read_b32, 2: ds_read_b64, 4: ds_read_b128, 22: ds_read2_b32 (two adjacent dwords), 24: two ds_read2_b32
__global__ __launch_bounds__(256) void k(unsigned *out, int iters, int stride, int off, unsigned long long *clk) {
    __shared__ unsigned lds[14336];
    for (int i = threadIdx.x; i < 14336; i += 256) lds[i] = 0xA0000000u + i;
    __syncthreads();
    const unsigned long long t0c = __builtin_amdgcn_s_memtime(), t0r = __builtin_amdgcn_s_memrealtime();
    const int l = threadIdx.x & 63;
    const unsigned base = (unsigned)(size_t)(const __attribute__((address_space(3))) void *)lds + 4u * (unsigned)(l * stride + off);
    unsigned acc = 0, bad = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) {      // four independent reads in flight, one wait
            const unsigned a0 = base, a1 = base + 4u * 64u * stride, a2 = base + 8u * 64u * stride, a3 = base + 12u * 64u * stride;
            if constexpr (W == 1) { unsigned v0, v1, v2, v3; asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %5\n\tds_read_b32 %2, %6\n\tds_read_b32 %3, %7\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3)); acc ^= v0 ^ v1 ^ v2 ^ v3; }
            if constexpr (W == 2) { v2u v0, v1, v2, v3; asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %6\n\tds_read_b64 %3, %7\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3)); acc ^= v0[0] ^ v1[1] ^ v2[0] ^ v3[1]; }
            if constexpr (W == 4) { v4u v0, v1, v2, v3; asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3)); acc ^= v0[0] ^ v1[1] ^ v2[2] ^ v3[3]; }
            if constexpr (W == 22) { v2u v0, v1, v2, v3; asm volatile("ds_read2_b32 %0, %4 offset1:1\n\tds_read2_b32 %1, %5 offset1:1\n\tds_read2_b32 %2, %6 offset1:1\n\tds_read2_b32 %3, %7 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3)); acc ^= v0[0] ^ v1[1] ^ v2[0] ^ v3[1]; }
            if constexpr (W == 264) { v4u v0, v1, v2, v3; asm volatile("ds_read2_b64 %0, %4 offset1:7\n\tds_read2_b64 %1, %5 offset1:7\n\tds_read2_b64 %2, %6 offset1:7\n\tds_read2_b64 %3, %7 offset1:7\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3)); acc ^= v0[0] ^ v1[1] ^ v2[2] ^ v3[3]; }
            if constexpr (W == 24) { v2u v0, v1, v2, v3; asm volatile("ds_read2_b32 %0, %4 offset1:1\n\tds_read2_b32 %1, %4 offset0:2 offset1:3\n\tds_read2_b32 %2, %5 offset1:1\n\tds_read2_b32 %3, %5 offset0:2 offset1:3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a0), "v"(a1)); acc ^= v0[0] ^ v1[1] ^ v2[0] ^ v3[1]; }
        }
    }
    // correctness of one read per lane
    {
        const unsigned idx = (unsigned)(l * stride + off);
        if constexpr (W == 2) { v2u v; asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base)); bad = (v[0] != 0xA0000000u + idx) | (v[1] != 0xA0000001u + idx); }
        if constexpr (W == 4) { v4u v; asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base));
                                bad = (v[0] != 0xA0000000u + idx) | (v[1] != 0xA0000001u + idx) | (v[2] != 0xA0000002u + idx) | (v[3] != 0xA0000003u + idx); }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (bad) atomicAdd(&out[1 << 20], 1u);
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0c; clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - t0r; }
}
template <int W>
static void run(const char *name, unsigned *d, unsigned long long *clk, int stride) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000;
    for (int off = 0; off < 4; ++off) {
        (void)hipMemset(d + (1 << 20), 0, 4);
        k<W><<<1024, 256>>>(d, 50, stride, off, clk);
        (void)hipEventRecord(e0);
        k<W><<<1024, 256>>>(d, iters, stride, off, clk);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned bad; (void)hipMemcpy(&bad, d + (1 << 20), 4, hipMemcpyDeviceToHost);
        unsigned long long h[2]; (void)hipMemcpy(h, clk + 200, sizeof(h), hipMemcpyDeviceToHost);
        const double ghz = (double)h[0] / (double)h[1] * 0.1;
        // 16 wave-level reads per iteration per wave (2x ds_read2_b32: 8 units of two instructions), 16 waves per CU in all
        printf("%-14s stride %2d dwords, offset %d dwords: %6.2f cycles per wave-instruction per CU, wrong lanes %u\n", name, stride, off,
               ms * 1e6 * ghz / iters / 16 / 16, bad);
    }
}
int main() {
    unsigned *d; (void)hipMalloc(&d, ((1 << 20) + 16) * 4);
    unsigned long long *clk; (void)hipMalloc(&clk, 4096 * 2 * sizeof(unsigned long long));
    for (int stride : {2, 4, 54}) {      // 2: b64-contiguous lanes; 4: b128-contiguous lanes; 54: the last layer's column pitch
        run<1>("ds_read_b32", d, clk, stride);
        run<22>("ds_read2_b32", d, clk, stride);
        run<2>("ds_read_b64", d, clk, stride);
        run<24>("2x ds_read2_b32", d, clk, stride);
        run<264>("ds_read2_b64", d, clk, stride);
        run<4>("ds_read_b128", d, clk, stride);
    }
    return 0;
}
