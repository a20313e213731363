// Device helpers shared by the MFMA kernels (per-layer: sesrq_mfma.hip, fused hidden trio: sesrq_trio.hip).
#pragma once
#include "sesrq_common.h"

namespace sesrq {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr float MAGIC = 12582912.f;   // 1.5 * 2^23

// accumulate modes
enum { MERGED = 0, GEN_STD = 1, GEN_ANY = 2, HYB = 3, GEN_TAP = 4, HYBS = 5 };   // GEN_STD: 18/20-bit clamps as literals; HYB: one risky PE
// GEN_TAP = GEN_ANY + the reference's PE dump taps (pe_outputK_P / pe_add_outputK, myQL/quan_func.py:372-378, 439-443) written by
// the MFMA per-PE kernels themselves (sesrq_forward_debug): the debug forward, never the production one
// HYBS = HYB of a 3-channel first layer on the 2:4 structured-sparse MFMA (v_smfmac_i32_16x16x128_i8: twice the K of the dense
// instruction in the same 16 cycles, measured): a pixel is one dword = one group of four K slots (3 channel bytes + a zero byte), the
// "other PEs" chain has two non-zero weights per group and the risky PE's chain one -- both are 2:4 images by construction, so each
// chain is ONE instruction over all 32 pixel taps instead of two dense ones.
__host__ __device__ constexpr bool mode_general(int m) { return m == GEN_STD || m == GEN_ANY || m == GEN_TAP; }
__host__ __device__ constexpr bool mode_biased(int m) { return m != GEN_ANY && m != GEN_TAP; }

__device__ __forceinline__ v4i mfma(v4i a, v4i b, v4i c) { return __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0); }
typedef int v8i __attribute__((ext_vector_type(8)));
// D += A_sparse(16 x 128, stored 16 x 64) x B(128 x 16).  Operand layout (measured, tools/smfmac_probe.hip): B lane (n, gb) register r
// = one group of four K slots; A lane (m, ga) stored bytes 2j, 2j+1 (j = 0..7) = the two kept elements of the group that B holds in
// lane group gb = 2 (ga & 1) + (j >> 2), register r = 4 (ga >> 1) + (j & 3); idx nibble j = position of element 0 | position of
// element 1 << 2 inside the group.
__device__ __forceinline__ v4i smfmac(v4i a, v8i b, v4i c, int idx) { return __builtin_amdgcn_smfmac_i32_16x16x128_i8(a, b, c, idx, 0, 0); }
__device__ __forceinline__ float med3(float v, float lo, float hi) { return __builtin_amdgcn_fmed3f(v, lo, hi); }
__device__ __forceinline__ int clampi3(int v, int lo, int hi) { return min(max(v, lo), hi); }
// the same for BIASED sums (bits = MAGIC_I + s = the float 1.5 * 2^23 + s, |s| < 2^22: one binade, so float order == integer order)
// with bounds that are not compile-time constants: one v_med3_f32 on the bit patterns.  (hipcc forms v_med3_i32 only for constant
// bounds -- v_max + v_min otherwise -- and an inline-asm v_med3_i32 would read an MFMA result without the wait states the compiler
// only inserts for instructions it models: 5617 wrong values in the first test.)
__device__ __forceinline__ int med3_biased(int v, int lo, int hi) {
    return __builtin_bit_cast(int, __builtin_amdgcn_fmed3f(__builtin_bit_cast(float, v), __builtin_bit_cast(float, lo), __builtin_bit_cast(float, hi)));
}
__device__ __forceinline__ v4i ld_frag(const int4 *p) { const int4 t = *p; v4i r = {t.x, t.y, t.z, t.w}; return r; }
__device__ __forceinline__ unsigned fbits(float f) { return __builtin_bit_cast(unsigned, f); }

// Input quantiser q0 = clamp8(rint(fl(fl(x/s) + z)))  (myQL/quan_func.py:225) as the low byte (two's complement) of the returned word.
// x / s is formed in three instructions, q = xc * r, fma(fma(-s, q, xc), r2, q), with NO test of fd.ok: the MFMA first-layer kernels run
// only when sesrq_create selected this form (fd.ok == 1; otherwise sesrq_api.hip sends layer 0 to the dot4 kernel, which divides) -- a
// wave-uniform test per value cut the staging code into 280 small blocks and cost the first layer 5 % (24.8 -> 23.5 us at 1080p).  The
// reciprocal form (option exact_div = 2) rides on the same instructions with r2 = 0: fma(e, 0, q) == q for the finite e the clamped x
// gives.  Round to nearest even by adding 1.5 * 2^23 -- no v_rndne / v_cvt_i32 -- and NO clamp of the result: x is clamped to
// [xlo, xhi], whose ends quantise to exactly -128 and 127 (sesrq_verify.hip proves the whole range, unclamped).
// Every constant sits in a VGPR (InQuantV, filled once per kernel through pin()): a scalar-operand v_mul / v_fma / v_add costs
// 2.1-2.3 ns per wave on gfx950, the all-VGPR form 1.4-1.5 (tools/op_cost_probe.hip) -- five such instructions per value -- and v_med3
// with two wave-uniform bounds needs one of them in a VGPR anyway (one SGPR per instruction).
struct InQuantV { float xlo, xhi, r, ns, r2, z, magic; };
__device__ __forceinline__ unsigned quantize_in_bits(float x, const InQuantV &c) {
    const float xc = med3(x, c.xlo, c.xhi);
    const float q = __fmul_rn(xc, c.r);
    const float q1 = __builtin_fmaf(__builtin_fmaf(c.ns, q, xc), c.r2, q);
    return __builtin_bit_cast(unsigned, __fadd_rn(__fadd_rn(q1, c.z), c.magic));
}

// A wave-uniform float pinned to a VGPR.  Why a VGPR: (1) hipcc 7.2 (clang 22) un-packs a v_pk_fma_f32 that sits in the shadow of an MFMA
// into two v_fma_f32; when the packed form read BOTH scalar operands out of one SGPR pair the halves become fma(v, s[n], s[n+1]) -- two
// scalar operands, illegal on gfx9 (a build error, never silent); (2) a scalar-operand v_fma / v_mul costs 2.2 ns per wave, the all-VGPR
// form 1.45 (tools/op_cost_probe.hip).
// How (round 4): the copy is the COMPILER's v_mov_b32; the asm statement is empty and only makes the value opaque.  Rounds 2-3 wrote the
// v_mov inside the asm string ("s_nop 1; v_mov_b32 %0, %1"): hipcc models no hazard of an instruction it cannot see, and in round 4's
// last-layer kernel the register allocator gave such a v_mov the dead fourth register of an MFMA result that was still in flight (the
// padding row of the 12-channel layer) -- three instructions later the MFMA wrote its result over the constant (XDL write -> VALU write of
// the same VGPR needs ~11 wait states): 9.5 M wrong bytes per 4K frame in the int8-only instance, found by bench.py's whole-frame parity.
// The same class as round 3's store-data hazard (a v_mov directly behind a buffer_store_dwordx4 of that register); both are gone by
// construction now: no VALU instruction of the kernels is hidden from the hazard recognizer except the SDWA adds of the residual merge,
// whose operands are live VALU results (tools/store_hazard_scan.py checks what remains).
__device__ __forceinline__ float in_vgpr(float x) {
    float r = x;
    asm("" : "+v"(r));
    return r;
}

// 0x80808080 (u8 <-> two's complement, four bytes at once) as a VGPR operand: v_xor with a literal costs 2.0 ns per wave, with a
// VGPR 1.4 (tools/op_cost_probe.hip); the statement is pure, so one v_mov per call site, hoisted out of the row loops
__device__ __forceinline__ unsigned flip80(unsigned w) {
    unsigned k = 0x80808080u;
    asm("" : "+v"(k));
    return w ^ k;
}

// low bytes of four words -> one word
__device__ __forceinline__ unsigned pack_lo_bytes(unsigned y0, unsigned y1, unsigned y2, unsigned y3) {
    const unsigned w01 = __builtin_amdgcn_perm(y1, y0, 0x0c0c0400u);
    const unsigned w23 = __builtin_amdgcn_perm(y3, y2, 0x0c0c0400u);
    return __builtin_amdgcn_perm(w23, w01, 0x05040100u);
}

constexpr int MAGIC_I = 0x4B400000;   // bit pattern of MAGIC: int s (|s| < 2^22) + MAGIC_I == bits of (float)(MAGIC + s)

// v[i] = fl(fl(s[i] * M) * 2^-n + zadd)   (un-rounded, two values per packed op)
// BIASED: s[i] already carries + MAGIC_I, i.e. its bits ARE the float MAGIC + s (exact); then
// fl(s*M) = fma(MAGIC + s, M, -MAGIC*M): the fma's product is exact, MAGIC*M = 3*M*2^22 is exactly
// representable (3*M < 2^18), so the single rounding is that of the exact s*M -- no v_cvt_f32_i32.
template <bool BIASED>
__device__ __forceinline__ void requant4(const int s[4], float Mf, float sh, float zadd, v2f &v01, v2f &v23) {
    const float zv = in_vgpr(zadd);       // see in_vgpr(): one v_mov per call site, hoisted out of the row loops
    const v2f M2 = {Mf, Mf}, sh2 = {sh, sh}, z2 = {zv, zv};
    if constexpr (BIASED) {
        const float c = -(MAGIC * Mf);
        const v2f c2 = {c, c};
        const v2f y01 = {__builtin_bit_cast(float, s[0]), __builtin_bit_cast(float, s[1])};
        const v2f y23 = {__builtin_bit_cast(float, s[2]), __builtin_bit_cast(float, s[3])};
        v01 = __builtin_elementwise_fma(__builtin_elementwise_fma(y01, M2, c2), sh2, z2);
        v23 = __builtin_elementwise_fma(__builtin_elementwise_fma(y23, M2, c2), sh2, z2);
    } else {
        const v2f f01 = {(float)s[0], (float)s[1]}, f23 = {(float)s[2], (float)s[3]};
        v01 = __builtin_elementwise_fma(f01 * M2, sh2, z2);
        v23 = __builtin_elementwise_fma(f23 * M2, sh2, z2);
    }
}

// clamp (un-rounded) then round-half-even to int8, 4 values -> packed word
__device__ __forceinline__ unsigned round_pack(v2f v01, v2f v23, float lo, float hi) {
    const v2f mg = {MAGIC, MAGIC};
    v2f c01 = {med3(v01[0], lo, hi), med3(v01[1], lo, hi)}, c23 = {med3(v23[0], lo, hi), med3(v23[1], lo, hi)};
    c01 = c01 + mg; c23 = c23 + mg;
    return pack_lo_bytes(fbits(c01[0]), fbits(c01[1]), fbits(c23[0]), fbits(c23[1]));
}

// The same for a value that already carries the zero point -128, v = fl(t - 128): one v_cvt_pk_u8_f32 per value does the clamp,
// the round-to-nearest-even and the byte insertion -- clamp8(rint(v)) + 128 == cvt_u8(fl(v + 128)) for EVERY fp32 t
// (exhaustive: tools/cvtpk_epilogue_probe.hip; v lies on the grid of t - 128, so v + 128 is exact where it matters) -- and one
// xor per word takes the four bytes back to two's complement: 2 pk_add + 4 cvt + 1 xor instead of 4 med3 + 2 pk_add + 3 perm.
// Only for the zero point -128 (every net calibrated on non-negative activations: all reference bundles), where ReLU's lower
// clamp max(z, -128) is the int8 clamp itself.
__device__ __forceinline__ unsigned round_pack_u8(v2f v01, v2f v23) {
    const v2f k = {128.f, 128.f};
    v01 = v01 + k; v23 = v23 + k;
    unsigned w = __builtin_amdgcn_cvt_pk_u8_f32(v01[0], 0, 0u);
    w = __builtin_amdgcn_cvt_pk_u8_f32(v01[1], 1, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(v23[0], 2, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(v23[1], 3, w);
    return flip80(w);
}

// w[i] = fma(bits(s[i]), M, c) for four biased sums, as four v_fma_f32 whose operands are ALL VGPRs (in_vgpr): round 3 issued two
// v_pk_fma_f32 with a scalar M.  Measured round 4 (tools/coissue2_probe.hip, tools/coissue3_probe.hip): beside MFMAs a packed fp32
// instruction costs 6.6 cycles of the SIMD (3.3 per value), a plain VALU instruction 2.1 from another wave and 3.4 from the MFMA's
// own wave; same-box A/B: trio -2.4 %, last layer -3.3 %.
__device__ __forceinline__ void fma4_biased(const int s[4], float Md, float Cd, float w[4]) {
    const float cv = in_vgpr(Cd), mv = in_vgpr(Md);
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = __builtin_fmaf(__builtin_bit_cast(float, s[i]), mv, cv);
}

// ---- epilogues (AT = any struct with the ConvArgs field names Mf, sh, z_next, Mres, shres, z_merge) ----
// U8: every zero point the epilogue adds is -128 (z_next; for the residual merge also z_merge): round_pack_u8

// hidden layer: q = clamp8(rint(relu(t) + z_next))            (myQL/quan_func.py:280)
// U8 == 2: the one-fma requant (prove_direct_requant, sesrq_verify.hip): w = fl(s * M) * 2^-n straight into cvt_pk_u8 -- no second
// fma, no "+ 128": per four values 2 pk_fma + 4 cvt + 1 xor instead of 4 pk_fma + 2 pk_add + 4 cvt + 1 xor
template <bool BIASED, int U8 = 0, class AT>
__device__ __forceinline__ unsigned epi_mid(const int s[4], const AT &a, float zlo) {
    v2f v01, v23;
    if constexpr (U8 == 2) {
        static_assert(BIASED, "one-fma requant: biased sums only");
        float t[4];
        fma4_biased(s, a.Md, a.Cd, t);
        unsigned w = __builtin_amdgcn_cvt_pk_u8_f32(t[0], 0, 0u);
        w = __builtin_amdgcn_cvt_pk_u8_f32(t[1], 1, w);
        w = __builtin_amdgcn_cvt_pk_u8_f32(t[2], 2, w);
        w = __builtin_amdgcn_cvt_pk_u8_f32(t[3], 3, w);
        return flip80(w);
    }
    if constexpr (U8 == 1) {
        requant4<BIASED>(s, a.Mf, a.sh, -128.f, v01, v23);
        return round_pack_u8(v01, v23);
    }
    requant4<BIASED>(s, a.Mf, a.sh, a.z_next, v01, v23);
    return round_pack(v01, v23, zlo, 127.f);
}
// layer-0 residual operand rc = clamp8(rint(relu(t) - 128))    (myQL/quan_func.py:250)
template <bool BIASED, class AT>
__device__ __forceinline__ unsigned epi_rc(const int s[4], const AT &a) {
    v2f v01, v23;
    requant4<BIASED>(s, a.Mf, a.sh, -128.f, v01, v23);
    return round_pack(v01, v23, -128.f, 127.f);
}
// layer L-2: long residual merged in the integer domain        (myQL/quan_func.py:249-270)
template <bool BIASED, int U8 = 0, class AT>
__device__ __forceinline__ unsigned epi_preres(const int s[4], unsigned rcword, const AT &a) {
    v2f v01, v23;
    requant4<BIASED>(s, a.Mf, a.sh, -128.f, v01, v23);
    const unsigned rcx = rcword ^ 0x80808080u;                 // rc + 128 as unsigned bytes
    // ic = rint(clamp(t - 128)) ; u = rc + ic + 256 = (rc + 128) + (ic + 128), an integer in [0, 510].
    // Adding 1.5*2^23 + 128 rounds to nearest even and leaves ic + 128 in the low mantissa bits; a plain integer add of the
    // rc byte then gives the bit pattern of the float 1.5*2^23 + u, which is exactly what the cvt-free requant
    // (requant4<true>) takes as its input: no v_rndne, no byte->float converts, no float adds.
    const v2f mg128 = {MAGIC + 128.f, MAGIC + 128.f};
    v2f c01 = {med3(v01[0], -128.f, 127.f), med3(v01[1], -128.f, 127.f)}, c23 = {med3(v23[0], -128.f, 127.f), med3(v23[1], -128.f, 127.f)};
    c01 = c01 + mg128; c23 = c23 + mg128;
    const int u[4] = {(int)(fbits(c01[0]) + (rcx & 0xffu)), (int)(fbits(c01[1]) + ((rcx >> 8) & 0xffu)),
                      (int)(fbits(c23[0]) + ((rcx >> 16) & 0xffu)), (int)(fbits(c23[1]) + (rcx >> 24))};
    v2f w01, w23;
    if constexpr (U8) {
        requant4<true>(u, a.Mres, a.shres, -128.f, w01, w23);
        return round_pack_u8(w01, w23);
    }
    requant4<true>(u, a.Mres, a.shres, a.z_merge, w01, w23);
    return round_pack(w01, w23, -128.f, 127.f);
}

// The same with the second requant read out of the 512-byte table q4(u) that sesrq_create built (fused trio: table in LDS at byte
// address lut_addr < 2^15).  The rounding constant also carries the table's address: c = ic + (MAGIC + 256 + lut_addr) is exact
// (ulp 1), its low 16 bits are lut_addr + (ic + 256), and adding the signed rc byte to them (one v_add_u32_sdwa) IS the LDS address
// of q4(u): per value 2 fma + med3 + add + sdwa-add + one byte read (+ 1/4 v_lshl_or), no second requant, no byte shuffles.
// ONE_FMA (prove_direct_requant for this layer's (M, n)): w = fl(s * M) * 2^-n out of one fma, ic + 128 = rint(clamp(w, 0, 255)) --
// the table index is then the sum of two unsigned bytes (below).
template <bool BIASED, bool ONE_FMA = false, class AT>
__device__ __forceinline__ unsigned epi_preres_lut(const int s[4], unsigned rcword, const AT &a, float lut_magic,
                                                   const unsigned char __attribute__((address_space(3))) *lut = nullptr) {
    typedef const unsigned char __attribute__((address_space(3))) *lds_u8_t;
    v2f v01, v23;
    v2f c01, c23;
    if constexpr (ONE_FMA) {
        static_assert(BIASED, "one-fma requant: biased sums only");
        float t[4];
        fma4_biased(s, a.Md, a.Cd, t);
        v01 = (v2f){t[0], t[1]}; v23 = (v2f){t[2], t[3]};
        // ic + 128 as four bytes of one word (cvt_pk_u8: clamp, rounding and insertion in one instruction per value), rc + 128 by one
        // xor per word; u = the sum of two unsigned bytes (v_add_u32_sdwa, BYTE_k + BYTE_k) IS the table index, the table's LDS address
        // rides in the read's offset field: 4 cvt + 1/4 xor instead of 4 med3 + 2 pk_add per four values
        unsigned icw = __builtin_amdgcn_cvt_pk_u8_f32(v01[0], 0, 0u);
        icw = __builtin_amdgcn_cvt_pk_u8_f32(v01[1], 1, icw);
        icw = __builtin_amdgcn_cvt_pk_u8_f32(v23[0], 2, icw);
        icw = __builtin_amdgcn_cvt_pk_u8_f32(v23[1], 3, icw);
        const unsigned rcu = flip80(rcword);
        unsigned u0, u1, u2, u3;      // (inline asm: hipcc builds v_bfe + v_add3 out of the C form; the operands are plain VALU results)
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0" : "=v"(u0) : "v"(icw), "v"(rcu));
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1" : "=v"(u1) : "v"(icw), "v"(rcu));
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_2" : "=v"(u2) : "v"(icw), "v"(rcu));
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3" : "=v"(u3) : "v"(icw), "v"(rcu));
        __builtin_assume(u0 < 511u); __builtin_assume(u1 < 511u); __builtin_assume(u2 < 511u); __builtin_assume(u3 < 511u);      // so the table's base folds into the reads' offset field
        typedef unsigned short v2us_ __attribute__((ext_vector_type(2)));
        const v2us_ x_ = {(unsigned short)lut[u0], (unsigned short)lut[u2]};
        const v2us_ y_ = {(unsigned short)lut[u1], (unsigned short)lut[u3]};
        return __builtin_bit_cast(unsigned, x_) | (__builtin_bit_cast(unsigned, y_) << 8);
    } else {
        requant4<BIASED>(s, a.Mf, a.sh, -128.f, v01, v23);
        const v2f mg = {lut_magic, lut_magic};                 // MAGIC + 256 + lut_addr: the low 16 bits of c are lut_addr + (ic + 256)
        c01 = (v2f){med3(v01[0], -128.f, 127.f), med3(v01[1], -128.f, 127.f)}; c23 = (v2f){med3(v23[0], -128.f, 127.f), med3(v23[1], -128.f, 127.f)};
        c01 = c01 + mg; c23 = c23 + mg;
    }
    // + the SIGNED rc byte (v_add_u32_sdwa, WORD_0 + sign-extended BYTE_k): lut_addr + (rc + ic + 256) = the address of q4(u)
    const unsigned a0 = (fbits(c01[0]) & 0xffffu) + (unsigned)(int)(signed char)(rcword), a1 = (fbits(c01[1]) & 0xffffu) + (unsigned)(int)(signed char)(rcword >> 8);
    const unsigned a2 = (fbits(c23[0]) & 0xffffu) + (unsigned)(int)(signed char)(rcword >> 16), a3 = (fbits(c23[1]) & 0xffffu) + (unsigned)((int)rcword >> 24);
    // bytes 0 / 2 and 1 / 3 as the halves of two registers, joined by one v_lshl_or.  (Written for ds_read_u8_d16 / _d16_hi pairs; gfx950 runs
    // with SRAM ECC, where a d16 load clears the register's other half, so hipcc emits four ds_read_u8 and two v_perm instead: three VALU
    // per four bytes, the minimum for four single-byte registers)
    typedef unsigned short v2us __attribute__((ext_vector_type(2)));
    const v2us x = {(unsigned short)*(lds_u8_t)(size_t)a0, (unsigned short)*(lds_u8_t)(size_t)a2};
    const v2us y = {(unsigned short)*(lds_u8_t)(size_t)a1, (unsigned short)*(lds_u8_t)(size_t)a3};
    return __builtin_bit_cast(unsigned, x) | (__builtin_bit_cast(unsigned, y) << 8);
}

// PE clamp / sum / adder clamp / add constant               (myQL/quan_func.py:370,380-386,437,491)
// NV: real rows of the lane (the last layer's three-row map leaves s[3] untouched: padding, never read)
template <int MODE, int NV = 4, class AT>
__device__ __forceinline__ void finish_sums(int s[4], const v4i *acc, const int4 ac, const AT &a) {
    const int acv[4] = {ac.x, ac.y, ac.z, ac.w};
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if constexpr (MODE == MERGED) {
            s[i] = acc[0][i];       // add constant already in the accumulator (C-in)
        } else if constexpr (MODE == HYB) {
            // acc[0] = add constant + the three PEs that cannot saturate (load-time proof: each stays inside 18 bits, so their
            // 18-bit clamps are no-ops), acc[1] = the risky PE.  |three safe sums + one clamped sum| <= 3*131072 + 131072
            // = 2^19: the 20-bit adder clamp cannot fire either (quan_func.py:437 is a no-op here).
            s[i] = acc[0][i] + clampi3(acc[1][i], -131072, 131071);
        } else if constexpr (MODE == GEN_STD) {
            const int t = clampi3(acc[0][i], -131072, 131071) + clampi3(acc[1][i], -131072, 131071) +
                          clampi3(acc[2][i], -131072, 131071) + clampi3(acc[3][i], -131072, 131071);
            s[i] = clampi3(t, -524288, 524287) + acv[i];
        } else {
            const int t = clampi3(acc[0][i], a.acc_lo, a.acc_hi) + clampi3(acc[1][i], a.acc_lo, a.acc_hi) +
                          clampi3(acc[2][i], a.acc_lo, a.acc_hi) + clampi3(acc[3][i], a.acc_lo, a.acc_hi);
            s[i] = clampi3(t, a.add_lo, a.add_hi) + acv[i];
        }
    }
}

// Debug taps of the per-PE MFMA kernels (GEN_TAP): lane (n, g) holds the four PE sums of accumulator rows 4g + i of pixel (gy, gx).
// pe_out = the PE sums after the accumulator clamp, pe_add = their sum after the adder clamp (before the add constant) -- what the
// reference dumps as pe_outputK_P.pt / pe_add_outputK.pt (myQL/quan_func.py:370-378, 437-443); same layout as the dot4 taps.
// lastnv: 0 = hidden / first layer (row 4g + i = channel g + 4i, PE-major), else the last layer's slot map (last_slot_oc).
template <int NV>
__device__ __forceinline__ void tap_sums(const v4i *acc, const ConvArgs &a, int n_img, int gy, int gx, int g, int lastnv) {
    if (gy >= a.H || gx >= a.W) return;
    const size_t HW = (size_t)a.H * a.W, px = (size_t)gy * a.W + gx;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int oc = lastnv ? last_slot_oc(lastnv, g, i, a.oc, a.ps) : g + 4 * i;
        if (oc >= a.oc) continue;
        int t = 0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int pe = clampi3(acc[p][i], a.acc_lo, a.acc_hi);
            t += pe;
            if (a.dbg_pe) a.dbg_pe[(((size_t)n_img * 4 + p) * a.oc + oc) * HW + px] = pe;
        }
        if (a.dbg_add) a.dbg_add[((size_t)n_img * a.oc + oc) * HW + px] = clampi3(t, a.add_lo, a.add_hi);
    }
}

// 4x4 transpose between lane groups (16 lanes each) and registers; its own inverse.
// in : w[r] in lane (n, g) = word g of row r        out: w[g'] in lane (n, r') = word g' of row r'
__device__ __forceinline__ void transpose4(unsigned w[4]) {
    v2u t;
    t = __builtin_amdgcn_permlane32_swap(w[0], w[2], false, false); w[0] = t[0]; w[2] = t[1];
    t = __builtin_amdgcn_permlane32_swap(w[1], w[3], false, false); w[1] = t[0]; w[3] = t[1];
    t = __builtin_amdgcn_permlane16_swap(w[0], w[1], false, false); w[0] = t[0]; w[1] = t[1];
    t = __builtin_amdgcn_permlane16_swap(w[2], w[3], false, false); w[2] = t[0]; w[3] = t[1];
}

// XCD-aware block -> (strip, run) map (round 4).  The dispatcher deals consecutive workgroup ids round-robin over the chip's 8 XCDs (blocks
// b and b + 8 share one, MI355X_MICROARCH.md), each with an L2 of its own: with the identity map the neighbours of a tile -- which read the
// same 128-byte lines at its edges and re-read its halo -- sit on OTHER XCDs, and every such line comes out of the Infinity Cache / HBM once
// per XCD.  Here XCD x gets the x-th contiguous eighth of the run-major tile order, so those reads hit the L2 the first one filled.
// Measured (same box, FETCH_SIZE x 2 as tools/fetch_calib_probe.hip calibrates it for 4-byte and 16-byte lanes alike): first layer 50.8 ->
// 25.8 MB fetched for its 24.9 MB fp32 frame -- a 64-pixel tile row is two 128-byte lines per channel plane and its halo touches two more --
// and 16.8 -> 14.6 us; trio 43.0 -> 35.0 MB (33.2 MB of input), last layer 43.0 -> 34.8 MB, times unchanged (they are issue-bound).
// Speed only: nothing depends on which XCD a block lands on.
// Every 64-byte line of the kernel-argument segment requested by the FIRST instructions of the kernel, in one batch.  hipcc loads kernel
// arguments where they are first used: the 540-byte ConvArgs reached the prologue as four to six DEPENDENT batches of scalar loads (block
// index -> frame pointers -> geometry -> epilogue constants ...), each a miss of the scalar cache to HBM with every workgroup of the launch
// starting at once -- 4.6 k cycles from kernel entry to the first frame load of the last layer (tools/stamps.py, round 4; with the segment
// in host memory, HIP_FORCE_DEV_KERNARG=0, the first layer is 5 us slower: five round trips).  After this the later loads hit the cache.
template <typename ARGS>
__device__ __forceinline__ void kernarg_warm() {
    typedef const int __attribute__((address_space(4))) *karg_t;
    karg_t kp = (karg_t)__builtin_amdgcn_kernarg_segment_ptr();
    // Only lines of the EXPLICIT argument struct (rounds 3-4 read 32 bytes past it, relying on the implicit arguments hipcc
    // appends; a build whose code object carries fewer would have read past the segment).  The implicit block counts gridDim reads
    // follow the struct directly and share its last line or the next one, which the kernel's own first use fetches.
    constexpr int LINES = ((int)sizeof(ARGS) + 63) / 64;
    static_assert(LINES <= 12 && sizeof(ARGS) % 4 == 0, "one asm operand per line");
    int t[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) t[i] = kp[(i < LINES ? i * 16 : 0)];
    const int last = kp[(int)sizeof(ARGS) / 4 - 1];
    asm volatile("" ::"s"(t[0]), "s"(t[1]), "s"(t[2]), "s"(t[3]), "s"(t[4]), "s"(t[5]), "s"(t[6]), "s"(t[7]), "s"(t[8]), "s"(t[9]), "s"(t[10]),
                 "s"(t[11]), "s"(last));
}

#ifdef SESRQ_STAMPS
struct BlockXY { int x, y, t_entry, c_entry; };      // diagnostic build: clocks read where the kernel fetches its block index
#else
struct BlockXY { int x, y; };
#endif
// inv_nx = ceil(2^32 / gridDim.x) from the host (0: divide): v / nx == mulhi(v, inv_nx) for v, nx < 2^16 (the error term nx * inv_nx - 2^32 < nx,
// times v, stays below 2^32) -- the prologue's 32-bit division was ~25 dependent scalar / vector instructions in front of the first load
__device__ __forceinline__ BlockXY xcd_block(unsigned inv_nx = 0) {
    BlockXY b = {(int)blockIdx.x, (int)blockIdx.y};
#ifdef SESRQ_STAMPS
    b.t_entry = (int)__builtin_amdgcn_s_memrealtime();
    b.c_entry = (int)__builtin_amdgcn_s_memtime();
#endif
    const unsigned nx = gridDim.x, total = nx * gridDim.y, per = total >> 3;
    const unsigned id = blockIdx.y * nx + blockIdx.x;
    if (id < per * 8) {      // the last total % 8 blocks keep their place
        const unsigned v = (id & 7) * per + (id >> 3);
        b.y = inv_nx ? (int)__umulhi(v, inv_nx) : (int)(v / nx);
        b.x = (int)(v - (unsigned)b.y * nx);
    }
    return b;
}

// Per-image NHWC16 tensor addressed through a buffer descriptor: rows/pixels outside the
// frame are dropped (stores) or read as zero (loads) by the hardware range check.
struct RowIO {
    __amdgpu_buffer_rsrc_t out, rc_in, rc_out;
    int voff;        // lane (n, r' = g): byte offset of pixel (y0 + g, gx) or out-of-range
    int row_bytes;   // W * 16
};
__device__ __forceinline__ RowIO make_rowio(const ConvArgs &a, int n_img, int y0, int gx, int g) {
    RowIO io;
    const size_t img = (size_t)a.H * a.W * 16;
    const int bytes = (int)img;
    io.out = __builtin_amdgcn_make_buffer_rsrc((char *)a.out + (size_t)n_img * img, 0, bytes, 0x00020000);
    io.rc_in = __builtin_amdgcn_make_buffer_rsrc((char *)a.rc_in + (size_t)n_img * img, 0, bytes, 0x00020000);
    io.rc_out = __builtin_amdgcn_make_buffer_rsrc((char *)a.rc_out + (size_t)n_img * img, 0, bytes, 0x00020000);
    io.row_bytes = a.W * 16;
    io.voff = (gx < a.W) ? ((y0 + g) * a.W + gx) * 16 : (int)0x80000000;
    return io;
}
__device__ __forceinline__ void store_rows4(__amdgpu_buffer_rsrc_t rs, const RowIO &io, int y4, unsigned w[4]) {
    // aux 16 = sc1: the activation tensor is only read again by the NEXT kernel; measured against the default policy,
    // sc0|sc1 and nt on 1080p: sc1 -6 % on the first layer, -3..5 % on the hidden layers when frames overlap; nt +12 %
    transpose4(w);
    const v4u v = {w[0], w[1], w[2], w[3]};
    // The row offset rides in the VECTOR offset, soffset = 0, on purpose: gfx950 needs one wait state between a dwordx4 store with
    // an SGPR soffset and a VALU write of its data registers (tools/store_hazard_probe.hip: 2000 of 8.4 M stores carried the
    // overwritten register at 0 wait states), LLVM's hazard recognizer assumes such a store needs none, and hipcc 7.2 did place a
    // v_mov / v_pk_fma of the data registers directly behind these stores (round 2's "garbage in the first output word").  For
    // soffset = 0 the compiler pads the two wait states itself.  One v_add per four rows.
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, io.voff + y4 * io.row_bytes, 0, 16);
}

// hidden-layer output of 4 rows: s4[r][i] -> requant -> transpose -> one 16-byte store per lane
template <int EPI, bool RC, bool BIASED, int U8 = 0, class AT>
__device__ __forceinline__ void emit_rows4(const int s4[4][4], const AT &a, const RowIO &io, int y4, float zlo) {
    unsigned w[4];
    if constexpr (EPI == EPI_PRERES) {
        unsigned rcw[4];
        const v4u rv = __builtin_amdgcn_raw_buffer_load_b128(io.rc_in, io.voff, y4 * io.row_bytes, 0);
        rcw[0] = rv[0]; rcw[1] = rv[1]; rcw[2] = rv[2]; rcw[3] = rv[3];
        transpose4(rcw);
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = epi_preres<BIASED, (U8 != 0)>(s4[r], rcw[r], a);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = epi_mid<BIASED, U8>(s4[r], a, zlo);
    }
    store_rows4(io.out, io, y4, w);
    if constexpr (RC) {
        unsigned rw[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) rw[r] = epi_rc<BIASED>(s4[r], a);
        store_rows4(io.rc_out, io, y4, rw);
    }
}

// residual-merging rows with the residual operand words already in compute layout (lane (n, g): word g of pixel n of row r)
template <bool BIASED, class AT>
__device__ __forceinline__ void emit_rows4_preres_rc(const int s4[4][4], const unsigned rcw[4], const AT &a, const RowIO &io, int y4) {
    unsigned w[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) w[r] = epi_preres<BIASED>(s4[r], rcw[r], a);
    store_rows4(io.out, io, y4, w);
}

}  // namespace sesrq
