// Device helpers shared by the MFMA engines (per-layer: sesrq_mfma.hip, fused: sesrq_fused.hip).
#pragma once
#include "sesrq_common.h"

namespace sesrq {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr float MAGIC = 12582912.f;   // 1.5 * 2^23

// accumulate modes
enum { MERGED = 0, GEN_STD = 1, GEN_ANY = 2, HYB = 3 };   // GEN_STD: 18/20-bit clamps as literals; HYB: one risky PE

__device__ __forceinline__ v4i mfma(v4i a, v4i b, v4i c) { return __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float med3(float v, float lo, float hi) { return __builtin_amdgcn_fmed3f(v, lo, hi); }
__device__ __forceinline__ int clampi3(int v, int lo, int hi) { return min(max(v, lo), hi); }
__device__ __forceinline__ v4i ld_frag(const int4 *p) { const int4 t = *p; v4i r = {t.x, t.y, t.z, t.w}; return r; }
__device__ __forceinline__ unsigned fbits(float f) { return __builtin_bit_cast(unsigned, f); }

// input quantiser q0 = clamp8(rint(fl(fl(x/s) + z)))  (myQL/quan_func.py:225), as a float in [-128,127]
__device__ __forceinline__ float quantize_in(float x, float s, float z, const FastDiv &fd) {
    float t;
    if (fd.ok) {                       // proven bit-identical for this (s, z): sesrq_verify.hip
        const float xc = med3(x, fd.xlo, fd.xhi);
        const float q = __fmul_rn(xc, fd.r);
        t = __builtin_fmaf(__builtin_fmaf(-s, q, xc), fd.r, q);
    } else {
        t = __fdiv_rn(x, s);
    }
    return med3(rintf(__fadd_rn(t, z)), -128.f, 127.f);
}
// same value as the low byte (two's complement) of the returned word: clamp the un-rounded value, then round to nearest
// even by adding 1.5 * 2^23 (clamp and rint commute for integer bounds) -- no v_rndne / v_cvt_i32
__device__ __forceinline__ unsigned quantize_in_bits(float x, float s, float z, const FastDiv &fd) {
    float t;
    if (fd.ok) {
        const float xc = med3(x, fd.xlo, fd.xhi);
        const float q = __fmul_rn(xc, fd.r);
        t = __builtin_fmaf(__builtin_fmaf(-s, q, xc), fd.r, q);
    } else {
        t = __fdiv_rn(x, s);
    }
    return __builtin_bit_cast(unsigned, __fadd_rn(med3(__fadd_rn(t, z), -128.f, 127.f), 12582912.f));
}

// general path: the MFMA operand of PE P = word P of four staged pixels.  Two v_pk_mov_b32 build the
// four consecutive operand registers (each moves one word out of two different source pairs) instead of
// four v_mov_b32.  op_sel/op_sel_hi = [s,s]: D.lo = src0.{lo|hi}, D.hi = src1.{lo|hi} (checked on gfx950).
typedef int v2i __attribute__((ext_vector_type(2)));
template <int P>
__device__ __forceinline__ v4i gather4(const int4 a, const int4 b, const int4 c, const int4 d) {
    const int av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w}, cv[4] = {c.x, c.y, c.z, c.w}, dv[4] = {d.x, d.y, d.z, d.w};
    constexpr int h = 2 * (P / 2);
    const v2i pa = {av[h], av[h + 1]}, pb = {bv[h], bv[h + 1]}, pc = {cv[h], cv[h + 1]}, pd = {dv[h], dv[h + 1]};
    v2i lo, hi;
    if constexpr ((P & 1) == 0) {
        asm("v_pk_mov_b32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0]" : "=v"(lo) : "v"(pa), "v"(pb));
        asm("v_pk_mov_b32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0]" : "=v"(hi) : "v"(pc), "v"(pd));
    } else {
        asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1]" : "=v"(lo) : "v"(pa), "v"(pb));
        asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1]" : "=v"(hi) : "v"(pc), "v"(pd));
    }
    const v4i r = {lo[0], lo[1], hi[0], hi[1]};
    return r;
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
    const v2f M2 = {Mf, Mf}, sh2 = {sh, sh}, z2 = {zadd, zadd};
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

}  // namespace sesrq
