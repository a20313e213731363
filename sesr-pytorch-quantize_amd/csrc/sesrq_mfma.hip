// sesrq MFMA engine: im2col-free implicit-GEMM INT8 convolution on the CDNA4 matrix cores
// (v_mfma_i32_16x16x64_i8), still "direct": the B operand of every MFMA is read straight
// out of the NHWC int8 input tile staged in LDS -- one ds_read_b128 = one 16-channel pixel =
// 16 of the 64 K-slots of a lane group -- nothing is ever materialised as an im2col matrix.
//
//   D[m][n] += sum_k A[m][k] * B[k][n]      M = 16 output-channel slots (weights, A)
//                                           N = 16 horizontally adjacent output pixels (B)
//                                           K = 64 = 4 lane groups g x 16 bytes
//   lane l:  n (or m) = l & 15, g = l >> 4;  D: lane holds rows m = 4g..4g+3 of column n.
//
// The K order inside the instruction is irrelevant: A and B are packed with the same
// (g, byte) -> (tap, channel) table (host: pack_mfma_frags in sesrq_api.hip).
//
//   merged  (load-time proof: no 18/20-bit saturation possible): a lane group = one tap,
//           16 bytes = the 16 channels; rows are re-used across ky (register rotation).
//   general (per-PE sums must be clamped separately, myQL/quan_func.py:370): a lane's 16
//           bytes = 4 taps x the 4 channels of ONE PE; one MFMA chain per PE; the operand is
//           word p of four staged pixels (PE-major channel order makes that a register pick).
//
// Output rows are ordered so that a lane's 4 accumulators are the 4 bytes of word g of the
// NHWC16 (PE-major) output pixel.  Four image rows are produced per step; a 4x4 transpose
// between lane groups and registers (v_permlane32_swap + v_permlane16_swap) then gives every
// lane one whole 16-byte pixel, stored with a single buffer_store_dwordx4.
//
// Epilogue arithmetic (same reference citations as sesrq_dot4.hip), VALU-lean but bit-identical:
//   * the multiplication by 2^-n and the zero-point add are one v_fma_f32 (the product
//     prod * 2^-n is exact, so fma(prod, 2^-n, z) == fl(fl(prod * 2^-n) + z));
//   * ReLU and the int8 clamp are one v_med3_f32 on the un-rounded value (integer bounds and a
//     monotone rounding commute with the clamp; fl(max(t,0)+z) == max(fl(t+z), z));
//   * rint + float->int8 is one add of 1.5*2^23 (round-to-nearest-even into the low mantissa
//     bits) followed by a byte pick; mul / fma / magic-add run as packed v_pk_*_f32.
#include "sesrq_common.h"

namespace sesrq {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int MTW = 64;   // tile width : 4 waves x 16 pixels
constexpr int MTH = 16;   // tile height: rows walked by every wave (multiple of 4)
constexpr float MAGIC = 12582912.f;   // 1.5 * 2^23

// accumulate modes
enum { MERGED = 0, GEN_STD = 1, GEN_ANY = 2 };   // GEN_STD: 18/20-bit clamps as literals

__device__ __forceinline__ v4i mfma(v4i a, v4i b, v4i c) { return __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float med3(float v, float lo, float hi) { return __builtin_amdgcn_fmed3f(v, lo, hi); }
__device__ __forceinline__ int clampi3(int v, int lo, int hi) { return min(max(v, lo), hi); }
__device__ __forceinline__ v4i ld_frag(const int4 *p) { const int4 t = *p; v4i r = {t.x, t.y, t.z, t.w}; return r; }
__device__ __forceinline__ unsigned fbits(float f) { return __builtin_bit_cast(unsigned, f); }

// low bytes of four words -> one word
__device__ __forceinline__ unsigned pack_lo_bytes(unsigned y0, unsigned y1, unsigned y2, unsigned y3) {
    const unsigned w01 = __builtin_amdgcn_perm(y1, y0, 0x0c0c0400u);
    const unsigned w23 = __builtin_amdgcn_perm(y3, y2, 0x0c0c0400u);
    return __builtin_amdgcn_perm(w23, w01, 0x05040100u);
}

// v[i] = fl(fl(s[i] * M) * 2^-n + zadd)   (un-rounded, two values per packed op)
__device__ __forceinline__ void requant4(const int s[4], float Mf, float sh, float zadd, v2f &v01, v2f &v23) {
    const v2f M2 = {Mf, Mf}, sh2 = {sh, sh}, z2 = {zadd, zadd};
    const v2f f01 = {(float)s[0], (float)s[1]}, f23 = {(float)s[2], (float)s[3]};
    v01 = __builtin_elementwise_fma(f01 * M2, sh2, z2);
    v23 = __builtin_elementwise_fma(f23 * M2, sh2, z2);
}

// clamp (un-rounded) then round-half-even to int8, 4 values -> packed word
__device__ __forceinline__ unsigned round_pack(v2f v01, v2f v23, float lo, float hi) {
    const v2f mg = {MAGIC, MAGIC};
    v2f c01 = {med3(v01[0], lo, hi), med3(v01[1], lo, hi)}, c23 = {med3(v23[0], lo, hi), med3(v23[1], lo, hi)};
    c01 = c01 + mg; c23 = c23 + mg;
    return pack_lo_bytes(fbits(c01[0]), fbits(c01[1]), fbits(c23[0]), fbits(c23[1]));
}

// hidden layer: q = clamp8(rint(relu(t) + z_next))            (myQL/quan_func.py:280)
__device__ __forceinline__ unsigned epi_mid(const int s[4], const ConvArgs &a, float zlo) {
    v2f v01, v23;
    requant4(s, a.Mf, a.sh, a.z_next, v01, v23);
    return round_pack(v01, v23, zlo, 127.f);
}
// layer-0 residual operand rc = clamp8(rint(relu(t) - 128))    (myQL/quan_func.py:250)
__device__ __forceinline__ unsigned epi_rc(const int s[4], const ConvArgs &a) {
    v2f v01, v23;
    requant4(s, a.Mf, a.sh, -128.f, v01, v23);
    return round_pack(v01, v23, -128.f, 127.f);
}
// layer L-2: long residual merged in the integer domain        (myQL/quan_func.py:249-270)
__device__ __forceinline__ unsigned epi_preres(const int s[4], unsigned rcword, const ConvArgs &a) {
    v2f v01, v23;
    requant4(s, a.Mf, a.sh, -128.f, v01, v23);
    const unsigned rcx = rcword ^ 0x80808080u;                 // rc + 128 as unsigned bytes
    const v2f k128 = {128.f, 128.f};
    // ic = rint(clamp(t - 128)) ; u = rc + ic + 256 = (rc + 128) + ic + 128   (all exact small integers)
    const v2f i01 = {rintf(med3(v01[0], -128.f, 127.f)), rintf(med3(v01[1], -128.f, 127.f))};
    const v2f i23 = {rintf(med3(v23[0], -128.f, 127.f)), rintf(med3(v23[1], -128.f, 127.f))};
    const v2f r01 = {(float)(rcx & 0xffu), (float)((rcx >> 8) & 0xffu)}, r23 = {(float)((rcx >> 16) & 0xffu), (float)(rcx >> 24)};
    const v2f u01 = (r01 + k128) + i01, u23 = (r23 + k128) + i23;
    const v2f M2 = {a.Mres, a.Mres}, sh2 = {a.shres, a.shres}, z2 = {a.z_merge, a.z_merge};
    const v2f w01 = __builtin_elementwise_fma(u01 * M2, sh2, z2), w23 = __builtin_elementwise_fma(u23 * M2, sh2, z2);
    return round_pack(w01, w23, -128.f, 127.f);
}

// last layer: requantise into the output domain + PixelShuffle(r) store (int8 and/or fp32)
__device__ __forceinline__ void epi_last(const int s[4], const ConvArgs &a, int g, int n, int gy, int gx, float zlo) {
    const int r = a.ps, r2 = r * r, Ho = a.H * r, Wo = a.W * r, cout = a.oc / r2;
    v2f v01, v23;
    requant4(s, a.Mf, a.sh, a.z_out, v01, v23);
    const float v[4] = {v01[0], v01[1], v23[0], v23[1]};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int o = 4 * g + i;
        if (o < a.oc) {
            const float q = rintf(med3(v[i], zlo, 127.f));
            const int c = o / r2, rem = o - c * r2, ii = rem / r, jj = rem - ii * r;
            const size_t off = (((size_t)n * cout + c) * Ho + (size_t)gy * r + ii) * Wo + (size_t)gx * r + jj;
            if (a.out_q) reinterpret_cast<signed char *>(a.out_q)[off] = (signed char)(int)q;
            if (a.out_f) a.out_f[off] = __fmul_rn(q - a.z_out, a.s_out);
        }
    }
}

// PE clamp / sum / adder clamp / add constant               (myQL/quan_func.py:370,380-386,437,491)
template <int MODE>
__device__ __forceinline__ void finish_sums(int s[4], const v4i *acc, const int4 ac, const ConvArgs &a) {
    const int acv[4] = {ac.x, ac.y, ac.z, ac.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if constexpr (MODE == MERGED) {
            s[i] = acc[0][i];       // add constant already in the accumulator (C-in)
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

// 4x4 transpose between lane groups (16 lanes each) and registers; its own inverse.
// in : w[r] in lane (n, g) = word g of row r        out: w[g'] in lane (n, r') = word g' of row r'
__device__ __forceinline__ void transpose4(unsigned w[4]) {
    v2u t;
    t = __builtin_amdgcn_permlane32_swap(w[0], w[2], false, false); w[0] = t[0]; w[2] = t[1];
    t = __builtin_amdgcn_permlane32_swap(w[1], w[3], false, false); w[1] = t[0]; w[3] = t[1];
    t = __builtin_amdgcn_permlane16_swap(w[0], w[1], false, false); w[0] = t[0]; w[1] = t[1];
    t = __builtin_amdgcn_permlane16_swap(w[2], w[3], false, false); w[2] = t[0]; w[3] = t[1];
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
    transpose4(w);
    const v4u v = {w[0], w[1], w[2], w[3]};
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, io.voff, y4 * io.row_bytes, 0);
}

// stage a (SH x SW) window of NHWC16 pixels into LDS, pad word outside the frame.
// All loads of a thread are issued back to back (buffer loads: out-of-range -> 0, then the pad
// word is selected in), and only then written to LDS: one memory round trip per tile, not one
// per loop iteration.
template <int SH, int SW, int R>
__device__ __forceinline__ void stage_nhwc16(int4 *tile, const ConvArgs &a, int n_img, int x0, int y0, int tid) {
    constexpr int NIT = (SH * SW + 255) / 256;
    const size_t img = (size_t)a.H * a.W * 16;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((char *)const_cast<void *>(a.in) + (size_t)n_img * img, 0, (int)img, 0x00020000);
    v4u v[NIT];
    bool ok[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = tid + it * 256;
        const int ty = i / SW, tx = i - ty * SW;
        const int gy = y0 - R + ty, gx = x0 - R + tx;
        ok[it] = (gy >= 0) & (gy < a.H) & (gx >= 0) & (gx < a.W) & (i < SH * SW);
        v[it] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok[it] ? (gy * a.W + gx) * 16 : (int)0x80000000, 0, 0);
    }
    const unsigned pw = (unsigned)a.pad_word;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = tid + it * 256;
        const v4u pad = {pw, pw, pw, pw};
        const v4u t = ok[it] ? v[it] : pad;
        if (i < SH * SW) tile[i] = make_int4((int)t[0], (int)t[1], (int)t[2], (int)t[3]);
    }
}

// hidden-layer output of 4 rows: s4[r][i] -> requant -> transpose -> one 16-byte store per lane
template <int EPI, bool RC>
__device__ __forceinline__ void emit_rows4(const int s4[4][4], const ConvArgs &a, const RowIO &io, int y4, float zlo) {
    unsigned w[4];
    if constexpr (EPI == EPI_PRERES) {
        const v4u rv = __builtin_amdgcn_raw_buffer_load_b128(io.rc_in, io.voff, y4 * io.row_bytes, 0);
        unsigned rcw[4] = {rv[0], rv[1], rv[2], rv[3]};
        transpose4(rcw);
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = epi_preres(s4[r], rcw[r], a);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = epi_mid(s4[r], a, zlo);
    }
    store_rows4(io.out, io, y4, w);
    if constexpr (RC) {
        unsigned rw[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) rw[r] = epi_rc(s4[r], a);
        store_rows4(io.rc_out, io, y4, rw);
    }
}

// ------------------------------------------------------------------ hidden 3x3, 16 -> 16 channels
template <int MODE, int EPI>
__global__ __launch_bounds__(256) void mfma_h3_kernel(const ConvArgs a) {
    constexpr bool GENERAL = MODE != MERGED;
    constexpr int SW = MTW + 4;                      // 1 left halo + 64 + 1 right halo + over-read
    constexpr int SH = MTH + 2 + (GENERAL ? 1 : 0);  // general reads row y+3 with zero weights
    __shared__ int4 tile[SH * SW];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, n = l & 15, g = l >> 4;
    const int x0 = blockIdx.x * MTW, y0 = blockIdx.y * MTH, n_img = blockIdx.z;
    stage_nhwc16<SH, SW, 1>(tile, a, n_img, x0, y0, tid);
    const int4 *fr = a.afrag;
    const int4 ac = fr[g];
    v4i A[GENERAL ? 4 : 3];
#pragma unroll
    for (int f = 0; f < (GENERAL ? 4 : 3); ++f) A[f] = ld_frag(fr + 4 + f * 64 + l);
    const float zlo = a.relu ? fmaxf(a.z_next, -128.f) : -128.f;
    const RowIO io = make_rowio(a, n_img, y0, x0 + 16 * w + n, g);
    __syncthreads();
    if constexpr (!GENERAL) {
        const int col = 16 * w + n + g;
        const v4i acc0 = {ac.x, ac.y, ac.z, ac.w};
        v4i B0 = ld_frag(tile + col), B1 = ld_frag(tile + SW + col);
#pragma unroll
        for (int y4 = 0; y4 < MTH; y4 += 4) {
            int s4[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const v4i B2 = ld_frag(tile + (y4 + r + 2) * SW + col);
                v4i acc[1];
                acc[0] = mfma(A[0], B0, acc0);
                acc[0] = mfma(A[1], B1, acc[0]);
                acc[0] = mfma(A[2], B2, acc[0]);
                B0 = B1; B1 = B2;
                finish_sums<MERGED>(s4[r], acc, ac, a);
            }
            emit_rows4<EPI, false>(s4, a, io, y4, zlo);
        }
    } else {
        const int col = 16 * w + n;
#pragma unroll 1
        for (int y4 = 0; y4 < MTH; y4 += 4) {
            int s4[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int4 *row = tile + (y4 + r + g) * SW + col;     // lane group g = kernel row ky
                const int4 P0 = row[0], P1 = row[1], P2 = row[2], P3 = row[3];
                const v4i zero = {0, 0, 0, 0};
                v4i acc[4];
                { const v4i b = {P0.x, P1.x, P2.x, P3.x}; acc[0] = mfma(A[0], b, zero); }
                { const v4i b = {P0.y, P1.y, P2.y, P3.y}; acc[1] = mfma(A[1], b, zero); }
                { const v4i b = {P0.z, P1.z, P2.z, P3.z}; acc[2] = mfma(A[2], b, zero); }
                { const v4i b = {P0.w, P1.w, P2.w, P3.w}; acc[3] = mfma(A[3], b, zero); }
                finish_sums<MODE>(s4[r], acc, ac, a);
            }
            emit_rows4<EPI, false>(s4, a, io, y4, zlo);
        }
    }
}

// ------------------------------------------------------------------ 5x5, 16 input channels
template <int MODE, int EPI>
__global__ __launch_bounds__(256) void mfma_h5_kernel(const ConvArgs a) {
    constexpr bool GENERAL = MODE != MERGED;
    constexpr int SW = MTW + 8;          // 2 + 64 + 2 halo, + over-read of the kx = 4..7 group
    constexpr int SH = MTH + 4;
    __shared__ int4 tile[SH * SW];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, n = l & 15, g = l >> 4;
    const int x0 = blockIdx.x * MTW, y0 = blockIdx.y * MTH, n_img = blockIdx.z;
    stage_nhwc16<SH, SW, 2>(tile, a, n_img, x0, y0, tid);
    const int4 *fr = a.afrag;
    const int4 ac = fr[g];
    const float zlo = a.relu ? fmaxf(EPI == EPI_LAST ? a.z_out : a.z_next, -128.f) : -128.f;
    const int gx = x0 + 16 * w + n;
    RowIO io;
    if constexpr (EPI != EPI_LAST) io = make_rowio(a, n_img, y0, gx, g);
    if constexpr (!GENERAL) {
        // frag (ky, h): lane group g = tap kx = 4h + g  (h = 1: only kx = 4 carries weights)
        v4i A[5][2];
#pragma unroll
        for (int ky = 0; ky < 5; ++ky)
#pragma unroll
            for (int h = 0; h < 2; ++h) A[ky][h] = ld_frag(fr + 4 + (ky * 2 + h) * 64 + l);
        __syncthreads();
        const int col = 16 * w + n + g;
        const v4i acc0 = {ac.x, ac.y, ac.z, ac.w};
        v4i B[5][2];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            B[r][0] = ld_frag(tile + r * SW + col);
            B[r][1] = ld_frag(tile + r * SW + col + 4);
        }
#pragma unroll
        for (int y4 = 0; y4 < MTH; y4 += 4) {
            int s4[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int y = y4 + r;
                B[(y + 4) % 5][0] = ld_frag(tile + (y + 4) * SW + col);
                B[(y + 4) % 5][1] = ld_frag(tile + (y + 4) * SW + col + 4);
                v4i acc[1];
                acc[0] = acc0;
#pragma unroll
                for (int ky = 0; ky < 5; ++ky) {
                    acc[0] = mfma(A[ky][0], B[(y + ky) % 5][0], acc[0]);
                    acc[0] = mfma(A[ky][1], B[(y + ky) % 5][1], acc[0]);
                }
                finish_sums<MERGED>(s4[r], acc, ac, a);
                if constexpr (EPI == EPI_LAST) {
                    if (y0 + y < a.H && gx < a.W) epi_last(s4[r], a, g, n_img, y0 + y, gx, zlo);
                }
            }
            if constexpr (EPI != EPI_LAST) emit_rows4<EPI, false>(s4, a, io, y4, zlo);
        }
    } else {
        // per PE p two K-chunks:  f = 0: group g = ky 0..3, words = kx 0..3
        //                         f = 1: g0 = (ky 4, kx 0..3)  g1 = (ky 0..3, kx 4)  g2 = (4,4)  g3 = none
        v4i A[2][4];
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int p = 0; p < 4; ++p) A[f][p] = ld_frag(fr + 4 + (f * 4 + p) * 64 + l);
        __syncthreads();
        const int col = 16 * w + n;
        int off1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) off1[i] = (g == 0) ? 4 * SW + i : (g == 1 ? i * SW + 4 : 4 * SW + 4);
#pragma unroll 1
        for (int y4 = 0; y4 < MTH; y4 += 4) {
            int s4[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int y = y4 + r;
                const v4i zero = {0, 0, 0, 0};
                v4i acc[4];
                {
                    const int4 *row = tile + (y + g) * SW + col;
                    const int4 P0 = row[0], P1 = row[1], P2 = row[2], P3 = row[3];
                    { const v4i b = {P0.x, P1.x, P2.x, P3.x}; acc[0] = mfma(A[0][0], b, zero); }
                    { const v4i b = {P0.y, P1.y, P2.y, P3.y}; acc[1] = mfma(A[0][1], b, zero); }
                    { const v4i b = {P0.z, P1.z, P2.z, P3.z}; acc[2] = mfma(A[0][2], b, zero); }
                    { const v4i b = {P0.w, P1.w, P2.w, P3.w}; acc[3] = mfma(A[0][3], b, zero); }
                }
                {
                    const int4 *base = tile + y * SW + col;
                    const int4 P0 = base[off1[0]], P1 = base[off1[1]], P2 = base[off1[2]], P3 = base[off1[3]];
                    { const v4i b = {P0.x, P1.x, P2.x, P3.x}; acc[0] = mfma(A[1][0], b, acc[0]); }
                    { const v4i b = {P0.y, P1.y, P2.y, P3.y}; acc[1] = mfma(A[1][1], b, acc[1]); }
                    { const v4i b = {P0.z, P1.z, P2.z, P3.z}; acc[2] = mfma(A[1][2], b, acc[2]); }
                    { const v4i b = {P0.w, P1.w, P2.w, P3.w}; acc[3] = mfma(A[1][3], b, acc[3]); }
                }
                finish_sums<MODE>(s4[r], acc, ac, a);
                if constexpr (EPI == EPI_LAST) {
                    if (y0 + y < a.H && gx < a.W) epi_last(s4[r], a, g, n_img, y0 + y, gx, zlo);
                }
            }
            if constexpr (EPI != EPI_LAST) emit_rows4<EPI, false>(s4, a, io, y4, zlo);
        }
    }
}

// ------------------------------------------------------------------ first layer 5x5, IC <= 4
// The frame is quantised while it is staged (q0 = clamp8(rint(x/s0 + z0)), quan_func.py:225);
// a pixel is one dword (byte c = channel c).  A lane's 16 bytes = 4 horizontally adjacent
// pixels, which start at an arbitrary pixel column -> the tile is kept in 4 copies shifted by
// 0..3 pixels so that every such group is one aligned ds_read_b128.
template <int MODE, int SRC, bool RC>
__global__ __launch_bounds__(256) void mfma_f5_kernel(const ConvArgs a) {
    constexpr bool GENERAL = MODE != MERGED;
    constexpr int SH = MTH + 4;
    constexpr int SWP = MTW + 8;         // staged pixel columns (2 halo + 64 + 2 halo + over-read)
    constexpr int SU = SWP / 4;          // 16-byte units per row per copy
    __shared__ int4 cp[4 * SH * SU];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, n = l & 15, g = l >> 4;
    const int x0 = blockIdx.x * MTW, y0 = blockIdx.y * MTH, n_img = blockIdx.z;
    const size_t HW = (size_t)a.H * a.W;
    int *cpw = reinterpret_cast<int *>(cp);
    {
        // all frame loads of a thread first (buffer loads, out-of-range -> 0), then quantise + LDS writes
        constexpr int NIT = (SH * SWP + 255) / 256;
        const size_t esz = (SRC == SRC_F32) ? 4 : 1;
        const size_t img = HW * a.ic * esz;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((char *)const_cast<void *>(a.in) + (size_t)n_img * img, 0, (int)img, 0x00020000);
        unsigned raw[NIT][4];
        bool ok[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            const int ty = i / SWP, tx = i - ty * SWP;
            const int gy = y0 - 2 + ty, gx = x0 - 2 + tx;
            ok[it] = (gy >= 0) & (gy < a.H) & (gx >= 0) & (gx < a.W) & (i < SH * SWP);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int off = (ok[it] && c < a.ic) ? (int)((c * (int)HW + gy * a.W + gx) * esz) : (int)0x80000000;
                if constexpr (SRC == SRC_F32) raw[it][c] = __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0);
                else raw[it][c] = (unsigned)(int)(signed char)__builtin_amdgcn_raw_buffer_load_b8(rs, off, 0, 0);
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            const int ty = i / SWP, tx = i - ty * SWP;
            int word = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                int q;
                if constexpr (SRC == SRC_F32)
                    q = (int)med3(rintf(__fadd_rn(__fdiv_rn(__builtin_bit_cast(float, raw[it][c]), a.s_in), a.z_in)), -128.f, 127.f);
                else
                    q = (int)raw[it][c];
                if (c < a.ic) word |= (q & 0xff) << (8 * c);
            }
            if (!ok[it]) word = a.pad_word;
            if (i < SH * SWP) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int t = tx - s;
                    if (t >= 0) cpw[((s * SH + ty) * SU + (t >> 2)) * 4 + (t & 3)] = word;
                }
            }
        }
    }
    const int4 *fr = a.afrag;
    const int4 ac = fr[g];
    constexpr int NPE = GENERAL ? 4 : 1;
    v4i A[3][NPE];
#pragma unroll
    for (int f = 0; f < 3; ++f)
#pragma unroll
        for (int p = 0; p < NPE; ++p) A[f][p] = ld_frag(fr + 4 + (f * NPE + p) * 64 + l);
    // lane group -> (kernel row, 4-pixel segment) per K-chunk; must match pack_mfma_frags (MFMA_F5)
    //   f0: (g,0)      f1: (4,0) (0,1) (1,1) (2,1)      f2: (3,1) (4,1) - -
    int addr[3];
    {
        const int rowofs[3] = {g, g == 0 ? 4 : g - 1, g == 0 ? 3 : (g == 1 ? 4 : 0)};
        const int seg[3] = {0, g == 0 ? 0 : 1, g < 2 ? 1 : 0};
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            const int c0 = 16 * w + n + 4 * seg[f];
            addr[f] = ((c0 & 3) * SH + rowofs[f]) * SU + (c0 >> 2);
        }
    }
    const float zlo = a.relu ? fmaxf(a.z_next, -128.f) : -128.f;
    const int gx = x0 + 16 * w + n;
    const RowIO io = make_rowio(a, n_img, y0, gx, g);
    __syncthreads();
#pragma unroll 1
    for (int y4 = 0; y4 < MTH; y4 += 4) {
        int s4[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int y = y4 + r;
            const v4i B0 = ld_frag(cp + addr[0] + y * SU), B1 = ld_frag(cp + addr[1] + y * SU), B2 = ld_frag(cp + addr[2] + y * SU);
            v4i acc[GENERAL ? 4 : 1];
            if constexpr (GENERAL) {
                const v4i zero = {0, 0, 0, 0};
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    acc[p] = mfma(A[0][p], B0, zero);
                    acc[p] = mfma(A[1][p], B1, acc[p]);
                    acc[p] = mfma(A[2][p], B2, acc[p]);
                }
            } else {
                const v4i acc0 = {ac.x, ac.y, ac.z, ac.w};
                acc[0] = mfma(A[0][0], B0, acc0);
                acc[0] = mfma(A[1][0], B1, acc[0]);
                acc[0] = mfma(A[2][0], B2, acc[0]);
            }
            finish_sums<MODE>(s4[r], acc, ac, a);
            if (a.dbg_q0 && g == 0 && y0 + y < a.H && gx < a.W) {
                const int word = cpw[((0 * SH + y + 2) * SU + ((16 * w + n + 2) >> 2)) * 4 + ((16 * w + n + 2) & 3)];
                for (int c = 0; c < a.ic; ++c)
                    a.dbg_q0[((size_t)n_img * a.ic + c) * HW + (size_t)(y0 + y) * a.W + gx] = (signed char)((word >> (8 * c)) & 0xff);
            }
        }
        emit_rows4<EPI_MID, RC>(s4, a, io, y4, zlo);
    }
}

template <typename K>
static void launch(K kern, const ConvArgs &a, hipStream_t st) {
    dim3 grid((a.W + MTW - 1) / MTW, (a.H + MTH - 1) / MTH, a.N);
    hipLaunchKernelGGL(kern, grid, dim3(256), 0, st, a);
}

#define SESRQ_BY_MODE(KERN, ...)                                                         \
    do {                                                                                 \
        if (mode == MERGED) launch(KERN<MERGED, __VA_ARGS__>, a, st);                    \
        else if (mode == GEN_STD) launch(KERN<GEN_STD, __VA_ARGS__>, a, st);             \
        else launch(KERN<GEN_ANY, __VA_ARGS__>, a, st);                                  \
    } while (0)

int launch_mfma(const LayerPlan &lp, const ConvArgs &a, int src, int epi, bool general, hipStream_t st) {
    if ((size_t)a.H * a.W * 16 >= ((size_t)1 << 31)) { set_error("mfma: frame too large for 32-bit buffer offsets"); return 1; }
    const bool std_bits = a.acc_lo == -131072 && a.acc_hi == 131071 && a.add_lo == -524288 && a.add_hi == 524287;
    const int mode = !general ? MERGED : (std_bits ? GEN_STD : GEN_ANY);
    switch (lp.mfma_kind) {
        case MFMA_H3:
            if (epi == EPI_MID) SESRQ_BY_MODE(mfma_h3_kernel, EPI_MID);
            else if (epi == EPI_PRERES) SESRQ_BY_MODE(mfma_h3_kernel, EPI_PRERES);
            else { set_error("mfma: 3x3 last layer not supported"); return 1; }
            break;
        case MFMA_H5:
            if (epi == EPI_MID) SESRQ_BY_MODE(mfma_h5_kernel, EPI_MID);
            else if (epi == EPI_PRERES) SESRQ_BY_MODE(mfma_h5_kernel, EPI_PRERES);
            else SESRQ_BY_MODE(mfma_h5_kernel, EPI_LAST);
            break;
        case MFMA_F5:
            if (src == SRC_F32) { if (a.rc_out) SESRQ_BY_MODE(mfma_f5_kernel, SRC_F32, true); else SESRQ_BY_MODE(mfma_f5_kernel, SRC_F32, false); }
            else { if (a.rc_out) SESRQ_BY_MODE(mfma_f5_kernel, SRC_I8, true); else SESRQ_BY_MODE(mfma_f5_kernel, SRC_I8, false); }
            break;
        default: set_error("mfma: layer shape not supported by the MFMA engine"); return 1;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("mfma launch failed: ") + hipGetErrorString(e)); return 1; }
    return 0;
}

}  // namespace sesrq
