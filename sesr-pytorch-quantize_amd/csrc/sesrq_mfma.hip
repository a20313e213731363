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
// NHWC16 (PE-major) output pixel: the epilogue packs them and stores one dword per lane.
// Epilogue arithmetic = sesrq_dot4.hip (same reference citations); the two multiplications by
// an exact power of two are folded into v_fma_f32, which is bit-identical (exact product).
#include "sesrq_common.h"

namespace sesrq {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int MTW = 64;   // tile width : 4 waves x 16 pixels
constexpr int MTH = 16;   // tile height: rows walked by every wave

__device__ __forceinline__ v4i mfma(v4i a, v4i b, v4i c) { return __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float med3(float v, float lo, float hi) { return __builtin_amdgcn_fmed3f(v, lo, hi); }
__device__ __forceinline__ int clampi3(int v, int lo, int hi) { return min(max(v, lo), hi); }
__device__ __forceinline__ v4i ld_frag(const int4 *p) { const int4 t = *p; v4i r = {t.x, t.y, t.z, t.w}; return r; }

__device__ __forceinline__ int pack4(const float q[4]) {
    return ((int)q[0] & 0xff) | (((int)q[1] & 0xff) << 8) | (((int)q[2] & 0xff) << 16) | ((int)q[3] << 24);
}

// s[i] = 20-bit-clamped PE sum + add constant for output slots 4g+i of one pixel.
// hidden layer, requantise into the next domain:  q = clamp8(rint(relu(t) + z))
__device__ __forceinline__ int epi_mid(const int s[4], const ConvArgs &a, float zlo, int *rcword) {
    float q[4], rc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float prod = __fmul_rn((float)s[i], a.Mf);
        q[i] = med3(rintf(__builtin_fmaf(prod, a.sh, a.z_next)), zlo, 127.f);
        rc[i] = med3(rintf(__builtin_fmaf(prod, a.sh, -128.f)), -128.f, 127.f);
    }
    if (rcword) *rcword = pack4(rc);
    return pack4(q);
}

// layer L-2: long residual merged in the integer domain (myQL/quan_func.py:249-270)
__device__ __forceinline__ int epi_preres(const int s[4], int rcword, const ConvArgs &a) {
    const unsigned rcx = (unsigned)rcword ^ 0x80808080u;      // bytes + 128 -> unsigned
    float q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float prod = __fmul_rn((float)s[i], a.Mf);
        const float ic = med3(rintf(__builtin_fmaf(prod, a.sh, -128.f)), -128.f, 127.f);
        const float u = (float)((rcx >> (8 * i)) & 0xffu) + ic + 128.f;   // rc + ic + 256
        const float pu = __fmul_rn(u, a.Mres);
        q[i] = med3(rintf(__builtin_fmaf(pu, a.shres, a.z_merge)), -128.f, 127.f);
    }
    return pack4(q);
}

// last layer: requantise into the output domain + PixelShuffle(r) store (int8 and/or fp32)
__device__ __forceinline__ void epi_last(const int s[4], const ConvArgs &a, int g, int n, int gy, int gx, float zlo) {
    const int r = a.ps, r2 = r * r, Ho = a.H * r, Wo = a.W * r, cout = a.oc / r2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int o = 4 * g + i;
        if (o < a.oc) {
            const float prod = __fmul_rn((float)s[i], a.Mf);
            const float q = med3(rintf(__builtin_fmaf(prod, a.sh, a.z_out)), zlo, 127.f);
            const int c = o / r2, rem = o - c * r2, ii = rem / r, jj = rem - ii * r;
            const size_t off = (((size_t)n * cout + c) * Ho + (size_t)gy * r + ii) * Wo + (size_t)gx * r + jj;
            if (a.out_q) reinterpret_cast<signed char *>(a.out_q)[off] = (signed char)(int)q;
            if (a.out_f) a.out_f[off] = __fmul_rn(q - a.z_out, a.s_out);
        }
    }
}

template <bool GENERAL>
__device__ __forceinline__ void finish_sums(int s[4], const v4i acc[GENERAL ? 4 : 1], const int4 ac, const ConvArgs &a) {
    const int acv[4] = {ac.x, ac.y, ac.z, ac.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if constexpr (GENERAL) {
            const int t = clampi3(acc[0][i], a.acc_lo, a.acc_hi) + clampi3(acc[1][i], a.acc_lo, a.acc_hi) +
                          clampi3(acc[2][i], a.acc_lo, a.acc_hi) + clampi3(acc[3][i], a.acc_lo, a.acc_hi);
            s[i] = clampi3(t, a.add_lo, a.add_hi) + acv[i];
        } else {
            s[i] = acc[0][i];       // add constant already in the accumulator (C-in)
        }
    }
}

template <int EPI>
__device__ __forceinline__ void store_pixel(const int s[4], const ConvArgs &a, int n_img, int gy, int gx, int g, float zlo) {
    const size_t pix = ((size_t)n_img * a.H + gy) * a.W + gx;
    if constexpr (EPI == EPI_MID) {
        int rcw;
        const int w = epi_mid(s, a, zlo, a.rc_out ? &rcw : nullptr);
        reinterpret_cast<int *>(a.out)[pix * 4 + g] = w;
        if (a.rc_out) reinterpret_cast<int *>(a.rc_out)[pix * 4 + g] = rcw;
    } else if constexpr (EPI == EPI_PRERES) {
        const int rcw = reinterpret_cast<const int *>(a.rc_in)[pix * 4 + g];
        reinterpret_cast<int *>(a.out)[pix * 4 + g] = epi_preres(s, rcw, a);
    } else {
        epi_last(s, a, g, n_img, gy, gx, zlo);
    }
}

// stage a (SH x SW) window of NHWC16 pixels into LDS, pad word outside the frame
template <int SH, int SW, int R>
__device__ __forceinline__ void stage_nhwc16(int4 *tile, const ConvArgs &a, int n_img, int x0, int y0, int tid) {
    const size_t HW = (size_t)a.H * a.W;
    const int4 *src = reinterpret_cast<const int4 *>(a.in) + (size_t)n_img * HW;
    for (int i = tid; i < SH * SW; i += 256) {
        const int ty = i / SW, tx = i - ty * SW;
        const int gy = y0 - R + ty, gx = x0 - R + tx;
        int4 v = make_int4(a.pad_word, a.pad_word, a.pad_word, a.pad_word);
        if ((gy >= 0) & (gy < a.H) & (gx >= 0) & (gx < a.W)) v = src[(size_t)gy * a.W + gx];
        tile[i] = v;
    }
}

// ------------------------------------------------------------------ hidden 3x3, 16 -> 16 channels
template <bool GENERAL, int EPI>
__global__ __launch_bounds__(256) void mfma_h3_kernel(const ConvArgs a) {
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
    __syncthreads();
    const int gx = x0 + 16 * w + n;
    if constexpr (!GENERAL) {
        const int col = 16 * w + n + g;
        const v4i acc0 = {ac.x, ac.y, ac.z, ac.w};
        v4i B0 = ld_frag(tile + col), B1 = ld_frag(tile + SW + col);
#pragma unroll
        for (int y = 0; y < MTH; ++y) {
            const v4i B2 = ld_frag(tile + (y + 2) * SW + col);
            v4i acc[1];
            acc[0] = mfma(A[0], B0, acc0);
            acc[0] = mfma(A[1], B1, acc[0]);
            acc[0] = mfma(A[2], B2, acc[0]);
            B0 = B1; B1 = B2;
            const int gy = y0 + y;
            if (gy < a.H && gx < a.W) {
                int s[4];
                finish_sums<false>(s, acc, ac, a);
                store_pixel<EPI>(s, a, n_img, gy, gx, g, zlo);
            }
        }
    } else {
        const int col = 16 * w + n;
#pragma unroll 4
        for (int y = 0; y < MTH; ++y) {
            const int4 *row = tile + (y + g) * SW + col;     // lane group g = kernel row ky
            const int4 P0 = row[0], P1 = row[1], P2 = row[2], P3 = row[3];
            const v4i zero = {0, 0, 0, 0};
            v4i acc[4];
            { const v4i b = {P0.x, P1.x, P2.x, P3.x}; acc[0] = mfma(A[0], b, zero); }
            { const v4i b = {P0.y, P1.y, P2.y, P3.y}; acc[1] = mfma(A[1], b, zero); }
            { const v4i b = {P0.z, P1.z, P2.z, P3.z}; acc[2] = mfma(A[2], b, zero); }
            { const v4i b = {P0.w, P1.w, P2.w, P3.w}; acc[3] = mfma(A[3], b, zero); }
            const int gy = y0 + y;
            if (gy < a.H && gx < a.W) {
                int s[4];
                finish_sums<true>(s, acc, ac, a);
                store_pixel<EPI>(s, a, n_img, gy, gx, g, zlo);
            }
        }
    }
}

// ------------------------------------------------------------------ 5x5, 16 input channels
template <bool GENERAL, int EPI>
__global__ __launch_bounds__(256) void mfma_h5_kernel(const ConvArgs a) {
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
        for (int y = 0; y < MTH; ++y) {
            B[(y + 4) % 5][0] = ld_frag(tile + (y + 4) * SW + col);
            B[(y + 4) % 5][1] = ld_frag(tile + (y + 4) * SW + col + 4);
            v4i acc[1];
            acc[0] = acc0;
#pragma unroll
            for (int ky = 0; ky < 5; ++ky) {
                acc[0] = mfma(A[ky][0], B[(y + ky) % 5][0], acc[0]);
                acc[0] = mfma(A[ky][1], B[(y + ky) % 5][1], acc[0]);
            }
            const int gy = y0 + y;
            if (gy < a.H && gx < a.W) {
                int s[4];
                finish_sums<false>(s, acc, ac, a);
                store_pixel<EPI>(s, a, n_img, gy, gx, g, zlo);
            }
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
#pragma unroll 2
        for (int y = 0; y < MTH; ++y) {
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
            const int gy = y0 + y;
            if (gy < a.H && gx < a.W) {
                int s[4];
                finish_sums<true>(s, acc, ac, a);
                store_pixel<EPI>(s, a, n_img, gy, gx, g, zlo);
            }
        }
    }
}

// ------------------------------------------------------------------ first layer 5x5, IC <= 4
// The frame is quantised while it is staged (q0 = clamp8(rint(x/s0 + z0)), quan_func.py:225);
// a pixel is one dword (byte c = channel c).  A lane's 16 bytes = 4 horizontally adjacent
// pixels, which start at an arbitrary pixel column -> the tile is kept in 4 copies shifted by
// 0..3 pixels so that every such group is one aligned ds_read_b128.
template <bool GENERAL, int SRC>
__global__ __launch_bounds__(256) void mfma_f5_kernel(const ConvArgs a) {
    constexpr int SH = MTH + 4;
    constexpr int SWP = MTW + 8;         // staged pixel columns (2 halo + 64 + 2 halo + over-read)
    constexpr int SU = SWP / 4;          // 16-byte units per row per copy
    __shared__ int4 cp[4 * SH * SU];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, n = l & 15, g = l >> 4;
    const int x0 = blockIdx.x * MTW, y0 = blockIdx.y * MTH, n_img = blockIdx.z;
    const size_t HW = (size_t)a.H * a.W;
    int *cpw = reinterpret_cast<int *>(cp);
    for (int i = tid; i < SH * SWP; i += 256) {
        const int ty = i / SWP, tx = i - ty * SWP;
        const int gy = y0 - 2 + ty, gx = x0 - 2 + tx;
        int word = a.pad_word;
        if ((gy >= 0) & (gy < a.H) & (gx >= 0) & (gx < a.W)) {
            word = 0;
            for (int c = 0; c < a.ic; ++c) {
                const size_t off = ((size_t)n_img * a.ic + c) * HW + (size_t)gy * a.W + gx;
                int q;
                if constexpr (SRC == SRC_F32)
                    q = (int)med3(rintf(__fadd_rn(__fdiv_rn(reinterpret_cast<const float *>(a.in)[off], a.s_in), a.z_in)), -128.f, 127.f);
                else
                    q = reinterpret_cast<const signed char *>(a.in)[off];
                word |= (q & 0xff) << (8 * c);
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int t = tx - s;
            if (t >= 0) cpw[((s * SH + ty) * SU + (t >> 2)) * 4 + (t & 3)] = word;
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
    // lane group -> (kernel row, 4-pixel segment) per K-chunk; must match pack_mfma_frags (KIND_F5)
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
    __syncthreads();
    const int gx = x0 + 16 * w + n;
#pragma unroll 4
    for (int y = 0; y < MTH; ++y) {
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
        const int gy = y0 + y;
        if (gy < a.H && gx < a.W) {
            int s[4];
            finish_sums<GENERAL>(s, acc, ac, a);
            store_pixel<EPI_MID>(s, a, n_img, gy, gx, g, zlo);
            if (a.dbg_q0 && g == 0) {
                const int word = cpw[((0 * SH + y + 2) * SU + ((16 * w + n + 2) >> 2)) * 4 + ((16 * w + n + 2) & 3)];
                for (int c = 0; c < a.ic; ++c)
                    a.dbg_q0[((size_t)n_img * a.ic + c) * HW + (size_t)gy * a.W + gx] = (signed char)((word >> (8 * c)) & 0xff);
            }
        }
    }
}

template <typename K>
static void launch(K kern, const ConvArgs &a, hipStream_t st) {
    dim3 grid((a.W + MTW - 1) / MTW, (a.H + MTH - 1) / MTH, a.N);
    hipLaunchKernelGGL(kern, grid, dim3(256), 0, st, a);
}

int launch_mfma(const LayerPlan &lp, const ConvArgs &a, int src, int epi, bool general, hipStream_t st) {
    switch (lp.mfma_kind) {
        case MFMA_H3:
            if (epi == EPI_MID) general ? launch(mfma_h3_kernel<true, EPI_MID>, a, st) : launch(mfma_h3_kernel<false, EPI_MID>, a, st);
            else if (epi == EPI_PRERES) general ? launch(mfma_h3_kernel<true, EPI_PRERES>, a, st) : launch(mfma_h3_kernel<false, EPI_PRERES>, a, st);
            else { set_error("mfma: 3x3 last layer not supported"); return 1; }
            break;
        case MFMA_H5:
            if (epi == EPI_MID) general ? launch(mfma_h5_kernel<true, EPI_MID>, a, st) : launch(mfma_h5_kernel<false, EPI_MID>, a, st);
            else if (epi == EPI_PRERES) general ? launch(mfma_h5_kernel<true, EPI_PRERES>, a, st) : launch(mfma_h5_kernel<false, EPI_PRERES>, a, st);
            else general ? launch(mfma_h5_kernel<true, EPI_LAST>, a, st) : launch(mfma_h5_kernel<false, EPI_LAST>, a, st);
            break;
        case MFMA_F5:
            if (src == SRC_F32) general ? launch(mfma_f5_kernel<true, SRC_F32>, a, st) : launch(mfma_f5_kernel<false, SRC_F32>, a, st);
            else general ? launch(mfma_f5_kernel<true, SRC_I8>, a, st) : launch(mfma_f5_kernel<false, SRC_I8>, a, st);
            break;
        default: set_error("mfma: layer shape not supported by the MFMA engine"); return 1;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("mfma launch failed: ") + hipGetErrorString(e)); return 1; }
    return 0;
}

}  // namespace sesrq
