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
//   hybrid  (exactly one PE can saturate): the merged chain over the other three PEs + that PE's chain.
//   general (per-PE sums must be clamped separately, myQL/quan_func.py:370): a lane's 16
//           bytes = 4 taps x the 4 channels of ONE PE; one MFMA chain per PE; the kernels stage a
//           PE-planar LDS image so that the operand is four plain dword reads.
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
#include <algorithm>
#include <map>
#include <mutex>

#include "sesrq_mfma_common.h"

namespace sesrq {

constexpr int MTW = 64;   // tile width : 4 waves x 16 pixels
constexpr int MTH = 8;    // tile height: rows walked by every wave (multiple of 4); 8 beats 12 and 16 on 1080p (latency hiding vs halo)

// last layer: requantise into the output domain + PixelShuffle(r) store (int8 and/or fp32).
// A lane owns output slots o = 4g..4g+3 of pixel (gy, gx); everything that depends only on the lane
// (channel / sub-pixel decode, column offset, validity) is worked out once per kernel, a row adds one
// scalar offset.  PixelShuffle(2) makes the 4 slots two 2-byte runs, PixelShuffle(4) one 4-byte run.
struct LastStore {
    __amdgpu_buffer_rsrc_t rq, rf;
    int vo[4];            // element offset of slot i inside image n_img (out of range if invalid)
    int va[4];            // anchor add: element offset of the slot's input pixel inside frame n_img (channel plane + column)
    const float *anc;
    __amdgpu_buffer_rsrc_t ra;      // OUTF == 2: the image's fp32 input frame (the x2 anchor), one descriptor over its Cin planes
    int r, row_elems;     // PixelShuffle factor, r * Wo
    // lane_row: the lane's output row relative to the row passed to store() (pe-split kernel: lane group = row)
    // NV: real accumulator rows per lane group (last_slot_oc, sesrq_common.h); slot i >= NV of a lane is padding
    template <int R, int NV>
    __device__ __forceinline__ void init_r(const ConvArgs &a, int n_img, int g, int gx, int lane_row) {
        r = R;
        constexpr int r2 = R * R;
        const int Ho = a.H * R, Wo = a.W * R, cout = a.oc / r2;
        const size_t img = (size_t)cout * Ho * Wo;
        // image n_img of the launch: the n_img-th slice of one batch, or a frame buffer of its own (ConvArgs::ft, sesrq_forward_many)
        char *bq = a.ft.n ? (char *)a.ft.out_q[n_img] : (char *)a.out_q + (size_t)n_img * img;
        char *bf = a.ft.n ? (char *)a.ft.out_f[n_img] : (char *)a.out_f + (size_t)n_img * img * 4;
        rq = __builtin_amdgcn_make_buffer_rsrc(bq, 0, a.out_q ? (int)img : 0, 0x00020000);
        rf = __builtin_amdgcn_make_buffer_rsrc(bf, 0, a.out_f ? (int)(img * 4) : 0, 0x00020000);
        row_elems = __builtin_amdgcn_readfirstlane(R * Wo);      // pinned to an SGPR: the row's store offset is then scalar arithmetic
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = last_slot_oc(NV, g, i, a.oc, R);
            const int c = o / r2, rem = o - c * r2, ii = rem / R, jj = rem - ii * R;
            vo[i] = (o < a.oc && gx < a.W) ? (c * Ho + ii + lane_row * R) * Wo + gx * R + jj : (int)0x10000000;   // stays out of range times 4
            va[i] = (o < a.oc && gx < a.W) ? c * a.H * a.W + gx + lane_row * a.W : 0;
        }
        anc = a.anchor ? (a.ft.n ? (const float *)a.ft.in[n_img] : a.anchor + (size_t)n_img * cout * a.H * a.W) : nullptr;
        ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(anc), 0, anc ? (int)((size_t)cout * a.H * a.W * 4) : 0, 0x00020000);
    }
    // OUTF == 2 (the x2 anchor add, test.py:148-155): the two input pixels this lane's three values of row gy add -- slots 0 / 1 share one
    // (channel slot/4, same pixel: the pair is two sub-pixels of it), slot 2 the other.  Issued BEFORE the row's MFMA chain; a row below the
    // frame reads out of range = 0.0f (its stores are dropped anyway).
    __device__ __forceinline__ void fetch_anchor(const ConvArgs &a, int gy, float &x01, float &x2) const {
        const int so = __builtin_amdgcn_readfirstlane(gy * a.W * 4);
        x01 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, va[0] * 4, so, 0));
        x2 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, va[2] * 4, so, 0));
    }
    // the shuffle factor is a constant in each branch: the slot decode costs shifts, not integer divisions
    // FAST = 2 / 4: the kernel instance is built for that shuffle factor (no switch, no dead arms in the prologue)
    template <int NV = 4, int FAST = 0>
    __device__ __forceinline__ void init(const ConvArgs &a, int n_img, int g, int gx, int lane_row = 0) {
        if constexpr (FAST != 0) { init_r<FAST, NV>(a, n_img, g, gx, lane_row); return; }
        switch (a.ps) {
            case 1: init_r<1, NV>(a, n_img, g, gx, lane_row); break;
            case 2: init_r<2, NV>(a, n_img, g, gx, lane_row); break;
            case 3: init_r<3, NV>(a, n_img, g, gx, lane_row); break;
            default: init_r<4, NV>(a, n_img, g, gx, lane_row); break;
        }
    }
    // gy: wave-uniform row; row_ok: false drops this lane's stores (lanes whose own row gy + lane_row is below the frame)
    // FAST = 2 / 4: PixelShuffle factor known at compile time, int8 output only, row_ok wave-uniform: no output-kind /
    // shuffle-factor / row branches and no per-row offset selects in the hot loop of the SESR last layer
    // NV = 3: the lane's three real values are s[0..2] (s[3] is the padding row: never read).  FAST = 2 with NV = 3 is the
    // pair map of last_slot_oc (12 channels, PixelShuffle(2)): bytes 0, 1 = one 2-byte run, byte 2 = a single.
    // FASTD = FAST + 10: the same with the one-fma requant (ConvArgs::direct: zero point -128, (M, n) proven): cvt_pk_u8 does clamp,
    // rounding and byte insertion -- per row 1 pk_fma + 1 fma + 3-4 cvt + 1 xor instead of 2 + 2 fma, 3-4 med3, 2 add, 1-3 perm
    // OUTF (round 5, FAST flavours only): the fp32 frame INSTEAD of the int8 one -- what the reference's model(inps) returns, y = (q - zero_L) *
    // f32(scale_L) (quan_func.py:594) -- with the same run structure as the int8 flavours: one 8-byte store per 2-value run, one 16-byte
    // store per 4-value run (the every-output-kind path issues one dword store per value behind output-kind branches: 66 us per 1080p
    // frame, 1.5 TB/s).  No anchor add here (that keeps the general path).
    // OUTF == 2: ... plus the x2 anchor add, y + x (one fp32 add per value; x01 / x2 from fetch_anchor): the pair map only (3 -> 12 channels,
    // PixelShuffle 2 -- the reference's one anchor topology, models/sesr_arch.py:171-205); other shapes keep the general path
    template <bool BIASED, int FASTD = 0, int NV = 4, int OUTF = 0>
    __device__ __forceinline__ void store(const int s[4], const ConvArgs &a, int gy, float zlo, bool row_ok = true, float x01 = 0.f, float x2 = 0.f) const {
        constexpr int FAST = FASTD % 10;
        static_assert(!OUTF || FAST != 0, "fp32-only store: FAST flavours");
        static_assert(OUTF != 2 || (NV == 3 && FAST == 2), "anchor flavour: the pair map only");
        if constexpr (OUTF != 0) {
            float yv[4] = {0.f, 0.f, 0.f, 0.f};
            const float sv = in_vgpr(a.s_out);
            if constexpr (FASTD >= 10) {
                // one-fma forms (zero_L == -128): q + 128 = clamp(rint(t), 0, 255) -- cvt_pk_u8's semantics, as a float -- IS q - zero_L
                const float cv = in_vgpr(FASTD >= 20 ? a.Cs : a.Cd), mv = in_vgpr(a.Md), hi = in_vgpr(255.f);
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    float t = __builtin_fmaf(__builtin_bit_cast(float, s[i]), mv, cv);
                    if constexpr (FASTD >= 20) t = __fadd_rn(t, 128.f);
                    yv[i] = __fmul_rn(__builtin_rintf(med3(t, 0.f, hi)), sv);
                }
            } else {
                v2f v01, v23;
                const int s3[4] = {s[0], s[1], s[2], NV == 4 ? s[3] : s[2]};
                requant4<BIASED>(s3, a.Mf, a.sh, a.z_out, v01, v23);
                const float v[4] = {v01[0], v01[1], v23[0], v23[1]};
                const float zo = in_vgpr(a.z_out);
#pragma unroll
                for (int i = 0; i < NV; ++i) yv[i] = __fmul_rn(__fsub_rn(__builtin_rintf(med3(v[i], zlo, 127.f)), zo), sv);
            }
            if constexpr (OUTF == 2) { yv[0] = __fadd_rn(yv[0], x01); yv[1] = __fadd_rn(yv[1], x01); yv[2] = __fadd_rn(yv[2], x2); }
            const int so = __builtin_amdgcn_readfirstlane(row_ok ? gy * (FAST * FAST * a.W) * 4 : 0x7fff0000);
            typedef float v2fs __attribute__((ext_vector_type(2)));
            typedef float v4fs __attribute__((ext_vector_type(4)));
            if constexpr (NV == 3) {
                static_assert(FAST == 2, "three real rows per lane group: the PixelShuffle(2) pair map only");
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, (v2fs){yv[0], yv[1]}), rf, this->vo[0] * 4, so, 0);
                __builtin_amdgcn_raw_buffer_store_b32(fbits(yv[2]), rf, this->vo[2] * 4, so, 0);
            } else if constexpr (FAST == 2) {
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, (v2fs){yv[0], yv[1]}), rf, this->vo[0] * 4, so, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, (v2fs){yv[2], yv[3]}), rf, this->vo[2] * 4, so, 0);
            } else {
                // 16-byte store: the row offset rides in the VECTOR offset, soffset = 0 (the store-data hazard of dwordx4 stores with an SGPR
                // soffset: store_rows4, sesrq_mfma_common.h -- the grouped-launch test caught 16 garbage floats per frame with the scalar form)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, (v4fs){yv[0], yv[1], yv[2], yv[3]}), rf, this->vo[0] * 4 + so, 0, 0);
            }
            return;
        }
        if constexpr (FASTD >= 10) {
            static_assert(BIASED && FAST != 0, "one-fma requant: biased sums, int8 output");
            // FASTD 2x (ConvArgs::direct == 2): the fma also subtracts the 128 (one rounding of s*M*2^-n - 128), one add brings it back
            // plain v_fma_f32 / v_add_f32 with every operand in a VGPR (see fma4_biased, sesrq_mfma_common.h)
            const float cv = in_vgpr(FASTD >= 20 ? a.Cs : a.Cd), mv = in_vgpr(a.Md);
            float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NV; ++i) t[i] = __builtin_fmaf(__builtin_bit_cast(float, s[i]), mv, cv);
            if constexpr (FASTD >= 20) {
                const float kv = in_vgpr(128.f);
#pragma unroll
                for (int i = 0; i < NV; ++i) t[i] = __fadd_rn(t[i], kv);
            }
            const v2f w01 = {t[0], t[1]}, w23 = {t[2], t[3]};
            unsigned w = __builtin_amdgcn_cvt_pk_u8_f32(w01[0], 0, 0u);
            w = __builtin_amdgcn_cvt_pk_u8_f32(w01[1], 1, w);
            w = __builtin_amdgcn_cvt_pk_u8_f32(w23[0], 2, w);
            if constexpr (NV == 4) w = __builtin_amdgcn_cvt_pk_u8_f32(w23[1], 3, w);
            w = flip80(w);
            const int so = __builtin_amdgcn_readfirstlane(row_ok ? gy * (FAST * FAST * a.W) : 0x7fff0000);
            if constexpr (NV == 3) {
                static_assert(FAST == 2, "three real rows per lane group: the PixelShuffle(2) pair map only");
                __builtin_amdgcn_raw_buffer_store_b16((unsigned short)w, rq, this->vo[0], so, 0);
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(w >> 16), rq, this->vo[2], so, 0);
            } else if constexpr (FAST == 2) {
                __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(w & 0xffffu), rq, this->vo[0], so, 0);
                __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(w >> 16), rq, this->vo[2], so, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b32(w, rq, this->vo[0], so, 0);
            }
            return;
        }
        v2f v01, v23;
        if constexpr (NV == 4) {
            requant4<BIASED>(s, a.Mf, a.sh, a.z_out, v01, v23);
        } else {
            const int s3[4] = {s[0], s[1], s[2], s[2]};
            requant4<BIASED>(s3, a.Mf, a.sh, a.z_out, v01, v23);
        }
        const v2f mg = {MAGIC, MAGIC};
        v2f c01 = {med3(v01[0], zlo, 127.f), med3(v01[1], zlo, 127.f)}, c23 = {med3(v23[0], zlo, 127.f), NV == 4 ? med3(v23[1], zlo, 127.f) : 0.f};
        c01 = c01 + mg;                    // low mantissa bits = rint(value), two's complement
        if constexpr (NV == 4) c23 = c23 + mg; else c23[0] = c23[0] + MAGIC;
        // scalar offset of the row: gy is wave-uniform by contract, but derives from threadIdx (the wave index), so the
        // compiler must be TOLD -- the GEN_STD last layer otherwise wraps every store in a waterfall loop (readfirstlane /
        // compare / saveexec / branch), which also keeps a row's stores from overlapping the next row's MFMA chain
        // FAST: row_ok is wave-uniform too (the whole row is inside the frame or not); a row outside gets an offset the
        // buffer's range check rejects (it adds voffset + soffset in wide arithmetic, tools/oob_probe.hip)
        // FAST: the row pitch is rebuilt from the kernel argument (row_elems went through the switch of init() and came back
        // in a VGPR: v_mul_lo + v_cndmask + v_readfirstlane per row instead of two scalar instructions)
        const int so = __builtin_amdgcn_readfirstlane((FAST == 0 || row_ok) ? gy * (FAST != 0 ? FAST * FAST * a.W : row_elems) : 0x7fff0000);
        if constexpr (FAST != 0 && NV == 3) {
            static_assert(FAST == 2, "three real rows per lane group: the PixelShuffle(2) pair map only");
            const unsigned w01 = __builtin_amdgcn_perm(fbits(c01[1]), fbits(c01[0]), 0x0c0c0400u);
            __builtin_amdgcn_raw_buffer_store_b16((unsigned short)w01, rq, this->vo[0], so, 0);
            __builtin_amdgcn_raw_buffer_store_b8((unsigned char)fbits(c23[0]), rq, this->vo[2], so, 0);
            return;
        } else if constexpr (FAST != 0) {
            const unsigned w = pack_lo_bytes(fbits(c01[0]), fbits(c01[1]), fbits(c23[0]), fbits(c23[1]));
            if constexpr (FAST == 2) {
                __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(w & 0xffffu), rq, this->vo[0], so, 0);
                __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(w >> 16), rq, this->vo[2], so, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b32(w, rq, this->vo[0], so, 0);
            }
            return;
        }
        int vo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) vo[i] = row_ok ? this->vo[i] : (int)0x10000000;
        if (a.out_q) {
            const unsigned w = pack_lo_bytes(fbits(c01[0]), fbits(c01[1]), fbits(c23[0]), fbits(c23[1]));
            if (NV == 4 && r == 2) {
                __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(w & 0xffffu), rq, vo[0], so, 0);
                __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(w >> 16), rq, vo[2], so, 0);
            } else if (NV == 4 && r == 4) {
                __builtin_amdgcn_raw_buffer_store_b32(w, rq, vo[0], so, 0);
            } else if (NV == 3 && last_pairmap(a.oc, r)) {
                __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(w & 0xffffu), rq, vo[0], so, 0);
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)((w >> 16) & 0xffu), rq, vo[2], so, 0);
            } else {
#pragma unroll
                for (int i = 0; i < NV; ++i) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)((w >> (8 * i)) & 0xffu), rq, vo[i], so, 0);
            }
        }
        if (a.out_f) {
            const v2f q01 = c01 - mg, q23 = c23 - mg;      // exact: back to the integer-valued float
            const float q[4] = {q01[0], q01[1], q23[0], q23[1]};
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                float yv = __fmul_rn(q[i] - a.z_out, a.s_out);
                if (anc && row_ok) yv = __fadd_rn(yv, anc[va[i] + gy * a.W]);   // + nearest-upsampled input (test.py:148-155)
                __builtin_amdgcn_raw_buffer_store_b32(fbits(yv), rf, vo[i] * 4, so * 4, 0);
            }
        }
    }
};

// Staging of a (SH x SW) window of NHWC16 pixels, split in two phases so that a persistent
// workgroup can keep the loads of its NEXT tile in flight while it computes the current one:
//   load():  all buffer loads of a thread issued back to back (out-of-range -> 0)
//   store(): pad word selected in for pixels outside the frame, then written to LDS
//   PW == 0: LDS image = SH x SW pixels of 16 bytes (merged / hybrid kernels: one ds_read_b128 = one pixel)
//   PW  > 0: PE-planar image for the per-PE (general) kernels: row pitch 4*PW dwords, dword [row][p][col] = word p
//            (the 4 channels of PE p) of the pixel -- a lane's MFMA operand (4 pixels of ONE PE) is then read
//            directly by dword loads with immediate plane offsets, no register shuffling.
//   CP  > 0: column-major PE-planar image (5x5 per-PE kernel): dword [p][col][row], column pitch CSP >= SH, plane pitch CP >= SW * CSP
//            -- vertical neighbours are 1 dword apart, horizontal ones CSP, the four planes CP: every operand of every row of a tile
//            is an immediate offset away from ONE lane-constant address.
template <int SH, int SW, int R, int PW = 0, int CP = 0, int CSP = 0>
struct StageNHWC16 {
    static constexpr int NIT = (SH * SW + 255) / 256;
    // element i of the window -> (row, column).  Row-major for the pixel-major images.  Column-major planar image (CP > 0): blocks of 8
    // columns x 4 rows per 32 lanes -- a dword store's 32 lanes then hit 32 different banks of the column-major image (column pitch 12:
    // 8 columns = 8 multiples of 4, + 4 rows), where 32 consecutive columns of one row hit 8 (4-way conflicts: measured, 6.3 M conflict
    // cycles per 1080p frame together with the reads' share); 8 lanes still load one whole 128-byte line
    __device__ __forceinline__ static void rc_of(int i, int &row, int &col) {
        if constexpr (CP > 0) {
            static_assert(CP == 0 || (SW % 8 == 0 && SH % 4 == 0), "8 x 4 staging blocks");
            const int u = i >> 5;
            row = (u / (SW / 8)) * 4 + ((i >> 3) & 3);
            col = (u % (SW / 8)) * 8 + (i & 7);
        } else {
            row = i / SW;
            col = i - row * SW;
        }
    }
    v4u v[NIT];
    bool ok[NIT];
    bool interior;               // wave-uniform: every staged pixel of the tile lies inside the frame -> no pad selects in store()
    int voff[NIT], ty[NIT];      // per-lane byte offset of the pixel in tile row space; tile row of the pixel
    __amdgpu_buffer_rsrc_t rs;
    int row_bytes;
    __device__ __forceinline__ void set_interior(const ConvArgs &a, int x0, int y0) {
        interior = (x0 - R >= 0) && (x0 - R + SW <= a.W) && (y0 - R >= 0) && (y0 - R + SH <= a.H);
    }
    // Addresses are lane constants + ONE scalar per tile (soffset = y0 * W * 16): the loop issues its
    // loads without touching a VGPR, so nothing forces a wait on the previous tile's stores.  Rows
    // outside the frame fall out of the buffer's range check (gfx950 checks voffset + soffset).
    __device__ __forceinline__ void init(const ConvArgs &a, int n_img, int x0, int tid) {
        const size_t img = (size_t)a.H * a.W * 16;
        rs = __builtin_amdgcn_make_buffer_rsrc((char *)const_cast<void *>(a.in) + (size_t)n_img * img, 0, (int)img, 0x00020000);
        row_bytes = a.W * 16;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            int tx;
            rc_of(i, ty[it], tx);
            const int gx = x0 - R + tx;
            const bool okx = (gx >= 0) & (gx < a.W) & (i < SH * SW);
            voff[it] = okx ? (ty[it] * a.W + gx) * 16 : (int)0x80000000;
            if (!okx) ty[it] = -(1 << 20);       // never a valid row -> pad
        }
    }
    // y0 >= R only (every tile but the first of the frame): offsets stay non-negative
    __device__ __forceinline__ void load(const ConvArgs &a, int n_img, int x0, int y0, int tid) {
        set_interior(a, x0, y0);
        const int soff = (y0 - R) * row_bytes;
        const int lo = R - y0, hi = a.H + R - y0;          // valid tile rows: lo <= ty < hi
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            v[it] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[it], soff, 0);
            ok[it] = (ty[it] >= lo) & (ty[it] < hi);
        }
    }
    // any y0 (prologue of a chunk): per-lane offsets, rows above the frame pushed out of range
    __device__ __forceinline__ void load_first(const ConvArgs &a, int n_img, int x0, int y0, int tid) {
        set_interior(a, x0, y0);
        const int lo = R - y0, hi = a.H + R - y0;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            ok[it] = (ty[it] >= lo) & (ty[it] < hi);
            v[it] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok[it] ? voff[it] + (y0 - R) * row_bytes : (int)0x80000000, 0, 0);
        }
    }
    // interior tiles (most of a frame) skip the per-pixel pad select: 8 of the ~10 vector instructions of an iteration
    __device__ __forceinline__ void store(int4 *tile, const ConvArgs &a, int tid) const {
        if (interior) store_t<false>(tile, a, tid);
        else store_t<true>(tile, a, tid);
    }
    template <bool PADSEL>
    __device__ __forceinline__ void store_t(int4 *tile, const ConvArgs &a, int tid) const {
        const unsigned pw = (unsigned)a.pad_word;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            const v4u pad = {pw, pw, pw, pw};
            const v4u t = (!PADSEL || ok[it]) ? v[it] : pad;
            if constexpr (CP > 0) {
                int *tw = reinterpret_cast<int *>(tile);
                int row, col;
                rc_of(i, row, col);
                if (i < SH * SW) {
                    int *d = tw + col * CSP + row;
                    d[0] = (int)t[0]; d[CP] = (int)t[1]; d[2 * CP] = (int)t[2]; d[3 * CP] = (int)t[3];
                }
            } else if constexpr (PW == 0) {
                if (i < SH * SW) tile[i] = make_int4((int)t[0], (int)t[1], (int)t[2], (int)t[3]);
            } else {
                int *tw = reinterpret_cast<int *>(tile);
                const int row = i / SW, col = i - row * SW;
                if (i < SH * SW) {
                    int *d = tw + row * (4 * PW) + col;
                    d[0] = (int)t[0]; d[PW] = (int)t[1]; d[2 * PW] = (int)t[2]; d[3 * PW] = (int)t[3];
                }
            }
        }
    }
};

// Persistent tile walk: a workgroup owns a 64-column strip and `a.chunk_tiles` vertically adjacent
// MTH-row tiles; tile k+1 is loaded (registers) while tile k is computed out of the other LDS buffer.
// Diagnostic build only (-DSESRQ_STAMPS): wave 0 of every workgroup records s_memrealtime /
// s_memtime at the phase boundaries into a buffer no other code reads (guide §7, in-kernel stamps).
#ifdef SESRQ_STAMPS
#define STAMP(k)                                                                                    \
    if (a.dbg_pe && tid == 0 && (k) < 15) {      /* slot 15 = kernel entry; runs longer than 4 tiles: the first 4 */ \
        int *sp = a.dbg_pe + ((bxy.y * gridDim.x + bxy.x) * 16 + (k)) * 2;                \
        sp[0] = (int)__builtin_amdgcn_s_memrealtime();                                              \
        sp[1] = (int)__builtin_amdgcn_s_memtime();                                                  \
    }
#define STAMP_ENTRY                                                                                 \
    if (a.dbg_pe && tid == 0) {                                                                     \
        int *sp = a.dbg_pe + ((bxy.y * gridDim.x + bxy.x) * 16 + 15) * 2;                          \
        sp[0] = bxy.t_entry;                                                                        \
        sp[1] = bxy.c_entry;                                                                        \
    }
#else
#define STAMP(k)
#define STAMP_ENTRY
#endif
#define SESRQ_TILE_WALK(STAGE_T, BUF0, BUF1, COMPUTE) SESRQ_TILE_WALK_H(MTH, STAGE_T, BUF0, BUF1, COMPUTE)
#define SESRQ_TILE_WALK_H(TILE_H, STAGE_T, BUF0, BUF1, COMPUTE)                                     \
    {                                                                                               \
        SESRQ_TILE_WALK_BEGIN(TILE_H, STAGE_T)                                                      \
        SESRQ_TILE_WALK_REST(TILE_H, BUF0, BUF1, COMPUTE)                                           \
    }
// BEGIN: the run's bounds and the loads of its first tile.  A kernel may place it at its very top, ahead of everything else it fetches
// (A fragments, tables, store geometry): the frame loads are then the FIRST vector-memory requests of the wave, and what the rest of the
// prologue waits for arrives beside them instead of in front of them (round 4: 2 us from kernel entry to the first frame load of the last
// layer, most of it dependent round trips to memory)
#define SESRQ_TILE_WALK_BEGIN(TILE_H, STAGE_T)                                                      \
    /* runs of (almost) equal length: the first run_rem runs are one tile longer (the host divided: launch()) */ \
    const int t_begin = bxy.y * a.run_q + min(bxy.y, a.run_rem);                                    \
    const int t_end = t_begin + a.run_q + (bxy.y < a.run_rem ? 1 : 0);                              \
    STAGE_T st;                                                                                     \
    st.init(a, n_img, x0, tid);                                                                     \
    STAMP(0)                                                                                        \
    STAMP_ENTRY                                                                                     \
    st.load_first(a, n_img, x0, t_begin * (TILE_H), tid);                                           \
    STAMP(1)
#define SESRQ_TILE_WALK_REST(TILE_H, BUF0, BUF1, COMPUTE)                                           \
    {                                                                                               \
        st.store(BUF0, a, tid);                                                                     \
        __syncthreads();                                                                            \
        STAMP(2)                                                                                    \
        for (int t = t_begin; t < t_end; ++t) {                                                     \
            const int y0 = t * (TILE_H);                                                            \
            const bool cur0 = ((t - t_begin) & 1) == 0;                                             \
            if (t + 1 < t_end) st.load(a, n_img, x0, y0 + (TILE_H), tid);                           \
            { const int4 *cur_tile = cur0 ? BUF0 : BUF1; COMPUTE(cur_tile) }                        \
            STAMP(3 + 3 * (t - t_begin))                                                            \
            if (t + 1 < t_end) { if (cur0) st.store(BUF1, a, tid); else st.store(BUF0, a, tid); }  \
            STAMP(4 + 3 * (t - t_begin))                                                            \
            __syncthreads();                                                                        \
            STAMP(5 + 3 * (t - t_begin))                                                            \
        }                                                                                           \
    }

// the rest of the walk (after SESRQ_TILE_WALK_BEGIN) for a stage type whose following tiles take their top halo rows from the tile under
// computation (StageFrame)
#define SESRQ_TILE_WALK_CARRY_REST(TILE_H, BUF0, BUF1, COMPUTE)                                     \
    {                                                                                               \
        st.store(BUF0, a, tid);                                                                     \
        __syncthreads();                                                                            \
        STAMP(2)                                                                                    \
        for (int t = t_begin; t < t_end; ++t) {                                                     \
            const int y0 = t * (TILE_H);                                                            \
            const bool cur0 = ((t - t_begin) & 1) == 0;                                             \
            if (t + 1 < t_end) st.load(a, n_img, x0, y0 + (TILE_H), tid);                           \
            { const int4 *cur_tile = cur0 ? BUF0 : BUF1; COMPUTE(cur_tile) }                        \
            STAMP(3 + 3 * (t - t_begin))                                                            \
            if (t + 1 < t_end) { if (cur0) st.store_next(BUF1, BUF0, a, tid); else st.store_next(BUF0, BUF1, a, tid); }  \
            STAMP(4 + 3 * (t - t_begin))                                                            \
            __syncthreads();                                                                        \
            STAMP(5 + 3 * (t - t_begin))                                                            \
        }                                                                                           \
    }

// ------------------------------------------------------------------ hidden 3x3, 16 -> 16 channels
template <int MODE, int EPI>
__global__ __launch_bounds__(256) void mfma_h3_kernel(const ConvArgs a) {
    constexpr bool GENERAL = mode_general(MODE);
    constexpr int SW = MTW + 4;                      // 1 left halo + 64 + 1 right halo + over-read
    constexpr int SH = MTH + 2 + (MODE != MERGED ? 1 : 0);  // per-PE chains read row y+3 with zero weights
    constexpr int PW = GENERAL ? SW : 0;             // planar image: row pitch 4*68 = 272 dwords = 16 mod 64 banks
    __shared__ int4 buf0[SH * SW], buf1[SH * SW];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, n = l & 15, g = l >> 4;
    kernarg_warm<ConvArgs>();
    const BlockXY bxy = xcd_block(a.inv_nx);
    const int x0 = bxy.x * MTW, n_img = blockIdx.z;
    const int4 *fr = a.afrag;
    constexpr bool BIASED = mode_biased(MODE);     // requant without v_cvt: sums carry + MAGIC_I (needs |s| < 2^22)
    int4 ac = fr[g];
    if constexpr (BIASED) { ac.x += MAGIC_I; ac.y += MAGIC_I; ac.z += MAGIC_I; ac.w += MAGIC_I; }
    v4i A[GENERAL ? 4 : 3];
#pragma unroll
    for (int f = 0; f < (GENERAL ? 4 : 3); ++f) A[f] = ld_frag(fr + 4 + f * 64 + l);
    v4i AR = {0, 0, 0, 0};
    if constexpr (MODE == HYB) AR = ld_frag(a.afrag2 + 4 + a.risky_pe * 64 + l);
    const float zlo = a.relu ? fmaxf(a.z_next, -128.f) : -128.f;
    const int gx = x0 + 16 * w + n;
    auto compute = [&](const int4 *tile, int y0) __attribute__((always_inline)) {
        const RowIO io = make_rowio(a, n_img, y0, gx, g);
        if constexpr (!GENERAL) {
            const int col = 16 * w + n + g;
            const v4i zero = {0, 0, 0, 0};
            const v4i acc0 = {ac.x, ac.y, ac.z, ac.w};
            const int *t32 = reinterpret_cast<const int *>(tile);
            const int rbase = (g * SW + 16 * w + n) * 4 + a.risky_pe;      // HYB: word risky_pe of pixel (row g, col)
            v4i B0 = ld_frag(tile + col), B1 = ld_frag(tile + SW + col);
#pragma unroll
            for (int y4 = 0; y4 < MTH; y4 += 4) {
                int s4[4][4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const v4i B2 = ld_frag(tile + (y4 + r + 2) * SW + col);
                    v4i acc[MODE == HYB ? 2 : 1];
                    acc[0] = mfma(A[0], B0, acc0);
                    acc[0] = mfma(A[1], B1, acc[0]);
                    acc[0] = mfma(A[2], B2, acc[0]);
                    B0 = B1; B1 = B2;
                    if constexpr (MODE == HYB) {
                        const int o = rbase + (y4 + r) * SW * 4;
                        const v4i br = {t32[o], t32[o + 4], t32[o + 8], t32[o + 12]};
                        acc[1] = mfma(AR, br, zero);
                    }
                    finish_sums<MODE>(s4[r], acc, ac, a);
                }
                emit_rows4<EPI, false, BIASED>(s4, a, io, y4, zlo);
            }
        } else {
            const int col = 16 * w + n;
#pragma unroll 1
            for (int y4 = 0; y4 < MTH; y4 += 4) {
                int s4[4][4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int *row = reinterpret_cast<const int *>(tile) + (y4 + r + g) * (4 * PW) + col;   // lane group g = kernel row ky
                    const v4i zero = {0, 0, 0, 0};
                    v4i acc[4];
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int *q = row + p * PW;                       // plane p: 4 horizontally adjacent pixels of PE p
                        const v4i b = {q[0], q[1], q[2], q[3]};
                        acc[p] = mfma(A[p], b, zero);
                    }
                    if constexpr (MODE == GEN_TAP) tap_sums<4>(acc, a, n_img, y0 + y4 + r, gx, g, 0);
                    finish_sums<MODE>(s4[r], acc, ac, a);
                }
                emit_rows4<EPI, false, BIASED>(s4, a, io, y4, zlo);
            }
        }
    };
#define SESRQ_COMPUTE(B) compute(B, y0);
    using Stage = StageNHWC16<SH, SW, 1, PW>;
    SESRQ_TILE_WALK(Stage, buf0, buf1, SESRQ_COMPUTE)
#undef SESRQ_COMPUTE
}

// ------------------------------------------------------------------ 5x5, 16 input channels
// FAST (EPI_LAST only): 2 / 4 = PixelShuffle factor, int8 output only (LastStore::store); 0 = every output kind
// 4 waves per SIMD: round 3, same-box A/B at 1080p: 5 waves (96 VGPRs, a handful of spills outside the row loop, a fifth workgroup per
// CU) ran the last layer 5 % SLOWER than 4 (29.3 vs 27.7 us) and the fused trio 13 % slower (47.1 vs 41.6 us).
// NV (EPI_LAST only): real accumulator rows per lane group, 3 for up to 12 output channels (last_slot_oc, sesrq_common.h)
// OUTF (EPI_LAST, FAST != 0): 1 = the fp32 frame instead of the int8 one, 2 = ... with the x2 anchor add (LastStore::store)
template <int MODE, int EPI, int FAST = 0, int NV = 4, int OUTF = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void mfma_h5_kernel(const ConvArgs a) {
    constexpr bool GENERAL = mode_general(MODE);
    constexpr int SW = MTW + 8;          // 2 + 64 + 2 halo, + over-read of the kx = 4..7 group
    constexpr int SH = MTH + 4;
    // per-PE (general) kernels: column-major PE-planar image [PE][col][row] (StageNHWC16, CP > 0), column pitch CS = 12 dwords (the tile's
    // rows), plane pitch CP = 72 columns.  A lane's operand per PE and K-chunk = two vertical pixel pairs that start on EVEN tile rows
    // (h5_pair, sesrq_common.h): 8-byte-aligned LDS accesses straight into the four operand registers -- K-chunk 0 one ds_read2_b64 of four
    // consecutive rows, K-chunk 1 two ds_read_b64 (round 3: four ds_read2_b32 of pairs of any alignment, 1.75 x the LDS cycles).  That needs
    // output rows of one parity per wave: wave w works on rows (w & 1) + 2t, t = 0..3, of TWO 16-column groups (w >> 1), with the A
    // fragments of its parity.  All 4 PEs x 4 rows of a column group are reached by immediate offsets from three lane-constant addresses;
    // the planes are more than a ds_read2_b64's offset range apart, so hipcc cannot pair reads of different PEs (it did, and then moved
    // 192 registers per tile into operand order).  Banks: h5_pair; the staging writes (dword stores 12 apart) are 4-way.
    constexpr int CS = GENERAL ? SH : 0;
    constexpr int CP = SW * CS;
    static_assert(!GENERAL || (CS % 8 == 4 && CP * 4 > 2040), "aligned pairs / bank rule / no ds_read2 across PE planes");
    constexpr int SHB = SH + (MODE == HYB ? 1 : 0);      // hybrid: the risky PE's pairs reach one row below the tile (zero weights)
    __shared__ int4 buf0[GENERAL ? CP : SHB * SW], buf1[GENERAL ? CP : SHB * SW];      // general: 4 planes of CP dwords
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, n = l & 15, g = l >> 4;
    kernarg_warm<ConvArgs>();
    const BlockXY bxy = xcd_block(a.inv_nx);
    const int x0 = bxy.x * MTW, n_img = blockIdx.z;
    using Stage = StageNHWC16<SH, SW, 2, 0, CP, CS>;
    SESRQ_TILE_WALK_BEGIN(MTH, Stage)
    const int4 *fr = a.afrag;
    constexpr bool BIASED = mode_biased(MODE);     // requant without v_cvt: sums carry + MAGIC_I (needs |s| < 2^22)
    int4 ac = fr[g];
    if constexpr (BIASED) { ac.x += MAGIC_I; ac.y += MAGIC_I; ac.z += MAGIC_I; ac.w += MAGIC_I; }
    const float zlo = a.relu ? fmaxf(EPI == EPI_LAST ? a.z_out : a.z_next, -128.f) : -128.f;
    const int gx = x0 + 16 * w + n;
    // merged: K-chunks 0..4 = kernel row f, lane group g = kx 0..3;  5 = column 4, lane group g = ky 0..3;  6 = tap (4,4)
    // general, per PE p two K-chunks of two vertical pixel pairs per lane group (h5_pair; pack_mfma_frags, MFMA_H5), one set per row parity
    constexpr int NF = GENERAL ? 8 : 7;
    // general: this wave's row parity and pair of 16-column groups -- wave-uniform, and TOLD so (readfirstlane): everything derived from
    // them (row offsets of the stores, column bases) is then scalar arithmetic instead of VALU + v_readfirstlane per row
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const int par = GENERAL ? (wu & 1) : 0, cg = wu >> 1;
    v4i A[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) A[f] = ld_frag(fr + 4 + (par * 8 + f) * 64 + l);
    // per-PE chains (general, and the risky PE's chain of the hybrid mode); must match pack_mfma_frags (MFMA_H5 general)
    int pcol[2][2], prow[2][2];                         // [chunk][pair]: column and first row of lane group g's pixel pair
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < 2; ++q) h5_pair(c, g, q, pcol[c][q], prow[c][q]);
    LastStore ls, ls1;                                  // general: one per column group of the wave
    if constexpr (EPI == EPI_LAST) {
        if constexpr (GENERAL) { ls.template init<NV, FAST % 10>(a, n_img, g, x0 + 32 * cg + n); ls1.template init<NV, FAST % 10>(a, n_img, g, x0 + 32 * cg + 16 + n); }
        else ls.template init<NV, FAST % 10>(a, n_img, g, gx);
    }
    v4i AR[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    if constexpr (MODE == HYB) {
        AR[0] = ld_frag(a.afrag2 + 4 + (0 * 4 + a.risky_pe) * 64 + l);
        AR[1] = ld_frag(a.afrag2 + 4 + (1 * 4 + a.risky_pe) * 64 + l);
    }
    // lane-constant byte offsets of the four pixel pairs inside a tile for both column groups, computed ONCE (pinned: hipcc re-derived them
    // -- 8 v_mul_lo + a dozen adds -- at the top of every tile)
    unsigned pboff[2][2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                pboff[j][c][q] = GENERAL ? (unsigned)(((32 * cg + 16 * j + n + pcol[c][q]) * CS + prow[c][q]) * 4) : 0u;      // [0][1] unused: pair 1 of K-chunk 0 = pair 0 + 2 rows
                asm volatile("" : "+v"(pboff[j][c][q]));
            }
    auto compute = [&](const int4 *tile, int y0) __attribute__((always_inline)) {
        RowIO io;
        if constexpr (EPI != EPI_LAST) io = make_rowio(a, n_img, y0, gx, g);
        if constexpr (!GENERAL) {
            const int col = 16 * w + n + g, colc = 16 * w + n + 4;
            const v4i zero = {0, 0, 0, 0};
            const v4i acc0 = {ac.x, ac.y, ac.z, ac.w};
            const int *t32 = reinterpret_cast<const int *>(tile);
            const int cb = (16 * w + n) * 4 + a.risky_pe;       // HYB: word risky_pe of column (16w + n), row 0
            v4i B[5];
#pragma unroll
            for (int r = 0; r < 4; ++r) B[r] = ld_frag(tile + r * SW + col);
#pragma unroll
            for (int y4 = 0; y4 < MTH; y4 += 4) {
                int s4[4][4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int y = y4 + r;
                    float x01 = 0.f, x2 = 0.f;
                    if constexpr (OUTF == 2) ls.fetch_anchor(a, y0 + y, x01, x2);
                    B[(y + 4) % 5] = ld_frag(tile + (y + 4) * SW + col);
                    const v4i C5 = ld_frag(tile + (y + g) * SW + colc);      // column 4: lane group g = kernel row g
                    const v4i C6 = ld_frag(tile + (y + 4) * SW + colc);      // tap (4,4)
                    v4i acc[MODE == HYB ? 2 : 1];
                    acc[0] = acc0;
#pragma unroll
                    for (int ky = 0; ky < 5; ++ky) acc[0] = mfma(A[ky], B[(y + ky) % 5], acc[0]);
                    acc[0] = mfma(A[5], C5, acc[0]);
                    acc[0] = mfma(A[6], C6, acc[0]);
                    if constexpr (MODE == HYB) {
                        int o[2][2];                                   // word risky_pe of the first pixel of pair [chunk][pair]
#pragma unroll
                        for (int c = 0; c < 2; ++c)
#pragma unroll
                            for (int q = 0; q < 2; ++q) o[c][q] = cb + ((y + prow[c][q]) * SW + pcol[c][q]) * 4;
                        const v4i b0 = {t32[o[0][0]], t32[o[0][0] + SW * 4], t32[o[0][1]], t32[o[0][1] + SW * 4]};
                        const v4i b1 = {t32[o[1][0]], t32[o[1][0] + SW * 4], t32[o[1][1]], t32[o[1][1] + SW * 4]};
                        acc[1] = mfma(AR[0], b0, zero);
                        acc[1] = mfma(AR[1], b1, acc[1]);
                    }
                    finish_sums<MODE, NV>(s4[r], acc, ac, a);
                    if constexpr (EPI == EPI_LAST) {
                        // a row below the frame is dropped by its offsets (FAST: the scalar one, else the lanes'), not by a
                        // branch: the four rows stay one basic block
                        ls.template store<BIASED, FAST, NV, OUTF>(s4[r], a, y0 + y, zlo, y0 + y < a.H, x01, x2);
                    }
                }
                if constexpr (EPI != EPI_LAST) emit_rows4<EPI, false, BIASED>(s4, a, io, y4, zlo);
            }
        } else {
            typedef int v2ia __attribute__((ext_vector_type(2)));                  // two adjacent dwords, 8-byte aligned: ds_read_b64
            typedef const v2ia __attribute__((address_space(3))) *lds_pair_t;
            const unsigned tb = (unsigned)(size_t)(const __attribute__((address_space(3))) void *)tile;     // LDS byte address of the tile
#pragma unroll
            for (int j = 0; j < 2; ++j) {                                // the wave's two 16-column groups
                const int gxj = x0 + 32 * cg + 16 * j + n;
                // LDS byte addresses (row 0, PE 0) of the four pairs.  K-chunk 0's second pair is the first one two rows down, but it gets an
                // address register of its own: reads off ONE register 8 bytes apart become a ds_read2_b64, which the LDS serves 16 lanes at a
                // time over 32 banks (column pitch 12: two-way conflicts, half the rate) where a ds_read_b64 goes 32 lanes at a time over 64
                unsigned pa0 = tb + pboff[j][0][0], pa1 = tb + pboff[j][0][0] + 8, p0 = tb + pboff[j][1][0], p1 = tb + pboff[j][1][1];
                int s4[4][4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {                            // tile rows par + 2t
                    const int gy = y0 + par + 2 * t;
                    float x01 = 0.f, x2 = 0.f;
                    if constexpr (OUTF == 2) { if (j == 0) ls.fetch_anchor(a, gy, x01, x2); else ls1.fetch_anchor(a, gy, x01, x2); }
                    const v4i zero = {0, 0, 0, 0};
                    v4i acc[4];
                    // a pair of row t + 1 is a pair of row t in another operand slot: hide the relation between the rows' addresses from the
                    // compiler, which otherwise keeps the pair and MOVES it into place (the reads are not what bounds this loop, vector issue is)
                    asm("" : "+v"(pa0), "+v"(pa1), "+v"(p0), "+v"(p1));
                    v4i b0[4], b1[4];
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int o = 4 * (2 * t + p * CP);
                        const v2ia a0 = *(lds_pair_t)(size_t)(pa0 + o), a1 = *(lds_pair_t)(size_t)(pa1 + o);
                        const v2ia c0 = *(lds_pair_t)(size_t)(p0 + o), c1 = *(lds_pair_t)(size_t)(p1 + o);
                        b0[p] = (v4i){a0[0], a0[1], a1[0], a1[1]};
                        b1[p] = (v4i){c0[0], c0[1], c1[0], c1[1]};
                    }
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        acc[p] = mfma(A[p], b0[p], zero);
                        acc[p] = mfma(A[4 + p], b1[p], acc[p]);
                    }
                    if constexpr (MODE == GEN_TAP) tap_sums<NV>(acc, a, n_img, gy, gxj, g, EPI == EPI_LAST ? NV : 0);
                    finish_sums<MODE, NV>(s4[t], acc, ac, a);
                    if constexpr (EPI == EPI_LAST) {
                        // a row below the frame is dropped by its offsets (FAST: the scalar one, else the lanes'), not by a branch
                        if (j == 0) ls.template store<BIASED, FAST, NV, OUTF>(s4[t], a, gy, zlo, gy < a.H, x01, x2);
                        else ls1.template store<BIASED, FAST, NV, OUTF>(s4[t], a, gy, zlo, gy < a.H, x01, x2);
                    }
                }
                if constexpr (EPI != EPI_LAST) {
                    // hidden 5x5 layer: the wave's four rows are two apart: lane (n, r' = g) stores pixel row y0 + par + 2g
                    RowIO ioj = make_rowio(a, n_img, y0, gxj, g);
                    ioj.voff = (gxj < a.W) ? ((y0 + par + 2 * g) * a.W + gxj) * 16 : (int)0x80000000;
                    emit_rows4<EPI, false, BIASED>(s4, a, ioj, 0, zlo);
                }
            }
        }
    };
#define SESRQ_COMPUTE(B) compute(B, y0);
    SESRQ_TILE_WALK_REST(MTH, buf0, buf1, SESRQ_COMPUTE)
#undef SESRQ_COMPUTE
}

// ------------------------------------------------------------------ last layer 5x5, 16 -> OC <= 4 (x2 nets), "pe-split"
// The 16 accumulator rows are (PE p, output channel o) = row 4p + o (pack_mfma_frags, MFMA_H5P): ONE chain
// over the full K (7 K-chunks, the B operands of the merged kernel) leaves PE g's four sums in lane group g.
// After the 18-bit clamp, a transpose-reduce over four pixel rows (12 permlane swaps + 12 adds) puts the
// complete adder sum of row r in lane group r: every lane then requantises and stores 4 real outputs.
//   K-chunk f < 5: lane group g = tap (ky f, kx g)     f = 5: (ky g, kx 4)     f = 6: g0 = (4, 4)
template <int MODE>
__global__ __launch_bounds__(256) void mfma_h5p_kernel(const ConvArgs a) {
    constexpr int SW = MTW + 8;
    constexpr int SH = MTH + 4;
    __shared__ int4 buf0[SH * SW], buf1[SH * SW];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, n = l & 15, g = l >> 4;
    kernarg_warm<ConvArgs>();
    const BlockXY bxy = xcd_block(a.inv_nx);
    const int x0 = bxy.x * MTW, n_img = blockIdx.z;
    const int4 *fr = a.afrag;
    constexpr bool BIASED = mode_biased(MODE);
    int4 ac = fr[0];
    if constexpr (BIASED) { ac.x += MAGIC_I; ac.y += MAGIC_I; ac.z += MAGIC_I; ac.w += MAGIC_I; }
    const float zlo = a.relu ? fmaxf(a.z_out, -128.f) : -128.f;
    const int gx = x0 + 16 * w + n;
    v4i A[7];
#pragma unroll
    for (int f = 0; f < 7; ++f) A[f] = ld_frag(fr + 4 + f * 64 + l);
    LastStore ls;
    ls.init(a, n_img, 0, gx, g);
    auto compute = [&](const int4 *tile, int y0) __attribute__((always_inline)) {
        const int col = 16 * w + n + g, colc = 16 * w + n + 4;
        const v4i zero = {0, 0, 0, 0};
        // merged: nothing can clamp -> the add constant rides in PE 0's accumulator rows
        const v4i acc0 = (MODE == MERGED && g == 0) ? (v4i){ac.x, ac.y, ac.z, ac.w} : zero;
        v4i B[5];
#pragma unroll
        for (int r = 0; r < 4; ++r) B[r] = ld_frag(tile + r * SW + col);
#pragma unroll
        for (int y4 = 0; y4 < MTH; y4 += 4) {
            unsigned s[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int y = y4 + r;
                B[(y + 4) % 5] = ld_frag(tile + (y + 4) * SW + col);
                const v4i C5 = ld_frag(tile + (y + g) * SW + colc);
                const v4i C6 = ld_frag(tile + (y + 4) * SW + colc);
                v4i acc = acc0;
#pragma unroll
                for (int ky = 0; ky < 5; ++ky) acc = mfma(A[ky], B[(y + ky) % 5], acc);
                acc = mfma(A[5], C5, acc);
                acc = mfma(A[6], C6, acc);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if constexpr (MODE == MERGED) s[r][i] = (unsigned)acc[i];
                    else if constexpr (MODE == GEN_ANY) s[r][i] = (unsigned)clampi3(acc[i], a.acc_lo, a.acc_hi);
                    else s[r][i] = (unsigned)clampi3(acc[i], -131072, 131071);
                }
            }
            int t[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // lane halves trade rows {0,1} against {2,3}: u0 = row 0 | row 2, u1 = row 1 | row 3 (PE g + PE g^2)
                v2u x = __builtin_amdgcn_permlane32_swap(s[0][i], s[2][i], false, false);
                const unsigned u0 = x[0] + x[1];
                x = __builtin_amdgcn_permlane32_swap(s[1][i], s[3][i], false, false);
                const unsigned u1 = x[0] + x[1];
                // odd lane groups trade with even ones: lane group r ends up with row r, all four PEs
                x = __builtin_amdgcn_permlane16_swap(u0, u1, false, false);
                t[i] = (int)(x[0] + x[1]);
            }
            int sf[4];
            const int acv[4] = {ac.x, ac.y, ac.z, ac.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (MODE == MERGED) sf[i] = t[i];
                else if constexpr (MODE == GEN_ANY) sf[i] = clampi3(t[i], a.add_lo, a.add_hi) + acv[i];
                else sf[i] = clampi3(t[i], -524288, 524287) + acv[i];
            }
            if (y0 + y4 < a.H) ls.store<BIASED>(sf, a, y0 + y4, zlo, y0 + y4 + g < a.H);
        }
    };
#define SESRQ_COMPUTE(B) compute(B, y0);
    using Stage = StageNHWC16<SH, SW, 2>;
    SESRQ_TILE_WALK(Stage, buf0, buf1, SESRQ_COMPUTE)
#undef SESRQ_COMPUTE
}

// ------------------------------------------------------------------ first layer 5x5, IC <= 4
// The frame is quantised while it is staged (q0 = clamp8(rint(x/s0 + z0)), quan_func.py:225);
// a pixel is one dword (byte c = channel c), the LDS image is COLUMN-major (dword [column][row], column pitch 18 = 2 mod 4):
// every operand dword of every row of a tile lies within ds_read2_b32's 8-bit dword offsets of two lane-constant addresses
// (round 2: row-major with an 80-dword pitch -- 4 address VALU + 2 operand moves per row), the 16 lanes of a group sit on 16
// distinct even banks and the two lane groups of a 32-lane half an odd number of rows apart, i.e. on the odd banks.
// NCH: input channels as a compile-time count (1 and 3 are the reference's nets), or 4 = "a.ic of them, tested per channel":
// the wave-uniform test put every channel's load and quantise code into a block of its own.
template <int SRC, int SH, int SWP, int PITCH, int NCH>
struct StageFrame {
    __device__ __forceinline__ static bool has_channel(const ConvArgs &a, int c) { return NCH < 4 ? c < NCH : c < a.ic; }
    static constexpr int NIT = (SH * SWP + 255) / 256;
    static constexpr int ESZ = (SRC == SRC_F32) ? 4 : 1;
    // ZPAD (fp32 frames): one buffer descriptor PER CHANNEL PLANE.  A pixel outside the frame -- column (lane offset 0x80000000), row
    // above (negative total offset) or below (past the plane: the range check sees voffset + soffset, tools/oob_probe.hip) -- then
    // loads 0.0f, and q0(0.0) = clamp8(z0) IS the pad value of the first layer (the zero point stands for 0.0): no row test, no
    // select of a pad word.  With one descriptor over the whole image a row below plane c would read plane c + 1.
    static constexpr bool ZPAD = SRC == SRC_F32;
    unsigned raw[NIT][4];
    bool ok[NIT];
    int voff[NIT], ty[NIT];
    __amdgpu_buffer_rsrc_t rs, rsp[ZPAD ? (NCH < 4 ? NCH : 4) : 1];
    int row_bytes, plane_bytes;
    InQuantV qc;
    __device__ __forceinline__ static float pin(float x) { float r = x; asm volatile("" : "+v"(r)); return r; }      // see in_vgpr()
    __device__ __forceinline__ void init(const ConvArgs &a, int n_img, int x0, int tid) {
        const size_t HW = (size_t)a.H * a.W;
        const size_t img = HW * a.ic * ESZ;
        char *frame = a.ft.n ? (char *)const_cast<void *>(a.ft.in[n_img]) : (char *)const_cast<void *>(a.in) + (size_t)n_img * img;      // ConvArgs::ft
        rs = __builtin_amdgcn_make_buffer_rsrc(frame, 0, (int)img, 0x00020000);
        row_bytes = a.W * ESZ;
        plane_bytes = (int)(HW * ESZ);
        if constexpr (ZPAD) {
#pragma unroll
            for (int c = 0; c < (NCH < 4 ? NCH : 4); ++c)
                rsp[c] = __builtin_amdgcn_make_buffer_rsrc(frame + (size_t)(c < a.ic ? c : 0) * plane_bytes, 0, plane_bytes, 0x00020000);
        }
        if constexpr (SRC != SRC_I8) {
            qc.xlo = pin(a.fd.xlo); qc.xhi = pin(a.fd.xhi); qc.r = pin(a.fd.r); qc.ns = pin(-a.s_in); qc.r2 = pin(a.fd.r2); qc.z = pin(a.z_in);
            qc.magic = pin(MAGIC);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            ty[it] = i / SWP;
            const int tx = i - ty[it] * SWP, gx = x0 - 2 + tx;
            const bool okx = (gx >= 0) & (gx < a.W) & (i < SH * SWP);
            voff[it] = okx ? (ty[it] * a.W + gx) * ESZ : (int)0x80000000;
            if (!okx) ty[it] = -(1 << 20);
        }
    }
    // Tiles of a run are stacked: the top CARRY = SH - TH rows of the next tile's staged window are the bottom CARRY rows of this
    // one's.  They are copied inside the LDS (store_next) instead of being loaded and quantised again: a following tile stages
    // only its TH new rows (element i -> row CARRY + i / SWP, same column), 864 pixels instead of 1152.
    static constexpr int CARRY = 4;
    static constexpr int NITN = ((SH - CARRY) * SWP + 255) / 256;
    template <bool FIRST>
    __device__ __forceinline__ void load_t(const ConvArgs &a, int y0) {
        constexpr int R0 = FIRST ? 0 : CARRY, N = FIRST ? NIT : NITN;
        const int soff = (y0 - 2 + R0) * row_bytes;        // FIRST: may be negative -> folded into the lane offset
        const int lo = 2 - y0 - R0, hi = a.H + 2 - y0 - R0;
#pragma unroll
        for (int it = 0; it < N; ++it) {
            int vo;
            if constexpr (ZPAD) {
                vo = FIRST ? voff[it] + soff : voff[it];      // no row test: the planes' own range checks return 0.0f = the pad value
            } else {
                ok[it] = (ty[it] >= lo) & (ty[it] < hi);
                vo = FIRST ? (ok[it] ? voff[it] + soff : (int)0x80000000) : voff[it];
            }
            if (!FIRST && CARRY && (it + 1) * 256 > (SH - CARRY) * SWP && threadIdx.x + it * 256 >= (SH - CARRY) * SWP) vo = (int)0x80000000;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                raw[it][c] = 0;
                if (has_channel(a, c)) {   // channel planes beyond ic are not loaded at all
                    const int so = (FIRST ? 0 : soff) + c * plane_bytes;
                    if constexpr (ZPAD) raw[it][c] = __builtin_amdgcn_raw_buffer_load_b32(rsp[c], vo, FIRST ? 0 : soff, 0);
                    else if constexpr (SRC == SRC_F32) raw[it][c] = __builtin_amdgcn_raw_buffer_load_b32(rs, vo, so, 0);
                    else raw[it][c] = (unsigned)(int)(signed char)__builtin_amdgcn_raw_buffer_load_b8(rs, vo, so, 0);
                }
            }
        }
    }
    __device__ __forceinline__ void load(const ConvArgs &a, int n_img, int x0, int y0, int tid) { load_t<false>(a, y0); }
    __device__ __forceinline__ void load_first(const ConvArgs &a, int n_img, int x0, int y0, int tid) { load_t<true>(a, y0); }
    template <bool FIRST>
    __device__ __forceinline__ void store_t(int4 *cp, const ConvArgs &a, int tid) const {
        constexpr int R0 = FIRST ? 0 : CARRY, N = FIRST ? NIT : NITN;
        int *cpw = reinterpret_cast<int *>(cp);
#pragma unroll
        for (int it = 0; it < N; ++it) {
            const int i = tid + it * 256;
            unsigned b[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if constexpr (SRC == SRC_F32)
                    b[c] = quantize_in_bits(__builtin_bit_cast(float, raw[it][c]), qc);
                else if constexpr (SRC == SRC_I8D)      // upstream net's int8 output: its float value, then this net's input quantiser
                    b[c] = quantize_in_bits(__fmul_rn((float)(int)raw[it][c] - a.z_prev, a.s_prev), qc);
                else
                    b[c] = raw[it][c];
                if (!has_channel(a, c)) b[c] = 0;
            }
            int word = (int)pack_lo_bytes(b[0], b[1], b[2], b[3]);
            if constexpr (!ZPAD) { if (!ok[it]) word = a.pad_word; }
            if (i < (SH - R0) * SWP) cpw[(i % SWP) * PITCH + (i / SWP) + R0] = word;      // column-major: dword [column][row], column pitch PITCH
        }
        // The last iteration's registers are loaded by the first tile only and are free afterwards; waves that skip that iteration
        // (no lane of theirs has an element) never wait for those loads, the registers are re-used inside the compute loop, and
        // hipcc's waitcnt pass then drains ALL loads in the loop's preheader -- the next tile's too, which are meant to fly during the
        // compute (first layer +0.6 us).  Retire them here, once per run.
        if constexpr (FIRST && CARRY > 0) __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0)
    }
    __device__ __forceinline__ void store(int4 *cp, const ConvArgs &a, int tid) const { store_t<true>(cp, a, tid); }
    // a following tile: rows SH-CARRY .. SH-1 of the tile under computation (cur, read-only by now) become rows 0 .. CARRY-1 of nxt
    // (a column's rows are adjacent dwords: two 8-byte moves per column), then the new rows
    __device__ __forceinline__ void store_next(int4 *nxt, const int4 *cur, const ConvArgs &a, int tid) const {
        static_assert(PITCH % 2 == 0 && (SH - 4) % 2 == 0 && 2 * SWP <= 256, "8-byte carry moves");
        const int cc = (tid >> 1) * PITCH + 2 * (tid & 1);      // the read is issued first, the quantiser's VALU work covers its latency
        int2 v = {0, 0};
        if (tid < 2 * SWP) v = *reinterpret_cast<const int2 *>(reinterpret_cast<const int *>(cur) + cc + (SH - 4));
        store_t<false>(nxt, a, tid);
        if (tid < 2 * SWP) *reinterpret_cast<int2 *>(reinterpret_cast<int *>(nxt) + cc) = v;
    }
};

constexpr int F5_TH = 12;      // first-layer tile height (rows): 8 is 1.3 us faster alone (shorter prologue), 12 and 16 re-quantise fewer halo pixels; with two frames in flight 12 gave +1 % (same-box A/B, round 2)
constexpr int F5_SH = F5_TH + 4;
constexpr int F5_SWP = MTW + 8;         // staged pixel columns (2 halo + 64 + 2 halo + over-read)
constexpr int F5_PITCH = F5_SH + 2;     // LDS column pitch in dwords (>= rows, = 2 mod 4)
static_assert(F5_PITCH % 4 == 2 && 3 * F5_PITCH + F5_SH < 256, "column pitch: bank rule / ds_read2_b32 offset range");
// RR (HYBS): the accumulator register whose rows can saturate (sesrq_api.hip: risky_reg), 4 = clamp all four
template <int MODE, int SRC, bool RC, int NCH, int RR = 4>
__device__ __forceinline__ void mfma_f5_body(const ConvArgs &a, int4 *buf0, int4 *buf1) {
    constexpr bool GENERAL = mode_general(MODE);
    constexpr int SH = F5_SH, SWP = F5_SWP, PITCH = F5_PITCH;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, n = l & 15, g = l >> 4;
    kernarg_warm<ConvArgs>();
    const BlockXY bxy = xcd_block(a.inv_nx);
    const int x0 = bxy.x * MTW, n_img = blockIdx.z;
    using Stage = StageFrame<SRC, SH, SWP, PITCH, NCH>;
    static_assert(F5_SH - F5_TH == 4, "StageFrame carries SH - TH = 4 rows");
    SESRQ_TILE_WALK_BEGIN(F5_TH, Stage)      // the frame loads of the first tile go out before anything else is fetched
    const int4 *fr = a.afrag;
    constexpr bool BIASED = mode_biased(MODE);     // requant without v_cvt: sums carry + MAGIC_I (needs |s| < 2^22)
    int4 ac = fr[g];
    if constexpr (BIASED) { ac.x += MAGIC_I; ac.y += MAGIC_I; ac.z += MAGIC_I; ac.w += MAGIC_I; }
    constexpr int NPE = GENERAL ? 4 : 1;
    v4i A[2][NPE];
    v4i AR[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    int idx_o = 0, idx_r = 0;           // HYBS: positions of the kept elements inside a pixel's four K slots, the same for every group
    if constexpr (MODE == HYBS) {       // A[0][0] = the other two channels' sparse image, AR[0] = the risky channel's
        A[0][0] = ld_frag(a.afrag_sp + 4 + l);
        A[1][0] = A[0][0];
        AR[0] = ld_frag(a.afrag_sp + 4 + 64 + l);
        const int pr = a.risky_pe, ca = pr == 0 ? 1 : 0, cb = pr == 2 ? 1 : 2;
        idx_o = (ca | (cb << 2)) * 0x11111111;
        idx_r = (pr | (((pr + 1) & 3) << 2)) * 0x11111111;
    } else {
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int p = 0; p < NPE; ++p) A[f][p] = ld_frag(fr + 4 + (f * NPE + p) * 64 + l);
    }
    if constexpr (MODE == HYB) {
#pragma unroll
        for (int f = 0; f < 2; ++f) AR[f] = ld_frag(a.afrag2 + 4 + (f * 4 + a.risky_pe) * 64 + l);
    }
    // lane group -> first pixel of its operand per K-chunk; must match pack_mfma_frags (MFMA_F5)
    //   chunk 0: row g, 4 horizontally adjacent pixels          (dwords +0 +P +2P +3P, P = column pitch)
    //   chunk 1: pattern {(0,0),(1,0),(1,1),(1,2)} translated by f5_tr(g)   (dwords +0 +1 +1+P +1+2P)
    int tr_r, tr_c;
    f5_tr(g, tr_r, tr_c);
    const int addr0 = ((16 * w + n) * PITCH + g) * 4, addr1 = ((16 * w + n + tr_c) * PITCH + tr_r) * 4;     // bytes
    const float zlo = a.relu ? fmaxf(a.z_next, -128.f) : -128.f;
    const int gx = x0 + 16 * w + n;
    // hybrid: s = (add constant + the three safe PEs) + clamp18(risky PE).  The risky PE's chain runs FIRST with the add constant as
    // its C input, is clamped against per-row bounds shifted by that constant, and is then the C input of the other chain: one
    // v_med3_i32 per value instead of a clamp and an add.
    int rlo[4], rhi[4];
    {
        const int acv[4] = {ac.x, ac.y, ac.z, ac.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { rlo[i] = acv[i] - 131072; rhi[i] = acv[i] + 131071; }
    }
    auto compute = [&](const int4 *cp, int y0) __attribute__((always_inline)) {
        const RowIO io = make_rowio(a, n_img, y0, gx, g);
        typedef const int __attribute__((address_space(3))) *lds_int_t;
        const unsigned tb = (unsigned)(size_t)(const __attribute__((address_space(3))) void *)cp;     // LDS byte address of the tile
        unsigned b0 = tb + addr0, b1 = tb + addr1;
#pragma unroll 1
        for (int y4 = 0; y4 < F5_TH; y4 += 4) {
            int s4[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // vertically adjacent pixels are adjacent dwords: keep the compiler from recognising that row y + 1 re-reads dwords of
                // row y -- it would load them once and MOVE them into the next operand (the reads are not what bounds this loop)
                asm("" : "+v"(b0), "+v"(b1));
                const lds_int_t p0 = (lds_int_t)(size_t)(b0 + 4 * r), p1 = (lds_int_t)(size_t)(b1 + 4 * r);
                const v4i B0 = {p0[0], p0[PITCH], p0[2 * PITCH], p0[3 * PITCH]}, B1 = {p1[0], p1[1], p1[1 + PITCH], p1[1 + 2 * PITCH]};
                v4i acc[GENERAL ? 4 : 1];
                const v4i zero = {0, 0, 0, 0};
                if constexpr (GENERAL) {
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        acc[p] = mfma(A[0][p], B0, zero);
                        acc[p] = mfma(A[1][p], B1, acc[p]);
                    }
                    if constexpr (MODE == GEN_TAP) tap_sums<4>(acc, a, n_img, y0 + y4 + r, gx, g, 0);
                    finish_sums<MODE>(s4[r], acc, ac, a);
                } else {
                    const v4i acc0 = {ac.x, ac.y, ac.z, ac.w};
                    if constexpr (MODE == HYBS) {      // one sparse MFMA per chain over all 32 pixel taps (B0 and B1 side by side)
                        const v8i B8 = {B0[0], B0[1], B0[2], B0[3], B1[0], B1[1], B1[2], B1[3]};
                        v4i rk = smfmac(AR[0], B8, acc0, idx_r);
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (RR == 4 || RR == i) rk[i] = med3_biased(rk[i], rlo[i], rhi[i]);
                        acc[0] = smfmac(A[0][0], B8, rk, idx_o);
                    } else if constexpr (MODE == HYB) {       // same B operands, A masked to the risky PE's channel
                        v4i rk = mfma(AR[0], B0, acc0);
                        rk = mfma(AR[1], B1, rk);
#pragma unroll
                        for (int i = 0; i < 4; ++i) rk[i] = med3_biased(rk[i], rlo[i], rhi[i]);
                        acc[0] = mfma(A[0][0], B0, rk);
                        acc[0] = mfma(A[1][0], B1, acc[0]);
                    } else {
                        acc[0] = mfma(A[0][0], B0, acc0);
                        acc[0] = mfma(A[1][0], B1, acc[0]);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) s4[r][i] = acc[0][i];
                }
            }
            // no separate residual tensor <=> zero[1] == -128 (sesrq_create) <=> this layer's z_next == -128: the cvt_pk_u8 epilogue
            if constexpr (!RC && BIASED) {      // wave-uniform: the one-fma requant where (M, n) passed its proof
                if (a.direct) emit_rows4<EPI_MID, RC, BIASED, 2>(s4, a, io, y4, zlo);
                else emit_rows4<EPI_MID, RC, BIASED, 1>(s4, a, io, y4, zlo);
            } else {
                emit_rows4<EPI_MID, RC, BIASED, !RC ? 1 : 0>(s4, a, io, y4, zlo);
            }
            b0 += 16; b1 += 16;
        }
    };
#define SESRQ_COMPUTE(B) compute(B, y0);
    SESRQ_TILE_WALK_CARRY_REST(F5_TH, buf0, buf1, SESRQ_COMPUTE)
#undef SESRQ_COMPUTE
}
// 4 waves per SIMD for the merged / hybrid first layer: with the default heuristics hipcc takes 141-153 VGPRs here (3 waves);
// told to fit 128 it does so without spilling and the extra wave hides more of the staging/quantisation latency (-5 %).
// NOT for the per-PE (general) variants: squeezed to 128 VGPRs, hipcc 7.2 produced a GEN_STD + residual-tensor instance whose
// first output word was garbage in lanes 28..31 (caught by the satw_zeros golden vectors; the same source without the
// attribute, or with unrelated extra code in the loop, is correct) -- those take the registers they ask for.
template <int MODE, int SRC, bool RC, int NCH, int RR = 4>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void mfma_f5_kernel_w4(const ConvArgs a) {
    __shared__ int4 buf0[(F5_SWP * F5_PITCH + 3) / 4], buf1[(F5_SWP * F5_PITCH + 3) / 4];      // SH rows of 4-byte pixels
    mfma_f5_body<MODE, SRC, RC, NCH, RR>(a, buf0, buf1);
}
template <int MODE, int SRC, bool RC, int NCH>
__global__ __launch_bounds__(256) void mfma_f5_kernel(const ConvArgs a) {
    __shared__ int4 buf0[(F5_SWP * F5_PITCH + 3) / 4], buf1[(F5_SWP * F5_PITCH + 3) / 4];
    mfma_f5_body<MODE, SRC, RC, NCH>(a, buf0, buf1);
}

#ifdef SESRQ_STAMPS
static int *g_stampbuf = nullptr;
extern "C" int sesrq_debug_fetch_stamps(void *host, size_t bytes) {
    return g_stampbuf ? (int)hipMemcpy(host, g_stampbuf, bytes, hipMemcpyDeviceToHost) : -1;
}
#endif
// Persistent walk geometry: one round of workgroups that exactly fits the chip.  The number of
// co-resident workgroups per CU comes from the occupancy API for THIS kernel (registers / LDS differ a
// lot between the merged and general variants); a strip's row tiles are then cut into the largest
// number of equal vertical runs that still fits.
template <auto KERN>
static void launch(ConvArgs a, hipStream_t st, int tile_h = MTH) {
    const auto kern = KERN;
    static std::mutex mu;
    static std::map<const void *, int> occ;          // per kernel (all instantiations share this function type)
    const int num_cu = device_cu_count();
    int blocks_per_cu;
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = occ.find((const void *)kern);
        if (it == occ.end()) {
            int b = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, kern, 256, 0) != hipSuccess || b < 1) b = 2;
            it = occ.emplace((const void *)kern, b).first;
        }
        blocks_per_cu = it->second;
    }
    const int strips = (a.W + MTW - 1) / MTW, row_tiles = (a.H + tile_h - 1) / tile_h;
    long long k = (a.wg_budget > 0 ? (long long)a.wg_budget : (long long)blocks_per_cu * num_cu) / ((long long)strips * a.N);
    k = std::max(1LL, std::min<long long>(k, row_tiles));
    a.chunk_tiles = (int)((row_tiles + k - 1) / k);
    a.run_q = (int)(row_tiles / k);
    a.run_rem = (int)(row_tiles % k);
    a.inv_nx = ((long long)strips * k < 65536 && strips < 65536) ? (unsigned)((0x100000000ULL + (unsigned)strips - 1) / (unsigned)strips) : 0u;
    dim3 grid(strips, (int)k, a.N);
    launch_kernel<KERN>(grid, dim3(256), 0, st, a);
}

// hybrid first layer.  3 input channels: always the sparse MFMA (sesrq_create packs the 2:4 images for every 3-channel layer with exactly
// one risky PE -- PE 3 holds no channel), by the register that needs the clamp; other channel counts: the dense hybrid kernel.
template <int SRC, bool RC, int NCH>
static int launch_f5_hybrid(const ConvArgs &a, hipStream_t st) {
    if constexpr (NCH == 3) {
        if (!a.afrag_sp) { set_error("mfma: hybrid 3-channel first layer without its sparse weight image"); return 1; }
        switch (a.risky_reg) {
            case 0: launch<mfma_f5_kernel_w4<HYBS, SRC, RC, 3, 0>>(a, st, F5_TH); break;
            case 1: launch<mfma_f5_kernel_w4<HYBS, SRC, RC, 3, 1>>(a, st, F5_TH); break;
            case 2: launch<mfma_f5_kernel_w4<HYBS, SRC, RC, 3, 2>>(a, st, F5_TH); break;
            case 3: launch<mfma_f5_kernel_w4<HYBS, SRC, RC, 3, 3>>(a, st, F5_TH); break;
            default: launch<mfma_f5_kernel_w4<HYBS, SRC, RC, 3, 4>>(a, st, F5_TH); break;
        }
    } else {
        launch<mfma_f5_kernel_w4<HYB, SRC, RC, NCH>>(a, st, F5_TH);
    }
    return 0;
}

#define SESRQ_BY_MODE(KERN, ...)                                                         \
    do {                                                                                 \
        if (mode == MERGED) launch<KERN<MERGED, __VA_ARGS__>>(a, st);                    \
        else if (mode == GEN_STD) launch<KERN<GEN_STD, __VA_ARGS__>>(a, st);             \
        else if (mode == HYB) launch<KERN<HYB, __VA_ARGS__>>(a, st);                     \
        else launch<KERN<GEN_ANY, __VA_ARGS__>>(a, st);                                  \
    } while (0)
// the one-fma last-layer flavours exist for the biased modes only (GEN_ANY sums carry no bias: it keeps the general store)
#define SESRQ_BY_MODE_B(KERN, ...)                                                       \
    do {                                                                                 \
        if (mode == MERGED) launch<KERN<MERGED, __VA_ARGS__>>(a, st);                    \
        else if (mode == GEN_STD) launch<KERN<GEN_STD, __VA_ARGS__>>(a, st);             \
        else launch<KERN<HYB, __VA_ARGS__>>(a, st);                                      \
    } while (0)

int launch_mfma(const LayerPlan &lp, const ConvArgs &a_in, int src, int epi, bool general, hipStream_t st, bool one_risky_pe, bool tap) {
    ConvArgs a = a_in;
#ifdef SESRQ_STAMPS
    if (!g_stampbuf) { (void)hipMalloc((void **)&g_stampbuf, 1 << 22); (void)hipMemset(g_stampbuf, 0, 1 << 22); }
    if (epi == (getenv("SESRQ_STAMP_EPI") ? atoi(getenv("SESRQ_STAMP_EPI")) : 0) && lp.mfma_kind == (getenv("SESRQ_STAMP_KIND") ? atoi(getenv("SESRQ_STAMP_KIND")) : MFMA_H3)) a.dbg_pe = g_stampbuf;
#endif
    if ((size_t)a.H * a.W * 16 >= ((size_t)1 << 28)) { set_error("mfma: frame too large for 32-bit buffer offsets (H*W must stay below 2^24 pixels)"); return 1; }
    const bool std_bits = a.acc_lo == -131072 && a.acc_hi == 131071 && a.add_lo == -524288 && a.add_hi == 524287;
    const int mode = tap ? GEN_TAP : (!general ? MERGED : (std_bits ? (one_risky_pe ? HYB : GEN_STD) : GEN_ANY));
    if (mode == GEN_TAP) {      // debug forward with PE taps: the per-PE kernels in their run-time-bounds form, every output kind
        if (lp.d_afrag_pesplit && a.afrag == lp.d_afrag_pesplit) { set_error("mfma: the pe-split last-layer kernel has no PE taps"); return 1; }
        switch (lp.mfma_kind) {
            case MFMA_H3:
                if (epi == EPI_MID) launch<mfma_h3_kernel<GEN_TAP, EPI_MID>>(a, st);
                else if (epi == EPI_PRERES) launch<mfma_h3_kernel<GEN_TAP, EPI_PRERES>>(a, st);
                else { set_error("mfma: 3x3 last layer not supported"); return 1; }
                break;
            case MFMA_H5:
                if (epi == EPI_MID) launch<mfma_h5_kernel<GEN_TAP, EPI_MID>>(a, st);
                else if (epi == EPI_PRERES) launch<mfma_h5_kernel<GEN_TAP, EPI_PRERES>>(a, st);
                else if (last_nv(a.oc) == 3) launch<mfma_h5_kernel<GEN_TAP, EPI_LAST, 0, 3>>(a, st);
                else launch<mfma_h5_kernel<GEN_TAP, EPI_LAST>>(a, st);
                break;
            case MFMA_F5:
                if (src == SRC_F32) { if (a.rc_out) launch<mfma_f5_kernel<GEN_TAP, SRC_F32, true, 4>>(a, st, F5_TH); else launch<mfma_f5_kernel<GEN_TAP, SRC_F32, false, 4>>(a, st, F5_TH); }
                else if (src == SRC_I8D) { if (a.rc_out) launch<mfma_f5_kernel<GEN_TAP, SRC_I8D, true, 4>>(a, st, F5_TH); else launch<mfma_f5_kernel<GEN_TAP, SRC_I8D, false, 4>>(a, st, F5_TH); }
                else { if (a.rc_out) launch<mfma_f5_kernel<GEN_TAP, SRC_I8, true, 4>>(a, st, F5_TH); else launch<mfma_f5_kernel<GEN_TAP, SRC_I8, false, 4>>(a, st, F5_TH); }
                break;
            default: set_error("mfma: layer shape not supported by the MFMA engine"); return 1;
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { set_error(std::string("mfma launch failed: ") + hipGetErrorString(e)); return 1; }
        return 0;
    }
    const bool dl = a.direct && mode != GEN_ANY && epi == EPI_LAST && a.z_out == -128.f && !a.anchor;
    const bool d1 = dl && a.direct == 1, d2 = dl && a.direct == 2;
    switch (lp.mfma_kind) {
        case MFMA_H3:
            if (epi == EPI_MID) SESRQ_BY_MODE(mfma_h3_kernel, EPI_MID);
            else if (epi == EPI_PRERES) SESRQ_BY_MODE(mfma_h3_kernel, EPI_PRERES);
            else { set_error("mfma: 3x3 last layer not supported"); return 1; }
            break;
        case MFMA_H5:
            if (epi == EPI_LAST && lp.d_afrag_pesplit && a.afrag == lp.d_afrag_pesplit) {
                if (mode == MERGED) launch<mfma_h5p_kernel<MERGED>>(a, st);
                else if (mode == GEN_ANY) launch<mfma_h5p_kernel<GEN_ANY>>(a, st);
                else launch<mfma_h5p_kernel<GEN_STD>>(a, st);
                break;
            }
            if (epi == EPI_MID) SESRQ_BY_MODE(mfma_h5_kernel, EPI_MID);
            else if (epi == EPI_PRERES) SESRQ_BY_MODE(mfma_h5_kernel, EPI_PRERES);
            // (below) d1: the output requant as one fma -- proven for this layer's (M, n), zero point -128, biased sums, int8 output only

            // fp32 frame + the x2 anchor add on the pair map (3 -> 12 channels, PixelShuffle 2): OUTF = 2; the one-fma forms do not care about the anchor
            else if (!a.out_q && a.out_f && a.anchor && last_nv(a.oc) == 3 && last_pairmap(a.oc, a.ps) && a.ic == 16) {
                const bool da = a.direct && mode != GEN_ANY && a.z_out == -128.f;
                if (da && a.direct == 1) SESRQ_BY_MODE_B(mfma_h5_kernel, EPI_LAST, 12, 3, 2); else if (da && a.direct == 2) SESRQ_BY_MODE_B(mfma_h5_kernel, EPI_LAST, 22, 3, 2); else SESRQ_BY_MODE(mfma_h5_kernel, EPI_LAST, 2, 3, 2);
            }
            // fp32 frame only (the reference's return type), no anchor: the OUTF flavours of the same FAST instances
            else if (!a.out_q && a.out_f && !a.anchor && last_nv(a.oc) == 3 && last_pairmap(a.oc, a.ps)) { if (d1) SESRQ_BY_MODE_B(mfma_h5_kernel, EPI_LAST, 12, 3, 1); else if (d2) SESRQ_BY_MODE_B(mfma_h5_kernel, EPI_LAST, 22, 3, 1); else SESRQ_BY_MODE(mfma_h5_kernel, EPI_LAST, 2, 3, 1); }
            else if (!a.out_q && a.out_f && !a.anchor && last_nv(a.oc) == 4 && a.ps == 2) { if (d1) SESRQ_BY_MODE_B(mfma_h5_kernel, EPI_LAST, 12, 4, 1); else if (d2) SESRQ_BY_MODE_B(mfma_h5_kernel, EPI_LAST, 22, 4, 1); else SESRQ_BY_MODE(mfma_h5_kernel, EPI_LAST, 2, 4, 1); }
            else if (!a.out_q && a.out_f && !a.anchor && last_nv(a.oc) == 4 && a.ps == 4) { if (d1) SESRQ_BY_MODE_B(mfma_h5_kernel, EPI_LAST, 14, 4, 1); else if (d2) SESRQ_BY_MODE_B(mfma_h5_kernel, EPI_LAST, 24, 4, 1); else SESRQ_BY_MODE(mfma_h5_kernel, EPI_LAST, 4, 4, 1); }
            else if (last_nv(a.oc) == 3) {       // up to 12 output channels: three real rows per lane group (must match pack_mfma_frags)
                if (a.out_q && !a.out_f && last_pairmap(a.oc, a.ps)) { if (d1) SESRQ_BY_MODE_B(mfma_h5_kernel, EPI_LAST, 12, 3); else if (d2) SESRQ_BY_MODE_B(mfma_h5_kernel, EPI_LAST, 22, 3); else SESRQ_BY_MODE(mfma_h5_kernel, EPI_LAST, 2, 3); }
                else SESRQ_BY_MODE(mfma_h5_kernel, EPI_LAST, 0, 3);
            }
            else if (a.out_q && !a.out_f && a.ps == 2) { if (d1) SESRQ_BY_MODE_B(mfma_h5_kernel, EPI_LAST, 12); else if (d2) SESRQ_BY_MODE_B(mfma_h5_kernel, EPI_LAST, 22); else SESRQ_BY_MODE(mfma_h5_kernel, EPI_LAST, 2); }
            else if (a.out_q && !a.out_f && a.ps == 4) { if (d1) SESRQ_BY_MODE_B(mfma_h5_kernel, EPI_LAST, 14); else if (d2) SESRQ_BY_MODE_B(mfma_h5_kernel, EPI_LAST, 24); else SESRQ_BY_MODE(mfma_h5_kernel, EPI_LAST, 4); }
            else SESRQ_BY_MODE(mfma_h5_kernel, EPI_LAST);
            break;
        case MFMA_F5:
#define SESRQ_F5(...)                                                                    \
    do {                                                                                 \
        if (mode == MERGED) launch<mfma_f5_kernel_w4<MERGED, __VA_ARGS__>>(a, st, F5_TH);       \
        else if (mode == HYB) { if (launch_f5_hybrid<__VA_ARGS__>(a, st)) return 1; }    \
        else if (mode == GEN_STD) launch<mfma_f5_kernel<GEN_STD, __VA_ARGS__>>(a, st, F5_TH);   \
        else launch<mfma_f5_kernel<GEN_ANY, __VA_ARGS__>>(a, st, F5_TH);                        \
    } while (0)
#define SESRQ_F5_NCH(...)                                                      \
    do {                                                                       \
        if (a.ic == 3) SESRQ_F5(__VA_ARGS__, 3);                               \
        else if (a.ic == 1) SESRQ_F5(__VA_ARGS__, 1);                          \
        else SESRQ_F5(__VA_ARGS__, 4);                                         \
    } while (0)
            if (src == SRC_F32) { if (a.rc_out) SESRQ_F5_NCH(SRC_F32, true); else SESRQ_F5_NCH(SRC_F32, false); }
            else if (src == SRC_I8D) { if (a.rc_out) SESRQ_F5_NCH(SRC_I8D, true); else SESRQ_F5_NCH(SRC_I8D, false); }
            else { if (a.rc_out) SESRQ_F5_NCH(SRC_I8, true); else SESRQ_F5_NCH(SRC_I8, false); }
#undef SESRQ_F5_NCH
#undef SESRQ_F5
            break;
        default: set_error("mfma: layer shape not supported by the MFMA engine"); return 1;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("mfma launch failed: ") + hipGetErrorString(e)); return 1; }
    return 0;
}

}  // namespace sesrq
