// sesrq fused front: the first 5x5 layer (with the input quantiser) AND the hidden trio in ONE launch -- the reference's conv 0..3
// (myQL/quan_func.py:217-293 input quantiser, :244-280 residual merge).  On top of what sesrq_trio.hip saves, the layer-0 output
// never goes to HBM at all: it is produced straight into the trio's input window in LDS, and the long-residual operand
// rc = clamp8(rint(relu(t0) - 128)) -- which IS the layer-0 output word when zero[1] == -128, the only case fused here -- is read
// back from that window eleven rows later.  Per 1080p frame: one launch, 100 MB of HBM traffic (33 MB write + 2 x 33 MB read) and
// the trio's whole global staging less than first-layer kernel + trio; every input pixel is quantised once per strip (the
// stand-alone first-layer kernel re-quantises its 4-row halo for every tile).
//
// Geometry = the trio's (strip of 64 computed / 60 valid columns, steps of 8 rows, 4 waves), plus a fourth rolling window:
//      RAW: quantised frame rows Y+1 .. Y+12, one dword per pixel (byte c = channel c), 80-dword pitch, columns x0-3 ..
//      IN : layer-0 rows  Y   .. Y+10   (11 rows: row Y is kept for the residual operand of output row Y)
//      A  : layer-a rows  Y   .. Y+9          B: layer-b rows Y-1 .. Y+8
// Step Y: phase 0 turns RAW positions i .. i+4 into IN position 3+i (8 rows x 66 columns -- the trio reads 66 input columns:
// 32 row-groups of 16 columns dealt to the four waves by ROW, two rows each, plus one group made of the two edge columns of all
// eight rows), then the trio's three phases run unchanged.
// Between steps the last 4 / 3 / 2 / 2 rows of RAW / IN / A / B move to the top of their windows.  4 barriers per step.
// A run starts cold with one step of phases 0, a, b without output (RAW loaded whole).
// First-layer accumulation modes: merged and hybrid (load-time proof, sesrq_api.hip); anything else runs unfused.
#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <type_traits>

#include "sesrq_mfma_common.h"

namespace sesrq {

constexpr int TV = 60, TH = 8, TP = 66, TR = TH + 2;      // the trio's geometry (sesrq_trio.hip)
constexpr int QIR = TR + 1;               // IN window rows
constexpr int QRR = TH + 4;               // RAW window rows
constexpr int QRP = 80;                   // RAW row pitch (dwords) = 16 mod 32 banks: operand rows g, g+1 hit disjoint banks
constexpr int QRC = 72;                   // staged RAW columns: 66 + 4 taps (+ 2 pad); columns beyond only feed dropped outputs
constexpr int OOB = (int)0x80000000;
constexpr int Q_WIN_I = QIR * TP, Q_WIN = TR * TP;                      // pixels (int4)
constexpr int Q_RAW_I4 = (QRR * QRP + 8) / 4;                           // int4 units (+ 8 dwords: the last rows' pattern over-read ends at dword 965)
constexpr int Q_FRAG_I4 = 4 + 4 * 64;                                    // first layer: add constants + 2 merged + 2 risky-PE fragments
constexpr int Q_ACC_I4 = 3 * 4;                                           // the trio's add constants (+ requant bias)
constexpr int QUAD_LDS_BYTES = (Q_WIN_I + 2 * Q_WIN + Q_RAW_I4 + Q_FRAG_I4 + Q_ACC_I4) * 16;      // 40 960 B: four workgroups per CU

// RAW staging: the frame's pixels of 8 (cold: 12) rows x 72 columns, quantised to one dword each
template <int SRC, int NCH>       // NCH: channels as a compile-time count (1, 3), or 4 = a.ic tested per channel (StageFrame)
struct QuadStage {
    __device__ __forceinline__ static bool has_channel(const QuadArgs &a, int c) { return NCH < 4 ? c < NCH : c < a.ic; }
    static constexpr int NIT_COLD = (QRR * QRC + 255) / 256;      // 4
    static constexpr int NIT = (TH * QRC + 255) / 256;            // 3
    static constexpr int ESZ = (SRC == SRC_F32) ? 4 : 1;
    unsigned raw[NIT_COLD][3];      // ic <= 3 in registers ... the 4th channel is loaded only if the net has one
    unsigned raw3[NIT_COLD];
    bool ok[NIT_COLD];
    int voff[NIT_COLD], ty[NIT_COLD], lds[NIT_COLD];
    __amdgpu_buffer_rsrc_t rs;
    int row_bytes, plane_bytes;
    __device__ __forceinline__ void init(const QuadArgs &a, int n_img, int x0, int tid) {
        const size_t HW = (size_t)a.t.H * a.t.W;
        const size_t img = HW * a.ic * ESZ;
        rs = __builtin_amdgcn_make_buffer_rsrc((char *)const_cast<void *>(a.frame) + (size_t)n_img * img, 0, (int)img, 0x00020000);
        row_bytes = a.t.W * ESZ;
        plane_bytes = (int)(HW * ESZ);
#pragma unroll
        for (int it = 0; it < NIT_COLD; ++it) {
            const int i = tid + it * 256;
            ty[it] = i / QRC;
            const int tx = i - ty[it] * QRC, gx = x0 - 3 + tx;
            lds[it] = ty[it] * QRP + tx;
            const bool okx = (gx >= 0) & (gx < a.t.W) & (i < QRR * QRC);
            voff[it] = okx ? (ty[it] * a.t.W + gx) * ESZ : OOB;
            if (!okx) ty[it] = -(1 << 20);
        }
    }
    // rows [y, y + nrows) of the frame (COLD: 12 rows, any y, lane-form offsets; else 8 rows, y > 0, one scalar offset)
    template <bool COLD>
    __device__ __forceinline__ void load(const QuadArgs &a, int y) {
        constexpr int nit = COLD ? NIT_COLD : NIT, nrows = COLD ? QRR : TH;
        const int lo = -y, hi = min(a.t.H - y, nrows);
        const int soff = y * row_bytes;
#pragma unroll
        for (int it = 0; it < nit; ++it) {
            ok[it] = (ty[it] >= lo) & (ty[it] < hi);
            const int vo = COLD ? (ok[it] ? voff[it] + soff : OOB) : ((ty[it] < nrows) ? voff[it] : OOB);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                unsigned v = 0;
                if (has_channel(a, c)) {   // channel planes beyond ic are not loaded at all
                    const int so = (COLD ? 0 : soff) + c * plane_bytes;
                    if constexpr (SRC == SRC_F32) v = __builtin_amdgcn_raw_buffer_load_b32(rs, vo, so, 0);
                    else v = (unsigned)(int)(signed char)__builtin_amdgcn_raw_buffer_load_b8(rs, vo, so, 0);
                }
                if (c < 3) raw[it][c] = v; else raw3[it] = v;
            }
        }
    }
    // quantise (myQL/quan_func.py:225) and write: COLD -> window rows 0 .. 11, else rows 4 .. 11
    template <bool COLD>
    __device__ __forceinline__ void store(int *win, const QuadArgs &a, int tid) const {
        constexpr int nit = COLD ? NIT_COLD : NIT, nrows = COLD ? QRR : TH, row0 = COLD ? 0 : 4;
#pragma unroll
        for (int it = 0; it < nit; ++it) {
            const int i = tid + it * 256;
            unsigned b[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const unsigned rv = c < 3 ? raw[it][c] : raw3[it];
                if constexpr (SRC == SRC_F32) b[c] = quantize_in_bits(__builtin_bit_cast(float, rv), a.s_in, a.z_in, a.fd);
                else if constexpr (SRC == SRC_I8D) b[c] = quantize_in_bits(__fmul_rn((float)(int)rv - a.z_prev, a.s_prev), a.s_in, a.z_in, a.fd);
                else b[c] = rv;
                if (!has_channel(a, c)) b[c] = 0;
            }
            int word = (int)pack_lo_bytes(b[0], b[1], b[2], b[3]);
            if (!ok[it]) word = a.pad_raw;
            if (i < nrows * QRC) win[row0 * QRP + lds[it]] = word;
        }
    }
};

struct QuadEpiC {
    float Mf, sh, z_next, Mres, shres, z_merge;
};
struct QuadEpi0 {
    float Mf, sh, z_next;
    int acc_lo, acc_hi, add_lo, add_hi;      // unused (merged / hybrid first layer); finish_sums' generic branch names them
};

template <int MODE0, int SRC, int NCH>
__global__ __launch_bounds__(256) void mfma_quad_kernel(const QuadArgs a) {
    extern __shared__ int4 quad_lds[];
    int4 *bufI = quad_lds, *bufA = bufI + Q_WIN_I, *bufB = bufA + Q_WIN;
    int *rawW = reinterpret_cast<int *>(bufB + Q_WIN);
    int4 *frg0 = bufB + Q_WIN + Q_RAW_I4;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, n = l & 15, g = l >> 4;
    const int n_img = blockIdx.z;
    const int H = a.t.H, W = a.t.W;
    const int x0 = blockIdx.x * TV - 2;              // frame column of computed column 0
    const int steps_total = (H + TH - 1) / TH;
    const int s_begin = (int)(((long long)blockIdx.y * steps_total) / gridDim.y);
    const int s_end = (int)(((long long)(blockIdx.y + 1) * steps_total) / gridDim.y);
    if (s_begin >= s_end) return;

    const int c = 16 * w + n, gx = x0 + c;
    const bool col_in = (gx >= 0) & (gx < W);
    const bool col_out = (c >= 2) & (c < 2 + TV) & (gx < W);
    const int rdcol = c + g;
    const int wrcol = (c + 1) * 4 + g;

    // Register budget (128 VGPRs = 4 waves per SIMD): a phase holds only ITS layer's three A fragments; the next phase's are
    // fetched from global memory (9 KB, L1/L2-resident) while the current phase runs, into the other of two register sets.  Add
    // constants and the first layer's fragments are read from LDS at the start of their phase.
    int4 *accL = frg0 + Q_FRAG_I4;
    if (tid < 12) {
        int4 v = a.t.l[tid >> 2].afrag[tid & 3];
        v.x += MAGIC_I; v.y += MAGIC_I; v.z += MAGIC_I; v.w += MAGIC_I;
        accL[tid] = v;
    }
    v4i WP[3], WQ[3];
    auto ldw = [&](v4i (&Wd)[3], int k) __attribute__((always_inline)) {
#pragma unroll
        for (int f = 0; f < 3; ++f) Wd[f] = ld_frag(a.t.l[k].afrag + 4 + f * 64 + l);
    };
    {   // first layer: [0..3] add constants (+ requant bias), [4 + f*64 + lane] merged fragments f = 0, 1, then the risky PE's two
        if (tid < 4) {
            int4 v = a.afrag0[tid];
            v.x += MAGIC_I; v.y += MAGIC_I; v.z += MAGIC_I; v.w += MAGIC_I;
            frg0[tid] = v;
        }
        if (tid < 128) frg0[4 + tid] = a.afrag0[4 + tid];
        else if (MODE0 == HYB) {
            const int f = (tid - 128) >> 6, ln = tid & 63;
            frg0[4 + 128 + (tid - 128)] = a.afrag0r[4 + (f * 4 + a.risky_pe) * 64 + ln];
        }
    }
    const size_t img = (size_t)H * W * 16;
    RowIO io;
    io.out = __builtin_amdgcn_make_buffer_rsrc((char *)a.t.out + (size_t)n_img * img, 0, (int)img, 0x00020000);
    io.rc_in = io.out;
    io.rc_out = io.out;
    io.row_bytes = W * 16;
    const int voff_c = col_out ? (g * W + gx) * 16 : OOB;
    const QuadEpiC ec = {a.t.l[2].Mf, a.t.l[2].sh, a.t.l[2].z_next, a.t.Mres, a.t.shres, a.t.z_merge};
    const QuadEpi0 e0 = {a.Mf0, a.sh0, a.z1, 0, 0, 0, 0};

    // ---- phase 0: first layer, RAW positions i .. i+4 -> IN position 3+i, rows i = 2w, 2w+1 of this wave, five column groups
    // lane group -> first pixel of its operand per K-chunk (must match pack_mfma_frags, MFMA_F5):
    //   chunk 0: row g, 4 horizontally adjacent pixels;  chunk 1: pattern {(0,0),(1,0),(1,1),(1,2)} translated by f5_tr(g)
    int tr_r, tr_c;
    f5_tr(g, tr_r, tr_c);
    const int a0 = g * QRP + n, a1 = tr_r * QRP + n + tr_c;
    unsigned colmask = 0;                 // bit gi: window column 16*gi + n is inside the frame
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
        const int ci = 16 * gi + n, x = x0 - 1 + ci;
        if (ci < TP && x >= 0 && x < W) colmask |= 1u << gi;
    }
    // one row-group: B operands at p0 / p1 (dword pointers of this lane), result word -> IN dword `dst` (or the pad word)
    auto item0 = [&](const int *p0, const int *p1, const v4i &F0, const v4i &F1, const v4i &R0, const v4i &R1, const v4i &ac,
                     const int4 &aci, bool keep, bool wr, unsigned *dst) __attribute__((always_inline)) {
        const v4i zero = {0, 0, 0, 0};
        const v4i B0 = {p0[0], p0[1], p0[2], p0[3]}, B1 = {p1[0], p1[QRP], p1[QRP + 1], p1[QRP + 2]};
        v4i acc[2];
        acc[0] = mfma(F0, B0, ac);
        acc[0] = mfma(F1, B1, acc[0]);
        if constexpr (MODE0 == HYB) {
            acc[1] = mfma(R0, B0, zero);
            acc[1] = mfma(R1, B1, acc[1]);
        }
        int s[4];
        finish_sums<MODE0>(s, acc, aci, e0);
        unsigned q = epi_mid<true>(s, e0, a.zlo0);
        q = keep ? q : (unsigned)a.t.pad_in;
        if (wr) *dst = q;
    };
    // 8 rows x 66 columns = 8 x 4 groups of 16 columns (two rows per wave) + ONE group made of the two edge columns 64, 65 of all
    // eight rows (lane n -> row n >> 1, column 64 + (n & 1): the B operand address is per lane anyway), done by wave 3
    const bool col_e = (x0 - 1 + 64 + (n & 1) >= 0) & (x0 - 1 + 64 + (n & 1) < W);
    auto phase0 = [&](int Y) __attribute__((always_inline)) {
        const v4i ac = ld_frag(frg0 + g);
        const int4 aci = make_int4(ac[0], ac[1], ac[2], ac[3]);
        const v4i F0 = ld_frag(frg0 + 4 + l), F1 = ld_frag(frg0 + 4 + 64 + l);
        v4i R0 = {0, 0, 0, 0}, R1 = {0, 0, 0, 0};
        if constexpr (MODE0 == HYB) { R0 = ld_frag(frg0 + 4 + 128 + l); R1 = ld_frag(frg0 + 4 + 192 + l); }
        unsigned *d = reinterpret_cast<unsigned *>(bufI);
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = 2 * w + ii;
            const int row = Y + 3 + i;
            const bool rok = (row >= 0) & (row < H);
#pragma unroll
            for (int gi = 0; gi < 4; ++gi)
                item0(rawW + a0 + i * QRP + 16 * gi, rawW + a1 + i * QRP + 16 * gi, F0, F1, R0, R1, ac, aci,
                      rok & ((colmask >> gi) & 1u), true, d + ((3 + i) * TP + 16 * gi + n) * 4 + g);
        }
        if (w == 3) {
            const int ie = n >> 1, ce = 64 + (n & 1), row = Y + 3 + ie;
            item0(rawW + (g + ie) * QRP + ce, rawW + (tr_r + ie) * QRP + ce + tr_c, F0, F1, R0, R1, ac, aci,
                  (row >= 0) & (row < H) & col_e, true, d + ((3 + ie) * TP + ce) * 4 + g);
        }
    };
    // ---- phases a, b: the trio's inner layers (sesrq_trio.hip)
    auto inner = [&](auto KC, auto I0, const int4 *src, int4 *dst, int row0, const v4i (&Wk)[3]) __attribute__((always_inline)) {
        constexpr int K = decltype(KC)::value, i0 = decltype(I0)::value;
        const TrioLayer &L = a.t.l[K];
        const int4 *p = src + rdcol;
        unsigned *d = reinterpret_cast<unsigned *>(dst) + wrcol;
        const v4i acck = ld_frag(accL + 4 * K + g);
        v4i B0 = ld_frag(p + (i0)*TP), B1 = ld_frag(p + (i0 + 1) * TP);
#pragma unroll
        for (int i = i0; i < TH; ++i) {
            const v4i B2 = ld_frag(p + (i + 2) * TP);
            v4i acc = mfma(Wk[0], B0, acck);
            acc = mfma(Wk[1], B1, acc);
            acc = mfma(Wk[2], B2, acc);
            B0 = B1; B1 = B2;
            const int s[4] = {acc[0], acc[1], acc[2], acc[3]};
            unsigned q = epi_mid<true>(s, L, L.zlo);
            const int row = row0 + i;
            const bool rok = (row >= 0) & (row < H);
            q = (rok & col_in) ? q : (unsigned)L.pad_next;
            d[(2 + i) * TP * 4] = q;
        }
    };
    // ---- phase c: output rows Y .. Y+7; rc[r] = this lane's residual operand word of row Y+r, read from IN before it shifts
    auto outer = [&](int Y, const unsigned rc012[3], const v4i (&Wk)[3]) __attribute__((always_inline)) {
        const int4 *p = bufB + rdcol;
        io.voff = col_out ? voff_c + Y * io.row_bytes : OOB;
        io.voffw = col_out ? gx * 16 + 4 * g + Y * io.row_bytes : OOB;
        const v4i acck = ld_frag(accL + 8 + g);
        // rows Y+3 .. Y+7 of the layer-0 output are still at IN positions 3 .. 7 (the shift only rewrote positions 0 .. 2)
        unsigned rc[TH];
#pragma unroll
        for (int r = 0; r < TH; ++r) rc[r] = r < 3 ? rc012[r] : reinterpret_cast<const unsigned *>(bufI)[(r * TP + c + 1) * 4 + g];
        v4i B0 = ld_frag(p), B1 = ld_frag(p + TP);
#pragma unroll
        for (int y4 = 0; y4 < TH; y4 += 4) {
            int s4[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const v4i B2 = ld_frag(p + (y4 + r + 2) * TP);
                v4i acc = mfma(Wk[0], B0, acck);
                acc = mfma(Wk[1], B1, acc);
                acc = mfma(Wk[2], B2, acc);
                B0 = B1; B1 = B2;
#pragma unroll
                for (int i = 0; i < 4; ++i) s4[r][i] = acc[i];
            }
            const unsigned rcw[4] = {rc[y4], rc[y4 + 1], rc[y4 + 2], rc[y4 + 3]};
            emit_rows4_preres_rc<true>(s4, rcw, ec, io, y4);
        }
    };
    using std::integral_constant;
    auto shift = [&](int4 *win) __attribute__((always_inline)) {      // rows TH, TH+1 of a 10-row window -> rows 0, 1
        if (tid < 2 * TP) { const int4 t = win[TH * TP + tid]; win[tid] = t; }
    };
    auto shift_in = [&]() __attribute__((always_inline)) {            // IN rows 8, 9, 10 -> 0, 1, 2
        if (tid < 3 * TP) { const int4 t = bufI[TH * TP + tid]; bufI[tid] = t; }
    };

    QuadStage<SRC, NCH> st;
    st.init(a, n_img, x0, tid);
    const int shl0 = (tid / QRC) * QRP + tid % QRC, shl1 = ((tid + 256) / QRC) * QRP + (tid + 256) % QRC;     // RAW shift: 4 rows x 72 dwords
    // one step; COLD: the step before the run's first one (RAW loaded whole, 4 / 2 rows of the inner layers, nothing stored)
    auto step = [&](auto COLDC, int Y, bool more) __attribute__((always_inline)) {
        constexpr bool COLD = decltype(COLDC)::value;
        if constexpr (COLD) {
            st.template load<true>(a, Y + 1);
            st.template store<true>(rawW, a, tid);
            __syncthreads();
        } else {
            shift(bufB);                                   // layer-b rows Y-1, Y (phase c of the previous step is done)
        }
        ldw(WP, 0);                                        // layer a's fragments arrive while phase 0 runs
        phase0(Y);
        if (more) st.template load<false>(a, Y + TH + 5);  // the next step's new frame rows Y+13 .. Y+20 (stored after barrier b)
        __syncthreads();                                   // (a) IN complete for this step
        ldw(WQ, 1);
        unsigned rc012[3];                                 // residual operand words of rows Y, Y+1, Y+2: IN positions 0 .. 2 are
#pragma unroll                                             // rewritten by the shift below, 3 .. 7 are read in phase c
        for (int r = 0; r < 3; ++r) rc012[r] = reinterpret_cast<const unsigned *>(bufI)[(r * TP + c + 1) * 4 + g];
        if constexpr (COLD) inner(integral_constant<int, 0>(), integral_constant<int, 4>(), bufI + TP, bufA, Y + 2, WP);
        else inner(integral_constant<int, 0>(), integral_constant<int, 0>(), bufI + TP, bufA, Y + 2, WP);
        int shR[2] = {0, 0};                               // RAW rows 8 .. 11 -> 0 .. 3 (288 dwords): read before, written after (b)
        shR[0] = rawW[TH * QRP + shl0];
        if (tid < 4 * QRC - 256) shR[1] = rawW[TH * QRP + shl1];
        __syncthreads();                                   // (b) layer a complete, IN free, RAW reads of the shift done
        ldw(WP, 2);
        if (more) {
            rawW[shl0] = shR[0];
            if (tid < 4 * QRC - 256) rawW[shl1] = shR[1];
            st.template store<false>(rawW, a, tid);
        }
        shift_in();
        if constexpr (COLD) inner(integral_constant<int, 1>(), integral_constant<int, 6>(), bufA, bufB, Y + 1, WQ);
        else inner(integral_constant<int, 1>(), integral_constant<int, 0>(), bufA, bufB, Y + 1, WQ);
        __syncthreads();                                   // (c) layer b complete, A free
        shift(bufA);
        if constexpr (!COLD) outer(Y, rc012, WP);
        __syncthreads();                                   // (d)
    };
    step(std::true_type(), (s_begin - 1) * TH, true);
    for (int s = s_begin; s < s_end; ++s) step(std::false_type(), s * TH, s + 1 < s_end);
}

// Launch geometry as sesrq_trio.hip: LDS padded to exactly `occ` workgroups per CU, strips cut into runs of (almost) equal length.
template <typename K>
static void launch_quad_k(K kern, QuadArgs a, hipStream_t st) {
    static const int occ = env_knob("SESRQ_QUAD_OCC", 4, 3, 4);      // tuning knob (workgroups per CU), read once
    const int num_cu = device_cu_count();
    const int lds = std::max(QUAD_LDS_BYTES, (160 * 1024 / occ) & ~1023);
    const int strips = (a.t.W + TV - 1) / TV, steps = (a.t.H + TH - 1) / TH;
    long long k = (a.t.wg_budget > 0 ? (long long)a.t.wg_budget : (long long)occ * num_cu) / ((long long)strips * a.t.N);
    k = std::max(1LL, std::min<long long>(k, steps));
    dim3 grid(strips, (int)k, a.t.N);
    launch_kernel(kern, grid, dim3(256), (unsigned)lds, st, a);
}

int launch_quad(const QuadArgs &a, bool hybrid, int src, hipStream_t st) {
    static_assert(QUAD_LDS_BYTES <= 40960, "four workgroups per CU");
    if ((size_t)a.t.H * a.t.W * 16 >= ((size_t)1 << 28)) { set_error("quad: frame too large for 32-bit buffer offsets (H*W must stay below 2^24 pixels)"); return 1; }
#define SESRQ_QUAD(SRC_, NCH_)                                                     \
    do {                                                                           \
        if (hybrid) launch_quad_k(mfma_quad_kernel<HYB, SRC_, NCH_>, a, st);       \
        else launch_quad_k(mfma_quad_kernel<MERGED, SRC_, NCH_>, a, st);           \
    } while (0)
#define SESRQ_QUAD_NCH(SRC_)                                                       \
    do {                                                                           \
        if (a.ic == 3) SESRQ_QUAD(SRC_, 3);                                        \
        else if (a.ic == 1) SESRQ_QUAD(SRC_, 1);                                   \
        else SESRQ_QUAD(SRC_, 4);                                                  \
    } while (0)
    if (src == SRC_F32) SESRQ_QUAD_NCH(SRC_F32);
    else if (src == SRC_I8D) SESRQ_QUAD_NCH(SRC_I8D);
    else SESRQ_QUAD_NCH(SRC_I8);
#undef SESRQ_QUAD_NCH
#undef SESRQ_QUAD
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("quad launch failed: ") + hipGetErrorString(e)); return 1; }
    return 0;
}

}  // namespace sesrq
