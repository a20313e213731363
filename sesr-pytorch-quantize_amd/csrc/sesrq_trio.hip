// sesrq fused hidden trio: three consecutive 3x3 16->16 layers (the reference's conv 1..3, the last one with
// the long-residual merge, myQL/quan_func.py:244-280) in ONE launch.  The two intermediate activation tensors
// never leave the CU: 2 x 66 MB of HBM traffic per 1080p frame and two synchronised kernel starts less than the
// per-layer kernels (sesrq_mfma.hip), and no global staging / transpose / 16-byte store work for the two inner
// layers (their MFMA results go to LDS as they come out: lane (n, g) owns word g of pixel n).
//
// Geometry: a workgroup (4 waves = 4 groups of 16 columns) owns a strip of 64 COMPUTED columns of which
// TV = 60 are valid outputs (a 3x3 layer eats one column per side; the two outermost computed columns of the
// inner layers are never read by a valid output) and walks down it in steps of TH = 8 rows.  Three LDS
// windows of TH + 2 rows each (IN, A, B; row pitch 66 pixels: columns -1 .. 64) hold
//      IN: input rows  Y+1 .. Y+10        A: layer-a rows Y .. Y+9        B: layer-b rows Y-1 .. Y+8
// while the step produces output rows Y .. Y+7: every layer lags the one before by one row, so all three
// phases have the same access pattern (window position 2+i from positions i .. i+2) and NO row is computed
// twice inside a chunk; between steps the last two rows of each window move to its top (132 pixels each).
// A chunk starts cold with a partial step (4 rows of layer a, 2 rows of layer b, nothing stored).
// Pixels outside the frame are the NEXT layer's pad value zc = max(zero, -128), exactly as the per-layer
// kernels pad (myQL/quan_func.py:351-356): the inner epilogues select the pad word there.
// Only the merged accumulation mode (load-time proof: no 18-/20-bit clamp can fire) is fused; anything else
// runs on the per-layer kernels.
#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <type_traits>

#include "sesrq_mfma_common.h"

namespace sesrq {

// Diagnostic build only (-DSESRQ_STAMPS, make stamps): EVERY wave of every workgroup records s_memtime at the phase boundaries of its
// first 12 full steps into a device array no other code reads (tools/trio_stamps.py) -- where a step's cycles go, barrier waits included.
#ifdef SESRQ_STAMPS
__device__ unsigned long long g_trio_stamps[1024 * 4 * 12 * 8];
#define TSTAMP(k)                                                                                                         \
    if (l == 0 && stamp_step < 12) {                                                                                      \
        const unsigned wg_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;                              \
        if (wg_ < 1024) g_trio_stamps[((wg_ * 4 + w) * 12 + stamp_step) * 8 + (k)] = __builtin_amdgcn_s_memtime();          \
    }
#else
#define TSTAMP(k)
#endif

constexpr int TV = 60;            // valid output columns per strip
constexpr int TH = 8;             // rows per step
constexpr int TP = 66;            // LDS row pitch (pixels): computed columns -1 .. 64
constexpr int TR = TH + 2;        // rows per LDS window
constexpr int OOB = (int)0x80000000;
constexpr int TRIO_WIN = TR * TP + 2;      // pixels per LDS window (+ 2: lane group 3 over-reads one pixel)
constexpr int TRIO_LUT_I4 = 32;               // 512-byte table of the residual merge behind the three windows
constexpr int TRIO_LDS_BYTES = 3 * TRIO_WIN * 16;      // dynamic part (the windows); the table is static LDS, TRIO_LUT_I4 * 16 bytes more

struct TrioStage {
    static constexpr int NIT = 3;         // 660 pixels (cold: 10 rows) or 528 (steady: 8 rows) over 256 threads
    v4u v[NIT];
    bool ok[NIT];
    int voff[NIT], ty[NIT], tx[NIT];
    int voff2s;                           // steady form of iteration 2: rows 8, 9 are not loaded
    __amdgpu_buffer_rsrc_t rs;
    int row_bytes;
    __device__ __forceinline__ void init(const TrioArgs &a, int n_img, int x0, int tid) {
        const size_t img = (size_t)a.H * a.W * 16;
        rs = __builtin_amdgcn_make_buffer_rsrc((char *)const_cast<void *>(a.in) + (size_t)n_img * img, 0, (int)img, 0x00020000);
        row_bytes = a.W * 16;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            ty[it] = i / TP;
            tx[it] = i - ty[it] * TP;
            const int gx = x0 - 1 + tx[it];
            const bool okx = (gx >= 0) & (gx < a.W) & (i < TR * TP);
            voff[it] = okx ? (ty[it] * a.W + gx) * 16 : OOB;
            if (!okx) ty[it] = -(1 << 20);          // never a valid row -> pad
        }
        voff2s = (ty[2] < TH) ? voff[2] : OOB;
    }
    // rows [y, y + nrows) of the frame.  COLD: any y (lane-form offsets, rows above the frame pushed out of range);
    // steady: y > 0, one scalar offset per step (gfx950 range-checks voffset + soffset)
    template <bool COLD>
    __device__ __forceinline__ void load(const TrioArgs &a, int y) {
        constexpr int nrows = COLD ? TR : TH;
        const int lo = -y, hi = min(a.H - y, nrows);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            ok[it] = (ty[it] >= lo) & (ty[it] < hi);
            if constexpr (COLD) v[it] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok[it] ? voff[it] + y * row_bytes : OOB, 0, 0);
            else v[it] = __builtin_amdgcn_raw_buffer_load_b128(rs, it == 2 ? voff2s : voff[it], y * row_bytes, 0);
        }
    }
    template <bool COLD>
    __device__ __forceinline__ void store(int4 *win, int pad_word, int tid) const {
        constexpr int nrows = COLD ? TR : TH, pos0 = COLD ? 0 : 2;
        const unsigned pw = (unsigned)pad_word;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            const v4u pad = {pw, pw, pw, pw};
            const v4u t = ok[it] ? v[it] : pad;
            if (i < nrows * TP) win[pos0 * TP + i] = make_int4((int)t[0], (int)t[1], (int)t[2], (int)t[3]);
        }
    }
};

struct TrioEpiC {           // the field names the shared epilogues read
    float Mf, sh, z_next, Mres, shres, z_merge, Md, Cd;
};

// U8, a bit mask: 1 = every zero point of the three epilogues is -128 (the launch checks): round_pack_u8 (sesrq_mfma_common.h);
// 2 = the requants of layers a and b passed prove_direct_requant: the one-fma form of epi_mid; 4 = so did the third layer's
// (residual merge: its first requant, into the fixed -128 domain of ic); 8 = the residual operand IS the trio's input tensor (the
// 5-conv nets: layer 0's output, zero[1] == -128): it is read out of the input window in LDS, already in compute layout (word g of pixel
// n), instead of being loaded again from global memory and transposed (2 loads + 8 v_permlane*_swap per step, and 33 MB per 1080p
// frame off the L2 / MALL).  Instances: 0, 1, 3, 7, 15
template <int EPI_C, int U8>
__global__ __launch_bounds__(256) void mfma_trio_kernel(const TrioArgs a) {
    extern __shared__ int4 trio_lds[];             // dynamic: the launch pads the size so that exactly `occ` workgroups fit a CU
    // the residual merge's table is the kernel's only STATIC LDS object: its address is the compile-time constant 0, so a table index
    // IS an LDS address (no base to add per lookup); the three windows are the dynamic part behind it
    __shared__ int4 trio_lut[TRIO_LUT_I4];
    int4 *lutp = trio_lut, *bufI = trio_lds, *bufA = bufI + TRIO_WIN, *bufB = bufI + 2 * TRIO_WIN;
    constexpr bool LUT = EPI_C == EPI_PRERES;
    constexpr bool RCW = LUT && (U8 & 8) != 0;
    // MAGIC + 256 + the table's LDS byte address (exact: < 2^24); see epi_preres_lut
    const float lut_magic = MAGIC + 256.f + (float)(unsigned)(size_t)(const __attribute__((address_space(3))) void *)lutp;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, n = l & 15, g = l >> 4;
    kernarg_warm<TrioArgs>();
    const BlockXY bxy = xcd_block(a.inv_nx);
    const int n_img = blockIdx.z;
    const int x0 = bxy.x * TV - 2;              // frame column of computed column 0
    // vertical runs of (almost) equal length: run c of n covers units [c*U/n, (c+1)*U/n) of a.run_unit rows -- whole steps (8), or
    // half steps (4) where the runs are short (the launch decides): such a run is walked in full steps plus, for an odd count,
    // one closing half step.  A 540p frame on a full chip has 1 - 2 steps per run: 8- and 16-row runs became 8- and 12-row
    // runs, 18.8 -> 14.8 us.  Long runs (1080p: 4 - 5 steps) gain nothing from it -- the workgroups that finish early leave
    // their issue slots to the others -- and a half step costs more than half a step, so they stay on whole steps.
    const int u_begin = bxy.y * a.run_q + min(bxy.y, a.run_rem);      // the host divided (launch_trio_k): no division in the prologue
    const int y_begin = a.run_unit * u_begin;
    const int y_end = a.run_unit * (u_begin + a.run_q + (bxy.y < a.run_rem ? 1 : 0));
    if (y_begin >= y_end) return;
    // The frame loads of the cold-start window are the FIRST vector-memory requests of the wave; the table and the A fragments follow and
    // arrive beside them (memory returns loads in order): the prologue used to wait for the table's round trip (global load -> LDS write)
    // before it had even asked for its first pixel.
    TrioStage st;
    st.init(a, n_img, x0, tid);
    st.load<true>(a, y_begin - TH + 1);
    int lut_word = 0;
    if constexpr (LUT) {
        if (threadIdx.x < 128) lut_word = a.merge_lut[threadIdx.x];
    }

    const int c = 16 * w + n, gx = x0 + c;
    const bool col_in = (gx >= 0) & (gx < a.W);                   // inner layers: inside the frame, else pad
    const bool col_out = (c >= 2) & (c < 2 + TV) & (gx < a.W);    // valid output columns of the strip
    const bool strip_in = (x0 >= 0) & (x0 + 63 < a.W);            // wave-uniform: every computed column of the strip is inside the frame
    const int rdcol = c + g;                                       // window pixel of tap kx = g (window column = computed column + 1)
    const int wrcol = (c + 1) * 4 + g;                             // window dword of this lane's output word

    v4i A[3][3], acc0[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int4 ac = a.l[k].afrag[g];
        acc0[k] = (v4i){ac.x + MAGIC_I, ac.y + MAGIC_I, ac.z + MAGIC_I, ac.w + MAGIC_I};     // add constant + cvt-free requant bias
#pragma unroll
        for (int f = 0; f < 3; ++f) A[k][f] = ld_frag(a.l[k].afrag + 4 + f * 64 + l);
    }
    const size_t img = (size_t)a.H * a.W * 16;
    RowIO io;
    io.out = __builtin_amdgcn_make_buffer_rsrc((char *)a.out + (size_t)n_img * img, 0, (int)img, 0x00020000);
    io.rc_in = __builtin_amdgcn_make_buffer_rsrc((char *)const_cast<void *>(a.rc_in) + (size_t)n_img * img, 0, (int)img, 0x00020000);
    io.rc_out = io.out;
    io.row_bytes = a.W * 16;
    const int voff_c = col_out ? (g * a.W + gx) * 16 : OOB;       // + Y * row_bytes per step
    const TrioEpiC ec = {a.l[2].Mf, a.l[2].sh, a.l[2].z_next, a.Mres, a.shres, a.z_merge, a.l[2].Md, a.l[2].Cd};

    // inner layer K: window position 2+i <- positions i .. i+2 of the source window; row0 = frame row of i = 0.
    // Written row by row: hipcc keeps ONE accumulator and serialises chain -> epilogue per row, the other three waves of the
    // SIMD fill the gaps.  A hand-pipelined variant (MFMAs of the next 4 rows issued before the epilogues of the previous 4,
    // weights in LDS to stay at 4 waves per SIMD) ran 11 % SLOWER alone (42 vs 38 us at 1080p) and the same with two frames in
    // flight (same-box A/B, round 2): the instruction count is what bounds this kernel, not the order inside one wave.
    // rows i0 .. i1-1 of the step: (0, 8) a full step, (4, 8) / (6, 8) the cold start, (0, 4) the half step that closes a run
    // PADC: std::true_type = the pad word is selected in for pixels outside the frame; std::false_type = the caller knows that every row
    // and column this call produces lies inside the frame (interior strips, steps away from the bottom edge: ~85 % of a 1080p frame):
    // no per-row compare / select (1 VALU + 4 SALU of the ~13 + 8 per row)
    auto inner = [&](auto KC, auto I0, auto I1, const int4 *src, int4 *dst, int row0, auto PADC) __attribute__((always_inline)) {
        constexpr int K = decltype(KC)::value, i0 = decltype(I0)::value, i1 = decltype(I1)::value;
        constexpr bool PAD = decltype(PADC)::value;
        const TrioLayer &L = a.l[K];
        const int4 *p = src + rdcol;
        unsigned *d = reinterpret_cast<unsigned *>(dst) + wrcol;
        v4i B0 = ld_frag(p + (i0)*TP), B1 = ld_frag(p + (i0 + 1) * TP);
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            const v4i B2 = ld_frag(p + (i + 2) * TP);
            v4i acc = mfma(A[K][0], B0, acc0[K]);
            acc = mfma(A[K][1], B1, acc);
            acc = mfma(A[K][2], B2, acc);
            B0 = B1; B1 = B2;
            const int s[4] = {acc[0], acc[1], acc[2], acc[3]};
            unsigned q = epi_mid<true, (U8 & 2) ? 2 : (U8 & 1)>(s, L, L.zlo);
            if constexpr (PAD) {
                const int row = row0 + i;
                const bool rok = (row >= 0) & (row < a.H);
                q = (rok & col_in) ? q : (unsigned)L.pad_next;
            }
            d[(2 + i) * TP * 4] = q;
        }
    };
    // outer layer: output rows Y .. Y+NR-1 (NR = 8, or 4 in a half step) from window positions 0 .. NR+1 of layer b
    // The residual operand of output rows Y .. Y+NR-1 (one 16-byte load per 4 rows).  Issued a whole phase before its use: written
    // next to the epilogue that consumes it, hipcc placed the load in front of the 4-row group and the s_waitcnt vmcnt(0) three
    // MFMAs later (the transposing swaps are scheduled early) -- an L2 / MALL round trip in the open, twice per step.
    auto rc_fetch = [&](auto NRC, int Y, v4u (&rcp)[2]) __attribute__((always_inline)) {
        constexpr int NR = decltype(NRC)::value;
        if constexpr (LUT) {
            const int vo = col_out ? voff_c + Y * io.row_bytes : OOB;
#pragma unroll
            for (int y4 = 0; y4 < NR; y4 += 4) rcp[y4 / 4] = __builtin_amdgcn_raw_buffer_load_b128(io.rc_in, vo, y4 * io.row_bytes, 0);
        }
    };
    auto outer = [&](auto NRC, int Y, const v4u (&rcp)[2]) __attribute__((always_inline)) {
        constexpr int NR = decltype(NRC)::value;
        const int4 *p = bufB + rdcol;
        io.voff = col_out ? voff_c + Y * io.row_bytes : OOB;
        v4i B0 = ld_frag(p), B1 = ld_frag(p + TP);
#pragma unroll
        for (int y4 = 0; y4 < NR; y4 += 4) {
            int s4[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const v4i B2 = ld_frag(p + (y4 + r + 2) * TP);
                v4i acc = mfma(A[2][0], B0, acc0[2]);
                acc = mfma(A[2][1], B1, acc);
                acc = mfma(A[2][2], B2, acc);
                B0 = B1; B1 = B2;
#pragma unroll
                for (int i = 0; i < 4; ++i) s4[r][i] = acc[i];
            }
            if constexpr (LUT) {
                const v4u rv = rcp[y4 / 4];
                unsigned rcw[4] = {rv[0], rv[1], rv[2], rv[3]}, wq[4];
                if constexpr (!RCW) transpose4(rcw);      // RCW: read from the window in compute layout
#pragma unroll
                for (int r = 0; r < 4; ++r) wq[r] = epi_preres_lut<true, (U8 & 4) != 0>(s4[r], rcw[r], ec, lut_magic,
                                                                                      (const unsigned char __attribute__((address_space(3))) *)lutp);
                store_rows4(io.out, io, y4, wq);
            } else {
                emit_rows4<EPI_C, false, true, (U8 & 4) ? 2 : (U8 & 1)>(s4, ec, io, y4, a.l[2].zlo);
            }
        }
    };
    // RCW: the residual operand of frame row (window position pos) of this lane's output pixel, straight from the input window
    auto rc_win = [&](int pos) __attribute__((always_inline)) { return (unsigned)reinterpret_cast<const int *>(bufI)[pos * TP * 4 + wrcol]; };
    // output rows Y .. Y+NR-1 <- the carried word (row Y: the previous window's position 7) and positions 0 .. NR-2; position NR-1 is
    // row Y+NR, the next step's first.  Called while bufI still holds rows Y+1 .. Y+10 (before the step's first barrier).
    unsigned rcc = 0;
    auto rc_from_window = [&](auto NRC, v4u (&rcp)[2]) __attribute__((always_inline)) {
        constexpr int NR = decltype(NRC)::value;
        unsigned t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < NR; ++i) t[i] = rc_win(i);
        rcp[0] = (v4u){rcc, t[0], t[1], t[2]};
        if constexpr (NR == 8) { rcp[1] = (v4u){t[3], t[4], t[5], t[6]}; rcc = t[7]; }
    };
    using std::integral_constant;
    auto shift = [&](int4 *win) __attribute__((always_inline)) {      // rows TH, TH+1 of a window -> rows 0, 1
        if (tid < 2 * TP) { const int4 t = win[TH * TP + tid]; win[tid] = t; }
    };

    using IC0 = integral_constant<int, 0>;
    using IC1 = integral_constant<int, 1>;
    using IC4 = integral_constant<int, 4>;
    using IC6 = integral_constant<int, 6>;
    using IC8 = integral_constant<int, 8>;
    // ---- cold start: the step before the run's first one, only the rows the first real step needs
    {
        const int Y = y_begin - TH;                              // its loads (rows Y + 1 ..) went out at the top of the kernel
        st.store<true>(bufI, a.pad_in, tid);
        if constexpr (LUT) {           // visible to every wave long before the first residual merge (barriers of the cold start)
            if (threadIdx.x < 128) reinterpret_cast<int *>(lutp)[threadIdx.x] = lut_word;
        }
        __syncthreads();
        if constexpr (RCW) rcc = rc_win(TH - 1);                // frame row y_begin = position 7 of the cold-start window (rows y_begin-7 ..)
        st.load<false>(a, Y + TH + 3);
        inner(IC0(), IC4(), IC8(), bufI, bufA, Y + 2, std::true_type());
        int4 shI = make_int4(0, 0, 0, 0);
        if (tid < 2 * TP) shI = bufI[TH * TP + tid];
        __syncthreads();
        if (tid < 2 * TP) bufI[tid] = shI;
        st.store<false>(bufI, a.pad_in, tid);
        inner(IC1(), IC6(), IC8(), bufA, bufB, Y + 1, std::true_type());
        __syncthreads();
        shift(bufA);
        __syncthreads();
    }
    int Y = y_begin;
#ifdef SESRQ_STAMPS
    int stamp_step = 0;
#endif
    // Window shifts (rows TH, TH+1 -> rows 0, 1) are split around a barrier each: the two rows are READ in front of the barrier behind which
    // they may be overwritten and WRITTEN behind it, so the LDS round trip runs while the wave waits for the others.  Stamps
    // (tools/trio_stamps.py): as "read; wait; write" at the top of phases a and c the shifts of bufB / bufA cost ~200 cycles each of a
    // 6000-cycle step (phase a 1870 cycles against phase b's 1370 for the same arithmetic).
    int4 shB = make_int4(0, 0, 0, 0);
    if (tid < 2 * TP) shB = bufB[TH * TP + tid];
    for (; y_end - Y >= TH; Y += TH) {
        TSTAMP(0)
        const bool more = Y + TH < y_end;                        // another step (full or half) follows
        if (more) st.load<false>(a, Y + TH + 3);                 // its new input rows, consumed after the first barrier
        v4u rcp[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        if constexpr (RCW) rc_from_window(IC8(), rcp);           // bufI = rows Y+1 .. Y+10 until the first barrier
        const bool nopad = strip_in && (Y + TH + 2 <= a.H);      // wave-uniform: rows Y+1 .. Y+9, all 64 columns inside
        if (nopad) inner(IC0(), IC0(), IC8(), bufI, bufA, Y + 2, std::false_type());
        else inner(IC0(), IC0(), IC8(), bufI, bufA, Y + 2, std::true_type());
        int4 shI = make_int4(0, 0, 0, 0);
        if (tid < 2 * TP) shI = bufI[TH * TP + tid];
        TSTAMP(1)
        __syncthreads();
        TSTAMP(2)
        if (tid < 2 * TP) bufB[tid] = shB;                       // layer-b rows Y-1, Y: every wave has left phase c of the previous step
        if (more && tid < 2 * TP) bufI[tid] = shI;
        if constexpr (!RCW) rc_fetch(IC8(), Y, rcp);             // in flight during phase b
        TSTAMP(3)
        if (nopad) inner(IC1(), IC0(), IC8(), bufA, bufB, Y + 1, std::false_type());
        else inner(IC1(), IC0(), IC8(), bufA, bufB, Y + 1, std::true_type());
        // The next step's input rows go into the window BEHIND phase b (nobody reads bufI between barrier 1 and the next step): their loads,
        // issued at the top of the step, then have two phases to arrive.  Stamps (tools/trio_stamps.py): written right behind barrier 1 the
        // store took 510 cycles of a 6050-cycle step (p90 912), most of it waiting for the loads.
        if (more) st.store<false>(bufI, a.pad_in, tid);
        int4 shA = make_int4(0, 0, 0, 0);
        if (tid < 2 * TP) shA = bufA[TH * TP + tid];             // layer-a rows Y+8, Y+9 (phase a, visible since barrier 1)
        TSTAMP(4)
        __syncthreads();
        TSTAMP(5)
        if (tid < 2 * TP) bufA[tid] = shA;                       // phase b has read rows 0, 1
        outer(IC8(), Y, rcp);
        if (tid < 2 * TP) shB = bufB[TH * TP + tid];             // layer-b rows Y+7, Y+8 (phase b, visible since barrier 2)
        TSTAMP(6)
        // NO barrier here (round 4: two per step instead of three).  What follows touches nothing phase c still reads: the next phase a reads
        // bufI (complete since barrier 2) and writes bufA rows 2.. (phase b is done with them since barrier 2), the next rc window reads
        // bufI; bufB -- the one buffer phase c reads -- is written again only behind the NEXT barrier 1 (rows 0, 1 from shB, then phase b).
        // A fast wave starts its next phase a (MFMA-heavy) beside the others' phase c (table look-ups, stores).
        TSTAMP(7)
#ifdef SESRQ_STAMPS
        ++stamp_step;
#endif
    }
    if (Y < y_end) {                                             // the closing half step: output rows Y .. Y+3
        v4u rcp[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        if constexpr (RCW) rc_from_window(IC4(), rcp);
        inner(IC0(), IC0(), IC4(), bufI, bufA, Y + 2, std::true_type());
        __syncthreads();
        if (tid < 2 * TP) bufB[tid] = shB;                       // behind the barrier: a slow wave may still have been in the last step's phase c
        if constexpr (!RCW) rc_fetch(IC4(), Y, rcp);
        inner(IC1(), IC0(), IC4(), bufA, bufB, Y + 1, std::true_type());
        __syncthreads();
        outer(IC4(), Y, rcp);
    }
}

// Launch geometry: the chip is filled EVENLY.  A workgroup lives for tens of microseconds here, so a compute unit that holds
// one workgroup more than its neighbours sets the kernel time while the others idle (measured: 864 workgroups on 1024 slots,
// SQ busy 1.47 x the average wave lifetime).  The dynamic LDS size is padded so that exactly `occ` workgroups fit a CU, and a
// strip is cut into floor(occ * CUs / (strips * N)) runs whose lengths differ by at most one step.
template <auto KERN>
static void launch_trio_k(TrioArgs a, hipStream_t st) {
    // 4 workgroups per CU: the LDS size a plain launch accepts and the kernel's registers allow (3 and 5 measured slower, rounds 2-3)
    constexpr int occ = 4;
    const int num_cu = device_cu_count();
    const int lds = std::max(TRIO_LDS_BYTES, ((160 * 1024 / occ) & ~1023) - TRIO_LUT_I4 * 16);      // static + dynamic: exactly occ workgroups per 160 KiB
    const int strips = (a.W + TV - 1) / TV, steps = (a.H + TH - 1) / TH;
    long long k = (a.wg_budget > 0 ? (long long)a.wg_budget : (long long)occ * num_cu) / ((long long)strips * a.N);
    k = std::max(1LL, std::min<long long>(k, steps));                     // a run is at least one full step on average
    a.chunk_steps = (int)((steps + k - 1) / k);
    a.run_unit = (steps < 3 * k) ? TH / 2 : TH;                           // short runs (< 3 steps) are cut in half-step units
    const int units_total = (a.H + a.run_unit - 1) / a.run_unit;
    a.run_q = (int)(units_total / k);
    a.run_rem = (int)(units_total % k);
    a.inv_nx = ((long long)strips * k < 65536 && strips < 65536) ? (unsigned)((0x100000000ULL + (unsigned)strips - 1) / (unsigned)strips) : 0u;
    dim3 grid(strips, (int)k, a.N);
    launch_kernel<KERN>(grid, dim3(256), (unsigned)lds, st, a);
}

#ifdef SESRQ_STAMPS
extern "C" int sesrq_debug_fetch_trio_stamps(void *host, size_t bytes) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trio_stamps), std::min(bytes, sizeof(g_trio_stamps)), 0, hipMemcpyDeviceToHost);
}
#endif

int launch_trio(const TrioArgs &a, int epi_c, hipStream_t st) {
    if ((size_t)a.H * a.W * 16 >= ((size_t)1 << 28)) { set_error("trio: frame too large for 32-bit buffer offsets (H*W must stay below 2^24 pixels)"); return 1; }
    // zero points all -128 (and ReLU's clamp therefore the int8 clamp): the cvt_pk_u8 epilogues
    bool u8 = a.l[0].z_next == -128.f && a.l[1].z_next == -128.f && a.l[0].zlo == -128.f && a.l[1].zlo == -128.f;
    if (epi_c == EPI_PRERES) u8 = u8 && a.z_merge == -128.f;
    else u8 = u8 && a.l[2].z_next == -128.f && a.l[2].zlo == -128.f;
    // sesrq_options.reduced_forms (TrioArgs::allow): 1 = the cvt_pk_u8 epilogues, 2 = one-fma requants of layers a and b, 4 = of the third
    // layer, 8 = the residual operand out of the input window -- each only where its proof / precondition holds
    u8 = u8 && (a.allow & 1);
    const bool ab = u8 && (a.allow & 2) && a.l[0].direct && a.l[1].direct, abc = ab && (a.allow & 4) && a.l[2].direct;      // one-fma requants (proof per layer)
    const int mode = abc ? 7 : (ab ? 3 : (u8 ? 1 : 0));
    if (epi_c == EPI_PRERES) {
        if (mode == 7 && a.rc_in == a.in && (a.allow & 8)) launch_trio_k<mfma_trio_kernel<EPI_PRERES, 15>>(a, st);      // the residual operand out of the input window
        else if (mode == 7) launch_trio_k<mfma_trio_kernel<EPI_PRERES, 7>>(a, st);
        else if (mode == 3) launch_trio_k<mfma_trio_kernel<EPI_PRERES, 3>>(a, st);
        else if (mode == 1) launch_trio_k<mfma_trio_kernel<EPI_PRERES, 1>>(a, st);
        else launch_trio_k<mfma_trio_kernel<EPI_PRERES, 0>>(a, st);
    } else if (epi_c == EPI_MID) {
        if (mode == 7) launch_trio_k<mfma_trio_kernel<EPI_MID, 7>>(a, st);
        else if (mode == 3) launch_trio_k<mfma_trio_kernel<EPI_MID, 3>>(a, st);
        else if (mode == 1) launch_trio_k<mfma_trio_kernel<EPI_MID, 1>>(a, st);
        else launch_trio_k<mfma_trio_kernel<EPI_MID, 0>>(a, st);
    } else { set_error("trio: the third layer must be a hidden layer"); return 1; }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("trio launch failed: ") + hipGetErrorString(e)); return 1; }
    return 0;
}

}  // namespace sesrq
