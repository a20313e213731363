// Internal declarations shared by the sesrq translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>
#include <string>
#include <vector>

#include "sesrq.h"

namespace sesrq {

// Measurement hook of sesrq_forward_timed: while start/stop are set (this thread), the next kernel launch of the forward carries
// them as its begin / end events (hipExtLaunchKernelGGL: timestamps of the dispatch itself, what a rocprofv3 kernel trace
// reports) instead of a hipEventRecord pair around it, which would add the dispatch latency of the launch to every interval.
struct KernelEvents { hipEvent_t start = nullptr, stop = nullptr; };
extern thread_local KernelEvents tl_kernel_events;
// Registry of the kernel instantiations the library can select (round 5).  EVERY launch of the library goes through launch_kernel<KERN>:
// instantiating that template for a kernel odr-uses KernelInstance<KERN>::id, whose dynamic initialiser registers the kernel's name when
// the library is loaded -- so the table is complete by construction: a kernel that can be launched is in it, whether or not any test has
// reached it yet.  sesrq_instance_count / _name / _launches (include/sesrq.h) expose the table and a per-process launch counter;
// tests/test_gpu_parity.py:test_every_kernel_instance_runs_on_reference_data fails for an instantiation no case of its matrix launched.
// (Round 4 shipped 9.5 M wrong bytes in an instantiation -- the int8-only last layer -- that 168 green tests never selected.)
int register_instance(const void *host_fn, const char *pretty_function);
void count_launch(int id);
template <auto KERN>
struct KernelInstance {
    static const char *pretty() { return __PRETTY_FUNCTION__; }      // "... [KERN = &sesrq::mfma_h5_kernel<1, 2, 22, 3>]"
    static const int id;
};
template <auto KERN>
const int KernelInstance<KERN>::id = register_instance((const void *)KERN, KernelInstance<KERN>::pretty());

template <auto KERN, typename... A>
inline void launch_kernel(dim3 grid, dim3 block, unsigned lds, hipStream_t st, const A &...a) {
    count_launch(KernelInstance<KERN>::id);
    if (tl_kernel_events.start) hipExtLaunchKernelGGL(KERN, grid, block, lds, st, tl_kernel_events.start, tl_kernel_events.stop, 0, a...);
    else hipLaunchKernelGGL(KERN, grid, block, lds, st, a...);
}

// Compute units of the CURRENT device, cached per device id (a process may hold nets on several devices: net->device).
inline int device_cu_count() {
    static int cus[64] = {0};                       // benign race: every writer stores the same value
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        cus[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return cus[dev];
}
// Integer tuning knob from the environment, read ONCE per process (first use), clamped to [lo, hi]; absent / out of range = dflt.
inline int env_knob(const char *name, int dflt, int lo, int hi) {
    const char *e = getenv(name);
    if (!e) return dflt;
    const int v = atoi(e);
    return (v < lo || v > hi) ? dflt : v;
}

enum Epi { EPI_MID = 0, EPI_PRERES = 1, EPI_LAST = 2 };

// First layer (MFMA_F5 images; kernel: mfma_f5_kernel): a pixel is one dword (byte c = channel c).  K-chunk 0 = the
// 4x4 block of taps (lane group g = kernel row g, dwords = kx 0..3); K-chunk 1 = the 9 taps of kernel row 4 and column 4, covered
// by four translates f5_tr(g) of ONE 4-pixel pattern {(0,0),(1,0),(1,1),(1,2)}: the two lane groups of a 32-lane half are an odd
// number of rows apart (3|0 and 3|2), which the column-major LDS image of mfma_f5_kernel needs for conflict-free dword reads.
__host__ __device__ inline void f5_tr(int g, int &row, int &col) {
    row = (g == 0 || g == 2) ? 3 : (g == 1 ? 0 : 2);
    col = g == 0 ? 0 : (g == 2 ? 2 : 4);
}
__host__ __device__ inline void f5_pt(int i, int &row, int &col) { row = i == 0 ? 0 : 1; col = i < 2 ? 0 : i - 1; }

// 5x5 per-PE chains (MFMA_H5 general image; kernels: mfma_h5_kernel).  A lane's 16 operand bytes per PE and K-chunk are TWO VERTICAL
// PAIRS of pixels.  Round 4: every pair starts on an EVEN row of the tile, so that in the column-major planar LDS image
// ([PE][col][row], column pitch even) a pair is ONE 8-byte-aligned LDS access: measured (tools/lds_unaligned_probe.hip) an aligned
// ds_read_b64 costs 3.45 cycles of the CU's LDS pipe and a ds_read2_b64 (two aligned pairs) 8.1, the ds_read2_b32 that fetched ONE pair of
// arbitrary alignment in round 3 costs 8.1 too, and a ds_read_b64 that is only dword-aligned 64.  An output row r of the tile reads tile
// rows r .. r + 4; the three aligned pairs that cover them start at r - par + s, s in {0, 2, 4}, par = r & 1: an even row uses taps
// ky = (0,1) (2,3) (4,-), an odd row (-,0) (1,2) (3,4).  The kernel therefore gives every wave rows of ONE parity (wave w: parity w & 1,
// two 16-column groups) and the weight image holds one set of A fragments per parity (pack_mfma_frags); the pixel addresses are the same
// for both parities.
// h5_pair(c, g, q): K-chunk c, lane group g, pair q covers column `col`, tile rows r - par + start, + 1.  15 pairs cover the 25 taps (five
// of them half empty), the 16th is padding.  K-chunk 0: the two pairs of a lane are vertically adjacent (4 consecutive rows of one
// column: ONE 16-byte read of 8-byte alignment = ds_read2_b64 straight into the operand registers); K-chunk 1: two separate pairs.
// Banks (8-byte accesses: 64 banks, two lane groups of 16 lanes per LDS cycle): with a column pitch of 4 mod 8 dwords the 16 lanes of a
// group sit 4 banks apart, each takes two, and the other lane group of the same 32-lane half fills the gaps iff its pair starts 2 rows
// (mod 4) away: true for K-chunk 0 (odd lane groups start 2 rows lower) and for the first pair of K-chunk 1; the second is two-way.
__host__ __device__ inline void h5_pair(int c, int g, int q, int &col, int &start) {
    //                         g0      g1      g2      g3
    const int tab[4][4][2] = {{{0, 0}, {1, 2}, {2, 0}, {3, 2}},      // chunk 0, pair 0
                              {{0, 2}, {1, 4}, {2, 2}, {3, 4}},      // chunk 0, pair 1 = pair 0 two rows down
                              {{0, 4}, {4, 2}, {2, 4}, {4, 2}},      // chunk 1, pair 0  (g3: padding, reads a pair that carries no weight)
                              {{1, 0}, {4, 4}, {3, 0}, {4, 0}}};     // chunk 1, pair 1
    col = tab[c * 2 + q][g][0];
    start = tab[c * 2 + q][g][1];
}
// tap of element e (0, 1) of that pair for output rows of parity par, or false if the element is padding
__host__ __device__ inline bool h5_tap(int c, int g, int q, int e, int par, int &ky, int &kx) {
    int col, start;
    h5_pair(c, g, q, col, start);
    ky = start + e - par; kx = col;
    return !(c == 1 && q == 0 && g == 3) && ky >= 0 && ky <= 4;
}

// Last layer on the MFMA engine (mfma_h5_kernel<.., EPI_LAST>): which output channel sits in accumulator row 4g + i, i.e. in
// register i of lane group g.  With 16 rows in channel order a 12-channel layer (SESR-x2: 3 colours x PixelShuffle(2)) leaves
// lane group 3 -- a quarter of every clamp / requant / store instruction -- working on padding.  nv = 3 puts the padding into
// REGISTER 3 of every lane instead (rows 4g + 3 carry no weights and are never looked at): all 64 lanes own three real
// channels.  For PixelShuffle(2) x 12 channels the three are a 2-byte run (channels 2g, 2g+1 = two horizontally adjacent
// sub-pixels) and one single (channel 8 + g): one 2-byte and one 1-byte store per lane and row.
// Single source of truth for pack_mfma_frags (host) and LastStore (device).
__host__ __device__ inline int last_nv(int oc) { return oc <= 12 ? 3 : 4; }
__host__ __device__ inline bool last_pairmap(int oc, int ps) { return oc == 12 && ps == 2; }
__host__ __device__ inline int last_slot_oc(int nv, int g, int i, int oc, int ps) {
    if (nv == 4) return 4 * g + i;
    if (i == 3) return 255;                                 // padding row
    if (last_pairmap(oc, ps)) return i < 2 ? 2 * g + i : 8 + g;
    return 3 * g + i;
}
enum Src { SRC_NHWC16 = 0, SRC_F32 = 1, SRC_I8 = 2, SRC_I8D = 3 };   // I8D: int8 frame in an upstream net's output domain
enum MfmaKind { MFMA_NONE = 0, MFMA_H3 = 1, MFMA_H5 = 2, MFMA_F5 = 3, MFMA_H5P = 4 };

// Verified fast division of the input quantiser (sesrq_verify.hip): q0(x) with x pre-clamped to
// [xlo, xhi] and x/s formed as fma(fma(-s, x*r, x), r, x*r); ok == 1 only after an exhaustive proof.
// ok == 0: the division instruction.  ok == 1: q = xc * r, x/s := fma(fma(-s, q, xc), r2, q) with xc = x clamped to [xlo, xhi]:
//   r2 == r is the form prove_fastdiv proves bit-identical to the division; r2 == 0 with r = fl(1/s) and an unbounded clamp is
//   the plain reciprocal multiply x * fl(1/s) of option exact_div = 2 (fd_reciprocal()).
struct FastDiv {
    int ok;
    float r, r2, xlo, xhi;
};
__host__ __device__ inline bool fd_reciprocal(const FastDiv &fd) { return fd.ok && fd.r2 == 0.f; }
FastDiv prove_fastdiv(float s, int zero);
bool prove_direct_requant(unsigned M, unsigned n);
bool prove_single_requant(unsigned M, unsigned n);
FastDiv reciprocal_form(float s, int zero);

// Frames of separate caller buffers as the images of ONE launch (sesrq_forward_many, round 4): n > 0 = image k of the launch reads its
// frame from in[k] and writes its result to out_q[k] / out_f[k] instead of the k-th slice of one contiguous batch.  Only the first layer
// (frame in) and the last layer (frame out, anchor) look at it; everything between is the library's own contiguous workspace.
constexpr int SESRQ_GROUP_MAX = 8;
struct FrameTable {
    int n;
    const void *in[SESRQ_GROUP_MAX];
    void *out_q[SESRQ_GROUP_MAX];
    float *out_f[SESRQ_GROUP_MAX];
};

// Per-launch arguments of one conv layer.  Lives in the kernarg segment (SGPR loads).
struct ConvArgs {
    const void *in;          // SRC_NHWC16: uint4 per pixel ; SRC_F32/SRC_I8: NCHW planes
    void *out;               // EPI_MID/PRERES: NHWC16 int8 ; EPI_LAST: unused
    const void *rc_in;       // EPI_PRERES: NHWC16 residual operand rc = clamp8(rint(short-128))
    void *rc_out;            // layer 0 only, when zero[1] != -128: separate rc tensor (else NULL)
    void *out_q;             // EPI_LAST: (N, C, H*r, W*r) int8 or NULL
    float *out_f;            // EPI_LAST: same shape fp32 or NULL
    const float *anchor;     // EPI_LAST: fp32 input frame (N,C,H,W) added, nearest-upsampled, to out_f; or NULL
    const int *wpk;          // dot4: packed weights [tap][OCP][4] dwords (see pack_weights)
    const int4 *afrag;       // mfma: [4] add-constant words (row order) + A fragments [F][64] (pack_mfma_frags)
    const int4 *afrag_sp;    // first layer, hybrid, 3 channels: 2:4-sparse images [4 header][others: 64 lanes][risky PE: 64 lanes] (pack_f5_sparse) or NULL
    const int4 *afrag2;      // mfma hybrid mode: afrag = merged image WITHOUT the risky PE, afrag2 = per-PE (general) image; risky_pe selects its chain
    int risky_pe;
    int risky_reg;           // hybrid first layer: the one accumulator register (0..3) that holds every channel that can saturate, or 4 = any
    int *dbg_pe;             // (N,4,OC,H,W) int32 or NULL
    int *dbg_add;            // (N,OC,H,W) int32 or NULL
    signed char *dbg_q0;     // (N,IC,H,W) int8: quantised input of layer 0 or NULL (dot4 kernels only)
    float *dbg_t;            // (N,OC,H,W) fp32: the layer's un-rounded requant output after its activation (layer 0: shortcut_tensor.pt), or NULL (dot4 kernels only)
    signed char *dbg_ic;     // (N,OC,H,W) int8: EPI_PRERES: ic = clamp8(rint(t - 128)) (input.4.spcial.pt), or NULL (dot4 kernels only)
    int *dbg_ovf;            // [2] counters: PE sums above / below the accumulator range before saturation, or NULL (dot4 general kernels only)
    int N, H, W;
    int chunk_tiles;         // mfma engine: vertically adjacent tiles walked by one workgroup
    int run_q, run_rem;      // mfma engine: run y of the strip covers tiles [y * run_q + min(y, run_rem), ... + run_q + (y < run_rem)): the host's division
    unsigned inv_nx;         // ceil(2^32 / gridDim.x) for xcd_block's block -> (strip, run) split without a division, or 0 (grid too large: divide)
    int wg_budget;           // mfma engine: workgroup slots the launch may fill (0 = one round of the chip)
    int ic, oc;              // real channel counts
    int pad_word;            // zc replicated into 4 bytes
    int acc_lo, acc_hi, add_lo, add_hi;
    float Mf, sh;            // (float)M, 2^-n
    int direct;              // this layer's requant into a -128 domain: 1 = as ONE fma (prove_direct_requant: Md, Cd), 2 = as one fma that
                             // also subtracts the 128, plus the add that brings it back (prove_single_requant: Md, Cs), 0 = the two-step form
    float Md, Cd, Cs;        // M * 2^-n, -(1.5 * 2^23) * M * 2^-n, Cd - 128
    float z_next;            // (float) zero of the domain this layer requantises into (+zero add)
    float Mres, shres;       // EPI_PRERES
    float z_merge;           // EPI_PRERES: zero of the last conv's input domain
    float s_in, z_in;        // SRC_F32: f32(scale_0), (float)zero_0
    float s_prev, z_prev;    // SRC_I8D: the upstream net's f32(scale_L), (float)zero_L: x = (q - z_prev) * s_prev, then the input quantiser
    FastDiv fd;              // SRC_F32 / SRC_I8D: how x / s_in is formed (FD_*)
    float s_out, z_out;      // EPI_LAST: f32(scale_L), (float) zero_L
    int relu;
    int ps;                  // EPI_LAST pixel shuffle factor
    int add_const[SESRQ_MAX_CH];
    const float2 *mn_oc;     // dot4 kernels: per-output-channel ((float)M, 2^-n) [oc] of a per-channel layer (sesrq_layer_desc.M_oc), or NULL
    FrameTable ft;           // MFMA first / last layer kernels only (ft.n == 0: one contiguous batch at in / out_q / out_f)
};

// fused hidden trio (sesrq_trio.hip): three consecutive 3x3 16->16 merged layers in one launch
struct TrioLayer {
    const int4 *afrag;       // merged A-fragment image of the layer (same as the per-layer MFMA kernel's)
    float Mf, sh, z_next;
    float zlo;               // lower clamp of the layer's output: relu ? max(z_next, -128) : -128
    float Md, Cd;            // the one-fma requant (ConvArgs::direct); the trio kernel's U8 == 2 instance uses it for layers a and b
    int direct;
    int pad_next;            // pad word (zc bytes) of the NEXT layer's input
};
struct TrioArgs {
    const void *in;          // NHWC16 input of the first layer
    void *out;               // NHWC16 output of the third layer
    const void *rc_in;       // EPI_PRERES: residual operand tensor
    int N, H, W;
    int chunk_steps;         // 8-row steps per run, rounded up
    int run_unit;            // trio: rows per partition unit of the vertical runs, 8 (whole steps) or 4 (half steps)
    int run_q, run_rem;      // run y of a strip covers units [y * run_q + min(y, run_rem), ... + run_q + (y < run_rem)) (the host's division)
    unsigned inv_nx;         // as ConvArgs::inv_nx
    int wg_budget;           // workgroup slots the launch may fill (0 = one round of the chip)
    int allow;               // sesrq_options.reduced_forms: which proven reduced forms the launch may select (bits 1, 2, 4, 8)
    int pad_in;              // pad word of the first layer's input
    float Mres, shres, z_merge;
    const int *merge_lut;    // 128 dwords = 512 bytes: q4 as a function of u = (rc + 128) + (ic + 128), the residual merge's second requant (sesrq_create)
    TrioLayer l[3];
};

struct LayerPlan {
    int k, ic, oc, ocp;
    bool general;            // per-PE accumulators + 18/20-bit clamps needed
    std::string engine;
    int *d_wpk_general = nullptr;   // device
    int *d_wpk_merged = nullptr;    // device
    int mfma_kind = MFMA_NONE;
    int4 *d_afrag_general = nullptr; // device
    int4 *d_afrag_merged = nullptr;  // device
    int4 *d_afrag_pesplit = nullptr; // device: last layer with OC <= 4 (MFMA_H5P image), else NULL
    int4 *d_afrag_sparse = nullptr;  // device: first layer, exactly one risky PE, 3 input channels: sparse hybrid images, else NULL
    int4 *d_afrag_others = nullptr;  // device: exactly one risky PE: merged image with that PE's channels zeroed, else NULL
    float2 *d_mn_oc = nullptr;       // device: per-output-channel ((float)M, 2^-n), per-channel layers only (they run on the dot4 kernels)
    std::string engine_dot4, engine_mfma;
    ConvArgs base;           // constant fields prefilled
    // static saturation analysis (per layer)
    long long worst_pe = 0, worst_sum = 0;
    int risky_mask = 0;      // PEs (bit p) whose 18-bit clamp can fire for some output channel
    int risky_oc = 0;        // output channels (bit o) with such a PE sum
};

void set_error(const std::string &msg);

// load-time proof (sesrq_verify.hip): can the 18-bit PE clamp / 20-bit adder clamp of this layer ever fire?
bool saturation_free(const sesrq_layer_desc &d, int zc, int acc_bits, int add_bits, long long &worst_pe, long long &worst_sum, int &risky_mask,
                     int *risky_oc = nullptr);

// launch planner (sesrq_plan.hip)
struct WsLayout {
    size_t act_bytes;      // one NHWC16 activation tensor
    size_t off_s, off_a, off_b, off_rc, total;
};
WsLayout ws_layout(const sesrq_net *net, int N, int H, int W);
bool groupable(const sesrq_net *net);      // can frames of separate caller buffers be the images of one launch (ConvArgs::ft)?
// ft != NULL: the launch's N = ft->n images are the frames ft->in[k] -> ft->out_q[k] / ft->out_f[k]
int forward_impl(const sesrq_net *net, const void *in, int in_dtype, void *out_q, void *out_f, int N, int H, int W, void *workspace,
                 size_t workspace_bytes, void *stream, const sesrq_taps *taps, hipEvent_t *ev, const FrameTable *ft = nullptr);

// dot4 engine
int launch_dot4(const LayerPlan &lp, bool general, const ConvArgs &a, int src, int epi, hipStream_t st);      // general: per-PE sums + clamps
// mfma engine
int launch_mfma(const LayerPlan &lp, const ConvArgs &a, int src, int epi, bool general, hipStream_t st, bool one_risky_pe = false, bool tap = false);
int launch_trio(const TrioArgs &a, int epi_c, hipStream_t st);
int launch_unpack_nhwc16(const void *nhwc, signed char *nchw, int N, int C, int H, int W, hipStream_t st);

}  // namespace sesrq

struct sesrq_net {
    int L = 0;
    std::vector<sesrq::LayerPlan> layers;
    std::vector<int> zero;
    float scale_in = 0.f, scale_out = 0.f;
    uint32_t M_res = 0, n_res = 0;
    int ps = 1;
    int acc_bits = 18, add_bits = 20;
    int engine = SESRQ_ENGINE_AUTO;     // options are fixed at sesrq_create: the net is immutable afterwards
    int force_general = 0;
    int div_mode = 0;        // sesrq_options.exact_div
    sesrq::FastDiv fd_proof = {0, 0.f, 0.f, 0.f, 0.f};        // what prove_fastdiv found, whatever form the options select
    int anchor_add = 0;
    int fuse_hidden = 1;
    int wg_budget = 0;
    int reduced_forms = -1;             // sesrq_options.reduced_forms, resolved (never -1 after sesrq_create)
    float i8_in_scale = 0.f;            // > 0: int8 input frames are in this (scale, zero) domain of an upstream net
    int i8_in_zero = 0;
    int *d_merge_lut = nullptr;         // device: 512-byte table of the residual merge (see TrioArgs::merge_lut)
    std::vector<int> trio_len;          // trio_len[k] == 3: layers k..k+2 are eligible for the fused hidden trio
    int device = 0;
    bool rc_separate = false;   // zero[1] != -128 -> layer 0 writes its own rc tensor
    sesrq::FastDiv fd = {0, 0.f, 0.f, 0.f, 0.f};
};
