// sesrq C ABI, part 2 of 4: workspace layout and the launch planner -- sesrq_forward / _debug / _timed are one walk over the net's layers
// (forward_impl) that picks, per layer or fused trio, the kernel family and fills its arguments.  See include/sesrq.h.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <new>

#include "sesrq_common.h"

namespace sesrq {

thread_local KernelEvents tl_kernel_events;

WsLayout ws_layout(const sesrq_net *net, int N, int H, int W) {
    WsLayout l;
    l.act_bytes = (((size_t)N * H * W * 16) + 255) & ~(size_t)255;
    l.off_s = 0;
    l.off_a = l.act_bytes;
    l.off_b = 2 * l.act_bytes;
    l.off_rc = 3 * l.act_bytes;
    l.total = (net->rc_separate ? 4 : 3) * l.act_bytes;
    return l;
}


// the fused hidden trio runs in the production forward (no debug taps), on the MFMA kernels, unless the per-PE path is forced
static bool trio_active(const sesrq_net *net, const sesrq_taps *taps) {
    return net->fuse_hidden && net->engine != SESRQ_ENGINE_DOT4 && !net->force_general && !taps;
}

// Can the frames of several caller buffers be the images of one launch (ConvArgs::ft)?  Only the MFMA first- and last-layer kernels read
// the table: the first layer needs its MFMA kernel (the proven division form), the last layer an MFMA shape.
bool groupable(const sesrq_net *net) {
    const LayerPlan &l0 = net->layers[0], &ll = net->layers[net->L - 1];
    return net->engine != SESRQ_ENGINE_DOT4 && l0.mfma_kind != MFMA_NONE && net->fd.ok && ll.mfma_kind != MFMA_NONE;
}

}  // namespace sesrq

using namespace sesrq;

extern "C" {

int sesrq_fast_division_proven(const sesrq_net *net) { return net ? net->fd_proof.ok : 0; }

const char *sesrq_layer_engine(const sesrq_net *net, int k) {
    if (!net || k < 0 || k >= net->L) return "";
    for (int j = std::max(1, k - 2); j <= k; ++j)
        if (trio_active(net, nullptr) && net->trio_len[j] == 3 && k < j + 3) return "mfma-trio-merged";
    return net->layers[k].engine.c_str();
}

int sesrq_layer_one_fma(const sesrq_net *net, int k) {
    if (!net || k < 0 || k >= net->L) return 0;
    // 1 = one fma, 2 = one fma + the add of 128 (output layer only).  First and output layer: sesrq_create has applied
    // sesrq_options.reduced_forms already; a hidden layer's proof is used by the fused trio only, under bit 2 (layers a, b) / bit 4 (third layer)
    int d = net->layers[k].base.direct;
    if (k > 0 && k < net->L - 1) {
        int bit = 2;
        for (int j = std::max(1, k - 2); j <= k; ++j)
            if (net->trio_len[j] == 3 && k == j + 2) bit = 4;
        if (!(net->reduced_forms & bit)) d = 0;
    }
    return d;
}

int sesrq_launch_plan(const sesrq_net *net, int *first, int *count) {
    if (!net) return 0;
    int n = 0;
    for (int k = 0; k < net->L;) {
        const int c = (trio_active(net, nullptr) && net->trio_len[k] == 3) ? 3 : 1;
        if (first) first[n] = k;
        if (count) count[n] = c;
        ++n;
        k += c;
    }
    return n;
}

int sesrq_net_shape(const sesrq_net *net, int *cin, int *cout, int *pixel_shuffle) {
    if (!net) { set_error("sesrq_net_shape: null net"); return 1; }
    if (cin) *cin = net->layers[0].ic;
    if (cout) *cout = net->layers[net->L - 1].oc;
    if (pixel_shuffle) *pixel_shuffle = net->ps;
    return 0;
}

size_t sesrq_workspace_bytes(const sesrq_net *net, int N, int H, int W) {
    if (!net || N < 1 || H < 1 || W < 1) return 0;
    return ws_layout(net, N, H, W).total;
}

}  // extern "C"

namespace sesrq {

// ft != NULL: the launch's N = ft->n images are the frames ft->in[k] -> ft->out_q[k] / ft->out_f[k] (in / out_q / out_f = frame 0's,
// for the null checks and as the "this output exists" flags)
int forward_impl(const sesrq_net *net, const void *in, int in_dtype, void *out_q, void *out_f, int N, int H, int W, void *workspace,
                 size_t workspace_bytes, void *stream, const sesrq_taps *taps, hipEvent_t *ev, const FrameTable *ft) {
    if (!net || !in || !workspace) { set_error("sesrq_forward: null argument"); return 1; }
    if (ft && (taps || !groupable(net) || ft->n != N || N > SESRQ_GROUP_MAX)) { set_error("sesrq_forward: frame table not applicable"); return 1; }
    if (!out_q && !out_f) { set_error("sesrq_forward: both outputs are NULL"); return 1; }
    if (net->anchor_add && in_dtype != SESRQ_F32) { set_error("sesrq_forward: anchor add needs the fp32 input frame"); return 1; }
    if (N < 1 || H < 1 || W < 1) { set_error("sesrq_forward: N, H, W must be positive"); return 1; }
    if ((size_t)N * H * W > (size_t)1 << 31) { set_error("sesrq_forward: frame batch too large (N*H*W > 2^31)"); return 1; }
    if (in_dtype != SESRQ_F32 && in_dtype != SESRQ_I8) { set_error("sesrq_forward: in_dtype must be SESRQ_F32 or SESRQ_I8"); return 1; }
    if ((uintptr_t)workspace & 15) { set_error("sesrq_forward: workspace must be 16-byte aligned"); return 1; }
    const WsLayout wl = ws_layout(net, N, H, W);
    if (workspace_bytes < wl.total) { set_error("sesrq_forward: workspace too small (see sesrq_workspace_bytes)"); return 1; }
    {   // the net's device copy of the bundle lives on net->device: a launch from another current device would read foreign pointers
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev != net->device) {
            set_error("sesrq_forward: the current HIP device (" + std::to_string(dev) + ") is not the device the net was created on (" +
                      std::to_string(net->device) + ")");
            return 1;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    const int L = net->L;
    if (taps && taps->overflow && hipMemsetAsync(taps->overflow, 0, (size_t)L * 2 * sizeof(int), st) != hipSuccess) {
        set_error("sesrq_forward: clearing the overflow counters failed"); return 1;
    }
    // buffers: S = layer-0 output (kept for the residual), A/B ping-pong, RC optional
    void *bufS = ws + wl.off_s, *bufA = ws + wl.off_a, *bufB = ws + wl.off_b;
    void *bufRC = net->rc_separate ? (void *)(ws + wl.off_rc) : bufS;
    const void *cur = in;
    int launch = 0;
    const int NL = taps ? L : sesrq_launch_plan(net, nullptr, nullptr);       // launches this forward may issue (ev[] holds 2 per launch)
    struct ClearKernelEvents { ~ClearKernelEvents() { tl_kernel_events = KernelEvents{}; } } clear_on_any_exit;
    for (int k = 0; k < L; ++launch) {
        const LayerPlan &lp = net->layers[k];
        if (trio_active(net, taps) && net->trio_len[k] == 3) {
            // ---- fused hidden trio: layers k, k+1, k+2 in one launch (sesrq_trio.hip)
            TrioArgs t;
            memset(&t, 0, sizeof(t));
            void *dst = (cur == bufA) ? bufB : bufA;
            t.in = cur; t.out = dst; t.rc_in = bufRC;
            t.merge_lut = net->d_merge_lut;
            t.N = N; t.H = H; t.W = W;
            t.wg_budget = net->wg_budget;
            t.allow = net->reduced_forms;
            t.pad_in = lp.base.pad_word;
            t.Mres = lp.base.Mres; t.shres = lp.base.shres; t.z_merge = lp.base.z_merge;
            for (int j = 0; j < 3; ++j) {
                const LayerPlan &lj = net->layers[k + j];
                t.l[j].afrag = lj.d_afrag_merged;
                t.l[j].Mf = lj.base.Mf; t.l[j].sh = lj.base.sh; t.l[j].z_next = lj.base.z_next; t.l[j].Md = lj.base.Md; t.l[j].Cd = lj.base.Cd; t.l[j].direct = lj.base.direct;
                t.l[j].zlo = lj.base.relu ? fmaxf(lj.base.z_next, -128.f) : -128.f;
                t.l[j].pad_next = net->layers[k + j + 1].base.pad_word;
            }
            if (launch >= NL) { set_error("sesrq_forward: more launches than sesrq_launch_plan reports"); return 1; }
            if (ev) tl_kernel_events = KernelEvents{ev[2 * launch], ev[2 * launch + 1]};     // begin / end events of the next kernel
            if (launch_trio(t, (k + 2 == L - 2) ? EPI_PRERES : EPI_MID, st)) return 1;
            tl_kernel_events = KernelEvents{};
            cur = dst;
            k += 3;
            continue;
        }
        ConvArgs a = lp.base;
        const bool dbg = taps && (taps->pe_out[k] || taps->pe_add[k] || taps->overflow);
        // the quantised input of layer 0 (input.0.pt) is a tap of the dot4 kernel: with it layer 0 runs there
        const bool q0tap = taps && k == 0 && (taps->act[0] || taps->shortcut);      // ... and so is shortcut_tensor.pt (layer 0's un-rounded output)
        const bool ictap = taps && k == L - 2 && taps->ic;                          // input.4.spcial.pt: the merging layer's ic, dot4 kernel too
        // PE taps on the MFMA engine: the per-PE kernels write them themselves (GEN_TAP); the overflow counters, the quantised
        // input tap and the pe-split last layer (OC <= 4) stay with the dot4 kernels
        const bool mfma_ok = net->engine != SESRQ_ENGINE_DOT4 && lp.mfma_kind != MFMA_NONE && (k > 0 || net->fd.ok);
        const bool tap_mfma = dbg && !taps->overflow && !q0tap && !ictap && mfma_ok && !lp.d_afrag_pesplit;
        const bool general = lp.general || net->force_general || dbg;      // per-PE sums + clamps
        a.wpk = general ? lp.d_wpk_general : lp.d_wpk_merged;
        a.N = N; a.H = H; a.W = W;
        a.wg_budget = net->wg_budget;
        a.in = cur;
        int src = (k == 0) ? (in_dtype == SESRQ_F32 ? SRC_F32 : (net->i8_in_scale > 0.f ? SRC_I8D : SRC_I8)) : SRC_NHWC16;
        a.s_prev = net->i8_in_scale; a.z_prev = (float)net->i8_in_zero;
        int epi = (k == L - 1) ? EPI_LAST : (k == L - 2 ? EPI_PRERES : EPI_MID);
        void *dst = nullptr;
        if (k == 0) { dst = bufS; a.rc_out = net->rc_separate ? bufRC : nullptr; }
        else if (k < L - 1) dst = (cur == bufA) ? bufB : bufA;
        a.out = dst;
        a.rc_in = bufRC;
        a.out_q = out_q; a.out_f = (float *)out_f;
        a.anchor = (net->anchor_add && in_dtype == SESRQ_F32) ? (const float *)in : nullptr;
        if (ft && (k == 0 || k == L - 1)) a.ft = *ft;
        if (taps) {
            a.dbg_pe = (int *)taps->pe_out[k];
            a.dbg_add = (int *)taps->pe_add[k];
            a.dbg_ovf = taps->overflow ? (int *)taps->overflow + 2 * k : nullptr;
            if (k == L - 2) a.dbg_ic = (signed char *)taps->ic;
            if (k == 0) { a.dbg_q0 = (signed char *)taps->act[0]; a.dbg_t = (float *)taps->shortcut; }
            else if (taps->act[k] && launch_unpack_nhwc16(cur, (signed char *)taps->act[k], N, lp.ic, H, W, st)) {
                set_error("sesrq_forward: debug unpack launch failed"); return 1;
            }
        }
        if (launch >= NL) { set_error("sesrq_forward: more launches than sesrq_launch_plan reports"); return 1; }
        if (ev) tl_kernel_events = KernelEvents{ev[2 * launch], ev[2 * launch + 1]};     // begin / end events of the next kernel
        const bool use_mfma = mfma_ok && (!dbg || tap_mfma) && !q0tap && !ictap;
        if (use_mfma) {
            a.afrag = general ? lp.d_afrag_general : lp.d_afrag_merged;
            // exactly one PE can saturate (and nothing forces the full per-PE path): merged chain + that PE's chain
            const bool one_pe = lp.general && !net->force_general && !dbg && lp.d_afrag_others && net->acc_bits == 18 && net->add_bits == 20;
            if (one_pe) {
                a.afrag = lp.d_afrag_others; a.afrag2 = lp.d_afrag_general; a.risky_pe = __builtin_ctz(lp.risky_mask); a.afrag_sp = lp.d_afrag_sparse;
                // hidden-layer rows: channel o sits in register o >> 2 of lane group o & 3.  If every channel that can saturate lives in
                // ONE register, the hybrid first layer clamps that register only (risky_reg), else all four (4)
                a.risky_reg = 4;
                for (int i = 0; i < 4; ++i)
                    if (lp.risky_oc && (lp.risky_oc & ~(0xf << (4 * i))) == 0) a.risky_reg = i;
            }
            if (lp.d_afrag_pesplit) a.afrag = lp.d_afrag_pesplit;
            if (launch_mfma(lp, a, src, epi, general, st, one_pe, tap_mfma)) return 1;
        } else if (launch_dot4(lp, general, a, src, epi, st)) return 1;
        tl_kernel_events = KernelEvents{};
        cur = dst;
        ++k;
    }
    return 0;
}

}  // namespace sesrq

extern "C" {

int sesrq_forward_debug(const sesrq_net *net, const void *in, int in_dtype, void *out_q, void *out_f, int N, int H, int W,
                        void *workspace, size_t workspace_bytes, void *stream, const sesrq_taps *taps) {
    return forward_impl(net, in, in_dtype, out_q, out_f, N, H, W, workspace, workspace_bytes, stream, taps, nullptr);
}

int sesrq_forward(const sesrq_net *net, const void *in, int in_dtype, void *out_q, void *out_f, int N, int H, int W,
                  void *workspace, size_t workspace_bytes, void *stream) {
    return forward_impl(net, in, in_dtype, out_q, out_f, N, H, W, workspace, workspace_bytes, stream, nullptr, nullptr);
}

int sesrq_forward_timed(const sesrq_net *net, const void *in, int in_dtype, void *out_q, void *out_f, int N, int H, int W,
                        void *workspace, size_t workspace_bytes, void *stream, int iters, float *launch_ms, float *forward_ms) {
    if (!net || iters < 1 || !launch_ms) { set_error("sesrq_forward_timed: bad argument"); return 1; }
    const int NL = sesrq_launch_plan(net, nullptr, nullptr);
    std::vector<hipEvent_t> ev((size_t)2 * NL * iters, nullptr);
    int rc = 0;
    for (auto &e : ev)
        if (hipEventCreate(&e) != hipSuccess) { set_error("sesrq_forward_timed: hipEventCreate failed"); e = nullptr; rc = 1; break; }
    for (int it = 0; it < iters && !rc; ++it)
        rc = forward_impl(net, in, in_dtype, out_q, out_f, N, H, W, workspace, workspace_bytes, stream, nullptr,
                          ev.data() + (size_t)2 * NL * it);
    if (!rc && hipStreamSynchronize((hipStream_t)stream) != hipSuccess) { set_error("hipStreamSynchronize failed"); rc = 1; }
    if (!rc) {
        for (int k = 0; k < NL; ++k) launch_ms[k] = 0.f;
        double fw = 0;
        for (int it = 0; it < iters; ++it) {
            hipEvent_t *e = ev.data() + (size_t)2 * NL * it;
            for (int k = 0; k < NL; ++k) {
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e[2 * k], e[2 * k + 1]);
                launch_ms[k] += ms / iters;
            }
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e[0], e[2 * NL - 1]);
            fw += ms;
        }
        if (forward_ms) *forward_ms = (float)(fw / iters);
    }
    for (auto &e : ev)
        if (e) (void)hipEventDestroy(e);
    return rc;
}

}  // extern "C"
