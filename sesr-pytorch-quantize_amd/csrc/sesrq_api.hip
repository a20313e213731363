// sesrq C ABI: bundle validation, weight repacking, static saturation proof, workspace
// layout and the per-layer launch sequence.  See include/sesrq.h for the contract.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <new>
#include <thread>

#include "sesrq_common.h"

namespace sesrq {

thread_local KernelEvents tl_kernel_events;

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }

#define HIP_OK(expr)                                                                        \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                  \
            return 1;                                                                       \
        }                                                                                   \
    } while (0)

// Packed weights for the dot4 engine: [tap][ocp][4] dwords.
//  16-channel input (IN_DW = 4): dword p = bytes j=0..3 -> W[oc][p + 4j][tap]
//  first layer (IC <= 4, IN_DW = 1): input dword byte c = channel c
//     general: dword p = W[oc][p][tap] in byte p only (each channel is its own PE)
//     merged : dword 0 = all channels
static void pack_weights(const sesrq_layer_desc &d, int ocp, bool first, std::vector<int> &gen, std::vector<int> &mer) {
    const int k = d.k, taps = k * k;
    gen.assign((size_t)taps * ocp * 4, 0);
    mer.assign((size_t)taps * ocp * 4, 0);
    for (int t = 0; t < taps; ++t)
        for (int o = 0; o < d.oc; ++o)
            for (int c = 0; c < d.ic; ++c) {
                const int w = d.w[((size_t)o * d.ic + c) * taps + t];
                const size_t base = ((size_t)t * ocp + o) * 4;
                if (!first) {
                    const int p = c & 3, j = c >> 2;
                    gen[base + p] |= (w & 0xff) << (8 * j);
                    mer[base + p] |= (w & 0xff) << (8 * j);
                } else {
                    gen[base + c] |= (w & 0xff) << (8 * c);
                    mer[base + 0] |= (w & 0xff) << (8 * c);
                }
            }
}

// Load-time proof that the 18-bit PE clamp and the 20-bit adder clamp can never fire:
// for q in [-128,127] (pad value included) the extreme PE sums are 127*S+ + 128*S- and
// -(128*S+ + 127*S-).  (myQL/quan_func.py:358-370,437 are then identities.)
// risky_oc (optional): bit o = some PE sum of output channel o can leave the accumulator range
static bool saturation_free(const sesrq_layer_desc &d, int zc, int acc_bits, int add_bits, long long &worst_pe,
                            long long &worst_sum, int &risky_mask, int *risky_oc = nullptr) {
    risky_mask = 0;
    if (risky_oc) *risky_oc = 0;
    const int taps = d.k * d.k;
    const long long acc_hi = (1LL << (acc_bits - 1)) - 1, add_hi = (1LL << (add_bits - 1)) - 1;
    worst_pe = worst_sum = 0;
    bool ok = (zc >= -128 && zc <= 127);
    if (!ok) { risky_mask = 15; if (risky_oc) *risky_oc = 0xffff; }
    for (int o = 0; o < d.oc; ++o) {
        long long tot_hi = 0, tot_lo = 0;
        for (int p = 0; p < 4; ++p) {
            long long sp = 0, sn = 0;
            for (int c = p; c < d.ic; c += 4)
                for (int t = 0; t < taps; ++t) {
                    const int w = d.w[((size_t)o * d.ic + c) * taps + t];
                    if (w > 0) sp += w; else sn -= w;
                }
            const long long hi = 127 * sp + 128 * sn, lo = 128 * sp + 127 * sn;
            worst_pe = std::max(worst_pe, std::max(hi, lo));
            if (hi > acc_hi || lo > acc_hi + 1) { ok = false; risky_mask |= 1 << p; if (risky_oc) *risky_oc |= 1 << o; }
            tot_hi += hi; tot_lo += lo;
        }
        worst_sum = std::max(worst_sum, std::max(tot_hi, tot_lo));
        if (tot_hi > add_hi || tot_lo > add_hi + 1) ok = false;
    }
    return ok;
}

// A-operand fragments of the MFMA engine.  Layout: 4 x int4 header = add constants in output-row
// order, then F fragments of 64 lanes x 16 bytes.  Lane (m = lane & 15, g = lane >> 4), byte b of
// fragment f carries W[ocmap(m)][ch][ky][kx] for the (ky, kx, ch) the kernel's B operand puts in
// the same (g, b) slot -- the tables below are the single source of truth for both sides
// (kernels: sesrq_mfma.hip).
// zero_pe >= 0: the channels of that PE carry no weights (hybrid kernels: the chain of the other three PEs)
// lastnv: 0 = hidden / first layer (PE-major channel order); 3 / 4 = last layer with that many real rows per lane group
// (last_slot_oc, sesrq_common.h), ps = its PixelShuffle factor
static void pack_mfma_frags(const sesrq_layer_desc &d, int kind, bool general, int lastnv, int ps, std::vector<int> &out, int zero_pe = -1) {
    const int taps = d.k * d.k;
    int F = 0;
    switch (kind) {
        case MFMA_H3: F = general ? 4 : 3; break;
        case MFMA_H5: F = general ? 16 : 7; break;      // general: two sets of 8, one per row parity (h5_pair, sesrq_common.h)
        case MFMA_F5: F = general ? 8 : 2; break;
        case MFMA_H5P: F = 7; break;
    }
    out.assign((size_t)16 + (size_t)F * 64 * 4, 0);
    // MFMA_H5P (last layer, OC <= 4): accumulator row m = (PE m/4, output channel m%4); a row only carries
    // the weights of its PE's channels, so one chain over the full K yields the four per-PE sums
    auto ocmap = [&](int m) { return kind == MFMA_H5P ? (m & 3) : (lastnv ? last_slot_oc(lastnv, m >> 2, m & 3, d.oc, ps) : (m >> 2) + 4 * (m & 3)); };
    for (int m = 0; m < 16; ++m) out[m] = (ocmap(m) < d.oc && !(kind == MFMA_H5P && m > 3)) ? d.add_const[ocmap(m)] : 0;
    auto chmap16 = [](int b) { return (b >> 2) + 4 * (b & 3); };
    signed char *bytes = reinterpret_cast<signed char *>(out.data() + 16);
    for (int f = 0; f < F; ++f)
        for (int lane = 0; lane < 64; ++lane)
            for (int b = 0; b < 16; ++b) {
                const int m = lane & 15, g = lane >> 4, i = b >> 2, j = b & 3;
                int ky = -1, kx = -1, ch = -1;
                if (kind == MFMA_H3 && !general) { ky = f; kx = g; ch = chmap16(b); if (g > 2) ky = -1; }
                else if (kind == MFMA_H3) { const int p = f; ky = g; kx = i; ch = p + 4 * j; if (g > 2 || i > 2) ky = -1; }
                else if (kind == MFMA_H5 && !general) {
                    // K-chunks 0..4: kernel row f, lane group g = kx 0..3 (the row operands are re-used across rows);
                    // chunk 5: column 4, lane group g = ky 0..3; chunk 6: tap (4,4) in lane group 0
                    ch = chmap16(b);
                    if (f < 5) { ky = f; kx = g; }
                    else if (f == 5) { ky = g; kx = 4; }
                    else if (g == 0) { ky = 4; kx = 4; }
                } else if (kind == MFMA_H5) {
                    // per row parity and PE p two K-chunks of two vertical pixel pairs per lane group: dword i = pair i / 2, element i % 2 (h5_tap)
                    const int par = f >> 3, fi = (f >> 2) & 1, p = f & 3;
                    ch = p + 4 * j;
                    if (!h5_tap(fi, g, i >> 1, i & 1, par, ky, kx)) ky = -1;
                }
                else if (kind == MFMA_H5P) {
                    ch = chmap16(b);
                    if (f < 5) { ky = f; kx = g; }
                    else if (f == 5) { ky = g; kx = 4; }
                    else if (g == 0) { ky = 4; kx = 4; }
                    if ((b >> 2) != (m >> 2)) ky = -1;              // byte group i = PE of the channel
                } else if (kind == MFMA_F5) {
                    // K-chunk 0: lane group g = kernel row g, dwords = kx 0..3.  K-chunk 1: the 9 remaining taps (row 4 and
                    // column 4) are covered by FOUR translates f5_tr(g) of ONE 4-pixel pattern f5_pt(i) (sesrq_common.h), so a single
                    // pair of ds_read2_b32 (same immediate offsets in every lane) fetches every lane group's operand.
                    const int npe = general ? 4 : 1, fi = f / npe, p = f % npe;
                    ch = j;
                    if (fi == 0) { ky = g; kx = i; }
                    else {
                        int tr_r, tr_c, pt_r, pt_c;
                        f5_tr(g, tr_r, tr_c);
                        f5_pt(i, pt_r, pt_c);
                        ky = tr_r + pt_r; kx = tr_c + pt_c;
                        const bool in_l = (ky == 4 && kx <= 4) || (kx == 4 && ky <= 4);   // taps not in K-chunk 0
                        const bool dup = (g == 2 && i == 1);                              // (4,2) belongs to lane group 0
                        if (!in_l || dup) ky = -1;
                    }
                    if (general && ch != p) ky = -1;
                }
                const int oc = ocmap(m);
                int w = 0;
                if (ky >= 0 && ky < d.k && kx >= 0 && kx < d.k && ch >= 0 && ch < d.ic && oc < d.oc && (ch & 3) != zero_pe)
                    w = d.w[((size_t)oc * d.ic + ch) * taps + ky * d.k + kx];
                bytes[((size_t)f * 64 + lane) * 16 + b] = (signed char)w;
            }
}

// Sparse hybrid images of a 3-channel first layer (HYBS, sesrq_mfma_common.h): header = add constants in row order, then the
// "other two channels" image and the risky channel's image, 64 lanes x 16 stored bytes each.  Stored byte 2j + e of lane (m, ga)
// = weight of channel ch_e at the tap that B lane group gb = 2 (ga & 1) + (j >> 2) holds in register r = 4 (ga >> 1) + (j & 3):
// r < 4: tap (ky = gb, kx = r); r >= 4: f5_tr(gb) + f5_pt(r - 4)  (the dense MFMA_F5 scheme, K-chunks 0 and 1 side by side).
static void pack_f5_sparse(const sesrq_layer_desc &d, int risky_pe, std::vector<int> &out) {
    const int taps = d.k * d.k;
    out.assign((size_t)16 + 2 * 64 * 4, 0);
    for (int m = 0; m < 16; ++m) { const int oc = (m >> 2) + 4 * (m & 3); out[m] = oc < d.oc ? d.add_const[oc] : 0; }
    int others[2], no = 0;
    for (int c = 0; c < 3; ++c) if (c != risky_pe) others[no++] = c;
    signed char *bytes = reinterpret_cast<signed char *>(out.data() + 16);
    for (int img = 0; img < 2; ++img)
        for (int lane = 0; lane < 64; ++lane)
            for (int s = 0; s < 16; ++s) {
                const int m = lane & 15, ga = lane >> 4, j = s >> 1, e = s & 1;
                const int gb = 2 * (ga & 1) + (j >> 2), r = 4 * (ga >> 1) + (j & 3);
                int ky, kx;
                if (r < 4) { ky = gb; kx = r; }
                else {
                    int tr_r, tr_c, pt_r, pt_c;
                    f5_tr(gb, tr_r, tr_c);
                    f5_pt(r - 4, pt_r, pt_c);
                    ky = tr_r + pt_r; kx = tr_c + pt_c;
                    const bool in_l = (ky == 4 && kx <= 4) || (kx == 4 && ky <= 4);
                    if (!in_l || (gb == 2 && r - 4 == 1)) ky = -1;          // (4,2) belongs to lane group 0
                }
                const int ch = img == 0 ? others[e] : (e == 0 ? risky_pe : -1);
                const int oc = (m >> 2) + 4 * (m & 3);
                int w = 0;
                if (ky >= 0 && ky < d.k && kx >= 0 && kx < d.k && ch >= 0 && ch < d.ic && oc < d.oc)
                    w = d.w[((size_t)oc * d.ic + ch) * taps + ky * d.k + kx];
                bytes[((size_t)img * 64 + lane) * 16 + s] = (signed char)w;
            }
}

static int replicate_byte(int v) {
    const int b = v & 0xff;
    return b | (b << 8) | (b << 16) | (b << 24);
}

struct WsLayout {
    size_t act_bytes;      // one NHWC16 activation tensor
    size_t off_s, off_a, off_b, off_rc, total;
};
static WsLayout ws_layout(const sesrq_net *net, int N, int H, int W) {
    WsLayout l;
    l.act_bytes = (((size_t)N * H * W * 16) + 255) & ~(size_t)255;
    l.off_s = 0;
    l.off_a = l.act_bytes;
    l.off_b = 2 * l.act_bytes;
    l.off_rc = 3 * l.act_bytes;
    l.total = (net->rc_separate ? 4 : 3) * l.act_bytes;
    return l;
}

}  // namespace sesrq

using namespace sesrq;

extern "C" {

const char *sesrq_last_error(void) { return g_err.c_str(); }
int sesrq_version(void) { return SESRQ_VERSION; }

void sesrq_default_options(sesrq_options *o) {
    if (!o) return;
    o->engine = SESRQ_ENGINE_AUTO;
    o->force_general = 0;
    o->exact_div = 0;
    o->anchor_add = 0;
    o->fuse_hidden = 1;
    o->wg_budget = 0;
    o->i8_in_scale = 0.f;
    o->i8_in_zero = 0;
}

int sesrq_create(const sesrq_net_desc *d, const sesrq_options *opts, sesrq_net **out) {
    if (!d || !out) { set_error("sesrq_create: null argument"); return 1; }
    *out = nullptr;
    sesrq_options o;
    sesrq_default_options(&o);
    if (opts) o = *opts;
    if (o.engine < SESRQ_ENGINE_AUTO || o.engine > SESRQ_ENGINE_MFMA) { set_error("sesrq_create: bad engine option"); return 1; }
    const int L = d->n_layers;
    if (L < 3 || L > SESRQ_MAX_LAYERS) { set_error("sesrq_create: n_layers must be in [3,16]"); return 1; }
    if (d->pe_num != 4) { set_error("sesrq_create: only pe_num == 4 is supported (define.py PE)"); return 1; }
    if (d->pe_acc_bits < 9 || d->pe_acc_bits > 31 || d->pe_add_bits < d->pe_acc_bits || d->pe_add_bits > 31) {
        set_error("sesrq_create: pe_acc_bits/pe_add_bits out of range"); return 1;
    }
    if (d->pixel_shuffle < 1 || d->pixel_shuffle > 4) { set_error("sesrq_create: pixel_shuffle must be 1..4"); return 1; }
    if (!d->layers || !d->zero) { set_error("sesrq_create: null layers/zero"); return 1; }
    if (d->M_res >= (1u << 16) || d->n_res > 32) { set_error("sesrq_create: residual requant constant out of range"); return 1; }
    if (!(d->scale_in > 0.f) || !(d->scale_out > 0.f)) { set_error("sesrq_create: scales must be positive"); return 1; }
    for (int k = 0; k <= L; ++k)
        if (d->zero[k] < -32768 || d->zero[k] > 127) { set_error("sesrq_create: zero point out of range [-32768,127]"); return 1; }
    for (int k = 0; k < L; ++k) {
        const sesrq_layer_desc &l = d->layers[k];
        if (l.k != 3 && l.k != 5) { set_error("sesrq_create: kernel size must be 3 or 5"); return 1; }
        if (l.ic < 1 || l.ic > SESRQ_MAX_CH || l.oc < 1 || l.oc > SESRQ_MAX_CH) { set_error("sesrq_create: channels must be 1..16"); return 1; }
        if (!l.w || !l.add_const) { set_error("sesrq_create: null weight/add_const"); return 1; }
        if (l.M >= (1u << 16) || l.n > 32) { set_error("sesrq_create: requant constant out of range (M < 2^16, n <= 32)"); return 1; }
        if (k > 0 && l.ic != d->layers[k - 1].oc) { set_error("sesrq_create: channel mismatch between consecutive layers"); return 1; }
        if (k < L - 1 && k > 0 && l.oc != 16 && l.oc > 16) { set_error("sesrq_create: hidden width > 16"); return 1; }
        for (int o = 0; o < l.oc; ++o)
            if (l.add_const[o] < -(1 << 24) || l.add_const[o] > (1 << 24)) { set_error("sesrq_create: add_const out of range"); return 1; }
    }
    if (d->layers[0].ic > 4) { set_error("sesrq_create: first layer supports 1..4 input channels"); return 1; }
    if (d->layers[0].oc != d->layers[L - 2].oc) { set_error("sesrq_create: residual source/destination width mismatch"); return 1; }
    const int r2 = d->pixel_shuffle * d->pixel_shuffle;
    if (d->layers[L - 1].oc % r2) { set_error("sesrq_create: last layer channels not divisible by pixel_shuffle^2"); return 1; }

    sesrq_net *net = new (std::nothrow) sesrq_net();
    if (!net) { set_error("sesrq_create: out of memory"); return 1; }
    net->L = L;
    net->zero.assign(d->zero, d->zero + L + 1);
    net->scale_in = d->scale_in;
    net->scale_out = d->scale_out;
    net->M_res = d->M_res;
    net->n_res = d->n_res;
    net->ps = d->pixel_shuffle;
    net->acc_bits = d->pe_acc_bits;
    net->add_bits = d->pe_add_bits;
    net->rc_separate = (d->zero[1] != -128);
    net->engine = o.engine;
    net->force_general = o.force_general ? 1 : 0;
    if (o.exact_div < 0 || o.exact_div > 2) { set_error("sesrq_create: exact_div must be 0, 1 or 2"); delete net; return 1; }
    net->div_mode = o.exact_div;
    if (o.fuse_hidden < 0 || o.fuse_hidden > 1) { set_error("sesrq_create: fuse_hidden must be 0 or 1"); delete net; return 1; }
    net->fuse_hidden = o.fuse_hidden;
    if (o.wg_budget < 0) { set_error("sesrq_create: wg_budget must be >= 0"); delete net; return 1; }
    net->wg_budget = o.wg_budget;
    // the int8 hand-off domain is an upstream net's OUTPUT domain (scale_L, zero[L]): an int8-range zero point
    if (!(o.i8_in_scale >= 0.f) || o.i8_in_zero < -128 || o.i8_in_zero > 127) { set_error("sesrq_create: bad int8 input domain (zero point must be in [-128, 127])"); delete net; return 1; }
    net->i8_in_scale = o.i8_in_scale;
    net->i8_in_zero = o.i8_in_zero;
    if (o.anchor_add && d->layers[0].ic * d->pixel_shuffle * d->pixel_shuffle != d->layers[L - 1].oc) {
        set_error("sesrq_create: anchor add needs as many output as input channels"); delete net; return 1;
    }
    net->anchor_add = o.anchor_add ? 1 : 0;
    if (hipGetDevice(&net->device) != hipSuccess) { set_error("sesrq_create: no HIP device"); delete net; return 1; }
    net->layers.resize(L);
    for (int k = 0; k < L; ++k) {
        const sesrq_layer_desc &l = d->layers[k];
        LayerPlan &lp = net->layers[k];
        lp.k = l.k; lp.ic = l.ic; lp.oc = l.oc;
        lp.ocp = (k == L - 1) ? ((l.oc + 3) & ~3) : 16;
        const int zc = std::max(d->zero[k], -128);
        lp.general = !saturation_free(l, zc, d->pe_acc_bits, d->pe_add_bits, lp.worst_pe, lp.worst_sum, lp.risky_mask, &lp.risky_oc);
        std::vector<int> gen, mer;
        pack_weights(l, lp.ocp, k == 0, gen, mer);
        const size_t bytes = gen.size() * sizeof(int);
        if (hipMalloc((void **)&lp.d_wpk_general, bytes) != hipSuccess || hipMalloc((void **)&lp.d_wpk_merged, bytes) != hipSuccess ||
            hipMemcpy(lp.d_wpk_general, gen.data(), bytes, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(lp.d_wpk_merged, mer.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) {
            set_error("sesrq_create: device upload failed");
            sesrq_destroy(net);
            return 1;
        }
        lp.mfma_kind = MFMA_NONE;
        if (k == 0) { if (l.k == 5 && l.ic <= 4) lp.mfma_kind = MFMA_F5; }
        else if (l.k == 3 && k < L - 1) lp.mfma_kind = MFMA_H3;
        else if (l.k == 5) lp.mfma_kind = MFMA_H5;
        if (lp.mfma_kind != MFMA_NONE) {
            const int lastnv = (k == L - 1) ? last_nv(l.oc) : 0;
            for (int gen = 0; gen < 2; ++gen) {
                std::vector<int> fr;
                pack_mfma_frags(l, lp.mfma_kind, gen == 1, lastnv, d->pixel_shuffle, fr);
                int4 **dst = gen ? &lp.d_afrag_general : &lp.d_afrag_merged;
                if (hipMalloc((void **)dst, fr.size() * sizeof(int)) != hipSuccess ||
                    hipMemcpy(*dst, fr.data(), fr.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
                    set_error("sesrq_create: device upload failed");
                    sesrq_destroy(net);
                    return 1;
                }
            }
            if (lp.general && __builtin_popcount(lp.risky_mask) == 1) {      // hybrid kernels: merged chain without the risky PE
                std::vector<int> fr;
                pack_mfma_frags(l, lp.mfma_kind, false, lastnv, d->pixel_shuffle, fr, __builtin_ctz(lp.risky_mask));
                if (hipMalloc((void **)&lp.d_afrag_others, fr.size() * sizeof(int)) != hipSuccess ||
                    hipMemcpy(lp.d_afrag_others, fr.data(), fr.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
                    set_error("sesrq_create: device upload failed");
                    sesrq_destroy(net);
                    return 1;
                }
            }
            if (lp.mfma_kind == MFMA_F5 && lp.d_afrag_others && l.ic == 3 && __builtin_ctz(lp.risky_mask) < 3) {
                std::vector<int> fr;
                pack_f5_sparse(l, __builtin_ctz(lp.risky_mask), fr);
                if (hipMalloc((void **)&lp.d_afrag_sparse, fr.size() * sizeof(int)) != hipSuccess ||
                    hipMemcpy(lp.d_afrag_sparse, fr.data(), fr.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
                    set_error("sesrq_create: device upload failed");
                    sesrq_destroy(net);
                    return 1;
                }
            }
            if (k == L - 1 && lp.mfma_kind == MFMA_H5 && l.oc <= 4) {
                std::vector<int> fr;
                pack_mfma_frags(l, MFMA_H5P, true, 4, d->pixel_shuffle, fr);
                if (hipMalloc((void **)&lp.d_afrag_pesplit, fr.size() * sizeof(int)) != hipSuccess ||
                    hipMemcpy(lp.d_afrag_pesplit, fr.data(), fr.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
                    set_error("sesrq_create: device upload failed");
                    sesrq_destroy(net);
                    return 1;
                }
            }
        }
        ConvArgs &a = lp.base;
        memset(&a, 0, sizeof(a));
        a.ic = l.ic; a.oc = l.oc;
        a.pad_word = replicate_byte(zc);
        a.acc_lo = -(1 << (d->pe_acc_bits - 1)); a.acc_hi = (1 << (d->pe_acc_bits - 1)) - 1;
        a.add_lo = -(1 << (d->pe_add_bits - 1)); a.add_hi = (1 << (d->pe_add_bits - 1)) - 1;
        a.Mf = (float)l.M;
        a.sh = ldexpf(1.0f, -(int)l.n);
        a.relu = l.relu;
        a.z_next = (float)d->zero[(k == 0 || k == L - 2) ? 1 : k + 1];
        a.Md = a.Mf * a.sh; a.Cd = -(12582912.f * a.Mf) * a.sh;
        {   // one-fma requant: the layer requantises into a -128 domain (z_next; the output layer: zero[L]) and (M, n) passes the proof
            static const int knob = env_knob("SESRQ_DIRECT", 1, 0, 1);
            const int zt = (k == L - 1) ? d->zero[L] : d->zero[(k == 0) ? 1 : k + 1];
            // the residual-merging layer L-2: its FIRST requant, into the fixed -128 domain of ic (quan_func.py:250), whatever the zero points
            // (the output layer has a second choice, form 2 -- LastStore, FASTD 2x: one fma that also subtracts the 128, and the add back)
            a.direct = (knob && (k == L - 2 || zt == -128)) ? sesrq_requant_form(l.M, l.n, k == L - 1) : 0;
            a.Cs = a.Cd - 128.f;
        }
        a.Mres = (float)d->M_res; a.shres = ldexpf(1.0f, -(int)d->n_res);
        a.z_merge = (float)d->zero[L - 1];
        a.s_in = d->scale_in; a.z_in = (float)d->zero[0];
        a.s_out = d->scale_out; a.z_out = (float)d->zero[L];
        a.ps = d->pixel_shuffle;
        for (int o = 0; o < l.oc; ++o) a.add_const[o] = l.add_const[o];
        lp.engine_dot4 = lp.general ? "dot4-general" : "dot4-merged";
        static const char *kn[] = {"", "mfma-h3", "mfma-h5", "mfma-f5"};
        const bool hyb = lp.general && __builtin_popcount(lp.risky_mask) == 1 && d->pe_acc_bits == 18 && d->pe_add_bits == 20;
        lp.engine_mfma = lp.mfma_kind == MFMA_NONE ? lp.engine_dot4 : std::string(kn[lp.mfma_kind]) + (hyb ? "-hybrid" : (lp.general ? "-general" : "-merged"));
        if (lp.d_afrag_pesplit) lp.engine_mfma = std::string("mfma-h5p-") + (lp.general ? "general" : "merged");
        lp.engine = (net->engine == SESRQ_ENGINE_DOT4) ? lp.engine_dot4 : lp.engine_mfma;
    }
    // fused hidden trios, greedy from the residual-merging layer L-2 backwards: three consecutive 3x3 16->16 layers whose
    // load-time proof allows the merged accumulation mode
    net->trio_len.assign(L, 0);
    auto trio_ok = [&](int k) {
        const LayerPlan &lp = net->layers[k];
        return k >= 1 && k <= L - 2 && lp.mfma_kind == MFMA_H3 && !lp.general && lp.ic == 16 && lp.oc == 16;
    };
    for (int k = L - 4; k >= 1 && trio_ok(k) && trio_ok(k + 1) && trio_ok(k + 2); k -= 3) net->trio_len[k] = 3;
    {   // Residual merge (myQL/quan_func.py:256-270): q4 = clamp8(rint(fl(fl(u * M_res) * 2^-n_res + zero[L-1]))) is a function of the
        // 9-bit integer u = rc + ic + 256 alone: a 511-entry byte table replaces the second requant of the fused trio's last phase
        // (2 fma + add + cvt per value) by one LDS byte read.  Same fp32 operations, same order, as requant4<true> + round_pack.
        unsigned char lut[512];
        const float Mres = (float)d->M_res, shres = ldexpf(1.0f, -(int)d->n_res), zm = (float)d->zero[L - 1];
        for (int u = 0; u < 512; ++u) {
            const float prod = (float)u * Mres;              // one rounding of the exact product, as fma(MAGIC + u, M, -MAGIC * M)
            float v = prod * shres;                          // exact (power of two)
            v = v + zm;                                      // one rounding, as fma(prod, 2^-n, z)
            v = fminf(fmaxf(v, -128.f), 127.f);
            lut[u] = (unsigned char)(signed char)(int)nearbyintf(v);
        }
        if (hipMalloc((void **)&net->d_merge_lut, sizeof(lut)) != hipSuccess ||
            hipMemcpy(net->d_merge_lut, lut, sizeof(lut), hipMemcpyHostToDevice) != hipSuccess) {
            set_error("sesrq_create: device upload failed");
            sesrq_destroy(net);
            return 1;
        }
    }
    net->fd_proof = prove_fastdiv(d->scale_in, d->zero[0]);
    net->fd = net->fd_proof;
    if (net->div_mode == 1) net->fd.ok = 0;
    if (net->div_mode == 2) {
        net->fd = reciprocal_form(d->scale_in, d->zero[0]);
        if (!net->fd.ok) { set_error("sesrq_create: exact_div = 2 needs a finite positive scale_in"); delete net; return 1; }
    }
    net->layers[0].base.fd = net->fd;
    if (!net->fd.ok) {      // no 3-instruction form for this (scale, zero), or exact_div = 1: layer 0 divides, on the dot4 kernel
        net->layers[0].engine = net->layers[0].engine_dot4;
    }
    *out = net;
    return 0;
}

void sesrq_destroy(sesrq_net *net) {
    if (!net) return;
    if (net->d_merge_lut) (void)hipFree(net->d_merge_lut);
    for (auto &lp : net->layers) {
        if (lp.d_wpk_general) (void)hipFree(lp.d_wpk_general);
        if (lp.d_wpk_merged) (void)hipFree(lp.d_wpk_merged);
        if (lp.d_afrag_general) (void)hipFree(lp.d_afrag_general);
        if (lp.d_afrag_merged) (void)hipFree(lp.d_afrag_merged);
        if (lp.d_afrag_pesplit) (void)hipFree(lp.d_afrag_pesplit);
        if (lp.d_afrag_others) (void)hipFree(lp.d_afrag_others);
        if (lp.d_afrag_sparse) (void)hipFree(lp.d_afrag_sparse);
    }
    delete net;
}

// the fused hidden trio runs in the production forward (no debug taps), on the MFMA kernels, unless the per-PE path is forced
static bool trio_active(const sesrq_net *net, const sesrq_taps *taps) {
    return net->fuse_hidden && net->engine != SESRQ_ENGINE_DOT4 && !net->force_general && !taps;
}

int sesrq_fast_division_proven(const sesrq_net *net) { return net ? net->fd_proof.ok : 0; }

const char *sesrq_layer_engine(const sesrq_net *net, int k) {
    if (!net || k < 0 || k >= net->L) return "";
    for (int j = std::max(1, k - 2); j <= k; ++j)
        if (trio_active(net, nullptr) && net->trio_len[j] == 3 && k < j + 3) return "mfma-trio-merged";
    return net->layers[k].engine.c_str();
}

int sesrq_layer_one_fma(const sesrq_net *net, int k) {
    if (!net || k < 0 || k >= net->L) return 0;
    return net->layers[k].base.direct;      // 1 = one fma, 2 = one fma + the add of 128 (output layer only)
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

size_t sesrq_workspace_bytes(const sesrq_net *net, int N, int H, int W) {
    if (!net || N < 1 || H < 1 || W < 1) return 0;
    return ws_layout(net, N, H, W).total;
}

// Can the frames of several caller buffers be the images of one launch (ConvArgs::ft)?  Only the MFMA first- and last-layer kernels read
// the table: the first layer needs its MFMA kernel (the proven division form), the last layer an MFMA shape.
static bool groupable(const sesrq_net *net) {
    const LayerPlan &l0 = net->layers[0], &ll = net->layers[net->L - 1];
    return net->engine != SESRQ_ENGINE_DOT4 && l0.mfma_kind != MFMA_NONE && net->fd.ok && ll.mfma_kind != MFMA_NONE;
}

// ft != NULL: the launch's N = ft->n images are the frames ft->in[k] -> ft->out_q[k] / ft->out_f[k] (in / out_q / out_f = frame 0's,
// for the null checks and as the "this output exists" flags)
static int forward_impl(const sesrq_net *net, const void *in, int in_dtype, void *out_q, void *out_f, int N, int H, int W,
                        void *workspace, size_t workspace_bytes, void *stream, const sesrq_taps *taps, hipEvent_t *ev,
                        const FrameTable *ft = nullptr) {
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
        const bool q0tap = taps && k == 0 && taps->act[0];
        // PE taps on the MFMA engine: the per-PE kernels write them themselves (GEN_TAP); the overflow counters, the quantised
        // input tap and the pe-split last layer (OC <= 4) stay with the dot4 kernels
        const bool mfma_ok = net->engine != SESRQ_ENGINE_DOT4 && lp.mfma_kind != MFMA_NONE && (k > 0 || net->fd.ok);
        const bool tap_mfma = dbg && !taps->overflow && !q0tap && mfma_ok && !lp.d_afrag_pesplit;
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
            if (k == 0) a.dbg_q0 = (signed char *)taps->act[0];
            else if (taps->act[k] && launch_unpack_nhwc16(cur, (signed char *)taps->act[k], N, lp.ic, H, W, st)) {
                set_error("sesrq_forward: debug unpack launch failed"); return 1;
            }
        }
        if (launch >= NL) { set_error("sesrq_forward: more launches than sesrq_launch_plan reports"); return 1; }
        if (ev) tl_kernel_events = KernelEvents{ev[2 * launch], ev[2 * launch + 1]};     // begin / end events of the next kernel
        const bool use_mfma = mfma_ok && (!dbg || tap_mfma) && !q0tap;
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

int sesrq_forward_debug(const sesrq_net *net, const void *in, int in_dtype, void *out_q, void *out_f, int N, int H, int W,
                        void *workspace, size_t workspace_bytes, void *stream, const sesrq_taps *taps) {
    return forward_impl(net, in, in_dtype, out_q, out_f, N, H, W, workspace, workspace_bytes, stream, taps, nullptr);
}

int sesrq_forward(const sesrq_net *net, const void *in, int in_dtype, void *out_q, void *out_f, int N, int H, int W,
                  void *workspace, size_t workspace_bytes, void *stream) {
    return forward_impl(net, in, in_dtype, out_q, out_f, N, H, W, workspace, workspace_bytes, stream, nullptr, nullptr);
}

}  // extern "C"  (the submission pool below is C++)

namespace {
// Submission pool of sesrq_forward_many: one persistent host thread per extra stream.  A HIP kernel launch costs the calling thread
// ~3.5 us (round 4, 540p workloads: 10.8-11.2 us per frame of three launches from one thread, where the device needs ~11 us per frame
// when fed): the launches of DIFFERENT streams are independent, so each stream's frames are enqueued by a thread of its own, in order.
struct SubmitJob {
    const sesrq_net *net; const sesrq_frame_io *frames; int count, first, stride, in_dtype, N, H, W;
    void *ws; size_t ws_bytes; void *stream;
    int group;               // frames of this stream per launch (1 = one sesrq_forward per frame)
    int next;                // the stream's next frame
    int rc = 0, bad = -1; std::string err;
};
// one launch sequence of the job: the next frame of its stream, or the next `group` frames as the images of one launch sequence (same
// kernels, N = g, pointer table).  false = nothing left (or an error: j.rc)
static bool job_step(SubmitJob &j) {
    const int k = j.next;
    if (k >= j.count || j.rc) return false;
    int g = 1;
    FrameTable ft;
    ft.n = 0;
    if (j.group > 1) {
        const sesrq_frame_io &f0 = j.frames[k];
        for (g = 0; g < j.group && k + g * j.stride < j.count; ++g) {
            const sesrq_frame_io &f = j.frames[k + g * j.stride];
            if (!f.in || (f.out_q != nullptr) != (f0.out_q != nullptr) || (f.out_f != nullptr) != (f0.out_f != nullptr)) break;
            ft.in[g] = f.in; ft.out_q[g] = f.out_q; ft.out_f[g] = (float *)f.out_f;
        }
        if (g < 1) g = 1;
        ft.n = g > 1 ? g : 0;
    }
    const sesrq_frame_io &f = j.frames[k];
    if (forward_impl(j.net, f.in, j.in_dtype, f.out_q, f.out_f, ft.n ? g : j.N, j.H, j.W, j.ws, j.ws_bytes, j.stream, nullptr, nullptr,
                     ft.n ? &ft : nullptr)) {
        j.rc = 1; j.bad = k; j.err = sesrq_last_error();
        return false;
    }
    j.next = k + g * j.stride;
    return true;
}
static int run_job(SubmitJob &j) {
    while (job_step(j)) {}
    return j.rc;
}
class SubmitPool {
    // A worker spins on its job slot for SPIN_US after finishing a job (the bench hands over a block of frames every ~1 ms: a condition
    // variable's wake-up would add 20-50 us to every block), then sleeps on the condition variable until the next call.
    static constexpr int SPIN_US = 2000;
    struct Worker {
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        std::atomic<SubmitJob *> job{nullptr};
        std::atomic<bool> done{true}, asleep{false};
        bool quit = false;
    };
    std::vector<std::unique_ptr<Worker>> workers;
    std::mutex call_mu;      // one sesrq_forward_many at a time uses the pool (a second concurrent caller enqueues on its own thread)
    static void loop(Worker *w) {
        int dev = -1;
        for (;;) {
            SubmitJob *j = nullptr;
            const auto t0 = std::chrono::steady_clock::now();
            while (!(j = w->job.load(std::memory_order_acquire))) {
                if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(SPIN_US)) {
                    std::unique_lock<std::mutex> lk(w->mu);
                    w->asleep.store(true, std::memory_order_seq_cst);
                    w->cv.wait(lk, [&] { return w->job.load(std::memory_order_seq_cst) || w->quit; });
                    w->asleep.store(false);
                    if (w->quit) return;
                } else {
                    __builtin_ia32_pause();
                }
            }
            if (dev != j->net->device) { dev = j->net->device; (void)hipSetDevice(dev); }      // a fresh thread starts on device 0
            run_job(*j);
            w->job.store(nullptr, std::memory_order_release);
            w->done.store(true, std::memory_order_release);
        }
    }
    void grow(size_t n) {
        while (workers.size() < n) {
            workers.emplace_back(new Worker());
            Worker *w = workers.back().get();
            w->th = std::thread(loop, w);
        }
    }
public:
    ~SubmitPool() {
        for (auto &w : workers) {
            { std::lock_guard<std::mutex> lk(w->mu); w->quit = true; }
            w->cv.notify_all();
            if (w->th.joinable()) w->th.join();
        }
    }
    // the threads exist (and spin) before the first batch that needs them: creating a thread costs more than the batch
    void ensure(size_t n) {
        std::unique_lock<std::mutex> call(call_mu, std::try_to_lock);
        if (call.owns_lock()) grow(n);
    }
    // jobs[0] runs on the caller's thread, jobs[1..] on the workers; false = the pool is busy (caller falls back to one thread)
    bool run(std::vector<SubmitJob> &jobs) {
        std::unique_lock<std::mutex> call(call_mu, std::try_to_lock);
        if (!call.owns_lock()) return false;
        grow(jobs.size() - 1);
        for (size_t i = 1; i < jobs.size(); ++i) {
            Worker *w = workers[i - 1].get();
            w->done.store(false, std::memory_order_relaxed);
            // seq_cst on purpose: "publish the job, then look whether the worker sleeps" against the worker's "say asleep, then look for a
            // job" is Dekker's pattern -- with a release store the load below may pass it (store buffer), both sides read the old value
            // and the worker sleeps on a pending job while this thread spins on `done`
            w->job.store(&jobs[i], std::memory_order_seq_cst);
            if (w->asleep.load(std::memory_order_seq_cst)) { std::lock_guard<std::mutex> lk(w->mu); w->cv.notify_all(); }
        }
        run_job(jobs[0]);
        for (size_t i = 1; i < jobs.size(); ++i) {
            Worker *w = workers[i - 1].get();
            while (!w->done.load(std::memory_order_acquire)) __builtin_ia32_pause();
        }
        return true;
    }
};
static SubmitPool &submit_pool() { static SubmitPool p; return p; }
}  // namespace

extern "C" {

int sesrq_forward_many(const sesrq_net *net, const sesrq_frame_io *frames, int count, int in_dtype, int N, int H, int W,
                       void *const *workspaces, size_t workspace_bytes, void *const *streams, int n_streams) {
    if (!net || !frames || !workspaces || !streams) { set_error("sesrq_forward_many: null argument"); return 1; }
    if (count < 0 || n_streams < 1 || n_streams > 64) { set_error("sesrq_forward_many: count must be >= 0 and n_streams in 1..64"); return 1; }
    for (int s = 0; s < std::min(n_streams, count); ++s)
        if (!workspaces[s]) { set_error("sesrq_forward_many: null workspace"); return 1; }
    // Frames per launch: a workspace that holds G > 1 single-image frames lets up to G consecutive frames of a stream share one launch
    // sequence (pointer table in the kernel arguments, ConvArgs::ft): the launches' fixed cost is paid once per group.
    int group = 1;
    if (N == 1 && groupable(net))
        while (group < SESRQ_GROUP_MAX && ws_layout(net, group + 1, H, W).total <= workspace_bytes) ++group;
    // SESRQ_SUBMIT_THREADS=0: everything from the calling thread; default: one thread per stream when a stream gets at least two
    // launch sequences (fewer: waking a thread costs more than the launches it takes over)
    static const int threads_knob = env_knob("SESRQ_SUBMIT_THREADS", 1, 0, 1);
    if (threads_knob && n_streams > 1) submit_pool().ensure((size_t)n_streams - 1);
    std::vector<SubmitJob> jobs;
    const bool pooled = threads_knob && n_streams > 1 && count >= 2 * n_streams * group;
    const int nj = std::min(n_streams, std::max(count, 1));
    for (int s = 0; s < nj; ++s)
        jobs.push_back(SubmitJob{net, frames, count, s, n_streams, in_dtype, N, H, W, workspaces[s], workspace_bytes, streams[s], group, s});
    if (!(pooled && submit_pool().run(jobs))) {      // one thread: the streams take turns, one launch sequence each
        for (bool any = true; any;) {
            any = false;
            for (auto &j : jobs) any |= job_step(j);
        }
    }
    int bad = -1;
    for (int j = 0; j < nj; ++j)
        if (jobs[j].rc && (bad < 0 || jobs[j].bad < jobs[bad].bad)) bad = j;
    if (bad >= 0) { set_error("sesrq_forward_many: frame " + std::to_string(jobs[bad].bad) + ": " + jobs[bad].err); return 1; }
    return 0;
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

/* ---------------------------------------------------------------- host scalar code */

int sesrq_requant_const(double r, int data_bit, int shift_max, uint32_t *M, uint32_t *n) {
    if (!M || !n) { set_error("sesrq_requant_const: null output"); return 1; }
    if (!(data_bit < shift_max)) { set_error("requan data bit must be less than shift_max"); return 1; }
    if (!(r > 0) || !isfinite(r)) { set_error("sesrq_requant_const: r must be positive and finite"); return 1; }
    int sh;
    const double ip = trunc(r);
    if (ip != 0) {
        // ceil(log2(ip + 1)) == bit length of ip for ip >= 1
        int bits = 0;
        for (double v = ip; v >= 1.0; v = floor(v / 2.0)) ++bits;
        sh = data_bit - bits;
    } else {
        double d = r * 2.0;
        int times = 0;
        while (trunc(d) == 0) { ++times; d *= 2.0; }
        sh = std::min(times + data_bit, shift_max);
    }
    *M = (uint32_t)(long long)trunc(ldexp(r, sh));
    *n = (uint32_t)sh;
    if (sh < 0) { set_error("sesrq_requant_const: multiplier >= 2^data_bit is not representable"); return 1; }
    return 0;
}

int sesrq_requant_form(uint32_t M, uint32_t n, int output_layer) {
    if (prove_direct_requant(M, n)) return 1;
    if (output_layer && prove_single_requant(M, n)) return 2;
    return 0;
}

int sesrq_quantize_weight(const float *w, size_t count, int width, int8_t *wq, double *scale) {
    if (!w || !wq || !scale || count == 0) { set_error("sesrq_quantize_weight: null/empty argument"); return 1; }
    if (width < 2 || width > 8) { set_error("sesrq_quantize_weight: width must be 2..8"); return 1; }
    float mx = w[0], mn = w[0];
    for (size_t i = 1; i < count; ++i) { mx = std::max(mx, w[i]); mn = std::min(mn, w[i]); }
    const double absmax = std::max(fabs((double)mx), fabs((double)mn));
    if (!(absmax > 0)) { set_error("Conv2d weight tensor is all zero"); return 1; }
    const int qmax = (1 << (width - 1)) - 1, qmin = -(1 << (width - 1));
    const double s = (absmax - (0 - absmax)) / (double)(qmax - qmin);
    const float sf = (float)s;
    for (size_t i = 0; i < count; ++i) {
        float q = rintf(w[i] / sf);
        q = std::min(std::max(q, (float)qmin), (float)qmax);
        wq[i] = (int8_t)q;
    }
    *scale = s;
    return 0;
}

int sesrq_add_const(const float *bias, const int8_t *wq, int oc, int per_oc, double s_in, int z_in, double s_w, int bias_width,
                    int32_t *out) {
    if (!bias || !wq || !out || oc < 1 || per_oc < 1) { set_error("sesrq_add_const: bad argument"); return 1; }
    if (bias_width < 2 || bias_width > 24) { set_error("sesrq_add_const: bias_width must be 2..24"); return 1; }
    const float lo = -(float)(1 << (bias_width - 1)), hi = (float)((1 << (bias_width - 1)) - 1);
    const float bs = (float)(s_in * s_w);
    for (int o = 0; o < oc; ++o) {
        float bq = rintf(bias[o] / bs);
        bq = std::min(std::max(bq, lo), hi);
        long long sw = 0;
        for (int i = 0; i < per_oc; ++i) sw += wq[(size_t)o * per_oc + i];
        const float app = (float)sw * (float)z_in;
        float v = bq - app;
        v = std::min(std::max(v, lo), hi);
        out[o] = (int32_t)v;
    }
    return 0;
}

int sesrq_calib_scale_zero(double min_val, double max_val, int width, double *scale, int *zero) {
    if (!scale || !zero) { set_error("sesrq_calib_scale_zero: null output"); return 1; }
    if (!(max_val != min_val)) { set_error("Input tensor is all equal"); return 1; }
    const int qmax = (1 << (width - 1)) - 1, qmin = -(1 << (width - 1));
    const double s = (max_val - min_val) / (double)(qmax - qmin);
    *scale = s;
    *zero = qmin - (int)nearbyint(min_val / s);   // python round(): half-to-even
    return 0;
}

}  // extern "C"
