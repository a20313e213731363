// sesrq C ABI, part 1 of 4: bundle validation + weight repacking (sesrq_create / sesrq_destroy), the error channel.
// The other parts: sesrq_plan.hip (workspace layout + launch planner = the forwards), sesrq_submit.hip (sesrq_forward_many and its
// submission threads), sesrq_scalar.hip (host scalar code of the path); load-time proofs: sesrq_verify.hip; the kernel-instance
// registry: sesrq_registry.hip.  See include/sesrq.h for the contract.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <new>

#include "sesrq_common.h"

namespace sesrq {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }

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
    o->reduced_forms = -1;
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
        if ((l.M_oc != nullptr) != (l.n_oc != nullptr)) { set_error("sesrq_create: per-channel requant constants need both M_oc and n_oc"); return 1; }
        for (int o = 0; l.M_oc && o < l.oc && o < SESRQ_MAX_CH; ++o)
            if (l.M_oc[o] >= (1u << 16) || l.n_oc[o] > 32) { set_error("sesrq_create: per-channel requant constant out of range (M < 2^16, n <= 32)"); return 1; }
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
    {   // which proven reduced forms the kernels may select: all by default; SESRQ_DIRECT=0 leaves the cvt_pk_u8 epilogues only
        static const int knob = env_knob("SESRQ_DIRECT", 1, 0, 1);
        if (o.reduced_forms < -1 || o.reduced_forms > 63) { set_error("sesrq_create: reduced_forms must be -1 or a mask of bits 1 | 2 | 4 | 8 | 16 | 32"); delete net; return 1; }
        net->reduced_forms = o.reduced_forms >= 0 ? o.reduced_forms : (knob ? 63 : (1 | 8));
    }
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
        if (l.M_oc) {      // per-output-channel requant constants: a device table for the dot4 kernels; no MFMA kernel, no trio, no grouping
            float2 mn[SESRQ_MAX_CH];
            for (int o = 0; o < SESRQ_MAX_CH; ++o) mn[o] = o < l.oc ? make_float2((float)l.M_oc[o], ldexpf(1.0f, -(int)l.n_oc[o])) : make_float2(0.f, 0.f);
            if (hipMalloc((void **)&lp.d_mn_oc, sizeof(mn)) != hipSuccess || hipMemcpy(lp.d_mn_oc, mn, sizeof(mn), hipMemcpyHostToDevice) != hipSuccess) {
                set_error("sesrq_create: device upload failed");
                sesrq_destroy(net);
                return 1;
            }
        }
        else if (k == 0) { if (l.k == 5 && l.ic <= 4) lp.mfma_kind = MFMA_F5; }
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
            const int zt = (k == L - 1) ? d->zero[L] : d->zero[(k == 0) ? 1 : k + 1];
            // the residual-merging layer L-2: its FIRST requant, into the fixed -128 domain of ic (quan_func.py:250), whatever the zero points
            // (the output layer has a second choice, form 2 -- LastStore, FASTD 2x: one fma that also subtracts the 128, and the add back)
            // sesrq_options.reduced_forms: bit 2 gates the first layer here, bits 16 / 32 the output layer's two forms; the hidden layers keep
            // their proof -- only the fused trio uses it, and applies bits 2 / 4 at launch (launch_trio)
            const int rf = net->reduced_forms;
            a.direct = 0;
            if (k == L - 2 || zt == -128) {
                if (k == L - 1) {
                    if ((rf & 16) && prove_direct_requant(l.M, l.n)) a.direct = 1;
                    else if ((rf & 32) && prove_single_requant(l.M, l.n)) a.direct = 2;
                } else if (k > 0 || (rf & 2)) {
                    a.direct = sesrq_requant_form(l.M, l.n, 0);
                }
            }
            a.Cs = a.Cd - 128.f;
        }
        if (l.M_oc) a.direct = 0;
        a.mn_oc = lp.d_mn_oc;
        a.Mres = (float)d->M_res; a.shres = ldexpf(1.0f, -(int)d->n_res);
        a.z_merge = (float)d->zero[L - 1];
        a.s_in = d->scale_in; a.z_in = (float)d->zero[0];
        a.s_out = d->scale_out; a.z_out = (float)d->zero[L];
        a.ps = d->pixel_shuffle;
        for (int o = 0; o < l.oc; ++o) a.add_const[o] = l.add_const[o];
        lp.engine_dot4 = std::string(lp.general ? "dot4-general" : "dot4-merged") + (l.M_oc ? "-perchannel" : "");
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
        if (lp.d_mn_oc) (void)hipFree(lp.d_mn_oc);
    }
    delete net;
}

}  // extern "C"
