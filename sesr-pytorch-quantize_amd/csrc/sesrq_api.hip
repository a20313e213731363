// sesrq C ABI: bundle validation, weight repacking, static saturation proof, workspace
// layout and the per-layer launch sequence.  See include/sesrq.h for the contract.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <new>

#include "sesrq_common.h"

namespace sesrq {

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
static bool saturation_free(const sesrq_layer_desc &d, int zc, int acc_bits, int add_bits, long long &worst_pe,
                            long long &worst_sum, int &risky_mask) {
    risky_mask = 0;
    const int taps = d.k * d.k;
    const long long acc_hi = (1LL << (acc_bits - 1)) - 1, add_hi = (1LL << (add_bits - 1)) - 1;
    worst_pe = worst_sum = 0;
    bool ok = (zc >= -128 && zc <= 127);
    if (!ok) risky_mask = 15;
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
            if (hi > acc_hi || lo > acc_hi + 1) { ok = false; risky_mask |= 1 << p; }
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
static void pack_mfma_frags(const sesrq_layer_desc &d, int kind, bool general, bool last, std::vector<int> &out, int zero_pe = -1) {
    const int taps = d.k * d.k;
    int F = 0;
    switch (kind) {
        case MFMA_H3: F = general ? 4 : 3; break;
        case MFMA_H5: F = general ? 8 : 7; break;
        case MFMA_H5L: F = general ? 8 : 10; break;
        case MFMA_F5: F = general ? 8 : 2; break;
        case MFMA_F5L: F = general ? 12 : 3; break;
        case MFMA_H5P: F = 7; break;
    }
    out.assign((size_t)16 + (size_t)F * 64 * 4, 0);
    // MFMA_H5P (last layer, OC <= 4): accumulator row m = (PE m/4, output channel m%4); a row only carries
    // the weights of its PE's channels, so one chain over the full K yields the four per-PE sums
    auto ocmap = [&](int m) { return kind == MFMA_H5P ? (m & 3) : (last ? m : (m >> 2) + 4 * (m & 3)); };
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
                    // per PE p two K-chunks: 0: lane group g = kernel row g, words = kx 0..3;  1: row 4 + column 4 by four
                    // translates of the pixel pattern {(0,0),(1,0),(2,0),(2,2)} (same scheme as MFMA_F5)
                    const int fi = f >> 2, p = f & 3;
                    static const int tr[4][2] = {{0, 4}, {2, 0}, {2, 1}, {2, 4}};
                    static const int pt[4][2] = {{0, 0}, {1, 0}, {2, 0}, {2, 2}};
                    ch = p + 4 * j;
                    if (fi == 0) { ky = g; kx = i; }
                    else {
                        ky = tr[g][0] + pt[i][0]; kx = tr[g][1] + pt[i][1];
                        const bool in_l = (ky == 4 && kx <= 4) || (kx == 4 && ky <= 4);
                        if (!in_l || (g == 3 && i == 0)) ky = -1;
                    }
                }
                else if (kind == MFMA_H5L && !general) { ky = f >> 1; kx = 4 * (f & 1) + g; ch = chmap16(b); if (kx > 4) ky = -1; }
                else if (kind == MFMA_H5L) {
                    const int fi = f >> 2, p = f & 3;
                    ch = p + 4 * j;
                    if (fi == 0) { ky = g; kx = i; }
                    else if (g == 0) { ky = 4; kx = i; }
                    else if (g == 1) { ky = i; kx = 4; }
                    else if (g == 2 && i == 0) { ky = 4; kx = 4; }
                } else if (kind == MFMA_H5P) {
                    ch = chmap16(b);
                    if (f < 5) { ky = f; kx = g; }
                    else if (f == 5) { ky = g; kx = 4; }
                    else if (g == 0) { ky = 4; kx = 4; }
                    if ((b >> 2) != (m >> 2)) ky = -1;              // byte group i = PE of the channel
                } else if (kind == MFMA_F5) {
                    // K-chunk 0: lane group g = kernel row g, dwords = kx 0..3.  K-chunk 1: the 9 remaining taps (row 4 and
                    // column 4) are covered by FOUR translates of ONE 4-pixel pattern {(0,0),(1,0),(2,0),(2,2)}, so a single
                    // pair of ds_read2_b32 (same immediate offsets in every lane) fetches every lane group's operand.
                    const int npe = general ? 4 : 1, fi = f / npe, p = f % npe;
                    static const int tr[4][2] = {{0, 4}, {2, 0}, {2, 1}, {2, 4}};       // (row, column) translation of lane group g
                    static const int pt[4][2] = {{0, 0}, {1, 0}, {2, 0}, {2, 2}};       // the pattern, dword i
                    ch = j;
                    if (fi == 0) { ky = g; kx = i; }
                    else {
                        ky = tr[g][0] + pt[i][0]; kx = tr[g][1] + pt[i][1];
                        const bool in_l = (ky == 4 && kx <= 4) || (kx == 4 && ky <= 4);   // taps not in K-chunk 0
                        const bool dup = (g == 3 && i == 0);                              // (2,4) belongs to lane group 0
                        if (!in_l || dup) ky = -1;
                    }
                    if (general && ch != p) ky = -1;
                } else if (kind == MFMA_F5L) {
                    const int npe = general ? 4 : 1, fi = f / npe, p = f % npe;
                    static const int tky[3][4] = {{0, 1, 2, 3}, {4, 0, 1, 2}, {3, 4, -1, -1}};
                    static const int tsg[3][4] = {{0, 0, 0, 0}, {0, 1, 1, 1}, {1, 1, 0, 0}};
                    ky = tky[fi][g]; kx = 4 * tsg[fi][g] + i; ch = j;
                    if (kx > 4) ky = -1;
                    if (general && ch != p) ky = -1;
                }
                const int oc = ocmap(m);
                int w = 0;
                if (ky >= 0 && ky < d.k && kx >= 0 && kx < d.k && ch >= 0 && ch < d.ic && oc < d.oc && (ch & 3) != zero_pe)
                    w = d.w[((size_t)oc * d.ic + ch) * taps + ky * d.k + kx];
                bytes[((size_t)f * 64 + lane) * 16 + b] = (signed char)w;
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

int sesrq_create(const sesrq_net_desc *d, sesrq_net **out) {
    if (!d || !out) { set_error("sesrq_create: null argument"); return 1; }
    *out = nullptr;
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
    if (hipGetDevice(&net->device) != hipSuccess) { set_error("sesrq_create: no HIP device"); delete net; return 1; }
    net->layers.resize(L);
    for (int k = 0; k < L; ++k) {
        const sesrq_layer_desc &l = d->layers[k];
        LayerPlan &lp = net->layers[k];
        lp.k = l.k; lp.ic = l.ic; lp.oc = l.oc;
        lp.ocp = (k == L - 1) ? ((l.oc + 3) & ~3) : 16;
        const int zc = std::max(d->zero[k], -128);
        lp.general = !saturation_free(l, zc, d->pe_acc_bits, d->pe_add_bits, lp.worst_pe, lp.worst_sum, lp.risky_mask);
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
            for (int gen = 0; gen < 2; ++gen) {
                std::vector<int> fr;
                pack_mfma_frags(l, lp.mfma_kind, gen == 1, k == L - 1, fr);
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
                pack_mfma_frags(l, lp.mfma_kind, false, k == L - 1, fr, __builtin_ctz(lp.risky_mask));
                if (hipMalloc((void **)&lp.d_afrag_others, fr.size() * sizeof(int)) != hipSuccess ||
                    hipMemcpy(lp.d_afrag_others, fr.data(), fr.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
                    set_error("sesrq_create: device upload failed");
                    sesrq_destroy(net);
                    return 1;
                }
            }
            if (lp.mfma_kind == MFMA_F5 || (lp.mfma_kind == MFMA_H5 && k == L - 1)) {
                for (int gen = 0; gen < 2; ++gen) {
                    std::vector<int> fr;
                    pack_mfma_frags(l, lp.mfma_kind == MFMA_F5 ? MFMA_F5L : MFMA_H5L, gen == 1, k == L - 1, fr);
                    int4 **dst = gen ? &lp.d_afrag_f5l_general : &lp.d_afrag_f5l_merged;
                    if (hipMalloc((void **)dst, fr.size() * sizeof(int)) != hipSuccess ||
                        hipMemcpy(*dst, fr.data(), fr.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
                        set_error("sesrq_create: device upload failed");
                        sesrq_destroy(net);
                        return 1;
                    }
                }
            }
            if (k == L - 1 && lp.mfma_kind == MFMA_H5 && l.oc <= 4) {
                std::vector<int> fr;
                pack_mfma_frags(l, MFMA_H5P, true, true, fr);
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
        lp.engine = lp.engine_mfma;
    }
    net->fd = prove_fastdiv(d->scale_in, d->zero[0]);
    net->layers[0].base.fd = net->fd;
    // fused engine: reference topology 5x5 / 3x3 x3 / 5x5, standard 18/20-bit PE model, z1 == -128
    net->fused_ok = (L == 5) && net->layers[0].mfma_kind == MFMA_F5 && net->layers[1].mfma_kind == MFMA_H3 &&
                    net->layers[2].mfma_kind == MFMA_H3 && net->layers[3].mfma_kind == MFMA_H3 &&
                    net->layers[4].mfma_kind == MFMA_H5 && !net->rc_separate && d->pe_acc_bits == 18 && d->pe_add_bits == 20 &&
                    d->layers[0].relu && d->layers[1].relu && d->layers[2].relu && d->layers[3].relu && !d->layers[4].relu;
    *out = net;
    return 0;
}

void sesrq_destroy(sesrq_net *net) {
    if (!net) return;
    for (auto &lp : net->layers) {
        if (lp.d_wpk_general) (void)hipFree(lp.d_wpk_general);
        if (lp.d_wpk_merged) (void)hipFree(lp.d_wpk_merged);
        if (lp.d_afrag_general) (void)hipFree(lp.d_afrag_general);
        if (lp.d_afrag_merged) (void)hipFree(lp.d_afrag_merged);
        if (lp.d_afrag_pesplit) (void)hipFree(lp.d_afrag_pesplit);
        if (lp.d_afrag_others) (void)hipFree(lp.d_afrag_others);
        if (lp.d_afrag_f5l_general) (void)hipFree(lp.d_afrag_f5l_general);
        if (lp.d_afrag_f5l_merged) (void)hipFree(lp.d_afrag_f5l_merged);
    }
    delete net;
}

int sesrq_set_option(sesrq_net *net, int option, int value) {
    if (!net) { set_error("sesrq_set_option: null net"); return 1; }
    switch (option) {
        case SESRQ_OPT_ENGINE:
            if (value < SESRQ_ENGINE_AUTO || value > SESRQ_ENGINE_FUSED) { set_error("sesrq_set_option: bad engine"); return 1; }
            if (value == SESRQ_ENGINE_FUSED && !net->fused_ok) { set_error("sesrq_set_option: this net is not eligible for the fused engine"); return 1; }
            net->engine = value;
            for (auto &lp : net->layers) lp.engine = (value == SESRQ_ENGINE_DOT4) ? lp.engine_dot4 : lp.engine_mfma;
            return 0;
        case SESRQ_OPT_FORCE_GENERAL: net->force_general = value ? 1 : 0; return 0;
        case SESRQ_OPT_EXACT_DIV: net->force_exact_div = value ? 1 : 0; return 0;
        case SESRQ_OPT_ANCHOR_ADD:
            if (value && net->layers[0].ic * net->ps * net->ps != net->layers[net->L - 1].oc) {
                set_error("sesrq_set_option: anchor add needs as many output as input channels"); return 1;
            }
            net->anchor_add = value ? 1 : 0; return 0;
    }
    set_error("sesrq_set_option: unknown option");
    return 1;
}

static bool use_fused(const sesrq_net *net, int in_dtype, const sesrq_taps *taps) {
    // opt-in only: correct, but slower than the per-layer MFMA kernels this round
    return net->fused_ok && net->engine == SESRQ_ENGINE_FUSED && in_dtype == SESRQ_F32 && !taps && !net->anchor_add;
}

int sesrq_fast_division_proven(const sesrq_net *net) { return net ? net->fd.ok : 0; }

const char *sesrq_layer_engine(const sesrq_net *net, int k) {
    if (!net || k < 0 || k >= net->L) return "";
    if (net->fused_ok && net->engine == SESRQ_ENGINE_FUSED) {
        static thread_local std::string s;
        const bool gen = net->layers[k].general || net->force_general;
        s = std::string("fused5-") + (gen ? "general" : "merged");
        return s.c_str();
    }
    return net->layers[k].engine.c_str();
}

size_t sesrq_workspace_bytes(const sesrq_net *net, int N, int H, int W) {
    if (!net || N < 1 || H < 1 || W < 1) return 0;
    return ws_layout(net, N, H, W).total;
}

static int forward_impl(const sesrq_net *net, const void *in, int in_dtype, void *out_q, void *out_f, int N, int H, int W,
                        void *workspace, size_t workspace_bytes, void *stream, const sesrq_taps *taps, hipEvent_t *ev) {
    if (!net || !in || !workspace) { set_error("sesrq_forward: null argument"); return 1; }
    if (!out_q && !out_f) { set_error("sesrq_forward: both outputs are NULL"); return 1; }
    if (net->anchor_add && in_dtype != SESRQ_F32) { set_error("sesrq_forward: anchor add needs the fp32 input frame"); return 1; }
    if (N < 1 || H < 1 || W < 1) { set_error("sesrq_forward: N, H, W must be positive"); return 1; }
    if ((size_t)N * H * W > (size_t)1 << 31) { set_error("sesrq_forward: frame batch too large (N*H*W > 2^31)"); return 1; }
    if (in_dtype != SESRQ_F32 && in_dtype != SESRQ_I8) { set_error("sesrq_forward: in_dtype must be SESRQ_F32 or SESRQ_I8"); return 1; }
    if ((uintptr_t)workspace & 15) { set_error("sesrq_forward: workspace must be 16-byte aligned"); return 1; }
    const WsLayout wl = ws_layout(net, N, H, W);
    if (workspace_bytes < wl.total) { set_error("sesrq_forward: workspace too small (see sesrq_workspace_bytes)"); return 1; }
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    const int L = net->L;
    if (use_fused(net, in_dtype, taps)) {
        FusedArgs f;
        memset(&f, 0, sizeof(f));
        f.in = in; f.out_q = out_q; f.out_f = (float *)out_f;
        f.N = N; f.H = H; f.W = W;
        f.ic = net->layers[0].ic; f.oc = net->layers[4].oc; f.ps = net->ps;
        const int strips = (W + 63) / 64;
        int nchunks = (int)((512 + (long long)strips * N / 2) / ((long long)strips * N));
        nchunks = std::max(1, std::min(nchunks, (H + 15) / 16));
        f.chunk = (H + nchunks - 1) / nchunks;
        const ConvArgs &a0 = net->layers[0].base, &a4 = net->layers[4].base;
        f.pad_in0 = a0.pad_word;
        f.fd = net->force_exact_div ? FastDiv{0, 0.f, 0.f, 0.f} : net->fd;
        f.s_in = a0.s_in; f.z_in = a0.z_in; f.s_out = a4.s_out; f.z_out = a4.z_out;
        f.Mres = a0.Mres; f.shres = a0.shres; f.z_merge = a0.z_merge;
        bool gen[5];
        for (int k = 0; k < 5; ++k) gen[k] = net->layers[k].general || net->force_general;
        const bool genh = gen[1] || gen[2] || gen[3];
        for (int k = 0; k < 5; ++k) {
            const LayerPlan &lp = net->layers[k];
            const bool g = (k == 0) ? gen[0] : (k == 4 ? gen[4] : genh);
            f.l[k].afrag = g ? lp.d_afrag_general : lp.d_afrag_merged;
            if (k == 0 || k == 4) f.l[k].afrag = g ? lp.d_afrag_f5l_general : lp.d_afrag_f5l_merged;
            f.l[k].Mf = lp.base.Mf; f.l[k].sh = lp.base.sh; f.l[k].z_next = lp.base.z_next;
            f.l[k].pad_next = (k < 4) ? net->layers[k + 1].base.pad_word : 0;
        }
        if (ev && hipEventRecord(ev[0], st) != hipSuccess) { set_error("hipEventRecord failed"); return 1; }
        if (launch_fused5(f, gen[0], genh, gen[4], st)) return 1;
        if (ev) {
            for (int k = 1; k < 2 * L; ++k)
                if (hipEventRecord(ev[k], st) != hipSuccess) { set_error("hipEventRecord failed"); return 1; }
        }
        return 0;
    }
    // buffers: S = layer-0 output (kept for the residual), A/B ping-pong, RC optional
    void *bufS = ws + wl.off_s, *bufA = ws + wl.off_a, *bufB = ws + wl.off_b;
    void *bufRC = net->rc_separate ? (void *)(ws + wl.off_rc) : bufS;
    const void *cur = in;
    for (int k = 0; k < L; ++k) {
        const LayerPlan &lp = net->layers[k];
        ConvArgs a = lp.base;
        const bool dbg = taps && (taps->pe_out[k] || taps->pe_add[k]);
        LayerPlan eff = lp;
        eff.general = lp.general || net->force_general || dbg;
        a.wpk = eff.general ? lp.d_wpk_general : lp.d_wpk_merged;
        a.N = N; a.H = H; a.W = W;
        if (net->force_exact_div) a.fd.ok = 0;
        a.in = cur;
        int src = (k == 0) ? (in_dtype == SESRQ_F32 ? SRC_F32 : SRC_I8) : SRC_NHWC16;
        int epi = (k == L - 1) ? EPI_LAST : (k == L - 2 ? EPI_PRERES : EPI_MID);
        void *dst = nullptr;
        if (k == 0) { dst = bufS; a.rc_out = net->rc_separate ? bufRC : nullptr; }
        else if (k < L - 1) dst = (cur == bufA) ? bufB : bufA;
        a.out = dst;
        a.rc_in = bufRC;
        a.out_q = out_q; a.out_f = (float *)out_f;
        a.anchor = (net->anchor_add && in_dtype == SESRQ_F32) ? (const float *)in : nullptr;
        if (taps) {
            a.dbg_pe = (int *)taps->pe_out[k];
            a.dbg_add = (int *)taps->pe_add[k];
            if (k == 0) a.dbg_q0 = (signed char *)taps->act[0];
            else if (taps->act[k] && launch_unpack_nhwc16(cur, (signed char *)taps->act[k], N, lp.ic, H, W, st)) {
                set_error("sesrq_forward: debug unpack launch failed"); return 1;
            }
        }
        if (ev && hipEventRecord(ev[2 * k], st) != hipSuccess) { set_error("hipEventRecord failed"); return 1; }
        const bool use_mfma = net->engine != SESRQ_ENGINE_DOT4 && lp.mfma_kind != MFMA_NONE && !dbg;
        if (use_mfma) {
            a.afrag = eff.general ? lp.d_afrag_general : lp.d_afrag_merged;
            // exactly one PE can saturate (and nothing forces the full per-PE path): merged chain + that PE's chain
            const bool one_pe = lp.general && !net->force_general && !dbg && lp.d_afrag_others && net->acc_bits == 18 && net->add_bits == 20;
            if (one_pe) { a.afrag = lp.d_afrag_others; a.afrag2 = lp.d_afrag_general; a.risky_pe = __builtin_ctz(lp.risky_mask); }
            if (lp.d_afrag_pesplit) a.afrag = lp.d_afrag_pesplit;
            if (launch_mfma(lp, a, src, epi, eff.general, st, one_pe)) return 1;
        } else if (launch_dot4(eff, a, src, epi, st)) return 1;
        if (ev && hipEventRecord(ev[2 * k + 1], st) != hipSuccess) { set_error("hipEventRecord failed"); return 1; }
        cur = dst;
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

int sesrq_forward_timed(const sesrq_net *net, const void *in, int in_dtype, void *out_q, void *out_f, int N, int H, int W,
                        void *workspace, size_t workspace_bytes, void *stream, int iters, float *layer_ms, float *forward_ms) {
    if (!net || iters < 1 || !layer_ms) { set_error("sesrq_forward_timed: bad argument"); return 1; }
    const int L = net->L;
    std::vector<hipEvent_t> ev((size_t)2 * L * iters);
    for (auto &e : ev) HIP_OK(hipEventCreate(&e));
    int rc = 0;
    for (int it = 0; it < iters && !rc; ++it)
        rc = forward_impl(net, in, in_dtype, out_q, out_f, N, H, W, workspace, workspace_bytes, stream, nullptr,
                          ev.data() + (size_t)2 * L * it);
    if (!rc && hipStreamSynchronize((hipStream_t)stream) != hipSuccess) { set_error("hipStreamSynchronize failed"); rc = 1; }
    if (!rc) {
        for (int k = 0; k < L; ++k) layer_ms[k] = 0.f;
        double fw = 0;
        for (int it = 0; it < iters; ++it) {
            hipEvent_t *e = ev.data() + (size_t)2 * L * it;
            for (int k = 0; k < L; ++k) {
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e[2 * k], e[2 * k + 1]);
                layer_ms[k] += ms / iters;
            }
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e[0], e[2 * L - 1]);
            fw += ms;
        }
        if (forward_ms) *forward_ms = (float)(fw / iters);
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
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
