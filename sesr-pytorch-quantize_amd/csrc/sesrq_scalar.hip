// sesrq C ABI, part 4 of 4: the host scalar code of the path (load time): requant-constant encoder, weight quantiser, bias constant,
// calibration finaliser.  Plain C++ -- no device code.  See include/sesrq.h.
#include <math.h>

#include <algorithm>

#include "sesrq_common.h"

using namespace sesrq;

extern "C" {

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

int sesrq_quantize_weight_per_channel(const float *w, int oc, size_t per_oc, int width, int8_t *wq, double *scale_oc) {
    if (!w || !wq || !scale_oc || oc < 1 || per_oc == 0) { set_error("sesrq_quantize_weight_per_channel: null/empty argument"); return 1; }
    for (int o = 0; o < oc; ++o)
        if (sesrq_quantize_weight(w + (size_t)o * per_oc, per_oc, width, wq + (size_t)o * per_oc, scale_oc + o)) {
            set_error("sesrq_quantize_weight_per_channel: output channel " + std::to_string(o) + ": " + sesrq_last_error());
            return 1;
        }
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
