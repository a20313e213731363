// Calibration pass (the reference's exe_mode 0; test.py:79-113,141-217): the float net is
// run with fake-quantised weights and activations while the running min/max of every conv input is
// observed.  It produces the activation domains (scale, zero) the integer path consumes; it is not a
// throughput path, so the kernels are simple (one lane = one pixel, all output channels).
//
// Per conv (reference: quantize_asymmetrical_by_tensor mode 0 -> reshape_input_for_hardware_pe -> Conv2d
// with Wq*sw -> PEs_and_bias_adder mode 0 -> activation), with this batch's (scale, zero) of the input:
//   q      = clamp8(rint(x/scale + zero)) - zero             integer; outside the frame 0     (quan_func.py:207,215)
//   pe_p   = sum_{ic = p mod 4, taps} Wq * q                  exact in int32                   (quan_func.py:298-318)
//   v_p    = clamp(f32(pe_p) * f32(scale*sw), fmin18, fmax18)                                  (quan_func.py:330-333)
//   v      = clamp(v_0 + v_1 + v_2 + v_3, fmin20, fmax20) + bias_q * f32(scale*sw)             (quan_func.py:431-434,459)
// The reference forms the same sums in fp32 (fl(Wq*sw) * fl(q*scale) accumulated by oneDNN in an
// unspecified order); here the integer sum is exact and scaled once, so results agree to fp32 rounding
// -- calibration is pinned to the reference within a tolerance, not bit for bit.
#include <algorithm>
#include <cmath>

#include "sesrq_common.h"

namespace sesrq {

struct CalibArgs {
    const float *in;        // (N, IC, H, W) fp32
    const float *skip;      // (N, OC, H, W) added after the activation (long residual) or NULL
    float *out;             // (N, OC, H, W)
    const int *w;           // [oc][ic][k][k] int32 (quantised weights)
    const float *qbias;     // [oc]  bias_q * f32(scale*sw)
    int N, H, W, ic, oc;
    float scale, zero;      // this batch's input domain
    float ss;               // f32(scale * sw)
    float acc_lo, acc_hi, add_lo, add_hi;
    int relu;
};

template <int K>
__global__ __launch_bounds__(256) void calib_conv_kernel(const CalibArgs a) {
    constexpr int R = K / 2, TW = 32, TH = 8, SW = TW + K - 1, SH = TH + K - 1;
    __shared__ short tile[SESRQ_MAX_CH][SH * SW];
    const int tid = threadIdx.x, lx = tid & 31, ly = tid >> 5;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, n = blockIdx.z;
    const size_t HW = (size_t)a.H * a.W;
    for (int c = 0; c < a.ic; ++c)
        for (int i = tid; i < SH * SW; i += 256) {
            const int ty = i / SW, tx = i - ty * SW, gy = y0 - R + ty, gx = x0 - R + tx;
            int q = 0;
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                const float xv = a.in[((size_t)n * a.ic + c) * HW + (size_t)gy * a.W + gx];
                const float r = fminf(fmaxf(rintf(__fadd_rn(__fdiv_rn(xv, a.scale), a.zero)), -128.f), 127.f);
                q = (int)(r - a.zero);
            }
            tile[c][i] = (short)q;
        }
    __syncthreads();
    const int gx = x0 + lx, gy = y0 + ly;
    if (gx >= a.W || gy >= a.H) return;
    for (int o = 0; o < a.oc; ++o) {
        float sum = 0.f;
        for (int p = 0; p < 4; ++p) {
            int acc = 0;
            for (int c = p; c < a.ic; c += 4) {
                const int *wp = a.w + ((size_t)o * a.ic + c) * K * K;
#pragma unroll
                for (int ky = 0; ky < K; ++ky)
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) acc += wp[ky * K + kx] * (int)tile[c][(ly + ky) * SW + lx + kx];
            }
            const float v = fminf(fmaxf(__fmul_rn((float)acc, a.ss), a.acc_lo), a.acc_hi);
            sum = (p == 0) ? v : __fadd_rn(sum, v);
        }
        float v = __fadd_rn(fminf(fmaxf(sum, a.add_lo), a.add_hi), a.qbias[o]);
        if (a.relu) v = fmaxf(v, 0.f);
        const size_t off = ((size_t)n * a.oc + o) * HW + (size_t)gy * a.W + gx;
        if (a.skip) v = __fadd_rn(v, a.skip[off]);
        a.out[off] = v;
    }
}

// order-preserving float <-> uint map so that min/max can use integer atomics
__device__ __forceinline__ unsigned f2ord(float f) { const unsigned u = __builtin_bit_cast(unsigned, f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__global__ void calib_minmax_kernel(const float *x, size_t n, unsigned *mm) {
    float lo = INFINITY, hi = -INFINITY;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        lo = fminf(lo, v); hi = fmaxf(hi, v);
    }
    for (int s = 32; s > 0; s >>= 1) { lo = fminf(lo, __shfl_xor(lo, s)); hi = fmaxf(hi, __shfl_xor(hi, s)); }
    if ((threadIdx.x & 63) == 0) { atomicMin(&mm[0], f2ord(lo)); atomicMax(&mm[1], f2ord(hi)); }
}
__global__ void calib_minmax_finish(const unsigned *mm, float *out) {
    for (int i = 0; i < 2; ++i) {
        const unsigned o = mm[i], u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
        out[i] = __builtin_bit_cast(float, u);
    }
}
__global__ void calib_fakequant_kernel(const float *in, float *out, size_t n, float scale, float zero) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float q = fminf(fmaxf(rintf(__fadd_rn(__fdiv_rn(in[i], scale), zero)), -128.f), 127.f);
        out[i] = __fmul_rn(q - zero, scale);
    }
}

// Histogram of a tensor over [lo, hi) in `bins` equal bins (bin = floor((x - lo) * bins / (hi - lo)), clamped to the range: values
// outside count in the edge bins), accumulated INTO hist.  One private copy per workgroup in LDS, merged with one atomic per
// non-empty bin: the entropy calibration variant's second pass (no reference counterpart).
constexpr int CALIB_MAX_BINS = 4096;
__global__ __launch_bounds__(256) void calib_hist_kernel(const float *x, size_t n, float lo, float inv_w, int bins, unsigned *hist) {
    __shared__ unsigned h[CALIB_MAX_BINS];
    for (int i = threadIdx.x; i < bins; i += 256) h[i] = 0;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = x[i];
        if (v == v) {                    // NaNs are not counted
            int b = (int)floorf(__fmul_rn(__fsub_rn(v, lo), inv_w));
            b = min(max(b, 0), bins - 1);
            atomicAdd(&h[b], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < bins; i += 256)
        if (h[i]) atomicAdd(&hist[i], h[i]);
}

}  // namespace sesrq

using namespace sesrq;

extern "C" {

int sesrq_calib_minmax(const float *x, size_t n, float *out_min_max, void *scratch8, void *stream) {
    if (!x || !out_min_max || !scratch8 || n == 0) { set_error("sesrq_calib_minmax: bad argument"); return 1; }
    hipStream_t st = (hipStream_t)stream;
    static const unsigned init[2] = {0xffffffffu, 0u};
    if (hipMemcpyAsync(scratch8, init, sizeof(init), hipMemcpyHostToDevice, st) != hipSuccess) { set_error("sesrq_calib_minmax: memcpy failed"); return 1; }
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 2048);
    launch_kernel<calib_minmax_kernel>(dim3(blocks), dim3(256), 0, st, x, n, (unsigned *)scratch8);
    launch_kernel<calib_minmax_finish>(dim3(1), dim3(1), 0, st, (const unsigned *)scratch8, out_min_max);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int sesrq_calib_conv(const sesrq_calib_conv_desc *d, const float *in, const float *skip, float *out, int N, int H, int W, void *stream) {
    if (!d || !in || !out || !d->w || !d->qbias) { set_error("sesrq_calib_conv: null argument"); return 1; }
    if ((d->k != 3 && d->k != 5) || d->ic < 1 || d->ic > SESRQ_MAX_CH || d->oc < 1 || d->oc > SESRQ_MAX_CH) { set_error("sesrq_calib_conv: unsupported layer shape"); return 1; }
    if (N < 1 || H < 1 || W < 1 || !(d->in_scale > 0.f)) { set_error("sesrq_calib_conv: bad size or scale"); return 1; }
    CalibArgs a;
    a.in = in; a.skip = skip; a.out = out; a.w = d->w; a.qbias = d->qbias;
    a.N = N; a.H = H; a.W = W; a.ic = d->ic; a.oc = d->oc;
    a.scale = d->in_scale; a.zero = (float)d->in_zero; a.ss = d->ss;
    a.acc_lo = d->acc_lo; a.acc_hi = d->acc_hi; a.add_lo = d->add_lo; a.add_hi = d->add_hi; a.relu = d->relu;
    dim3 grid((W + 31) / 32, (H + 7) / 8, N);
    if (d->k == 3) launch_kernel<calib_conv_kernel<3>>(grid, dim3(256), 0, (hipStream_t)stream, a);
    else launch_kernel<calib_conv_kernel<5>>(grid, dim3(256), 0, (hipStream_t)stream, a);
    if (hipGetLastError() != hipSuccess) { set_error("sesrq_calib_conv: launch failed"); return 1; }
    return 0;
}

int sesrq_calib_histogram(const float *x, size_t n, float lo, float hi, int bins, uint32_t *hist, void *stream) {
    if (!x || !hist || n == 0) { set_error("sesrq_calib_histogram: bad argument"); return 1; }
    if (bins < 2 || bins > CALIB_MAX_BINS) { set_error("sesrq_calib_histogram: bins must be 2..4096"); return 1; }
    if (!(hi > lo) || !std::isfinite(lo) || !std::isfinite(hi)) { set_error("sesrq_calib_histogram: need finite lo < hi"); return 1; }
    const float inv_w = (float)bins / (hi - lo);
    const int blocks = (int)std::min<size_t>((n + 256 * 16 - 1) / (256 * 16), 2048);
    launch_kernel<calib_hist_kernel>(dim3(std::max(blocks, 1)), dim3(256), 0, (hipStream_t)stream, x, n, lo, inv_w, bins, (unsigned *)hist);
    if (hipGetLastError() != hipSuccess) { set_error("sesrq_calib_histogram: launch failed"); return 1; }
    return 0;
}

int sesrq_calib_fakequant(const float *in, float *out, size_t n, float scale, int zero, void *stream) {
    if (!in || !out || n == 0 || !(scale > 0.f)) { set_error("sesrq_calib_fakequant: bad argument"); return 1; }
    launch_kernel<calib_fakequant_kernel>(dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out, n, scale, (float)zero);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

}  // extern "C"
