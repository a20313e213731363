// sesrq dot4 engine: im2col-free direct k x k INT8 convolution for gfx950 (CDNA4).
//
// One lane = one output pixel x all output channels.  The input tile (+halo) is staged once
// through LDS as NHWC int8 with the channels in PE-major order
//     byte(c) = 4*(c % 4) + c / 4      ->  32-bit word p of a pixel = the four channels of PE p
// so that one v_dot4_i32_i8 is exactly "one tap of one PE for one output channel", with the
// weight word in an SGPR (wave-uniform scalar loads).  The four PE partial sums are kept in
// separate accumulators whenever the load-time proof cannot rule out 18-/20-bit saturation
// (GENERAL); otherwise they share one accumulator.
//
// Epilogue = the reference's integer tail, fused (paths relative to the reference root):
//   18-bit PE clamp, PE sum            myQL/quan_func.py:370,380-386
//   20-bit clamp, + add constant       myQL/quan_func.py:437,491
//   t = f32(acc*M) * 2^-n              myQL/quan_func.py:529,560,584,605   (fp32 product rounding kept)
//   ReLU                               models/model_utils_pt.py:24-26
//   q = clamp8(rint(t + zero_next))    myQL/quan_func.py:280
//   residual merge                     myQL/quan_func.py:249-270
//   output requant + dequant + shuffle myQL/quan_func.py:584-594, models/sesr_sim.py:49
// and, for layer 0, the input quantiser  q0 = clamp8(rint(x/s0 + z0))  (quan_func.py:225)
// fused into the LDS staging.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off   (no fast-math: the fp32
// multiply-then-add sequence of the reference must not be contracted).
#include "sesrq_common.h"

namespace sesrq {

constexpr int TW = 32;   // output tile width  (lanes 0..31 of a half-wave row)
constexpr int TH = 8;    // output tile height (256 threads = 4 waves, 2 rows per wave)

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }
__device__ __forceinline__ float q8f(float v) { return fminf(fmaxf(rintf(v), -128.f), 127.f); }

template <int K, int IN_DW, bool GENERAL, int EPI, int OCP, int SRC>
__global__ __launch_bounds__(256) void conv_dot4_kernel(const ConvArgs a) {
    constexpr int R = K / 2;
    constexpr int SW = TW + K - 1;             // staged tile width  (pixels)
    constexpr int SH = TH + K - 1;             // staged tile height
    constexpr int NACC = GENERAL ? 4 : 1;
    __shared__ __attribute__((aligned(16))) int lds[SH * SW * IN_DW];

    const int tid = threadIdx.x;
    const int lx = tid & (TW - 1), ly = tid / TW;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int n = blockIdx.z;
    const int H = a.H, W = a.W;
    const size_t HW = (size_t)H * W;

    // ---- stage the input tile (+halo) into LDS; outside the frame = pad word (zc bytes)
    for (int i = tid; i < SH * SW; i += 256) {
        const int ty = i / SW, tx = i - ty * SW;
        const int gy = y0 - R + ty, gx = x0 - R + tx;
        const bool inside = (gy >= 0) & (gy < H) & (gx >= 0) & (gx < W);
        if constexpr (SRC == SRC_NHWC16) {
            int4 v = make_int4(a.pad_word, a.pad_word, a.pad_word, a.pad_word);
            if (inside) v = reinterpret_cast<const int4 *>(a.in)[(size_t)n * HW + (size_t)gy * W + gx];
            reinterpret_cast<int4 *>(lds)[i] = v;
        } else {
            int word = a.pad_word;
            if (inside) {
                word = 0;
                for (int c = 0; c < a.ic; ++c) {
                    const size_t off = ((size_t)n * a.ic + c) * HW + (size_t)gy * W + gx;
                    int q;
                    if constexpr (SRC == SRC_F32) {
                        const float xv = reinterpret_cast<const float *>(a.in)[off];
                        q = (int)q8f(__fadd_rn((fd_reciprocal(a.fd) ? __fmul_rn(xv, a.fd.r) : __fdiv_rn(xv, a.s_in)), a.z_in));
                    } else if constexpr (SRC == SRC_I8D) {
                        const float xv = __fmul_rn((float)(int)reinterpret_cast<const signed char *>(a.in)[off] - a.z_prev, a.s_prev);
                        q = (int)q8f(__fadd_rn((fd_reciprocal(a.fd) ? __fmul_rn(xv, a.fd.r) : __fdiv_rn(xv, a.s_in)), a.z_in));
                    } else {
                        q = reinterpret_cast<const signed char *>(a.in)[off];
                    }
                    word |= (q & 0xff) << (8 * c);
                }
            }
            lds[i] = word;
        }
    }
    __syncthreads();

    // ---- accumulate: taps x PEs x output channels, weights from SGPRs
    int acc[NACC][OCP];
#pragma unroll
    for (int p = 0; p < NACC; ++p)
#pragma unroll
        for (int o = 0; o < OCP; ++o) acc[p][o] = 0;

    const int *__restrict__ wp = a.wpk;
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const int li = (ly + ky) * SW + lx + kx;
            int px[4];
            if constexpr (IN_DW == 4) {
                const int4 v = reinterpret_cast<const int4 *>(lds)[li];
                px[0] = v.x; px[1] = v.y; px[2] = v.z; px[3] = v.w;
            } else {
                px[0] = px[1] = px[2] = px[3] = lds[li];
            }
            const int tap = ky * K + kx;
#pragma unroll
            for (int o = 0; o < OCP; ++o) {
                if constexpr (IN_DW == 1 && !GENERAL) {
                    acc[0][o] = __builtin_amdgcn_sdot4(px[0], wp[(tap * OCP + o) * 4], acc[0][o], false);
                } else {
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        acc[GENERAL ? p : 0][o] =
                            __builtin_amdgcn_sdot4(px[p], wp[(tap * OCP + o) * 4 + p], acc[GENERAL ? p : 0][o], false);
                }
            }
        }
    }

    const int gx = x0 + lx, gy = y0 + ly;
    if (gx >= W || gy >= H) return;
    const size_t pix = (size_t)n * HW + (size_t)gy * W + gx;
    // (all global stores come after the last weight load, so the weight loads stay scalar)
    if constexpr (SRC != SRC_NHWC16) {
        if (a.dbg_q0) {
            const int word = lds[(ly + R) * SW + lx + R];
            for (int c = 0; c < a.ic; ++c)
                a.dbg_q0[((size_t)n * a.ic + c) * HW + (size_t)gy * W + gx] = (signed char)((word >> (8 * c)) & 0xff);
        }
    }

    // ---- epilogue
    float t[OCP];
    int n_hi = 0, n_lo = 0;
#pragma unroll
    for (int o = 0; o < OCP; ++o) {
        int s;
        if constexpr (GENERAL) {
            int pe[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) pe[p] = clampi(acc[p][o], a.acc_lo, a.acc_hi);
            if (a.dbg_ovf && o < a.oc) {       // the reference's 'max_overflow' / 'min_overflow' events (myQL/quan_func.py:358-361)
#pragma unroll
                for (int p = 0; p < 4; ++p) { n_hi += acc[p][o] > a.acc_hi; n_lo += acc[p][o] < a.acc_lo; }
            }
            s = clampi(pe[0] + pe[1] + pe[2] + pe[3], a.add_lo, a.add_hi);
            if (a.dbg_pe && o < a.oc) {
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    a.dbg_pe[(((size_t)n * 4 + p) * a.oc + o) * HW + (size_t)gy * W + gx] = pe[p];
            }
            if (a.dbg_add && o < a.oc) a.dbg_add[((size_t)n * a.oc + o) * HW + (size_t)gy * W + gx] = s;
        } else {
            s = acc[0][o];
        }
        s += a.add_const[o];
        // per-output-channel requant constants (sesrq_layer_desc.M_oc; wave-uniform: scalar loads) or the layer's one (M, n)
        const float Mf_o = a.mn_oc ? a.mn_oc[o < SESRQ_MAX_CH ? o : 0].x : a.Mf, sh_o = a.mn_oc ? a.mn_oc[o < SESRQ_MAX_CH ? o : 0].y : a.sh;
        float v = __fmul_rn((float)s, Mf_o) * sh_o;
        if (a.relu) v = fmaxf(v, 0.f);
        t[o] = v;
        if (a.dbg_t && o < a.oc) a.dbg_t[((size_t)n * a.oc + o) * HW + (size_t)gy * W + gx] = v;      // layer 0: shortcut_tensor.pt (quan_func.py:529-549)
    }

    if constexpr (GENERAL) {
        if (a.dbg_ovf) {
            if (n_hi) atomicAdd(a.dbg_ovf, n_hi);
            if (n_lo) atomicAdd(a.dbg_ovf + 1, n_lo);
        }
    }

    if constexpr (EPI == EPI_MID || EPI == EPI_PRERES) {
        static_assert(OCP == 16 || EPI == EPI_LAST, "hidden layers are padded to 16 channels");
        int rcw[4] = {0, 0, 0, 0};
        if constexpr (EPI == EPI_PRERES) {
            const int4 v = reinterpret_cast<const int4 *>(a.rc_in)[pix];
            rcw[0] = v.x; rcw[1] = v.y; rcw[2] = v.z; rcw[3] = v.w;
        }
        int ow[4] = {0, 0, 0, 0}, rw[4] = {0, 0, 0, 0};
#pragma unroll
        for (int o = 0; o < OCP; ++o) {
            const int p = o & 3, j = o >> 2;         // PE-major byte position
            float q;
            if constexpr (EPI == EPI_PRERES) {
                const float rc = (float)(signed char)((rcw[p] >> (8 * j)) & 0xff);
                const float ic = q8f(__fadd_rn(t[o], -128.f));
                if (a.dbg_ic && o < a.oc) a.dbg_ic[((size_t)n * a.oc + o) * HW + (size_t)gy * W + gx] = (signed char)(int)ic;      // input.4.spcial.pt (quan_func.py:250,254)
                const float u = rc + ic + 256.f;
                const float v = __fmul_rn(u, a.Mres) * a.shres;
                q = q8f(__fadd_rn(v, a.z_merge));
            } else {
                q = q8f(__fadd_rn(t[o], a.z_next));
                if (a.rc_out) rw[p] |= (((int)q8f(__fadd_rn(t[o], -128.f))) & 0xff) << (8 * j);
            }
            ow[p] |= (((int)q) & 0xff) << (8 * j);
        }
        reinterpret_cast<int4 *>(a.out)[pix] = make_int4(ow[0], ow[1], ow[2], ow[3]);
        if constexpr (EPI == EPI_MID) {
            if (a.rc_out) reinterpret_cast<int4 *>(a.rc_out)[pix] = make_int4(rw[0], rw[1], rw[2], rw[3]);
        }
    } else {
        // last conv: requantise into the output domain, PixelShuffle(r) fused into the store
        const int r = a.ps, r2 = r * r;
        const int Ho = H * r, Wo = W * r;
        const int cout = a.oc / r2;
#pragma unroll
        for (int o = 0; o < OCP; ++o) {
            if (o < a.oc) {
                const float q = q8f(__fadd_rn(t[o], a.z_out));
                const int c = o / r2, rem = o - c * r2, i = rem / r, j = rem - i * r;
                const size_t off = (((size_t)n * cout + c) * Ho + (size_t)gy * r + i) * Wo + (size_t)gx * r + j;
                if (a.out_q) reinterpret_cast<signed char *>(a.out_q)[off] = (signed char)(int)q;
                if (a.out_f) {
                    float yv = __fmul_rn(q - a.z_out, a.s_out);
                    if (a.anchor) yv = __fadd_rn(yv, a.anchor[((size_t)n * cout + c) * HW + (size_t)gy * W + gx]);   // test.py:148-155
                    a.out_f[off] = yv;
                }
            }
        }
    }
}

// NHWC16 (PE-major) -> NCHW int8, debug taps only
__global__ void unpack_nhwc16_kernel(const int4 *__restrict__ in, signed char *__restrict__ out, int C, size_t HW,
                                     size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t n = i / HW, hw = i - n * HW;
    const int4 v = in[i];
    const int w[4] = {v.x, v.y, v.z, v.w};
    for (int c = 0; c < C; ++c) out[(n * C + c) * HW + hw] = (signed char)((w[c & 3] >> (8 * (c >> 2))) & 0xff);
}

int launch_unpack_nhwc16(const void *nhwc, signed char *nchw, int N, int C, int H, int W, hipStream_t st) {
    const size_t HW = (size_t)H * W, total = HW * N;
    launch_kernel<unpack_nhwc16_kernel>(dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const int4 *>(nhwc), nchw, C, HW, total);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

template <int K, int IN_DW, bool GENERAL, int EPI, int OCP, int SRC>
static void launch_one(const ConvArgs &a, hipStream_t st) {
    dim3 grid((a.W + TW - 1) / TW, (a.H + TH - 1) / TH, a.N);
    launch_kernel<conv_dot4_kernel<K, IN_DW, GENERAL, EPI, OCP, SRC>>(grid, dim3(256), 0, st, a);
}

template <int K, bool GENERAL>
static int dispatch(const LayerPlan &lp, const ConvArgs &a, int src, int epi, hipStream_t st) {
    if (src != SRC_NHWC16) {
        if (epi != EPI_MID) { set_error("dot4: first layer must be a hidden layer"); return 1; }
        if (src == SRC_F32) launch_one<K, 1, GENERAL, EPI_MID, 16, SRC_F32>(a, st);
        else if (src == SRC_I8D) launch_one<K, 1, GENERAL, EPI_MID, 16, SRC_I8D>(a, st);
        else launch_one<K, 1, GENERAL, EPI_MID, 16, SRC_I8>(a, st);
        return 0;
    }
    if (epi == EPI_MID) { launch_one<K, 4, GENERAL, EPI_MID, 16, SRC_NHWC16>(a, st); return 0; }
    if (epi == EPI_PRERES) { launch_one<K, 4, GENERAL, EPI_PRERES, 16, SRC_NHWC16>(a, st); return 0; }
    switch (lp.ocp) {
        case 4: launch_one<K, 4, GENERAL, EPI_LAST, 4, SRC_NHWC16>(a, st); return 0;
        case 8: launch_one<K, 4, GENERAL, EPI_LAST, 8, SRC_NHWC16>(a, st); return 0;
        case 12: launch_one<K, 4, GENERAL, EPI_LAST, 12, SRC_NHWC16>(a, st); return 0;
        case 16: launch_one<K, 4, GENERAL, EPI_LAST, 16, SRC_NHWC16>(a, st); return 0;
    }
    set_error("dot4: unsupported padded channel count");
    return 1;
}

int launch_dot4(const LayerPlan &lp, bool general, const ConvArgs &a, int src, int epi, hipStream_t st) {
    int rc;
    if (lp.k == 3) rc = general ? dispatch<3, true>(lp, a, src, epi, st) : dispatch<3, false>(lp, a, src, epi, st);
    else if (lp.k == 5) rc = general ? dispatch<5, true>(lp, a, src, epi, st) : dispatch<5, false>(lp, a, src, epi, st);
    else { set_error("dot4: kernel size must be 3 or 5"); return 1; }
    if (rc) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("dot4 launch failed: ") + hipGetErrorString(e)); return 1; }
    return 0;
}

}  // namespace sesrq
