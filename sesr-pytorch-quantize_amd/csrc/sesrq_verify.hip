// Load-time proofs of sesrq_create: the input quantiser's fast division (below), the one-fma requant forms (prove_direct_requant,
// prove_single_requant) and the static saturation bound of a layer (saturation_free).
//
// Verified fast division for the input quantiser.
//
// The reference quantises the frame with a TRUE fp32 division (myQL/quan_func.py:225):
//     q0(x) = clamp8(rint(fl(fl(x / s) + z)))
// An IEEE-correct division costs ~12 VALU instructions per element on gfx950 and is half of the
// first layer's instruction stream.  Because s is one scalar per net, the quotient can instead be
// formed with the correctly rounded reciprocal r = RN(1/s) and one Newton/Markstein correction,
//     q = x*r ;  rem = fma(-s, q, x) ;  q' = fma(rem, r, q)            (3 instructions)
// which equals RN(x/s) for all but (possibly) pathological operands.  "Possibly" is not good
// enough for a bit-exact path, so sesrq_create PROVES it for the net's own (s, z): a kernel
// enumerates EVERY fp32 value x in [xlo, xhi] -- the range whose ends quantise to exactly -128 and
// 127, inputs are clamped to it first -- and compares the UNCLAMPED rint of the fast form with the
// clamped true division: the chain is monotonic in x, so the clamp of x is the int8 clamp and the
// kernels need no second one (round 3; before, the range ended 8 steps beyond and the result was
// clamped again).  ~1e9 values, about a millisecond of GPU time, cached per (s, z).  Any mismatch
// (or ends that do not land on -128 / 127) disables the fast path for that net.
#include <cmath>
#include <map>
#include <mutex>
#include <utility>

#include "sesrq_common.h"

namespace sesrq {

__device__ __forceinline__ float q8_exact(float x, float s, float z) {
    return __builtin_amdgcn_fmed3f(rintf(__fadd_rn(__fdiv_rn(x, s), z)), -128.f, 127.f);
}
__device__ __forceinline__ float q8_fast(float x, float s, float r, float z) {
    const float q = __fmul_rn(x, r);
    const float rem = __builtin_fmaf(-s, q, x);
    const float q1 = __builtin_fmaf(rem, r, q);
    return rintf(__fadd_rn(q1, z));        // no clamp: x comes from [xlo, xhi] only
}

// bit patterns [b0, b1] (same sign, increasing magnitude)
__global__ void verify_fastdiv_kernel(unsigned b0, unsigned b1, float s, float r, float z, unsigned long long *bad) {
    const unsigned long long n = (unsigned long long)b1 - b0 + 1;
    unsigned long long local = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __builtin_bit_cast(float, (unsigned)(b0 + i));
        local += (q8_exact(x, s, z) != q8_fast(x, s, r, z)) ? 1 : 0;
    }
    if (local) atomicAdd(bad, local);
}

static std::mutex g_mu;
static std::map<std::pair<unsigned, int>, FastDiv> g_cache;

// Option exact_div = 2: x * fl(1/s) on the proven form's instructions (r2 = 0).  x is clamped to the same saturating range
// first -- multiplication by r > 0 is monotonic and the bounds quantise to exactly -128 and 127 (checked), so clamping x IS the int8
// clamp, and it keeps x * r finite (fma(-inf, 0, inf) would be NaN).  ok == 0 (division) if the range is degenerate.
FastDiv reciprocal_form(float s, int zero) {
    FastDiv fd;
    fd.ok = 0; fd.r = 0.f; fd.r2 = 0.f; fd.xlo = 0.f; fd.xhi = 0.f;
    if (!(s > 0.f) || !std::isfinite(s)) return fd;
    fd.r = 1.0f / s;
    fd.xlo = (float)((-128.0 - (double)zero) * (double)s);
    fd.xhi = (float)((127.0 - (double)zero) * (double)s);
    const float z = (float)zero;
    const bool ok = std::isfinite(fd.r) && fd.r > 0.f && std::isfinite(fd.xlo) && std::isfinite(fd.xhi) && fd.xlo < fd.xhi &&
                    rintf(fd.xlo * fd.r + z) == -128.f && rintf(fd.xhi * fd.r + z) == 127.f;
    fd.ok = ok ? 1 : 0;
    return fd;
}

FastDiv prove_fastdiv(float s, int zero) {
    FastDiv fd;
    fd.ok = 0; fd.r = 0.f; fd.r2 = 0.f; fd.xlo = 0.f; fd.xhi = 0.f;
    if (!(s > 0.f) || !std::isfinite(s)) return fd;
    const std::pair<unsigned, int> key(__builtin_bit_cast(unsigned, s), zero);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_cache.find(key);
        if (it != g_cache.end()) return it->second;
    }
    const float z = (float)zero;
    fd.r = fd.r2 = (float)(1.0L / (long double)s);
    fd.xlo = (float)((-128.0 - (double)zero) * (double)s);
    fd.xhi = (float)((127.0 - (double)zero) * (double)s);
    bool ok = std::isfinite(fd.r) && std::isfinite(fd.xlo) && std::isfinite(fd.xhi) && fd.r > 0.f && fd.xlo < fd.xhi;
    unsigned long long *d_bad = nullptr;
    if (ok && hipMalloc((void **)&d_bad, sizeof(*d_bad)) == hipSuccess && hipMemset(d_bad, 0, sizeof(*d_bad)) == hipSuccess) {
        auto run = [&](float a, float b) {   // all floats between a and b, same sign, |a| <= |b|
            const unsigned ua = __builtin_bit_cast(unsigned, a), ub = __builtin_bit_cast(unsigned, b);
            launch_kernel<verify_fastdiv_kernel>(dim3(4096), dim3(256), 0, 0, ua, ub, s, fd.r, z, d_bad);
        };
        if (fd.xlo < 0.f) { run(-0.0f, fd.xlo); if (fd.xhi >= 0.f) run(0.0f, fd.xhi); else run(fd.xhi, fd.xlo); }   // (both negative: |xhi| <= |xlo|)
        else run(fd.xlo, fd.xhi);
        unsigned long long bad = 1;
        ok = hipMemcpy(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost) == hipSuccess && bad == 0;
        // the ends of the clamp range must land on the ends of the int8 range (host check with the exact formula, unclamped)
        ok = ok && rintf(fd.xlo / s + z) == -128.f && rintf(fd.xhi / s + z) == 127.f;
    } else {
        ok = false;
    }
    if (d_bad) (void)hipFree(d_bad);
    fd.ok = ok ? 1 : 0;
    std::lock_guard<std::mutex> lk(g_mu);
    g_cache[key] = fd;
    return fd;
}

// One-fma requant into a domain whose zero point is -128 (hidden layers behind ReLU, the output layer: every reference bundle).
// The reference value is q = clamp8(rint(fl(t' - 128))) with t' = fl(s * M) * 2^-n (myQL/quan_func.py:280; relu's clamp is the int8
// clamp here); the kernels' cvt_pk_u8 form evaluates cvt_u8(fl(fl(t' - 128) + 128)) - 128, proven equal for EVERY fp32 t'
// (tools/cvtpk_epilogue_probe.hip).  fl(t' - 128) is exact for t' >= 64 (Sterbenz) and rounds t' to the 2^-17 grid below, so
// cvt_u8(t') itself -- t' straight out of ONE fma, fl((MAGIC + s) * (M 2^-n) - MAGIC * M 2^-n) = fl(s * M) * 2^-n, no "- 128",
// no "+ 128" -- differs only where some reachable t' < 64 lies within 2^-18 of a half-integer without being one.  Whether that
// happens depends on (M, n) alone: all s whose t' lies in [-2, 258] are enumerated (a few 1e5 values; outside both forms
// saturate), with the device's operations restated in host fp32 (this file is built with -ffp-contract=off).
bool prove_direct_requant(unsigned M, unsigned n) {
    if (M == 0 || 3ull * M >= (1ull << 18) || n > 40) return false;          // MAGIC * M must be exact (the biased form's own condition)
    const float Mf = (float)M, sh = ldexpf(1.0f, -(int)n), Md = Mf * sh, Cd = -(12582912.f * Mf) * sh;
    const double scale = (double)M * ldexp(1.0, -(int)n);
    long long s_lo = (long long)floor(-2.0 / scale), s_hi = (long long)ceil(258.0 / scale);
    const long long lim = (1ll << 22) - 1;
    if (s_lo < -lim) s_lo = -lim;
    if (s_hi > lim) s_hi = lim;
    auto cvt_u8 = [](float w) { const float r = rintf(w); return r < 0.f ? 0.f : (r > 255.f ? 255.f : r); };
    for (long long s = s_lo; s <= s_hi; ++s) {
        const float y = 12582912.f + (float)s;                                 // bits = MAGIC_I + s: exact for |s| < 2^22
        const float t = fmaf(y, Mf, -(12582912.f * Mf));                       // fl(s * M)
        const float v = fmaf(t, sh, -128.f);                                   // the kernels' second fma
        const float ref = cvt_u8(v + 128.f);
        const float one = cvt_u8(fmaf(y, Md, Cd));
        if (ref != one) return false;
    }
    return true;
}

// Second candidate where the first fails: v = fl(s * M * 2^-n - 128) out of ONE fma (a single rounding of the exact value; the
// reference rounds s * M first), then cvt_u8(fl(v + 128)) as in the two-step form: one fma less than that.  The two roundings
// differ for other s than the one-fma form's (e.g. (M, n) = (32865, 24): one-fma fails for one s of 132 729, this form for none;
// (44669, 25): the other way round), so a layer takes whichever its (M, n) allows.  Same enumeration.
bool prove_single_requant(unsigned M, unsigned n) {
    if (M == 0 || 3ull * M >= (1ull << 18) || n > 38) return false;
    const float Mf = (float)M, sh = ldexpf(1.0f, -(int)n), Md = Mf * sh, Cd = -(12582912.f * Mf) * sh, Cs = Cd - 128.f;
    if ((double)Cs != (double)Cd - 128.0) return false;                        // the fma's addend must be exact
    const double scale = (double)M * ldexp(1.0, -(int)n);
    long long s_lo = (long long)floor(-2.0 / scale), s_hi = (long long)ceil(258.0 / scale);
    const long long lim = (1ll << 22) - 1;
    if (s_lo < -lim) s_lo = -lim;
    if (s_hi > lim) s_hi = lim;
    auto cvt_u8 = [](float w) { const float r = rintf(w); return r < 0.f ? 0.f : (r > 255.f ? 255.f : r); };
    for (long long s = s_lo; s <= s_hi; ++s) {
        const float y = 12582912.f + (float)s;
        const float t = fmaf(y, Mf, -(12582912.f * Mf));
        const float ref = cvt_u8(fmaf(t, sh, -128.f) + 128.f);
        const float one = cvt_u8(fmaf(y, Md, Cs) + 128.f);
        if (ref != one) return false;
    }
    return true;
}


// Load-time proof that the 18-bit PE clamp and the 20-bit adder clamp can never fire:
// for q in [-128,127] (pad value included) the extreme PE sums are 127*S+ + 128*S- and
// -(128*S+ + 127*S-).  (myQL/quan_func.py:358-370,437 are then identities.)
// risky_oc (optional): bit o = some PE sum of output channel o can leave the accumulator range
bool saturation_free(const sesrq_layer_desc &d, int zc, int acc_bits, int add_bits, long long &worst_pe,
                            long long &worst_sum, int &risky_mask, int *risky_oc) {
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


}  // namespace sesrq
