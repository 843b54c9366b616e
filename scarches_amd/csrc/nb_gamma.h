// Gamma-function terms of the negative-binomial likelihood for one stored count (shared by recon_loss.hip and
// out_nb_fused.hip; scarches/models/trvae/losses.py:93-100: lgamma(x+theta) - lgamma(theta) - lgamma(x+1) and, for the
// theta gradient, digamma(x+theta) - digamma(theta)).
#pragma once
#include "common.h"

namespace {
// ---- gamma-function terms of one stored count ---------------------------------------------------------------
//   lgam = lgamma(x+theta) - lgamma(theta) - lgamma(x+1),   dig = digamma(x+theta) - digamma(theta)
// Heavy path (any real x > 0): fp64 Stirling series on arguments shifted to >= 8 by the 8-term recurrence; the
// three shift products collapse into ONE log of their ratio and log(w) is shared between lgamma and digamma.
struct ShiftD { double w, p, rs; };          // shifted argument, product y(y+1)..(y+7) (1 if unshifted), sum 1/(y+k)
__device__ __forceinline__ ShiftD shift8(double y, bool want_rs) {
    ShiftD s{y, 1.0, 0.0};
    if (y < 8.0) {
        double num = 0.0, den = 1.0;          // sum_k 1/(y+k) = num/den, built without divisions
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const double f = y + (double)k;
            num = fma(num, f, den);
            den *= f;
        }
        s.p = den;
        s.w = y + 8.0;
        if (want_rs) s.rs = num / den;
    }
    return s;
}
__device__ __forceinline__ double stirling_lgamma(double w, double lw, double iw) {   // w >= 8
    const double i2 = iw * iw;
    const double ser = iw * (1.0 / 12.0 + i2 * (-1.0 / 360.0 + i2 * (1.0 / 1260.0 + i2 * (-1.0 / 1680.0))));
    return (w - 0.5) * lw - w + 0.91893853320467274178 + ser;
}
__device__ __forceinline__ double asym_digamma(double lw, double iw) {                  // w >= 8
    const double i2 = iw * iw;
    return lw - 0.5 * iw - i2 * (1.0 / 12.0 - i2 * (1.0 / 120.0 - i2 * (1.0 / 252.0 - i2 * (1.0 / 240.0))));
}
__device__ __noinline__ void nb_gamma_terms_heavy(float x, float th, float* lgam, float* dig) {
    const double xd = (double)x, td = (double)th;
    const ShiftD a = shift8(xd + td, true), b = shift8(td, true), c = shift8(xd + 1.0, false);
    const double la = log(a.w), lb = log(b.w), lc = log(c.w);
    const double ia = 1.0 / a.w, ib = 1.0 / b.w, ic = 1.0 / c.w;
    const double lg = stirling_lgamma(a.w, la, ia) - stirling_lgamma(b.w, lb, ib) - stirling_lgamma(c.w, lc, ic)
                      - log(a.p / (b.p * c.p));
    const double dg = (asym_digamma(la, ia) - a.rs) - (asym_digamma(lb, ib) - b.rs);
    *lgam = (float)lg;
    *dig = (float)dg;
}
// Fast path: integer count x <= 16 (the bulk of scRNA-seq counts) and moderate theta:
//   lgam = log prod_{k<x} (theta+k)/(k+1),  dig = sum_{k<x} 1/(theta+k)   (exact identities, fp32, no cancellation)
__device__ __forceinline__ bool nb_gamma_fast_ok(float x, float th) {
    return x <= 16.0f && x == floorf(x) && th < 128.0f && th > 1e-30f;
}

}  // namespace
