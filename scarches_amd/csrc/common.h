// Device-side helpers shared by the scarches_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SCA_WAVE 64

#define SCA_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return (int)e__;              \
    } while (0)

// ----------------------------------------------------------------------------- reductions
// Wave-wide (64 lanes) reductions, result in every lane.  Within a row of 16 lanes: four DPP steps (quad_perm xor 1,
// xor 2, row_half_mirror, row_mirror - no LDS round trip, no lane-address arithmetic); across the four rows: the row
// results through v_readlane.  (The __shfl_xor butterfly this replaces is six ds_bpermute round trips plus five address
// instructions each: ~40 instructions and ~500 cycles of LDS latency per reduction - a third of the likelihood kernel's
// per-row time went there.)
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f32<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(v);     // row_half_mirror
    v += dpp_f32<0x140>(v);     // row_mirror
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_f32<0xB1>(v));
    v = fmaxf(v, dpp_f32<0x4E>(v));
    v = fmaxf(v, dpp_f32<0x141>(v));
    v = fmaxf(v, dpp_f32<0x140>(v));
    return v;
}
__device__ __forceinline__ float lane_f32(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    return (lane_f32(v, 0) + lane_f32(v, 16)) + (lane_f32(v, 32) + lane_f32(v, 48));
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    v = row16_max(v);
    return fmaxf(fmaxf(lane_f32(v, 0), lane_f32(v, 16)), fmaxf(lane_f32(v, 32), lane_f32(v, 48)));
}
// log through v_log_f32 (log2, 1 ulp) and one multiply: __logf() expands to the 12-instruction denormal-safe, fma-
// corrected form, which the arguments of the likelihood kernels (>= 1e-8, never denormal) do not need
__device__ __forceinline__ float fast_logf(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994530942f; }

// Block-wide reductions for blockDim.x <= 1024 (<= 16 waves). `scratch` needs 16 entries.
// All threads get the result.  Contains two barriers; safe to call repeatedly with the same scratch.
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += scratch[i];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* scratch) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    float r = scratch[0];
    for (int i = 1; i < nw; ++i) r = fmaxf(r, scratch[i]);
    return r;
}
__device__ __forceinline__ double block_sum_d(double v, double* scratch) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum_d(v);
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    double r = 0.0;
    for (int i = 0; i < nw; ++i) r += scratch[i];
    return r;
}

// ----------------------------------------------------------------------------- Philox4x32-10
struct Philox4 {
    uint32_t x, y, z, w;
};
__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                 uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    return Philox4{c0, c1, c2, c3};
}
__device__ __forceinline__ float u32_to_unit(uint32_t u) {   // (0,1]
    return ((float)(u >> 8) + 1.0f) * (1.0f / 16777216.0f);
}
// keep-mask bit for dropout: stream id = layer tag, element = row*width + col
__device__ __forceinline__ bool dropout_keep(uint64_t seed, uint32_t stream, uint32_t row, uint32_t col, float p) {
    const Philox4 r = philox4x32_10(row, col >> 2, stream, 0x5ca1ab1eu, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint32_t u = (col & 3) == 0 ? r.x : (col & 3) == 1 ? r.y : (col & 3) == 2 ? r.z : r.w;
    return u32_to_unit(u) > p;
}
// standard normal (Box-Muller), two per call
__device__ __forceinline__ void philox_normal2(uint64_t seed, uint32_t stream, uint32_t idx, float& a, float& b) {
    const Philox4 r = philox4x32_10(idx, 0u, stream, 0x0badcafeu, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float u1 = u32_to_unit(r.x), u2 = u32_to_unit(r.y);
    const float rad = sqrtf(-2.0f * logf(u1));
    float s, c;
    sincosf(6.283185307179586f * u2, &s, &c);
    a = rad * c;
    b = rad * s;
}

// ----------------------------------------------------------------------------- special functions (f64, nnz-only work)
// digamma(x) for x > 0: upward recurrence to x >= 6, then the asymptotic series.
__device__ __forceinline__ double digamma_pos(double x) {
    double r = 0.0;
    while (x < 6.0) {
        r -= 1.0 / x;
        x += 1.0;
    }
    const double f = 1.0 / (x * x);
    const double t = f * (-1.0 / 12.0 + f * (1.0 / 120.0 + f * (-1.0 / 252.0 + f * (1.0 / 240.0 + f * (-1.0 / 132.0)))));
    return r + log(x) - 0.5 / x + t;
}
