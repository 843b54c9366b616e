// BatchNorm1d(affine=True) variant of the layer block (use_bn=True), gfx950.
//
// Reference: nn.BatchNorm1d between Linear and ReLU (scarches/models/trvae/modules.py:79,146,161): train mode
// normalises with the (biased) batch statistics and moves the running statistics with momentum 0.1 (unbiased
// variance); eval mode normalises with the running statistics.  Frozen surgery models switch
// `track_running_stats` off (trainers/trvae/trainer.py:264-270): batch statistics are still used in train mode, the
// running buffers stay untouched.
//
// Unlike LayerNorm this is a reduction over the BATCH, so the block is split: the row kernel (layers.hip, norm=0,
// relu=0) materialises the pre-activation, column statistics are reduced with fp64 atomics (one 64-column stripe
// per workgroup column, rows split over workgroups), then a second row pass normalises / activates.
#include "common.h"

namespace {

constexpr int BN_ROWS = 64;   // rows per workgroup in the column reductions

// acc[0][h] += sum_b v1(b,h), acc[1][h] += sum_b v2(b,h)
template <class F>
__device__ __forceinline__ void column_reduce2(int B, int H, double* acc, F f) {
    __shared__ double part[2][4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int h = blockIdx.x * 64 + cx;
    const int r0 = blockIdx.y * BN_ROWS, r1 = min(B, r0 + BN_ROWS);
    double s1 = 0.0, s2 = 0.0;
    if (h < H)
        for (int r = r0 + ry; r < r1; r += 4) {
            float a, b;
            f(r, h, a, b);
            s1 += (double)a;
            s2 += (double)b;
        }
    part[0][ry][cx] = s1;
    part[1][ry][cx] = s2;
    __syncthreads();
    if (ry == 0 && h < H) {
        atomicAdd(&acc[h], part[0][0][cx] + part[0][1][cx] + part[0][2][cx] + part[0][3][cx]);
        atomicAdd(&acc[H + h], part[1][0][cx] + part[1][1][cx] + part[1][2][cx] + part[1][3][cx]);
    }
}

__global__ __launch_bounds__(256) void bn_colstats_kernel(const float* __restrict__ a, int B, int H,
                                                          double* __restrict__ acc) {
    column_reduce2(B, H, acc, [&](int r, int h, float& v1, float& v2) {
        const float x = a[(long long)r * H + h];
        v1 = x;
        v2 = x * x;
    });
}

// stats[0][h] = mean, stats[1][h] = rstd.  use_batch: batch statistics (train mode), else running statistics.
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ acc, int B, int H, int use_batch,
                                                          int update_running, float momentum,
                                                          float* __restrict__ running_mean,
                                                          float* __restrict__ running_var, float* __restrict__ stats) {
    const int h = blockIdx.x * 256 + threadIdx.x;
    if (h >= H) return;
    float mean, var;
    if (use_batch) {
        const double m = acc[h] / B;
        const double v = fmax(acc[H + h] / B - m * m, 0.0);
        mean = (float)m;
        var = (float)v;
        if (update_running) {
            const double vu = B > 1 ? v * (double)B / (double)(B - 1) : v;
            running_mean[h] = (1.0f - momentum) * running_mean[h] + momentum * mean;
            running_var[h] = (1.0f - momentum) * running_var[h] + momentum * (float)vu;
        }
    } else {
        mean = running_mean[h];
        var = running_var[h];
    }
    stats[h] = mean;
    stats[H + h] = 1.0f / sqrtf(var + 1e-5f);
}

// a (in place) <- xhat = (a - mean) * rstd ; h = dropout(relu(xhat * gamma + beta))
__global__ __launch_bounds__(256) void bn_apply_kernel(float* __restrict__ a, int B, int H,
                                                       const float* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float p, const uint8_t* __restrict__ mask, uint64_t seed,
                                                       uint32_t rng_stream, const uint32_t* __restrict__ step,
                                                       float* __restrict__ hout) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * H) return;
    const int b = (int)(i / H), hh = (int)(i - (long long)b * H);
    const float xh = (a[i] - stats[hh]) * stats[H + hh];
    a[i] = xh;
    float y = fmaxf(fmaf(xh, gamma[hh], beta[hh]), 0.f);
    if (p > 0.f) {
        const uint32_t stp = step ? step[0] : 0u;
        const bool keep = mask ? (mask[i] != 0) : dropout_keep(seed + stp, rng_stream, (uint32_t)b, (uint32_t)hh, p);
        y = keep ? y * (1.0f / (1.0f - p)) : 0.f;
    }
    hout[i] = y;
}

struct BnBwdArgs {
    const float* dh; int S; long long slab_stride; const float* extra;
    int B, H;
    const float* xhat; const float* stats; const float* gamma; const float* beta;
    float p; const uint8_t* mask; uint64_t seed; uint32_t rng_stream; const uint32_t* step;
    float* dy;              // [B][H] scratch = output buffer (dy first, then da in place)
    double* acc;            // [2][H]: sum dy, sum dy*xhat
    float* dgamma; float* dbeta;
};

// dy = upstream gradient through dropout and ReLU; column sums of dy and dy * xhat
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const BnBwdArgs a) {
    const int H = a.H;
    const float keep_scale = a.p > 0.f ? 1.0f / (1.0f - a.p) : 1.f;
    const uint32_t stp = a.step ? a.step[0] : 0u;
    column_reduce2(a.B, H, a.acc, [&](int r, int h, float& v1, float& v2) {
        const long long o = (long long)r * H + h;
        float g = 0.f;
        for (int k = 0; k < a.S; ++k) g += a.dh[k * a.slab_stride + o];
        if (a.extra) g += a.extra[o];
        if (a.p > 0.f) {
            const bool keep = a.mask ? (a.mask[o] != 0)
                                     : dropout_keep(a.seed + stp, a.rng_stream, (uint32_t)r, (uint32_t)h, a.p);
            g = keep ? g * keep_scale : 0.f;
        }
        const float xh = a.xhat[o];
        if (!(fmaf(xh, a.gamma[h], a.beta[h]) > 0.f)) g = 0.f;
        a.dy[o] = g;
        v1 = g;
        v2 = g * xh;
    });
}

// da = gamma * rstd * (dy - mean_b(dy) - xhat * mean_b(dy * xhat)) (batch statistics) ; dgamma, dbeta
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BnBwdArgs a, int batch_stats) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int H = a.H;
    if (i >= (long long)a.B * H) return;
    const int hh = (int)(i % H);
    const float s1 = (float)a.acc[hh], s2 = (float)a.acc[H + hh];
    const float g = a.gamma[hh] * a.stats[H + hh];
    const float dy = a.dy[i];
    a.dy[i] = batch_stats ? g * (dy - s1 / (float)a.B - a.xhat[i] * (s2 / (float)a.B)) : g * dy;
    if (i < H) {
        if (a.dgamma) a.dgamma[hh] = s2;
        if (a.dbeta) a.dbeta[hh] = s1;
    }
}

}  // namespace

// a: [B][H] pre-activation (output of sca_layer_fwd with norm=0, relu=0); overwritten with xhat.
extern "C" int sca_bn_fwd(float* a, int B, int H, const float* gamma, const float* beta, float* running_mean,
                          float* running_var, int use_batch_stats, int update_running, float momentum, float p,
                          const uint8_t* mask, uint64_t seed, uint32_t rng_stream, const uint32_t* step, double* acc,
                          float* stats, float* h, void* stream) {
    if (B <= 0 || H <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (use_batch_stats) {
        hipError_t e = hipMemsetAsync(acc, 0, sizeof(double) * 2 * (size_t)H, st);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(bn_colstats_kernel, dim3((H + 63) / 64, (B + BN_ROWS - 1) / BN_ROWS), dim3(256), 0, st, a,
                           B, H, acc);
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((H + 255) / 256), dim3(256), 0, st, acc, B, H, use_batch_stats,
                       update_running, momentum, running_mean, running_var, stats);
    const long long n = (long long)B * H;
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, B, H, stats, gamma,
                       beta, p, mask, seed, rng_stream, step, h);
    SCA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sca_bn_bwd(const float* dh, int S, long long slab_stride, const float* extra, int B, int H,
                          const float* xhat, const float* stats, const float* gamma, const float* beta,
                          int batch_stats, float p, const uint8_t* mask, uint64_t seed, uint32_t rng_stream,
                          const uint32_t* step, double* acc, float* da, float* dgamma, float* dbeta, void* stream) {
    if (B <= 0 || H <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(acc, 0, sizeof(double) * 2 * (size_t)H, st);
    if (e != hipSuccess) return (int)e;
    BnBwdArgs a{dh, S, slab_stride, extra, B, H, xhat, stats, gamma, beta, p, mask, seed, rng_stream, step, da, acc,
                dgamma, dbeta};
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((H + 63) / 64, (B + BN_ROWS - 1) / BN_ROWS), dim3(256), 0, st, a);
    const long long n = (long long)B * H;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, batch_stats);
    SCA_CHECK_LAUNCH();
    return 0;
}
