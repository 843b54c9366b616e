// Optimiser step and loss bookkeeping (gfx950).
//
// Adam exactly as torch.optim.Adam(lr, betas=(0.9, 0.999), eps, weight_decay, amsgrad=False) applies it - the
// configuration of scarches/trainers/trvae/trainer.py:221-223 (eps = 0.01, weight_decay = 0.04, L2 folded into the
// gradient, NOT AdamW) - fused over a flat parameter range, with the optional clip_grad_value_ of trainer.py:277-278.
// Per-step scalars (lr, bias corrections, KL weight) are read from a small device buffer so that a captured HIP graph
// of the whole train step can be replayed without re-recording:
//   hyper[0] = alpha_kl   hyper[1] = lr   hyper[2] = 1 - beta1^t   hyper[3] = sqrt(1 - beta2^t)
//   hyper[4] = adam eps   hyper[5] = clip value (<= 0: off)
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long long n,
                                                   const float* __restrict__ hyper, float beta1, float beta2,
                                                   float weight_decay) {
    const float lr = hyper[1], bc1 = hyper[2], bc2s = hyper[3], eps = hyper[4], clip = hyper[5];
    const float step_size = lr / bc1;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float gi = g[i];
        if (clip > 0.f) gi = fminf(fmaxf(gi, -clip), clip);
        const float pi = p[i];
        gi = fmaf(weight_decay, pi, gi);
        float mi = m[i], vi = v[i];
        mi = mi + (1.0f - beta1) * (gi - mi);           // torch: exp_avg.lerp_(grad, 1 - beta1)
        vi = beta2 * vi + (1.0f - beta2) * gi * gi;
        const float denom = sqrtf(vi) / bc2s + eps;
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - step_size * (mi / denom);
    }
}

// losses[0..3] = recon, kl, mmd, total (= recon + alpha*kl + mmd);  accum[0..3] += w * the same; accum[4] += w, w = hyper[6].
__global__ __launch_bounds__(256) void finalize_losses_kernel(const float* __restrict__ row_recon, float recon_sign,
                                                              const float* __restrict__ row_kl, int B,
                                                              const double* __restrict__ mmd_part, int n_part,
                                                              const float* __restrict__ hyper,
                                                              float* __restrict__ losses, double* __restrict__ accum) {
    __shared__ double red[16];
    double r = 0.0, k = 0.0, mm = 0.0;
    for (int i = threadIdx.x; i < B; i += 256) {
        r += (double)row_recon[i];
        if (row_kl) k += (double)row_kl[i];
    }
    for (int i = threadIdx.x; i < n_part; i += 256) mm += mmd_part[i];
    r = block_sum_d(r, red);
    k = block_sum_d(k, red);
    mm = block_sum_d(mm, red);
    if (threadIdx.x == 0) {
        const double recon = recon_sign * r / (double)B, kl = k / (double)B;
        const double tot = recon + (double)hyper[0] * kl + mm;
        losses[0] = (float)recon; losses[1] = (float)kl; losses[2] = (float)mm; losses[3] = (float)tot;
        if (accum) {
            // the reference logs float32 `.item()` values per iteration and averages them per epoch
            // hyper[6] = weight of this minibatch's means in the epoch log (1; n_local / n_global on a data-parallel
            // rank, so that the all-reduced sums give the means over the GLOBAL minibatches also for ragged shards)
            // (a zero-initialised hyper[8] - valid before the weight existed - reads as weight 1)
            const double w = hyper[6] == 0.f ? 1.0 : (double)hyper[6];
            accum[0] += w * (double)(float)recon; accum[1] += w * (double)(float)kl; accum[2] += w * (double)(float)mm;
            accum[3] += w * (double)(float)tot; accum[4] += w;
        }
    }
}

// Start of an optimisation step: hyper[0..5] <- (alpha_kl, lr, 1-beta1^t, sqrt(1-beta2^t), eps, clip), step_ctr += 1.
__global__ void begin_step_kernel(float* __restrict__ hyper, uint32_t* __restrict__ step_ctr, float a, float lr,
                                  float bc1, float bc2s, float eps, float clip) {
    if (threadIdx.x == 0) {
        hyper[0] = a; hyper[1] = lr; hyper[2] = bc1; hyper[3] = bc2s; hyper[4] = eps; hyper[5] = clip;
        if (step_ctr) step_ctr[0] += 1u;
    }
}

// begin_step + staging of the minibatch ids / conditions into the static buffers a recorded launch plan or HIP graph
// reads (one launch instead of a kernel and two copies at the head of every step).
__global__ __launch_bounds__(256) void begin_step_batch_kernel(float* __restrict__ hyper,
                                                               uint32_t* __restrict__ step_ctr, float a, float lr,
                                                               float bc1, float bc2s, float eps, float clip,
                                                               const int64_t* __restrict__ rows_src,
                                                               int64_t* __restrict__ rows_dst,
                                                               const int32_t* __restrict__ cond_src,
                                                               int32_t* __restrict__ cond_dst, int B,
                                                               const int64_t* __restrict__ next_src,
                                                               int64_t* __restrict__ next_dst, int B_next) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) {
        hyper[0] = a; hyper[1] = lr; hyper[2] = bc1; hyper[3] = bc2s; hyper[4] = eps; hyper[5] = clip;
        if (step_ctr) step_ctr[0] += 1u;
    }
    if (i < B) {
        if (rows_dst) rows_dst[i] = rows_src[i];
        if (cond_dst) cond_dst[i] = cond_src[i];
    }
    if (i < B_next) next_dst[i] = next_src[i];
}

}  // namespace

extern "C" int sca_begin_step_batch_next(float*, uint32_t*, float, float, float, float, float, float, const int64_t*,
                                         int64_t*, const int32_t*, int32_t*, int, const int64_t*, int64_t*, int, void*);
extern "C" int sca_begin_step_batch(float* hyper, uint32_t* step_ctr, float alpha_kl, float lr, float bias_corr1,
                                    float bias_corr2_sqrt, float eps, float clip, const int64_t* rows_src,
                                    int64_t* rows_dst, const int32_t* cond_src, int32_t* cond_dst, int B,
                                    void* stream) {
    return sca_begin_step_batch_next(hyper, step_ctr, alpha_kl, lr, bias_corr1, bias_corr2_sqrt, eps, clip, rows_src,
                                     rows_dst, cond_src, cond_dst, B, nullptr, nullptr, 0, stream);
}

// ... and the row ids of the NEXT minibatch into the static buffer the step's trailing gather reads (the engine gathers
// minibatch n + 1 beside the gradient tail and Adam of step n: CVAEEngine.train_step(next_rows=...)).
extern "C" int sca_begin_step_batch_next(float* hyper, uint32_t* step_ctr, float alpha_kl, float lr, float bias_corr1,
                                         float bias_corr2_sqrt, float eps, float clip, const int64_t* rows_src,
                                         int64_t* rows_dst, const int32_t* cond_src, int32_t* cond_dst, int B,
                                         const int64_t* next_src, int64_t* next_dst, int B_next, void* stream) {
    if (B < 0 || B_next < 0 || (rows_dst && !rows_src) || (cond_dst && !cond_src)) return -1;
    if (B_next > 0 && (!next_src || !next_dst)) return -1;
    const int n = B > B_next ? B : B_next;
    const int nb = n > 0 ? (n + 255) / 256 : 1;
    hipLaunchKernelGGL(begin_step_batch_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, hyper, step_ctr, alpha_kl,
                       lr, bias_corr1, bias_corr2_sqrt, eps, clip, rows_src, rows_dst, cond_src, cond_dst, B, next_src,
                       next_dst, B_next);
    SCA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sca_begin_step(float* hyper, uint32_t* step_ctr, float alpha_kl, float lr, float bias_corr1,
                              float bias_corr2_sqrt, float eps, float clip, void* stream) {
    hipLaunchKernelGGL(begin_step_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, hyper, step_ctr, alpha_kl, lr,
                       bias_corr1, bias_corr2_sqrt, eps, clip);
    SCA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sca_fill_zero(void* p, long long bytes, void* stream) {
    if (bytes <= 0) return 0;
    hipError_t e = hipMemsetAsync(p, 0, (size_t)bytes, (hipStream_t)stream);
    return e == hipSuccess ? 0 : (int)e;
}

extern "C" int sca_adam_step(float* p, const float* g, float* m, float* v, long long n, const float* hyper,
                             float beta1, float beta2, float weight_decay, void* stream) {
    if (n <= 0) return 0;
    long long nb = (n + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, hyper, beta1,
                       beta2, weight_decay);
    SCA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sca_finalize_losses(const float* row_recon, float recon_sign, const float* row_kl, int B,
                                   const double* mmd_part, int n_part, const float* hyper, float* losses,
                                   double* accum, void* stream) {
    hipLaunchKernelGGL(finalize_losses_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, row_recon, recon_sign,
                       row_kl, B, mmd_part, n_part, hyper, losses, accum);
    SCA_CHECK_LAUNCH();
    return 0;
}
