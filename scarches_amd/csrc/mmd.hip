// Fused multi-scale RBF MMD between condition blocks of the minibatch, forward + backward (gfx950).
//
// Reference: scarches/models/trvae/losses.py:166-270 + _utils.py:18-24.  The reference partitions the batch by
// condition and loops over condition pairs, re-evaluating every diagonal block C-1 times:
//     mmd = beta * sum_{pairs (a,b)} [ mean K(a,a) + mean K(b,b) - 2 mean K(a,b) ],
//     K(x,y) = sum_{alpha in 19 bandwidths} exp(-|x-y|^2 / (2 alpha))
//   pairs: a > b with n_a,n_b >= 1 (no boundary) or a < boundary <= b with n_a,n_b >= 2 (losses.py:255-268).
// Here every B x B kernel entry is evaluated ONCE and weighted:
//     mmd = sum_{i,j} Wt[c_i][c_j] * K(z_i, z_j),   Wt[a][b] = beta * Q[a][b] / (n_a n_b)
//   with Q[a][a] = number of pairs containing a, Q[a][b] = -1 for each selected pair (SURVEY a18 identity).
// The gradient d mmd / d z_i = 4 sum_j Wt[c_i][c_j] K'(d2_ij) (z_i - z_j) is accumulated in the same pass
// (K' = -sum_alpha exp(-d2/2alpha) / (2 alpha)), written as per-column-chunk slabs (no atomics).
// The kernel is order-agnostic (any row order); it is exp-throughput bound: 19 v_exp per pair.
#include "common.h"

namespace {

__constant__ float MMD_BETA[19] = {
    // 1 / (2 alpha) in float32, alpha as float32 (the reference builds torch.FloatTensor(alphas))
    1.f / (2.f * 1e-6f), 1.f / (2.f * 1e-5f), 1.f / (2.f * 1e-4f), 1.f / (2.f * 1e-3f), 1.f / (2.f * 1e-2f),
    1.f / (2.f * 1e-1f), 1.f / (2.f * 1.f),   1.f / (2.f * 5.f),   1.f / (2.f * 10.f),  1.f / (2.f * 15.f),
    1.f / (2.f * 20.f),  1.f / (2.f * 25.f),  1.f / (2.f * 30.f),  1.f / (2.f * 35.f),  1.f / (2.f * 100.f),
    1.f / (2.f * 1e3f),  1.f / (2.f * 1e4f),  1.f / (2.f * 1e5f),  1.f / (2.f * 1e6f)};

// Single workgroup: condition counts -> weight table Wt[C][C].
__global__ __launch_bounds__(256) void mmd_prepare_kernel(const int32_t* __restrict__ cond, int B, int C, int boundary,
                                                          float beta, float* __restrict__ Wt,
                                                          int32_t* __restrict__ counts_out) {
    extern __shared__ int cnt[];           // [C] counts, then [C] valid flags
    int* valid = cnt + C;
    __shared__ int nlow, nhigh;
    for (int i = threadIdx.x; i < C; i += 256) cnt[i] = 0;
    if (threadIdx.x == 0) { nlow = 0; nhigh = 0; }
    __syncthreads();
    for (int i = threadIdx.x; i < B; i += 256) {
        const int c = cond[i];
        if (c >= 0 && c < C) atomicAdd(&cnt[c], 1);
    }
    __syncthreads();
    const int need = boundary >= 0 ? 2 : 1;
    for (int i = threadIdx.x; i < C; i += 256) {
        const int v = cnt[i] >= need;
        valid[i] = v;
        if (counts_out) counts_out[i] = cnt[i];
        if (v) {
            if (boundary >= 0 && i >= boundary) atomicAdd(&nhigh, 1);
            else atomicAdd(&nlow, 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * C; i += 256) {
        const int a = i / C, b = i - a * C;
        float q = 0.f;
        if (valid[a] && valid[b]) {
            if (boundary < 0) {
                q = (a == b) ? (float)(nlow - 1) : -1.f;
            } else {
                const bool al = a < boundary, bl = b < boundary;
                if (a == b) q = al ? (float)nhigh : (float)nlow;
                else if (al != bl) q = -1.f;
            }
            q = beta * q / ((float)cnt[a] * (float)cnt[b]);
        }
        Wt[i] = q;
    }
}

// K = sum_a exp(-beta_a d2), Kp = sum_a beta_a exp(-beta_a d2) over the 19 bandwidths: straight-line code,
// one v_mul + one raw v_exp_f32 (2^x; results below 2^-126 flush to 0, i.e. < 1.2e-38 absolute) + two accumulates
// per bandwidth.  (Per-term wave-uniform skip branches were measured slower than the exps they save.)
__device__ __forceinline__ void rbf19(float d2, float& K, float& Kp) {
    constexpr float LOG2E = 1.4426950408889634f;
    K = 0.f; Kp = 0.f;
#pragma unroll
    for (int a = 0; a < 19; ++a) {
        const float beta = MMD_BETA[a];
        const float e = __builtin_amdgcn_exp2f(-(beta * LOG2E) * d2);
        K += e;
        Kp = fmaf(beta, e, Kp);
    }
}

constexpr int MMD_JC = 64;    // columns (j) per workgroup chunk (== wave size: run ends come from one ballot)

// Small feature dimension (latent z): z_i, diff and the gradient accumulator live in registers.
template <int DMAX>
__global__ __launch_bounds__(256) void mmd_small_kernel(const float* __restrict__ z, int ldz,
                                                        const int32_t* __restrict__ cond, int B, int D, int C,
                                                        const float* __restrict__ Wt, double* __restrict__ loss_part,
                                                        float* __restrict__ grad_slabs, long long slab_stride) {
    __shared__ float zj[MMD_JC * DMAX];
    __shared__ int cj[MMD_JC];
    __shared__ int rend[MMD_JC];
    __shared__ double redd[16];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int j0 = blockIdx.y * MMD_JC, j1 = min(B, j0 + MMD_JC);
    for (int t = threadIdx.x; t < (j1 - j0) * D; t += 256) {
        const int jj = t / D, k = t - jj * D;
        zj[jj * DMAX + k] = z[(long long)(j0 + jj) * ldz + k];
    }
    for (int t = threadIdx.x; t < j1 - j0; t += 256) cj[t] = cond[j0 + t];
    __syncthreads();
    if (threadIdx.x < 64) {      // rend[t] = one past the end of t's condition run (MMD_JC == 64: one wave, one ballot)
        const int t = threadIdx.x, n = j1 - j0;
        const bool last = t < n && (t == n - 1 || cj[t + 1] != cj[t]);
        const unsigned long long m = __ballot(last);
        if (t < n) rend[t] = t + __ffsll((long long)(m >> t));
    }
    __syncthreads();

    float zi[DMAX], gr[DMAX];
    const bool act = i < B;
    const int ci = act ? cond[i] : 0;
#pragma unroll
    for (int k = 0; k < DMAX; ++k) {
        zi[k] = (act && k < D) ? z[(long long)i * ldz + k] : 0.f;
        gr[k] = 0.f;
    }
    const bool ci_ok = ci >= 0 && ci < C;
    const float* wrow = Wt + (long long)(ci_ok ? ci : 0) * C;
    float lsum = 0.f;
    if (act && ci_ok) {
        // columns are walked run by run (a run = consecutive columns of one condition; condition-sorted
        // minibatches have <= C runs): the weight is fetched once per run and the inner loop is pure VALU + LDS
        int jj = 0;
        const int nj = j1 - j0;
        while (jj < nj) {
            const int c2 = __builtin_amdgcn_readfirstlane(cj[jj]);
            const int je = __builtin_amdgcn_readfirstlane(rend[jj]);
            const float w = (c2 >= 0 && c2 < C) ? wrow[c2] : 0.f;
            for (; jj < je; ++jj) {
                float diff[DMAX];
                float d2 = 0.f;
#pragma unroll
                for (int k = 0; k < DMAX; ++k) {
                    diff[k] = zi[k] - zj[jj * DMAX + k];     // k >= D: LDS pad is never read into the result
                    if (k < D) d2 = fmaf(diff[k], diff[k], d2);
                }
                float K, Kp;
                rbf19(d2, K, Kp);
                lsum = fmaf(w, K, lsum);
                const float coef = -4.0f * w * Kp;
#pragma unroll
                for (int k = 0; k < DMAX; ++k)
                    if (k < D) gr[k] = fmaf(coef, diff[k], gr[k]);
            }
        }
    }
    const double tot = block_sum_d((double)lsum, redd);
    if (threadIdx.x == 0) loss_part[blockIdx.y * gridDim.x + blockIdx.x] = tot;
    if (act && grad_slabs) {
        float* g = grad_slabs + blockIdx.y * slab_stride + (long long)i * D;
#pragma unroll
        for (int k = 0; k < DMAX; ++k)
            if (k < D) g[k] = gr[k];
    }
}

// Any feature dimension (MMD on the first decoder layer, mmd_on='y'): 16 x 16 pair tiles through LDS.
// A workgroup owns 16 rows i and a chunk of columns j; the gradient for its rows is accumulated in registers.
constexpr int MMD_GD = 64;    // feature chunk staged per pass
__global__ __launch_bounds__(256) void mmd_generic_kernel(const float* __restrict__ y, int ldy,
                                                          const int32_t* __restrict__ cond, int B, int D, int C,
                                                          const float* __restrict__ Wt, int jchunk,
                                                          double* __restrict__ loss_part,
                                                          float* __restrict__ grad_slabs, long long slab_stride) {
    extern __shared__ float sm[];          // yi [16][D] | yj [16][D] | coef [16][17]
    float* yi = sm;
    float* yj = sm + 16 * D;
    float* coef = sm + 32 * D;
    __shared__ double redd[16];
    const int t = threadIdx.x, ti = t >> 4, tj = t & 15;
    const int i0 = blockIdx.x * 16;
    const int jbeg = blockIdx.y * jchunk, jend = min(B, jbeg + jchunk);
    for (int q = t; q < 16 * D; q += 256) {
        const int r = q / D, k = q - r * D;
        yi[q] = (i0 + r < B) ? y[(long long)(i0 + r) * ldy + k] : 0.f;
    }
    const int gi = i0 + ti;
    const int ci = gi < B ? cond[gi] : -1;
    // gradient accumulators: thread owns row ti (of 16) and features k = tj + 16*m
    constexpr int MAXM = 64;               // D <= 1024
    float gacc[MAXM];
#pragma unroll
    for (int m = 0; m < MAXM; ++m) gacc[m] = 0.f;
    float lsum = 0.f;
    for (int jt = jbeg; jt < jend; jt += 16) {
        __syncthreads();
        for (int q = t; q < 16 * D; q += 256) {
            const int r = q / D, k = q - r * D;
            yj[q] = (jt + r < jend) ? y[(long long)(jt + r) * ldy + k] : 0.f;
        }
        __syncthreads();
        const int gj = jt + tj;
        float d2 = 0.f;
        for (int k = 0; k < D; ++k) {
            const float d = yi[ti * D + k] - yj[tj * D + k];
            d2 = fmaf(d, d, d2);
        }
        float cf = 0.f;
        if (gi < B && gj < jend && ci >= 0 && ci < C) {
            const int c2 = cond[gj];
            const float w = (c2 >= 0 && c2 < C) ? Wt[(long long)ci * C + c2] : 0.f;
            float K, Kp;
            rbf19(d2, K, Kp);
            lsum = fmaf(w, K, lsum);
            cf = -4.0f * w * Kp;
        }
        coef[ti * 17 + tj] = cf;
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MAXM; ++m) {
            const int k = tj + 16 * m;
            if (k < D) {
                float s = 0.f;
                const float a0 = yi[ti * D + k];
#pragma unroll
                for (int q = 0; q < 16; ++q) s = fmaf(coef[ti * 17 + q], a0 - yj[q * D + k], s);
                gacc[m] += s;
            }
        }
    }
    const double tot = block_sum_d((double)lsum, redd);
    if (t == 0) loss_part[blockIdx.y * gridDim.x + blockIdx.x] = tot;
    if (gi < B && grad_slabs) {
        float* g = grad_slabs + blockIdx.y * slab_stride + (long long)gi * D;
#pragma unroll
        for (int m = 0; m < MAXM; ++m) {
            const int k = tj + 16 * m;
            if (k < D) g[k] = gacc[m];
        }
    }
}

}  // namespace

// Number of gradient slabs / loss partials the main kernel will produce for (B, D).
extern "C" int sca_mmd_num_slabs(int B, int D) {
    if (D <= 16) return (B + MMD_JC - 1) / MMD_JC;
    // generic kernel: aim for >= 512 workgroups
    const int itiles = (B + 15) / 16;
    int chunks = (512 + itiles - 1) / itiles;
    const int jt = (B + 15) / 16;
    if (chunks > jt) chunks = jt;
    if (chunks < 1) chunks = 1;
    return chunks;
}
extern "C" int sca_mmd_num_partials(int B, int D) {
    if (D <= 16) return sca_mmd_num_slabs(B, D) * ((B + 255) / 256);
    return sca_mmd_num_slabs(B, D) * ((B + 15) / 16);
}

// Wt: [C][C] floats; counts: [C] int32 (optional); loss_part: sca_mmd_num_partials doubles;
// grad_slabs: sca_mmd_num_slabs x [B][D] floats (slab_stride elements apart), may be null (forward only).
extern "C" int sca_mmd_fwd_bwd(const float* feat, int ld, const int32_t* cond, int B, int D, int C, int boundary,
                               float beta, float* Wt, int32_t* counts, double* loss_part, float* grad_slabs,
                               long long slab_stride, void* stream) {
    if (B <= 0 || C <= 0) return 0;
    if (C > 4096 || D > 1024) return -2;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(mmd_prepare_kernel, dim3(1), dim3(256), 2 * C * sizeof(int), st, cond, B, C, boundary, beta, Wt,
                       counts);
    const int ns = sca_mmd_num_slabs(B, D);
    if (D <= 16) {
        dim3 grid((B + 255) / 256, ns);
        if (D == 10)        // the default latent width: exact-size instantiation, all k < D guards fold away
            hipLaunchKernelGGL((mmd_small_kernel<10>), grid, dim3(256), 0, st, feat, ld, cond, B, D, C, Wt, loss_part,
                               grad_slabs, slab_stride);
        else
            hipLaunchKernelGGL((mmd_small_kernel<16>), grid, dim3(256), 0, st, feat, ld, cond, B, D, C, Wt, loss_part,
                               grad_slabs, slab_stride);
    } else {
        const int jt = (B + 15) / 16;
        const int jchunk = ((jt + ns - 1) / ns) * 16;
        dim3 grid((B + 15) / 16, ns);
        const size_t shm = (size_t)(32 * D + 16 * 17) * sizeof(float);
        if (shm > 48 * 1024)
            hipFuncSetAttribute((const void*)mmd_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        hipLaunchKernelGGL(mmd_generic_kernel, grid, dim3(256), shm, st, feat, ld, cond, B, D, C, Wt, jchunk, loss_part,
                           grad_slabs, slab_stride);
    }
    SCA_CHECK_LAUNCH();
    return 0;
}
