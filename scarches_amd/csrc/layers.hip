// Fused epilogues of the conditional-VAE layers (gfx950): everything between two GEMMs.
//
// Forward of one block  Linear(+condition) -> LayerNorm(no affine) -> ReLU -> Dropout
//   (scarches/models/trvae/modules.py:21-27 CondLayers, :78-84 / :145-151 / :160-166 norm+act+dropout):
//   the GEMM leaves split-K slabs; this kernel sums them, adds the bias and the condition column
//   Wc[:, c_b] (the reference's one-hot GEMM is a column gather), normalises the row, applies ReLU and the
//   dropout keep-mask and writes both the normalised pre-activation (needed by backward) and the output.
// Backward: dh -> da through dropout, ReLU and LayerNorm in one pass.
// Reductions over the batch that the reference gets from autograd of the one-hot GEMM / bias broadcast
//   (dWc[:, c] = sum_{b: c_b = c} da[b, :],  db = sum_b da[b, :]) are a segmented column sum.
// Latent block: mean/log-var heads -> reparameterised sample + KL (models/base/_base.py:391-407,
//   models/trvae/trvae.py:143-147) and its backward.
//
// One wave (64 lanes) per cell row; rows are independent, all reductions are wave shuffles.
#include "common.h"

namespace {

struct LayerFwdArgs {
    const float* slabs; int S; long long slab_stride;   // [S][B][H] GEMM partial sums
    int B, H;
    const float* bias;          // [H] or null
    const float* Wc; int C;     // [H][C] condition weights or null
    const int32_t* cond;        // [B]
    const float* cvec; int E;   // optional dense condition vectors [B][E] with Wc = [H][E] (scPoli embeddings)
    int norm;                   // 0 none, 1 LayerNorm(eps 1e-5, no affine)
    int relu;
    float p;                    // dropout rate (0 = off)
    const uint8_t* mask;        // optional injected keep-mask [B][H]; else Philox(seed, rng_stream, step)
    uint64_t seed; uint32_t rng_stream;
    const uint32_t* step;       // device step counter mixed into the dropout stream (null = 0)
    float* ynorm;               // [B][H] normalised pre-activation (or pre-activation when norm == 0)
    float* rstd;                // [B]
    float* h;                   // [B][H] output
};

template <int EPL>
__global__ __launch_bounds__(256) void layer_fwd_kernel(const LayerFwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= a.B) return;
    const int H = a.H;
    float v[EPL];
    const int c = (a.Wc && !a.cvec) ? a.cond[b] : 0;
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
        const int hh = lane + 64 * j;
        float s = 0.f;
        if (hh < H) {
            const float* p = a.slabs + (long long)b * H + hh;
            for (int k = 0; k < a.S; ++k) s += p[k * a.slab_stride];
            if (a.bias) s += a.bias[hh];
            if (a.Wc) {
                if (a.cvec) {
                    const float* e = a.cvec + (long long)b * a.E;
                    const float* w = a.Wc + (long long)hh * a.E;
                    float d = 0.f;
                    for (int q = 0; q < a.E; ++q) d = fmaf(e[q], w[q], d);
                    s += d;
                } else {
                    s += a.Wc[(long long)hh * a.C + c];
                }
            }
        }
        v[j] = s;
    }
    float rs = 1.f;
    if (a.norm == 1) {
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < EPL; ++j) sum += (lane + 64 * j < H) ? v[j] : 0.f;
        const float mean = wave_sum(sum) / (float)H;
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const float d = (lane + 64 * j < H) ? v[j] - mean : 0.f;
            sq = fmaf(d, d, sq);
        }
        const float var = wave_sum(sq) / (float)H;
        rs = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
        for (int j = 0; j < EPL; ++j) v[j] = (v[j] - mean) * rs;
    }
    if (lane == 0 && a.rstd) a.rstd[b] = rs;
    const float keep_scale = a.p > 0.f ? 1.0f / (1.0f - a.p) : 1.f;
    const uint32_t stp = a.step ? a.step[0] : 0u;
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
        const int hh = lane + 64 * j;
        if (hh >= H) continue;
        const long long o = (long long)b * H + hh;
        if (a.ynorm) a.ynorm[o] = v[j];
        float out = a.relu ? fmaxf(v[j], 0.f) : v[j];
        if (a.p > 0.f) {
            const bool keep = a.mask ? (a.mask[o] != 0)
                                     : dropout_keep(a.seed + stp, a.rng_stream, (uint32_t)b, (uint32_t)hh, a.p);
            out = keep ? out * keep_scale : 0.f;
        }
        a.h[o] = out;
    }
}

struct LayerBwdArgs {
    const float* dh; int S; long long slab_stride;   // [S][B][H] upstream gradient (slabs are summed)
    const float* extra;                               // optional additional upstream gradient [B][H]
    int B, H;
    const float* ynorm; const float* rstd;
    int norm, relu;
    float p; const uint8_t* mask; uint64_t seed; uint32_t rng_stream; const uint32_t* step;
    float* da;                                        // [B][H] gradient w.r.t. the Linear output
};

template <int EPL>
__global__ __launch_bounds__(256) void layer_bwd_kernel(const LayerBwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= a.B) return;
    const int H = a.H;
    float g[EPL], y[EPL];
    const float keep_scale = a.p > 0.f ? 1.0f / (1.0f - a.p) : 1.f;
    const uint32_t stp = a.step ? a.step[0] : 0u;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
        const int hh = lane + 64 * j;
        float gv = 0.f, yv = 0.f;
        if (hh < H) {
            const long long o = (long long)b * H + hh;
            for (int k = 0; k < a.S; ++k) gv += a.dh[k * a.slab_stride + o];
            if (a.extra) gv += a.extra[o];
            yv = a.ynorm[o];
            if (a.p > 0.f) {
                const bool keep = a.mask ? (a.mask[o] != 0)
                                         : dropout_keep(a.seed + stp, a.rng_stream, (uint32_t)b, (uint32_t)hh, a.p);
                gv = keep ? gv * keep_scale : 0.f;
            }
            if (a.relu && !(yv > 0.f)) gv = 0.f;
        }
        g[j] = gv; y[j] = yv;
        s1 += gv;
        s2 = fmaf(gv, yv, s2);
    }
    if (a.norm == 1) {
        const float m1 = wave_sum(s1) / (float)H, m2 = wave_sum(s2) / (float)H;
        const float rs = a.rstd[b];
#pragma unroll
        for (int j = 0; j < EPL; ++j) g[j] = rs * (g[j] - m1 - y[j] * m2);
    }
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
        const int hh = lane + 64 * j;
        if (hh < H) a.da[(long long)b * H + hh] = g[j];
    }
}

// Segmented column sum: segT[c][h] += sum over rows b with cond[b] == c of da[b][h]  (cond == null: one segment).
// A workgroup owns ROWS consecutive rows; a thread owns a column and keeps a running sum for the current
// condition run in a register, flushing with one atomic per (run, column) - coalesced 256-byte atomics.
// With condition-sorted minibatches (the loader emits them that way) that is ~B/ROWS + C flushes per column.
constexpr int COLSUM_ROWS = 32;
__global__ __launch_bounds__(256) void cond_colsum_kernel(const float* __restrict__ da, int B, int H,
                                                          const int32_t* __restrict__ cond, float* __restrict__ segT) {
    const int r0 = blockIdx.x * COLSUM_ROWS, r1 = min(B, r0 + COLSUM_ROWS);
    for (int hh = blockIdx.y * 256 + threadIdx.x; hh < H; hh += gridDim.y * 256) {
        int cur = cond ? cond[r0] : 0;
        float acc = 0.f;
        for (int r = r0; r < r1; ++r) {
            const int c = cond ? cond[r] : 0;
            if (c != cur) {
                atomicAdd(&segT[(long long)cur * H + hh], acc);
                acc = 0.f;
                cur = c;
            }
            acc += da[(long long)r * H + hh];
        }
        atomicAdd(&segT[(long long)cur * H + hh], acc);
    }
}

// dWc[h][c] = segT[c][h];  db[h] = sum_c segT[c][h].
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ segT, int C, int H,
                                                            float* __restrict__ dWc, float* __restrict__ db) {
    const int hh = blockIdx.x * 256 + threadIdx.x;
    if (hh >= H) return;
    float s = 0.f;
    for (int c = 0; c < C; ++c) {
        const float v = segT[(long long)c * H + hh];
        s += v;
        if (dWc) dWc[(long long)hh * C + c] = v;
    }
    if (db) db[hh] = s;
}

struct LatentArgs {
    const float* heads; int S; long long slab_stride;   // [S][B][2Z] = [mean | log_var] GEMM slabs
    const float* bias;                                   // [2Z]
    int B, Z;
    const float* eps;                                    // optional injected noise [B][Z]
    uint64_t seed; uint32_t rng_stream; const uint32_t* step;
    int sample;                                          // 0: z = mean (get_latent(mean=True))
    float* mu_lv;                                        // [B][2Z] saved for backward
    float* eps_out;                                      // [B][Z] noise actually used (saved for backward)
    float* z;                                            // [B][Z]
    float* kl_row;                                       // [B]
};

// One thread per (row, latent dim): coalesced over the [B][2Z] head rows; the row's KL is reduced across the Z
// neighbouring lanes through LDS (a workgroup owns 256 / Zp whole rows, Zp = Z rounded up to a power of two).
__global__ __launch_bounds__(256) void latent_fwd_kernel(const LatentArgs a, int Zp) {
    __shared__ float klp[256];
    const int Z = a.Z;
    const int rows_per_blk = 256 / Zp;
    const int rl = threadIdx.x / Zp, k = threadIdx.x - rl * Zp;
    const int b = blockIdx.x * rows_per_blk + rl;
    const bool act = rl < rows_per_blk && b < a.B && k < Z;
    float kl = 0.f;
    if (act) {
        const uint32_t stp = a.step ? a.step[0] : 0u;
        float mu = a.bias ? a.bias[k] : 0.f, lv = a.bias ? a.bias[Z + k] : 0.f;
        for (int s = 0; s < a.S; ++s) {
            const float* p = a.heads + s * a.slab_stride + (long long)b * 2 * Z;
            mu += p[k];
            lv += p[Z + k];
        }
        const float var = expf(lv) + 1e-4f;
        const float sd = sqrtf(var);
        float e = 0.f;
        if (a.sample) {
            if (a.eps) {
                e = a.eps[(long long)b * Z + k];
            } else {
                float n0, n1;
                philox_normal2(a.seed + stp, a.rng_stream, (uint32_t)(b * ((Z + 1) / 2) + (k >> 1)), n0, n1);
                e = (k & 1) ? n1 : n0;
            }
        }
        const float var_ratio = sd * sd;       // torch kl_normal_normal squares the scale again
        kl = 0.5f * (var_ratio + mu * mu - 1.0f - logf(var_ratio));
        if (a.mu_lv) {
            a.mu_lv[(long long)b * 2 * Z + k] = mu;
            a.mu_lv[(long long)b * 2 * Z + Z + k] = lv;
        }
        if (a.eps_out) a.eps_out[(long long)b * Z + k] = e;
        a.z[(long long)b * Z + k] = fmaf(e, sd, mu);
    }
    if (!a.kl_row) return;
    klp[threadIdx.x] = kl;
    __syncthreads();
    if (act && k == 0) {
        float t = 0.f;
        for (int q = 0; q < Z; ++q) t += klp[rl * Zp + q];      // same order as the per-row loop of the reference sum
        a.kl_row[b] = t;
    }
}

// Wide latents (Z > 256, e.g. expiMap: one latent per gene program): one wave per row, lanes walk the row (coalesced),
// the KL row sum is one wave reduction.
__global__ __launch_bounds__(256) void latent_fwd_row_kernel(const LatentArgs a) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= a.B) return;                  // wave-uniform
    const int Z = a.Z;
    const uint32_t stp = a.step ? a.step[0] : 0u;
    float kl = 0.f;
    for (int k = lane; k < Z; k += 64) {
        float mu = a.bias ? a.bias[k] : 0.f, lv = a.bias ? a.bias[Z + k] : 0.f;
        for (int s = 0; s < a.S; ++s) {
            const float* p = a.heads + s * a.slab_stride + (long long)b * 2 * Z;
            mu += p[k];
            lv += p[Z + k];
        }
        const float var = expf(lv) + 1e-4f;
        const float sd = sqrtf(var);
        float e = 0.f;
        if (a.sample) {
            if (a.eps) {
                e = a.eps[(long long)b * Z + k];
            } else {
                float n0, n1;
                philox_normal2(a.seed + stp, a.rng_stream, (uint32_t)(b * ((Z + 1) / 2) + (k >> 1)), n0, n1);
                e = (k & 1) ? n1 : n0;
            }
        }
        const float var_ratio = sd * sd;
        kl += 0.5f * (var_ratio + mu * mu - 1.0f - logf(var_ratio));
        if (a.mu_lv) {
            a.mu_lv[(long long)b * 2 * Z + k] = mu;
            a.mu_lv[(long long)b * 2 * Z + Z + k] = lv;
        }
        if (a.eps_out) a.eps_out[(long long)b * Z + k] = e;
        a.z[(long long)b * Z + k] = fmaf(e, sd, mu);
    }
    kl = wave_sum(kl);
    if (lane == 0 && a.kl_row) a.kl_row[b] = kl;
}

struct LatentBwdArgs {
    const float* dz; int S; long long slab_stride;     // decoder dX slabs [S][B][Z]
    const float* dz2; int S2; long long slab_stride2;  // optional MMD gradient slabs [S2][B][Z]
    const float* mu_lv; const float* eps;
    const float* hyper;                                // hyper[0] = alpha_kl
    int B, Z;
    float* dheads;                                     // [B][2Z]
};

__global__ __launch_bounds__(256) void latent_bwd_kernel(const LatentBwdArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.B * a.Z) return;
    const int b = i / a.Z, k = i - b * a.Z;
    float g = 0.f;
    for (int s = 0; s < a.S; ++s) g += a.dz[s * a.slab_stride + i];
    for (int s = 0; s < a.S2; ++s) g += a.dz2[s * a.slab_stride2 + i];
    const float mu = a.mu_lv[(long long)b * 2 * a.Z + k], lv = a.mu_lv[(long long)b * 2 * a.Z + a.Z + k];
    const float ev = expf(lv), var = ev + 1e-4f, sd = sqrtf(var);
    const float akl = a.hyper[0] / (float)a.B;
    // z = mu + eps*sd ; KL = 0.5 (var + mu^2 - 1 - log var)
    const float dmu = g + akl * mu;
    const float dlv = g * a.eps[i] * (0.5f * ev / sd) + akl * 0.5f * (1.0f - 1.0f / var) * ev;
    a.dheads[(long long)b * 2 * a.Z + k] = dmu;
    a.dheads[(long long)b * 2 * a.Z + a.Z + k] = dlv;
}

template <typename Args, void (*K2)(Args), void (*K4)(Args), void (*K8)(Args), void (*K16)(Args), void (*K32)(Args)>
int launch_rows(const Args& a, int B, int H, hipStream_t st) {
    if (H > 64 * 32) return -2;
    dim3 grid((B + 3) / 4), block(256);
    if (H <= 128) hipLaunchKernelGGL(K2, grid, block, 0, st, a);
    else if (H <= 256) hipLaunchKernelGGL(K4, grid, block, 0, st, a);
    else if (H <= 512) hipLaunchKernelGGL(K8, grid, block, 0, st, a);
    else if (H <= 1024) hipLaunchKernelGGL(K16, grid, block, 0, st, a);
    else hipLaunchKernelGGL(K32, grid, block, 0, st, a);
    return 0;
}

}  // namespace

extern "C" int sca_layer_fwd(const float* slabs, int S, long long slab_stride, int B, int H, const float* bias,
                             const float* Wc, int C, const int32_t* cond, const float* cvec, int E, int norm, int relu,
                             float p, const uint8_t* mask, uint64_t seed, uint32_t rng_stream, const uint32_t* step,
                             float* ynorm, float* rstd, float* h, void* stream) {
    if (B <= 0) return 0;
    if (norm < 0 || norm > 1) return -3;
    LayerFwdArgs a{slabs, S, slab_stride, B, H, bias, Wc, C, cond, cvec, E, norm, relu, p, mask, seed, rng_stream,
                   step, ynorm, rstd, h};
    const int rc = launch_rows<LayerFwdArgs, layer_fwd_kernel<2>, layer_fwd_kernel<4>, layer_fwd_kernel<8>,
                               layer_fwd_kernel<16>, layer_fwd_kernel<32>>(a, B, H, (hipStream_t)stream);
    if (rc) return rc;
    SCA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sca_layer_bwd(const float* dh, int S, long long slab_stride, const float* extra, int B, int H,
                             const float* ynorm, const float* rstd, int norm, int relu, float p, const uint8_t* mask,
                             uint64_t seed, uint32_t rng_stream, const uint32_t* step, float* da, void* stream) {
    if (B <= 0) return 0;
    if (norm < 0 || norm > 1) return -3;
    LayerBwdArgs a{dh, S, slab_stride, extra, B, H, ynorm, rstd, norm, relu, p, mask, seed, rng_stream, step, da};
    const int rc = launch_rows<LayerBwdArgs, layer_bwd_kernel<2>, layer_bwd_kernel<4>, layer_bwd_kernel<8>,
                               layer_bwd_kernel<16>, layer_bwd_kernel<32>>(a, B, H, (hipStream_t)stream);
    if (rc) return rc;
    SCA_CHECK_LAUNCH();
    return 0;
}

// segT must hold max(C,1)*H floats; it is zeroed here (memset node), filled by atomics, then finished.
extern "C" int sca_cond_colsum(const float* da, int B, int H, const int32_t* cond, int C, float* segT, float* dWc,
                               float* db, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int Cs = (cond && C > 0) ? C : 1;
    hipError_t e = hipMemsetAsync(segT, 0, (size_t)Cs * H * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    if (B > 0) {
        dim3 grid((B + COLSUM_ROWS - 1) / COLSUM_ROWS, (H + 255) / 256);
        hipLaunchKernelGGL(cond_colsum_kernel, grid, dim3(256), 0, st, da, B, H, (cond && C > 0) ? cond : nullptr, segT);
    }
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((H + 255) / 256), dim3(256), 0, st, segT, Cs,
                       H, (cond && C > 0) ? dWc : nullptr, db);
    SCA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sca_latent_fwd(const float* heads, int S, long long slab_stride, const float* bias, int B, int Z,
                              const float* eps, uint64_t seed, uint32_t rng_stream, const uint32_t* step, int sample,
                              float* mu_lv, float* eps_out, float* z, float* kl_row, void* stream) {
    if (B <= 0) return 0;
    LatentArgs a{heads, S, slab_stride, bias, B, Z, eps, seed, rng_stream, step, sample, mu_lv, eps_out, z, kl_row};
    if (Z <= 256) {
        int Zp = 1;
        while (Zp < Z) Zp <<= 1;
        const int rpb = 256 / Zp;
        hipLaunchKernelGGL(latent_fwd_kernel, dim3((B + rpb - 1) / rpb), dim3(256), 0, (hipStream_t)stream, a, Zp);
    } else {
        hipLaunchKernelGGL(latent_fwd_row_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
    }
    SCA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sca_latent_bwd(const float* dz, int S, long long slab_stride, const float* dz2, int S2,
                              long long slab_stride2, const float* mu_lv, const float* eps, const float* hyper, int B,
                              int Z, float* dheads, void* stream) {
    if (B <= 0) return 0;
    LatentBwdArgs a{dz, S, slab_stride, dz2, S2, slab_stride2, mu_lv, eps, hyper, B, Z, dheads};
    hipLaunchKernelGGL(latent_bwd_kernel, dim3((B * Z + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    SCA_CHECK_LAUNCH();
    return 0;
}
