// scPoli-specific device work (gfx950): condition embeddings, prototype losses, per-class latent moments.
//
//   * embeddings   nn.Embedding(n_i, E_i, max_norm) lookups hstacked into one dense condition vector per cell
//                  (scarches/models/scpoli/scpoli.py:99-101,137).  The lookup renormalises the TOUCHED rows in place
//                  (L2 norm > max_norm -> row *= max_norm / (norm + 1e-7)); the gradient is a scatter-add.
//   * prototypes   prototype_labeled_loss / prototype_unlabeled_loss (trainers/scpoli/trainer.py:721-773) with their
//                  gradients w.r.t. the latent rows (and, for the epoch-end update, the unlabeled prototypes).
//   * moments      per-class count / sum z / sum z z^T in fp64 for update_labeled_prototypes (trainer.py:667-719,
//                  cov: trainers/scpoli/_utils.py:237-282); the three arrays are what DP ranks all-reduce.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void embed_mark_kernel(const int32_t* __restrict__ idx, int stride, int B, int n,
                                                         uint8_t* __restrict__ flags) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const int r = idx[(long long)b * stride];
    if (r >= 0 && r < n) flags[r] = 1;      // same-value byte stores: benign
}

// one wave per embedding row: renormalise if it was touched, clear the flag
__global__ __launch_bounds__(256) void embed_renorm_kernel(float* __restrict__ W, int n, int E, float max_norm,
                                                           uint8_t* __restrict__ flags) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n || !flags[r]) return;
    float ss = 0.f;
    for (int e = lane; e < E; e += 64) { const float w = W[(long long)r * E + e]; ss = fmaf(w, w, ss); }
    const float norm = sqrtf(wave_sum(ss));
    if (norm > max_norm) {
        const float sc = max_norm / (norm + 1e-7f);
        for (int e = lane; e < E; e += 64) W[(long long)r * E + e] *= sc;
    }
    if (lane == 0) flags[r] = 0;
}

__global__ __launch_bounds__(256) void embed_gather_kernel(const float* __restrict__ W, int n, int E,
                                                           const int32_t* __restrict__ idx, int stride, int B,
                                                           float* __restrict__ out, int ldo) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * E) return;
    const int b = (int)(i / E), e = (int)(i - (long long)b * E);
    const int r = idx[(long long)b * stride];
    out[(long long)b * ldo + e] = (r >= 0 && r < n) ? W[(long long)r * E + e] : 0.f;
}

__global__ __launch_bounds__(256) void embed_scatter_kernel(const float* __restrict__ ga, const float* __restrict__ gb,
                                                            int ldg, int B, int E, int n,
                                                            const int32_t* __restrict__ idx, int stride,
                                                            float* __restrict__ dW) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * E) return;
    const int b = (int)(i / E), e = (int)(i - (long long)b * E);
    const int r = idx[(long long)b * stride];
    if (r < 0 || r >= n) return;
    float g = 0.f;
    if (ga) g += ga[(long long)b * ldg + e];
    if (gb) g += gb[(long long)b * ldg + e];
    atomicAdd(&dW[(long long)r * E + e], g);
}

// ------------------------------------------------------------------------------------------------- prototypes
struct ProtoArgs {
    const float* z; int ld; int B, Z;
    const float* P; int K;                       // labeled prototypes [K][Z] (may be null)
    const int32_t* labels; int n_levels;         // [B][n_levels], -1 = unknown
    const uint8_t* labeled;                      // [B] (null = every row labeled)
    const float* Pu; int Ku;                     // unlabeled prototypes [Ku][Z] (may be null)
    int* cnt;                                    // [K + Ku]
    int* assign; float* mind;                    // [B]
    float* rl; float* ru;                        // [B] per-row loss contributions
    float wl, wu;                                // gradient weights (eta, eta * unlabeled_weight)
    float* dz; int lddz;                         // [B][Z] written (not accumulated); may be null
    float* dPu;                                  // [Ku][Z] accumulated (pre-zeroed); may be null
};

__global__ __launch_bounds__(256) void proto_count_kernel(const ProtoArgs a) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= a.B) return;
    const float* zb = a.z + (long long)b * a.ld;
    if (a.P && (!a.labeled || a.labeled[b]))
        for (int l = 0; l < a.n_levels; ++l) {
            const int v = a.labels[(long long)b * a.n_levels + l];
            if (v >= 0 && v < a.K) atomicAdd(&a.cnt[v], 1);
        }
    if (a.Pu) {
        float best = INFINITY; int bk = 0;
        for (int k = 0; k < a.Ku; ++k) {
            float d = 0.f;
            for (int q = 0; q < a.Z; ++q) { const float t = zb[q] - a.Pu[(long long)k * a.Z + q]; d = fmaf(t, t, d); }
            if (d < best) { best = d; bk = k; }          // first minimum wins, as torch.min
        }
        a.assign[b] = bk;
        a.mind[b] = sqrtf(best);
        atomicAdd(&a.cnt[a.K + bk], 1);
    }
}

__global__ __launch_bounds__(256) void proto_grad_kernel(const ProtoArgs a) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= a.B) return;
    const float* zb = a.z + (long long)b * a.ld;
    float* dzb = a.dz ? a.dz + (long long)b * a.lddz : nullptr;
    if (dzb) for (int q = 0; q < a.Z; ++q) dzb[q] = 0.f;
    float rl = 0.f, ru = 0.f;
    if (a.P && (!a.labeled || a.labeled[b]))
        for (int l = 0; l < a.n_levels; ++l) {
            const int v = a.labels[(long long)b * a.n_levels + l];
            if (v < 0 || v >= a.K) continue;
            const float* p = a.P + (long long)v * a.Z;
            float d2 = 0.f;
            for (int q = 0; q < a.Z; ++q) { const float t = zb[q] - p[q]; d2 = fmaf(t, t, d2); }
            const float d = sqrtf(d2), inv_n = 1.0f / (float)a.cnt[v];
            rl += d * inv_n;
            if (dzb && d > 0.f) {
                const float c = a.wl * inv_n / d;
                for (int q = 0; q < a.Z; ++q) dzb[q] = fmaf(c, zb[q] - p[q], dzb[q]);
            }
        }
    if (a.Pu) {
        int M = 0;
        for (int k = 0; k < a.Ku; ++k) M += a.cnt[a.K + k] > 0;
        const int k = a.assign[b];
        const float d = a.mind[b];
        const float s = 1.0f / ((float)a.cnt[a.K + k] * (float)M);
        ru = d * s;
        if (d > 0.f && (dzb || a.dPu)) {
            const float* p = a.Pu + (long long)k * a.Z;
            for (int q = 0; q < a.Z; ++q) {
                const float g = (zb[q] - p[q]) / d * s;
                if (dzb) dzb[q] = fmaf(a.wu, g, dzb[q]);
                if (a.dPu) atomicAdd(&a.dPu[(long long)k * a.Z + q], -g);
            }
        }
    }
    a.rl[b] = rl;
    a.ru[b] = ru;
}

__global__ __launch_bounds__(256) void proto_finish_kernel(const float* __restrict__ rl, const float* __restrict__ ru,
                                                           int B, float wl, float wu, double* __restrict__ out_w,
                                                           float* __restrict__ raw, int accumulate_raw) {
    __shared__ double red[4][2];
    double l = 0, u = 0;
    for (int i = threadIdx.x; i < B; i += 256) { l += rl[i]; u += ru[i]; }
    for (int o = 32; o > 0; o >>= 1) { l += __shfl_down(l, o); u += __shfl_down(u, o); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = l; red[threadIdx.x >> 6][1] = u; }
    __syncthreads();
    if (threadIdx.x == 0) {
        l = red[0][0] + red[1][0] + red[2][0] + red[3][0];
        u = red[0][1] + red[1][1] + red[2][1] + red[3][1];
        if (out_w) out_w[0] = wl * l + wu * u;
        if (raw) {
            raw[0] = (accumulate_raw ? raw[0] : 0.f) + (float)l;
            raw[1] = (accumulate_raw ? raw[1] : 0.f) + (float)u;
        }
    }
}

// ------------------------------------------------------------------------------------------------- classify
// scpoli.classify (models/scpoli/scpoli.py:270-315, p = 2): distances of every latent row to the prototypes of the
// current hierarchy level (classes [Kc] = rows of P), optionally softmax(-d) over the level, and the row's winner
// (first minimum distance / first maximum probability, as torch.min / torch.max).  One thread per row; the level's
// prototypes are staged in LDS.  The same distance arithmetic as the prototype losses (sum of squared differences).
__global__ __launch_bounds__(256) void proto_classify_kernel(const float* __restrict__ z, int ld, int B, int Z,
                                                             const float* __restrict__ P,
                                                             const int32_t* __restrict__ classes, int Kc,
                                                             int get_prob, float* __restrict__ dists,
                                                             int32_t* __restrict__ pred, float* __restrict__ uncert) {
    extern __shared__ float pl[];                 // [Kc][Z]
    for (int i = threadIdx.x; i < Kc * Z; i += 256) pl[i] = P[(long long)classes[i / Z] * Z + (i % Z)];
    __syncthreads();
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float* zb = z + (long long)b * ld;
    float* db = dists + (long long)b * Kc;
    float best = INFINITY;
    int bk = 0;
    for (int k = 0; k < Kc; ++k) {
        float d2 = 0.f;
        for (int q = 0; q < Z; ++q) { const float t = zb[q] - pl[k * Z + q]; d2 = fmaf(t, t, d2); }
        const float d = sqrtf(d2);
        db[k] = d;
        if (d < best) { best = d; bk = k; }
    }
    if (get_prob) {
        // softmax(-d): the largest probability belongs to the smallest distance (first one on ties)
        float se = 0.f;
        for (int k = 0; k < Kc; ++k) se += expf(best - db[k]);
        const float inv = 1.0f / se;
        float pb = -1.f;
        for (int k = 0; k < Kc; ++k) {
            const float pr = expf(best - db[k]) * inv;
            db[k] = pr;
            if (pr > pb) { pb = pr; bk = k; }
        }
        best = pb;
    }
    pred[b] = classes[bk];
    uncert[b] = best;
}

// ------------------------------------------------------------------------------------------------- moments
__global__ __launch_bounds__(256) void class_moments_kernel(const float* __restrict__ z, int ld, int N, int Z,
                                                            const int32_t* __restrict__ labels, int n_levels,
                                                            const uint8_t* __restrict__ sel, int K,
                                                            double* __restrict__ cnt, double* __restrict__ sum,
                                                            double* __restrict__ outer) {
    // one wave per row; lanes cover the Z x Z outer product
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= N || (sel && !sel[b])) return;
    const float* zb = z + (long long)b * ld;
    for (int l = 0; l < n_levels; ++l) {
        const int v = labels[(long long)b * n_levels + l];
        if (v < 0 || v >= K) continue;
        if (lane == 0) atomicAdd(&cnt[v], 1.0);
        for (int q = lane; q < Z; q += 64) atomicAdd(&sum[(long long)v * Z + q], (double)zb[q]);
        for (int q = lane; q < Z * Z; q += 64)
            atomicAdd(&outer[(long long)v * Z * Z + q], (double)zb[q / Z] * (double)zb[q % Z]);
    }
}

__global__ __launch_bounds__(256) void class_moments_finish_kernel(const double* __restrict__ cnt,
                                                                   const double* __restrict__ sum,
                                                                   const double* __restrict__ outer, int K, int Z,
                                                                   const float* __restrict__ prev_mean,
                                                                   const float* __restrict__ prev_cov,
                                                                   const uint8_t* __restrict__ update,
                                                                   float* __restrict__ mean, float* __restrict__ cov) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= K * Z * Z) return;
    const int v = i / (Z * Z), r = (i / Z) % Z, c = i % Z;
    const double n = cnt[v];
    const bool upd = n > 0 && (!update || update[v]);
    if (upd) {
        const double mr = sum[v * Z + r] / n, mc = sum[v * Z + c] / n;
        cov[i] = (float)((outer[i] - n * mr * mc) / (n - 1.0));
        if (c == 0) mean[v * Z + r] = (float)mr;
    } else {
        cov[i] = prev_cov ? prev_cov[i] : 0.f;
        if (c == 0) mean[v * Z + r] = prev_mean ? prev_mean[v * Z + r] : 0.f;
    }
}

}  // namespace

extern "C" int sca_embed_lookup(float* W, int n, int E, const int32_t* idx, int stride, int B, float max_norm,
                                uint8_t* flags, float* out, int ldo, void* stream) {
    if (B <= 0 || E <= 0) return 0;
    if (ldo < E || stride < 1) return -1;
    hipStream_t st = (hipStream_t)stream;
    if (max_norm > 0.f) {
        if (!flags) return -1;
        hipLaunchKernelGGL(embed_mark_kernel, dim3((B + 255) / 256), dim3(256), 0, st, idx, stride, B, n, flags);
        hipLaunchKernelGGL(embed_renorm_kernel, dim3((n + 3) / 4), dim3(256), 0, st, W, n, E, max_norm, flags);
    }
    const long long tot = (long long)B * E;
    hipLaunchKernelGGL(embed_gather_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, W, n, E, idx, stride,
                       B, out, ldo);
    SCA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sca_embed_scatter_add(const float* ga, const float* gb, int ldg, int B, int E, int n,
                                     const int32_t* idx, int stride, float* dW, void* stream) {
    if (B <= 0 || E <= 0) return 0;
    if (ldg < E || stride < 1) return -1;
    const long long tot = (long long)B * E;
    hipLaunchKernelGGL(embed_scatter_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ga,
                       gb, ldg, B, E, n, idx, stride, dW);
    SCA_CHECK_LAUNCH();
    return 0;
}

extern "C" long long sca_proto_workspace_bytes(int B, int K, int Ku) {
    return 4LL * ((long long)K + Ku + 4LL * B + 16);
}

extern "C" int sca_proto_loss(const float* z, int ld, int B, int Z, const float* P, int K, const int32_t* labels,
                              int n_levels, const uint8_t* labeled, float weight, const float* Pu, int Ku,
                              float weight_u, void* work, double* loss_weighted, float* raw, int accumulate_raw,
                              float* dz, int lddz, float* dPu, void* stream) {
    if (B <= 0) return 0;
    if (ld < Z || (dz && lddz < Z) || (P && !labels)) return -1;
    hipStream_t st = (hipStream_t)stream;
    if (!P) K = 0;
    if (!Pu) Ku = 0;
    int* cnt = (int*)work;
    int* assign = cnt + K + Ku;
    float* mind = (float*)(assign + B);
    float* rl = mind + B;
    float* ru = rl + B;
    hipError_t e = hipMemsetAsync(cnt, 0, sizeof(int) * (size_t)(K + Ku), st);
    if (e != hipSuccess) return (int)e;
    ProtoArgs a{z, ld, B, Z, P, K, labels, n_levels, labeled, Pu, Ku, cnt, assign, mind, rl, ru, weight, weight_u,
                dz, lddz, dPu};
    hipLaunchKernelGGL(proto_count_kernel, dim3((B + 255) / 256), dim3(256), 0, st, a);
    hipLaunchKernelGGL(proto_grad_kernel, dim3((B + 255) / 256), dim3(256), 0, st, a);
    hipLaunchKernelGGL(proto_finish_kernel, dim3(1), dim3(256), 0, st, rl, ru, B, weight, weight_u, loss_weighted, raw,
                       accumulate_raw);
    SCA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sca_proto_classify(const float* z, int ld, int B, int Z, const float* P, int K,
                                  const int32_t* classes, int Kc, int get_prob, float* dists, int32_t* pred,
                                  float* uncert, void* stream) {
    if (B <= 0) return 0;
    if (ld < Z || Z <= 0 || Kc <= 0 || K <= 0 || !P || !classes) return -1;
    const size_t shm = sizeof(float) * (size_t)Kc * Z;
    if (shm > 48 * 1024) return -2;
    hipLaunchKernelGGL(proto_classify_kernel, dim3((B + 255) / 256), dim3(256), shm, (hipStream_t)stream, z, ld, B, Z, P,
                       classes, Kc, get_prob, dists, pred, uncert);
    SCA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sca_class_moments(const float* z, int ld, int N, int Z, const int32_t* labels, int n_levels,
                                 const uint8_t* sel, int K, double* cnt, double* sum, double* outer, void* stream) {
    if (N <= 0 || K <= 0) return 0;
    if (ld < Z) return -1;
    hipLaunchKernelGGL(class_moments_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, z, ld, N, Z, labels,
                       n_levels, sel, K, cnt, sum, outer);
    SCA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sca_class_moments_finish(const double* cnt, const double* sum, const double* outer, int K, int Z,
                                        const float* prev_mean, const float* prev_cov, const uint8_t* update,
                                        float* mean, float* cov, void* stream) {
    if (K <= 0) return 0;
    const int n = K * Z * Z;
    hipLaunchKernelGGL(class_moments_finish_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, cnt, sum,
                       outer, K, Z, prev_mean, prev_cov, update, mean, cov);
    SCA_CHECK_LAUNCH();
    return 0;
}
