// expiMap-specific device work (gfx950): masked decoder weights, proximal operators, HSIC.
//
//   * masked copy        MaskedLinear.forward uses weight*mask (scarches/models/expimap/modules.py:24-25); the engine
//                        keeps ONE packed decoder weight [G][L + n_ext_m + n_ext] = [expr*mask | ext_m*mask_m | ext]
//                        (column order of MaskedCondLayers.forward, modules.py:62-85) so the decoder is a single GEMM;
//                        the same kernel scatters the packed weight gradient back through the masks.
//   * proximal operators ProxGroupLasso / ProxL1 (scarches/trainers/expimap/regularized.py:5-58), in place on the
//                        [G][L] decoder weights after each Adam step (regularized.py:283-298).
//   * HSIC               scarches/models/expimap/losses.py:9-31: value and gradient w.r.t. the latent rows.
#include "common.h"

namespace {

// dst[r*ldd + c] = src[r*lds + c] * (mask ? mask[r*ldm + c] : 1)
__global__ __launch_bounds__(256) void masked_copy_kernel(const float* __restrict__ src, int lds_,
                                                          const float* __restrict__ mask, int ldm,
                                                          float* __restrict__ dst, int ldd, int rows, int cols) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
    float v = src[(long long)r * lds_ + c];
    if (mask) v *= mask[(long long)r * ldm + c];
    dst[(long long)r * ldd + c] = v;
}

// ProxL1: soft threshold by alpha; with I (same shape as W) only where I == 0.
__global__ __launch_bounds__(256) void prox_l1_kernel(float* __restrict__ W, long long n, float alpha,
                                                      const float* __restrict__ I) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (I && I[i] != 0.f) return;
    const float w = W[i];
    float o;
    if (w >= alpha) o = w - alpha;
    else if (w <= -alpha) o = w + alpha;
    else o = w - w;                 // the reference subtracts W from itself (keeps NaN semantics)
    W[i] = o;
}

// ProxGroupLasso over the columns of W [rows][cols]: n_c = ||W[:, c]||_2, coeff_c = alpha * (omega ? omega[c] : 1);
//   W[:, c] <- (W[:, c] - W[:, c] / s_c) * [n_c > coeff_c],  s_c = n_c / coeff_c (+1 when not > 0).
// Three small launches over the whole matrix instead of one workgroup per 64 columns (5 workgroups walked the
// 5000 x 280 decoder of the expiMap benchmark in 546 us): clear the column sums, accumulate them from 64-column x
// 64-row tiles (one float atomic per column and tile), apply.  The column sums live in a file-scope device array
// (kernels name it directly; the host never takes its address): calls that overlap on one GPU would race on it -
// the engine issues its proximal steps on one stream.  Matrices with more columns take the one-launch kernel.
// sca_prox_group_lasso_ws takes the column sums as a caller-owned workspace of `cols` floats instead (no library state:
// calls on different streams may overlap when their workspaces differ).
constexpr int PROX_MAX_COLS = 16384;
constexpr int PROX_ROWS = 64;              // rows per tile of the accumulate / apply kernels
__device__ float g_prox_colsq[PROX_MAX_COLS];

__global__ __launch_bounds__(256) void prox_gl_clear_kernel(int cols, float* __restrict__ ws) {
    float* colsq = ws ? ws : g_prox_colsq;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < cols) colsq[c] = 0.f;
}

__global__ __launch_bounds__(256) void prox_gl_sumsq_kernel(const float* __restrict__ W, int rows, int cols,
                                                            float* __restrict__ ws) {
    float* colsq = ws ? ws : g_prox_colsq;
    __shared__ float part[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const int r0 = blockIdx.y * PROX_ROWS, r1 = min(rows, r0 + PROX_ROWS);
    float ss = 0.f;
    if (c < cols)
#pragma unroll 4
        for (int r = r0 + ry; r < r1; r += 4) {
            const float w = W[(long long)r * cols + c];
            ss = fmaf(w, w, ss);
        }
    part[ry][cx] = ss;
    __syncthreads();
    if (ry == 0 && c < cols) atomicAdd(&colsq[c], (part[0][cx] + part[1][cx]) + (part[2][cx] + part[3][cx]));
}

__global__ __launch_bounds__(256) void prox_gl_apply_kernel(float* __restrict__ W, int rows, int cols, float alpha,
                                                            const float* __restrict__ omega,
                                                            const float* __restrict__ ws) {
    const float* colsq = ws ? ws : g_prox_colsq;
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    if (c >= cols) return;
    const int r0 = blockIdx.y * PROX_ROWS, r1 = min(rows, r0 + PROX_ROWS);
    const float norm = sqrtf(colsq[c]);
    const float coeff = alpha * (omega ? omega[c] : 1.f);
    const float keep = norm > coeff ? 1.f : 0.f;
    float s = norm / coeff;
    if (!(s > 0.f)) s += 1.f;
#pragma unroll 4
    for (int r = r0 + ry; r < r1; r += 4) {
        const long long o = (long long)r * cols + c;
        float w = W[o];
        w -= w / s;
        W[o] = w * keep;
    }
}

// One-launch variant (any number of columns): a workgroup owns 64 columns, lanes = columns, the 4 waves split the rows.
__global__ __launch_bounds__(256) void prox_group_lasso_kernel(float* __restrict__ W, int rows, int cols, float alpha,
                                                               const float* __restrict__ omega) {
    __shared__ float part[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    float ss = 0.f;
    if (c < cols)
        for (int r = ry; r < rows; r += 4) {
            const float w = W[(long long)r * cols + c];
            ss = fmaf(w, w, ss);
        }
    part[ry][cx] = ss;
    __syncthreads();
    if (c >= cols) return;
    const float norm = sqrtf(part[0][cx] + part[1][cx] + part[2][cx] + part[3][cx]);
    const float coeff = alpha * (omega ? omega[c] : 1.f);
    const float keep = norm > coeff ? 1.f : 0.f;
    float s = norm / coeff;
    if (!(s > 0.f)) s += 1.f;
    for (int r = ry; r < rows; r += 4) {
        const long long o = (long long)r * cols + c;
        float w = W[o];
        w -= w / s;
        W[o] = w * keep;
    }
}

// ---------------------------------------------------------------------------------------------------- HSIC
// side[k]: 0 = column belongs to the first argument of hsic(), 1 = to the second, anything else = ignored.
__global__ __launch_bounds__(256) void hsic_kmat_kernel(const float* __restrict__ z, int ld, int B, int D,
                                                        const uint8_t* __restrict__ side, float sigA, float sigB,
                                                        float* __restrict__ KA, float* __restrict__ KB) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * B) return;
    const int i = (int)(idx / B), j = (int)(idx - (long long)i * B);
    const float* zi = z + (long long)i * ld;
    const float* zj = z + (long long)j * ld;
    float da = 0.f, db = 0.f;
    for (int k = 0; k < D; ++k) {
        const float d = zi[k] - zj[k];
        const uint8_t s = side[k];
        if (s == 0) da = fmaf(d, d, da);
        else if (s == 1) db = fmaf(d, d, db);
    }
    KA[idx] = expf(-sigA * da);
    KB[idx] = expf(-sigB * db);
}

// one wave per row: rA_i = mean_j KA_ij, rB_i = mean_j KB_ij, pAB_i = sum_j KA_ij KB_ij
__global__ __launch_bounds__(256) void hsic_rowstats_kernel(const float* __restrict__ KA, const float* __restrict__ KB,
                                                            int B, float* __restrict__ rA, float* __restrict__ rB,
                                                            float* __restrict__ pAB) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= B) return;
    float sa = 0.f, sb = 0.f, sab = 0.f;
    for (int j = lane; j < B; j += 64) {
        const float a = KA[(long long)i * B + j], b = KB[(long long)i * B + j];
        sa += a; sb += b; sab = fmaf(a, b, sab);
    }
    sa = wave_sum(sa); sb = wave_sum(sb); sab = wave_sum(sab);
    if (lane == 0) { rA[i] = sa / (float)B; rB[i] = sb / (float)B; pAB[i] = sab; }
}

// scal[0] = hsic, scal[1] = weight*hsic, scal[2] = dLoss/dh = weight / (2 sqrt h), scal[3] = mean KA, scal[4] = mean KB.
// out_loss (optional, double): out_loss[0] (+)= weight*hsic, out_raw[0] (+)= hsic  (accumulate = one-vs-all sums).
__global__ __launch_bounds__(256) void hsic_scalar_kernel(const float* __restrict__ rA, const float* __restrict__ rB,
                                                          const float* __restrict__ pAB, int B, float weight,
                                                          float* __restrict__ scal, double* __restrict__ out_weighted,
                                                          float* __restrict__ out_raw, int accumulate) {
    __shared__ double red[4][4];
    double a = 0, b = 0, ab = 0, rr = 0;
    for (int i = threadIdx.x; i < B; i += 256) {
        a += rA[i]; b += rB[i]; ab += pAB[i]; rr += (double)rA[i] * rB[i];
    }
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_down(a, o); b += __shfl_down(b, o); ab += __shfl_down(ab, o); rr += __shfl_down(rr, o);
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wv][0] = a; red[wv][1] = b; red[wv][2] = ab; red[wv][3] = rr; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = red[0][0] + red[1][0] + red[2][0] + red[3][0];
        b = red[0][1] + red[1][1] + red[2][1] + red[3][1];
        ab = red[0][2] + red[1][2] + red[2][2] + red[3][2];
        rr = red[0][3] + red[1][3] + red[2][3] + red[3][3];
        const double mA = a / B, mB = b / B;
        const double h = ab / ((double)B * B) + mA * mB - 2.0 * rr / B;
        const double l = sqrt(h);
        scal[0] = (float)l;
        scal[1] = (float)(weight * l);
        scal[2] = h > 0 ? (float)(weight / (2.0 * l)) : 0.f;
        scal[3] = (float)mA;
        scal[4] = (float)mB;
        if (out_weighted) out_weighted[0] = (accumulate ? out_weighted[0] : 0.0) + weight * l;
        if (out_raw) out_raw[0] = (accumulate ? out_raw[0] : 0.f) + (float)l;
    }
}

// dz[i][k] += sum_j c^{side(k)}_ij (z_ik - z_jk),
//   cA_ij = gL (2 KB_ij + 2 mB - 2 rB_i - 2 rB_j)/B^2 * KA_ij * (-2 sigA)   (and A<->B swapped for side 1).
// One workgroup per row i; j is walked in chunks of 256 whose coefficients are staged in LDS.
__global__ __launch_bounds__(256) void hsic_grad_kernel(const float* __restrict__ z, int ld, int B, int D,
                                                        const uint8_t* __restrict__ side, float sigA, float sigB,
                                                        const float* __restrict__ KA, const float* __restrict__ KB,
                                                        const float* __restrict__ rA, const float* __restrict__ rB,
                                                        const float* __restrict__ scal, float* __restrict__ dz,
                                                        int lddz) {
    __shared__ float cA[256], cB[256];
    const int i = blockIdx.x, t = threadIdx.x;
    const float gL = scal[2], mA = scal[3], mB = scal[4];
    const float invB2 = 1.0f / ((float)B * (float)B);
    const float rAi = rA[i], rBi = rB[i];
    for (int k0 = 0; k0 < D; k0 += 256) {
        const int k = k0 + t;
        const int sd = k < D ? side[k] : 2;
        const float zik = k < D ? z[(long long)i * ld + k] : 0.f;
        float acc = 0.f;
        for (int j0 = 0; j0 < B; j0 += 256) {
            __syncthreads();
            const int j = j0 + t;
            if (j < B) {
                const float a = KA[(long long)i * B + j], b = KB[(long long)i * B + j];
                cA[t] = gL * (2.f * b + 2.f * mB - 2.f * rBi - 2.f * rB[j]) * invB2 * a * (-2.f * sigA);
                cB[t] = gL * (2.f * a + 2.f * mA - 2.f * rAi - 2.f * rA[j]) * invB2 * b * (-2.f * sigB);
            }
            __syncthreads();
            if (sd <= 1) {
                const float* cc = sd == 0 ? cA : cB;
                const int jn = min(256, B - j0);
                for (int q = 0; q < jn; ++q) acc = fmaf(cc[q], zik - z[(long long)(j0 + q) * ld + k], acc);
            }
        }
        if (sd <= 1) dz[(long long)i * lddz + k] += acc;
    }
}

}  // namespace

extern "C" int sca_masked_copy(const float* src, int lds_, const float* mask, int ldm, float* dst, int ldd, int rows,
                               int cols, void* stream) {
    if (rows <= 0 || cols <= 0) return 0;
    if (lds_ < cols || ldd < cols || (mask && ldm < cols)) return -1;
    const long long n = (long long)rows * cols;
    hipLaunchKernelGGL(masked_copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                       lds_, mask, ldm, dst, ldd, rows, cols);
    SCA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sca_prox_l1(float* W, long long n, float alpha, const float* I, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(prox_l1_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, n,
                       alpha, I);
    SCA_CHECK_LAUNCH();
    return 0;
}

static int prox_group_lasso_impl(float* W, int rows, int cols, float alpha, const float* omega, float* ws,
                                 void* stream) {
    if (rows <= 0 || cols <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if ((ws || cols <= PROX_MAX_COLS) && rows > 4 * PROX_ROWS) {
        const dim3 grid((cols + 63) / 64, (rows + PROX_ROWS - 1) / PROX_ROWS);
        hipLaunchKernelGGL(prox_gl_clear_kernel, dim3((cols + 255) / 256), dim3(256), 0, st, cols, ws);
        hipLaunchKernelGGL(prox_gl_sumsq_kernel, grid, dim3(256), 0, st, W, rows, cols, ws);
        hipLaunchKernelGGL(prox_gl_apply_kernel, grid, dim3(256), 0, st, W, rows, cols, alpha, omega, ws);
    } else {
        hipLaunchKernelGGL(prox_group_lasso_kernel, dim3((cols + 63) / 64), dim3(256), 0, st, W, rows, cols, alpha,
                           omega);
    }
    SCA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sca_prox_group_lasso(float* W, int rows, int cols, float alpha, const float* omega, void* stream) {
    return prox_group_lasso_impl(W, rows, cols, alpha, omega, nullptr, stream);
}

// The same operator with the column sums of squares in a caller-owned workspace of `cols` floats (any contents on
// entry): nothing is kept in the library, so calls on different streams with different workspaces may overlap.
extern "C" int sca_prox_group_lasso_ws(float* W, int rows, int cols, float alpha, const float* omega, float* colsq_ws,
                                       void* stream) {
    if (!colsq_ws) return -1;
    return prox_group_lasso_impl(W, rows, cols, alpha, omega, colsq_ws, stream);
}

extern "C" long long sca_hsic_workspace_floats(int B) { return 2LL * B * B + 3LL * B + 8; }

extern "C" int sca_hsic_fwd_bwd(const float* z, int ld, int B, int D, const uint8_t* side, float sigma_a,
                                float sigma_b, float weight, float* work, double* loss_weighted, float* loss_raw,
                                int accumulate, float* dz, int lddz, void* stream) {
    if (B <= 0 || D <= 0) return 0;
    if (ld < D || (dz && lddz < D)) return -1;
    hipStream_t st = (hipStream_t)stream;
    float* KA = work;
    float* KB = KA + (long long)B * B;
    float* rA = KB + (long long)B * B;
    float* rB = rA + B;
    float* pAB = rB + B;
    float* scal = pAB + B;
    const long long n = (long long)B * B;
    hipLaunchKernelGGL(hsic_kmat_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, z, ld, B, D, side,
                       sigma_a, sigma_b, KA, KB);
    hipLaunchKernelGGL(hsic_rowstats_kernel, dim3((B + 3) / 4), dim3(256), 0, st, KA, KB, B, rA, rB, pAB);
    hipLaunchKernelGGL(hsic_scalar_kernel, dim3(1), dim3(256), 0, st, rA, rB, pAB, B, weight, scal, loss_weighted,
                       loss_raw, accumulate);
    if (dz)
        hipLaunchKernelGGL(hsic_grad_kernel, dim3(B), dim3(256), 0, st, z, ld, B, D, side, sigma_a, sigma_b, KA, KB,
                           rA, rB, scal, dz, lddz);
    SCA_CHECK_LAUNCH();
    return 0;
}
