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
// (K' = -sum_alpha exp(-d2/2alpha) / (2 alpha)).  The kernels are order-agnostic (any row order) and bound by the
// number of vector instructions; see rbf19 for how the 19 exps per ordered pair become 3 exps + one polynomial pair.
#include "common.h"
#include <type_traits>
#include <algorithm>
#include <atomic>
#include <stdlib.h>

namespace {

__constant__ float MMD_BETA[19] = {
    // 1 / (2 alpha) in float32, alpha as float32 (the reference builds torch.FloatTensor(alphas))
    1.f / (2.f * 1e-6f), 1.f / (2.f * 1e-5f), 1.f / (2.f * 1e-4f), 1.f / (2.f * 1e-3f), 1.f / (2.f * 1e-2f),
    1.f / (2.f * 1e-1f), 1.f / (2.f * 1.f),   1.f / (2.f * 5.f),   1.f / (2.f * 10.f),  1.f / (2.f * 15.f),
    1.f / (2.f * 20.f),  1.f / (2.f * 25.f),  1.f / (2.f * 30.f),  1.f / (2.f * 35.f),  1.f / (2.f * 100.f),
    1.f / (2.f * 1e3f),  1.f / (2.f * 1e4f),  1.f / (2.f * 1e5f),  1.f / (2.f * 1e6f)};

// 1 / (2 alpha) as compile-time constants (for the straight-line code below; same values as MMD_BETA)
__host__ __device__ constexpr float mmd_beta(int a) {
    constexpr float alpha[19] = {1e-6f, 1e-5f, 1e-4f, 1e-3f, 1e-2f, 1e-1f, 1.f,  5.f,  10.f, 15.f,
                                 20.f,  25.f,  30.f,  35.f,  100.f, 1e3f,  1e4f, 1e5f, 1e6f};
    return 1.f / (2.f * alpha[a]);
}
// The ELEVEN bandwidths alpha >= 10 (beta <= 0.05) as one polynomial pair of degree 11 in t = d2 / 48 - 1 on
// d2 <= 96: Chebyshev interpolation of  f = sum_a exp(-beta_a d2)  and  f' = sum_a beta_a exp(-beta_a d2)  in fp64
// (scripts/mmd_poly_fit.py prints these arrays; float32 bandwidths as in MMD_BETA).  Fit error 4.6e-9 on f in [5.3, 11]
// and 2.3e-10 on f' in [0.018, 0.165]; evaluated by an fp32 Horner chain the error is 1.3e-6 max / 2.5e-7 rms on f - one
// rounding of a sum of that size, what the exp-by-exp accumulation leaves as well - and 2.1e-8 / 3.7e-9 on f'.
constexpr int MMD_WIDE0 = 14;          // first of the five widest bandwidths (alpha >= 100)
constexpr int MMD_PC_DEG = 11;
constexpr float MMD_PC_MAX_D2 = 96.f;
constexpr float MMD_PC_K[12] = {6.690013409e+00f, -2.189525604e+00f, 1.198144674e+00f, -5.572726727e-01f,
                                2.325278521e-01f, -8.855332434e-02f, 3.069831803e-02f, -9.630272165e-03f,
                                2.707465552e-03f, -6.992304698e-04f, 1.876339084e-04f, -3.961673065e-05f};
constexpr float MMD_PC_Kp[12] = {4.561511800e-02f, -4.992268234e-02f, 3.482955322e-02f, -1.937758736e-02f,
                                 9.224207141e-03f, -3.835581476e-03f, 1.404727227e-03f, -4.556471540e-04f,
                                 1.307036582e-04f, -3.418900087e-05f, 9.257660167e-06f, -1.963592013e-06f};

// Condition counts -> weight table Wt[C][C] (workgroup 0); the other workgroups clear `zero_buf` (the gradient
// accumulator of the small-D kernel).  Counting looks at run boundaries so that condition-sorted minibatches cost
// two LDS atomics per run, not one per cell.
__global__ __launch_bounds__(256) void mmd_prepare_kernel(const int32_t* __restrict__ cond, int B, int C, int boundary,
                                                          float beta, float* __restrict__ Wt,
                                                          int32_t* __restrict__ counts_out,
                                                          float* __restrict__ zero_buf, long long zero_n) {
    if (blockIdx.x > 0) {
        const long long i = ((long long)(blockIdx.x - 1) * 256 + threadIdx.x) * 4;
        if (i + 3 < zero_n) *reinterpret_cast<float4*>(zero_buf + i) = make_float4(0.f, 0.f, 0.f, 0.f);
        else for (long long q = i; q < zero_n; ++q) zero_buf[q] = 0.f;
        return;
    }
    extern __shared__ int cnt[];           // [C] counts, then [C] valid flags
    int* valid = cnt + C;
    __shared__ int nlow, nhigh;
    for (int i = threadIdx.x; i < C; i += 256) cnt[i] = 0;
    if (threadIdx.x == 0) { nlow = 0; nhigh = 0; }
    __syncthreads();
    {
        // a run [s, e] of one condition adds (e + 1) - s: two LDS atomics per run, any row order (the sums wrap modulo
        // 2^32 for heavily fragmented batches; their difference does not).  Conditions are staged 4096 at a time with
        // all of a thread's loads in flight at once: a single workgroup is latency-bound, not bandwidth-bound.
        constexpr int CH = 4096;
        __shared__ int buf[CH + 2];
        unsigned* ucnt = reinterpret_cast<unsigned*>(cnt);
        for (int base = 0; base < B; base += CH) {
            const int n = min(CH, B - base);
            int v[CH / 256];
#pragma unroll
            for (int q = 0; q < CH / 256; ++q) {
                const int j = threadIdx.x + 256 * q;
                v[q] = j < n ? cond[base + j] : 0;
            }
            if (threadIdx.x == 0) {
                buf[0] = base > 0 ? cond[base - 1] : INT32_MIN;
                buf[n + 1] = base + n < B ? cond[base + n] : INT32_MIN;
            }
#pragma unroll
            for (int q = 0; q < CH / 256; ++q)
                if (threadIdx.x + 256 * q < n) buf[1 + threadIdx.x + 256 * q] = v[q];
            __syncthreads();
#pragma unroll
            for (int q = 0; q < CH / 256; ++q) {
                const int j = threadIdx.x + 256 * q, c = v[q];
                if (j < n && c >= 0 && c < C) {
                    if (buf[j + 2] != c) atomicAdd(&ucnt[c], (unsigned)(base + j + 1));
                    if (buf[j] != c) atomicSub(&ucnt[c], (unsigned)(base + j));
                }
            }
            __syncthreads();
        }
    }
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

// K = sum_a exp(-beta_a d2), Kp = sum_a beta_a exp(-beta_a d2) over the 19 bandwidths: three exps + one polynomial
// pair in the common case - see rbf19 below.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }   // v_pk_fma_f32

// KK[u] = (K, Kp) of pair u.  Written for the packed fp32 VALU ops of gfx950 (two lanes of work per instruction):
// the exp arguments of two pairs share one v_pk_mul_f32, each exp feeds (K, Kp) with one v_pk_fma_f32, and the two
// Horner chains of the polynomial pair (rbf19) run as one packed chain.
template <int U>
__device__ __forceinline__ void rbf_terms(int a, const float (&d2)[U], v2f (&KK)[U]) {
    constexpr float LOG2E = 1.4426950408889634f;
    const float beta = MMD_BETA[a];
    const float c = -(beta * LOG2E);
    const v2f ob = {1.0f, beta};
    if constexpr (U % 2 == 0) {
#pragma unroll
        for (int u = 0; u < U; u += 2) {
            const v2f arg = v2f{d2[u], d2[u + 1]} * v2f{c, c};
            const float e0 = __builtin_amdgcn_exp2f(arg.x), e1 = __builtin_amdgcn_exp2f(arg.y);
            KK[u] = pk_fma(v2f{e0, e0}, ob, KK[u]);
            KK[u + 1] = pk_fma(v2f{e1, e1}, ob, KK[u + 1]);
        }
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float e = __builtin_amdgcn_exp2f(c * d2[u]);
            KK[u] = pk_fma(v2f{e, e}, ob, KK[u]);
        }
    }
}
// The nine middle bandwidths (alpha = 0.1 ... 35) with SIX exps: 1/(2 alpha) of alpha = 20 -> 10 -> 5 and of
// alpha = 30 -> 15 double at every step, so those kernels are successive squares of exp(-d2/40) and exp(-d2/60)
// (the argument rounding of the fp32 product beta*d2 dominates the error of either form).
template <int U>
__device__ __forceinline__ void rbf_mid(const float (&d2)[U], v2f (&KK)[U]) {
    constexpr float LOG2E = 1.4426950408889634f;
    rbf_terms<U>(5, d2, KK);     // alpha 0.1
    rbf_terms<U>(6, d2, KK);     // alpha 1
    rbf_terms<U>(11, d2, KK);    // alpha 25
    rbf_terms<U>(13, d2, KK);    // alpha 35
    constexpr float c20 = -(mmd_beta(10) * LOG2E), c30 = -(mmd_beta(12) * LOG2E);
    const v2f b20 = {1.0f, mmd_beta(10)}, b10 = {1.0f, mmd_beta(8)}, b5 = {1.0f, mmd_beta(7)};
    const v2f b30 = {1.0f, mmd_beta(12)}, b15 = {1.0f, mmd_beta(9)};
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const float e20 = __builtin_amdgcn_exp2f(c20 * d2[u]), e30 = __builtin_amdgcn_exp2f(c30 * d2[u]);
        const float e10 = e20 * e20, e15 = e30 * e30;
        const float e5 = e10 * e10;
        KK[u] = pk_fma(v2f{e20, e20}, b20, KK[u]);
        KK[u] = pk_fma(v2f{e30, e30}, b30, KK[u]);
        KK[u] = pk_fma(v2f{e10, e10}, b10, KK[u]);
        KK[u] = pk_fma(v2f{e15, e15}, b15, KK[u]);
        KK[u] = pk_fma(v2f{e5, e5}, b5, KK[u]);
    }
}
// U independent pairs per call: the per-pair chains (exp -> accumulate, Horner) are latency-bound on their own.
// Common case (every lane 1.75 <= d2 <= 96): THREE exps (alpha = 0.1, 1, 5) + the degree-11 polynomial pair for the
// eleven bandwidths alpha >= 10 = 22 vector instructions per pair (the exp-by-exp form of the same sum took 42; the
// kernels are bound by the number of vector instructions).
//   * alpha <= 0.01 (beta >= 50): 2^(-beta log2e d2) flushes to exactly 0 once beta log2e d2 > 126, i.e. for every
//     lane with d2 > 1.75 - one ballot skips the five exps (taken for all but near-coincident pairs);
//   * some lane beyond the polynomial's range: every bandwidth by its own exp (squared chains for 20 -> 10 -> 5, 30 -> 15).
// (the opaque copies pin the rarely needed exps inside their branches: hipcc otherwise speculates them)
template <int U>
__device__ __forceinline__ void rbf19(const float (&d2)[U], v2f (&KK)[U]) {
    bool near = false, far = false;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        KK[u] = v2f{0.f, 0.f};
        near |= d2[u] < 1.75f;
        far |= d2[u] > MMD_PC_MAX_D2;
    }
    if (__ballot(far) == 0ull) {
        rbf_terms<U>(5, d2, KK);
        rbf_terms<U>(6, d2, KK);
        rbf_terms<U>(7, d2, KK);
        v2f kp[U], tt[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float t = fmaf(d2[u], 2.0f / MMD_PC_MAX_D2, -1.0f);
            tt[u] = v2f{t, t};
            kp[u] = v2f{MMD_PC_K[MMD_PC_DEG], MMD_PC_Kp[MMD_PC_DEG]};
        }
#pragma unroll
        for (int k = MMD_PC_DEG - 1; k >= 0; --k)
#pragma unroll
            for (int u = 0; u < U; ++u) kp[u] = pk_fma(kp[u], tt[u], v2f{MMD_PC_K[k], MMD_PC_Kp[k]});
#pragma unroll
        for (int u = 0; u < U; ++u) KK[u] += kp[u];
    } else {
        float dw[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { dw[u] = d2[u]; asm volatile("" : "+v"(dw[u])); }
        rbf_mid<U>(dw, KK);
#pragma unroll
        for (int a = MMD_WIDE0; a < 19; ++a) rbf_terms<U>(a, dw, KK);
    }
    if (__ballot(near) != 0ull) {
        float dn[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { dn[u] = d2[u]; asm volatile("" : "+v"(dn[u])); }
#pragma unroll
        for (int a = 0; a < 5; ++a) rbf_terms<U>(a, dn, KK);
    }
}
__device__ __forceinline__ void rbf19(float d2, float& K, float& Kp) {
    const float d[1] = {d2};
    v2f kk[1];
    rbf19<1>(d, kk);
    K = kk[0].x; Kp = kk[0].y;
}

constexpr int MMD_T = 64;     // rows per wave (== wave size)
constexpr int MMD_TC = 32;    // columns per workgroup chunk
constexpr int MMD_PLD = 65;   // row pitch of the coefficient tile (conflict-free both ways)

// Small feature dimension (latent z): K(i,j) = K(j,i) and Wt is symmetric, so only tiles on or above the diagonal
// are evaluated.  A wave owns 64 rows i (z_i, diff and the row gradient in registers) and walks the 32 columns of the
// workgroup's chunk, four columns per iteration; for an off-diagonal tile the pair coefficient coef_ij = -4 Wt K' is
// parked in LDS and a second, cheap pass with the roles swapped (lane = column, 2 lanes x D/2 features per column)
// accumulates the column gradient  g_j = z_j sum_i coef_ij - sum_i coef_ij z_i, so every exp serves both ordered
// pairs.  Column gradients are added into slab 0 of `grad` with float atomics (zeroed by mmd_prepare_kernel), once
// per chunk after an LDS sum over the four waves; row gradients are plain stores into the slab of the workgroup's
// column range (workgroups below the diagonal store zeros), so the consumer sums 1 + gridDim.y slabs.
//
// ROWS = true is the row-block form for data-parallel training (SURVEY 8(e) exception 1): the launch owns rows
// [row_lo, row_lo + row_n) of the all-gathered minibatch and walks ALL columns - every tile is treated like a diagonal
// tile (ordered pairs, weight 1, no transposed pass, no column gradients), loss partials are the block's share
// sum_{i in block} sum_j Wt K and `grad` holds gridDim.y slabs of [row_n][D] row gradients (local row index).
template <int DMAX, bool ROWS>
__global__ __launch_bounds__(256) void mmd_small_kernel(const float* __restrict__ z, int ldz,
                                                        const int32_t* __restrict__ cond, int B, int D, int C,
                                                        const float* __restrict__ Wt, double* __restrict__ loss_part,
                                                        float* __restrict__ grad, long long slab_stride, int JR,
                                                        int row_lo, int row_n) {
    constexpr int KH = (DMAX + 1) / 2;        // features per half-wave in the transposed pass
    constexpr int ZS = ((KH + 1 + 3) / 4) * 4;                          // floats per (row, half) in the staged rows
    // (the row-block form has no transposed pass: its staging rows and coefficient tiles shrink to nothing, which
    //  leaves the CU's LDS to the kernels that share the chip with it)
    __shared__ __attribute__((aligned(16))) float zis[ROWS ? 1 : 4][ROWS ? 4 : MMD_T * 2 * ZS];
    __shared__ __attribute__((aligned(8))) float zj[MMD_TC * DMAX];
    __shared__ int cj[MMD_TC];
    __shared__ float Pt[ROWS ? 1 : 4][ROWS ? 1 : MMD_TC * MMD_PLD];  // per wave: coef tile [column][row]; later the wave's column gradients
    __shared__ double redd[16];
    static_assert(MMD_TC * MMD_PLD >= MMD_TC * DMAX, "column-gradient staging aliases the coefficient tile");
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nrows = ROWS ? row_n : B;
    const int nb = (nrows + MMD_T - 1) / MMD_T, nbc = (B + MMD_TC - 1) / MMD_TC;
    constexpr int CPR = MMD_T / MMD_TC;        // column chunks per row block
    const int Jend = min(nbc, ((int)blockIdx.y + 1) * JR);
    const int Jbeg = ROWS ? (int)blockIdx.y * JR : max((int)blockIdx.y * JR, (int)blockIdx.x * 4 * CPR);
    const int I = blockIdx.x * 4 + w;
    const int il = I * MMD_T + lane;           // row within the launch's row range
    const int i = (ROWS ? row_lo : 0) + il;
    const bool act = il < nrows;
    // row gradients go to the slab of this column range (symmetric form: slab 0 collects the column gradients)
    float* gslab = grad ? grad + ((ROWS ? 0 : 1) + (long long)blockIdx.y) * slab_stride + (long long)il * D : nullptr;
    if (Jbeg >= Jend) {                        // every tile of this workgroup lies below the diagonal
        if (threadIdx.x == 0) loss_part[blockIdx.y * gridDim.x + blockIdx.x] = 0.0;
        if (act && gslab)
            for (int k = 0; k < D; ++k) gslab[k] = 0.f;
        return;
    }
    const int ci = act ? cond[i] : -1;
    const bool ok = ci >= 0 && ci < C;
    const float* wrow = Wt + (long long)(ok ? ci : 0) * C;
    static_assert(DMAX % 2 == 0, "features are processed as packed pairs");
    constexpr int DP = DMAX / 2;
    float zi[DMAX];
    v2f zi2[DP], gr2[DP];
#pragma unroll
    for (int k = 0; k < DMAX; ++k) zi[k] = (act && k < D) ? z[(long long)i * ldz + k] : 0.f;
#pragma unroll
    for (int k = 0; k < DP; ++k) {
        zi2[k] = v2f{zi[2 * k], zi[2 * k + 1]};
        gr2[k] = v2f{0.f, 0.f};
    }
    // staged rows for the transposed pass: [i][half][ZS] = {1, z_i[half*KH .. half*KH + KH)} (the leading 1 makes
    // sum_i coef part of the same packed FMA chain)
    if constexpr (!ROWS)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < ZS; ++q) {
            float v = q == 0 ? 1.0f : 0.f;
#pragma unroll
            for (int k = 0; k < DMAX; ++k)
                if (q >= 1 && q <= KH && k == h * KH + q - 1) v = zi[k];
            zis[w][lane * (2 * ZS) + h * ZS + q] = v;
        }
    float lsum = 0.f;
    const int jp = lane & (MMD_TC - 1), kh = lane >> 5;       // transposed pass: column of the chunk, feature half
    for (int J = Jbeg; J < Jend; ++J) {
        const int j0 = J * MMD_TC, nj = min(B - j0, MMD_TC);
        __syncthreads();                       // previous chunk fully consumed
        for (int t = threadIdx.x; t < nj * DMAX; t += 256) {      // features past D are stored as zeros
            const int jj = t / DMAX, k = t - jj * DMAX;
            zj[jj * DMAX + k] = k < D ? z[(long long)(j0 + jj) * ldz + k] : 0.f;
        }
        if (threadIdx.x < MMD_TC) cj[threadIdx.x] = threadIdx.x < nj ? cond[j0 + threadIdx.x] : -1;
        __syncthreads();
        // bit t of `ends` = column t closes a condition run (every wave derives it for itself: one ballot)
        const unsigned long long ends =
            __ballot(lane < nj && (lane == nj - 1 || cj[min(lane + 1, MMD_TC - 1)] != cj[min(lane, MMD_TC - 1)]));
        const int Ij = J / CPR;                // row block the chunk's columns belong to
        const bool mine = (ROWS || I <= Ij) && I < nb;   // wave-uniform
        const bool diag = ROWS || I == Ij;
        const bool park = mine && !diag && grad != nullptr;
        float gj[KH];
#pragma unroll
        for (int q = 0; q < KH; ++q) gj[q] = 0.f;
        if (mine) {
            float wgt = 0.f, wl = 0.f, w4 = 0.f;
            // U columns of one condition run: U independent diff -> d2 -> exp -> accumulate chains
            auto columns = [&](auto UU, int jj) {
                constexpr int U = decltype(UU)::value;
                v2f diff[U][DP], KK[U];
                float d2[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const v2f* zc = reinterpret_cast<const v2f*>(&zj[(jj + u) * DMAX]);
                    v2f acc = {0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < DP; ++k) {
                        diff[u][k] = zi2[k] - zc[k];
                        acc = pk_fma(diff[u][k], diff[u][k], acc);
                    }
                    d2[u] = acc.x + acc.y;
                }
                rbf19<U>(d2, KK);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    lsum = fmaf(wl, KK[u].x, lsum);
                    const float coef = w4 * KK[u].y;
                    const v2f cc = {coef, coef};
#pragma unroll
                    for (int k = 0; k < DP; ++k) gr2[k] = pk_fma(cc, diff[u][k], gr2[k]);
                    if constexpr (!ROWS) { if (park) Pt[w][(jj + u) * MMD_PLD + lane] = coef; }
                }
            };
            int jj = 0;
            while (jj < nj) {
                // columns are walked run by run (condition-sorted minibatches have few runs): the weight is fetched
                // once per run and the inner loop is pure VALU + LDS
                const int c2 = __builtin_amdgcn_readfirstlane(cj[jj]);
                const int je = jj + __ffsll((long long)(ends >> jj));
                wgt = (ok && c2 >= 0 && c2 < C) ? wrow[c2] : 0.f;
                wl = diag ? wgt : 2.0f * wgt;
                w4 = -4.0f * wgt;
                for (; jj + 4 <= je; jj += 4) columns(std::integral_constant<int, 4>{}, jj);
                for (; jj < je; ++jj) columns(std::integral_constant<int, 1>{}, jj);
            }
            if constexpr (!ROWS) if (park) {
                // roles swapped: lane = (column jp, feature half kh); rows of this wave come from LDS
                v2f acc2[ZS / 2];                  // (sum coef, acc_0), (acc_1, acc_2), ...
#pragma unroll
                for (int q = 0; q < ZS / 2; ++q) acc2[q] = v2f{0.f, 0.f};
                const float* prow = &Pt[w][jp * MMD_PLD];
                const float* zr = &zis[w][kh * ZS];
#pragma unroll 8
                for (int i2 = 0; i2 < MMD_T; ++i2) {
                    const float c = prow[i2];
                    const v2f cc = {c, c};
#pragma unroll
                    for (int q4 = 0; q4 < ZS / 4; ++q4) {
                        const float4 a = *reinterpret_cast<const float4*>(zr + i2 * (2 * ZS) + 4 * q4);
                        acc2[2 * q4] = pk_fma(cc, v2f{a.x, a.y}, acc2[2 * q4]);
                        acc2[2 * q4 + 1] = pk_fma(cc, v2f{a.z, a.w}, acc2[2 * q4 + 1]);
                    }
                }
                const float csum = acc2[0].x;
#pragma unroll
                for (int q = 0; q < KH; ++q) {
                    const int k = kh * KH + q;
                    const float aq = ((q + 1) & 1) ? acc2[(q + 1) / 2].y : acc2[(q + 1) / 2].x;
                    gj[q] = (jp < nj && k < D) ? fmaf(zj[jp * DMAX + k], csum, -aq) : 0.f;
                }
            }
        }
        if constexpr (!ROWS) if (grad != nullptr) {
            // column gradients of the four waves: staged in each wave's own (now dead) coefficient tile, summed, added
#pragma unroll
            for (int q = 0; q < KH; ++q) {
                const int k = kh * KH + q;
                if (k < DMAX) Pt[w][jp * DMAX + k] = gj[q];
            }
            __syncthreads();
            for (int t = threadIdx.x; t < nj * D; t += 256) {
                const int jj = t / D, k = t - jj * D;
                const float s = (Pt[0][jj * DMAX + k] + Pt[1][jj * DMAX + k]) + (Pt[2][jj * DMAX + k] + Pt[3][jj * DMAX + k]);
                if (s != 0.f) atomicAdd(&grad[(long long)(j0 + jj) * D + k], s);
            }
        }
    }
    const double tot = block_sum_d((double)lsum, redd);
    if (threadIdx.x == 0) loss_part[blockIdx.y * gridDim.x + blockIdx.x] = tot;
    if (act && gslab) {
#pragma unroll
        for (int k = 0; k < DMAX; ++k)
            if (k < D) gslab[k] = (k & 1) ? gr2[k / 2].y : gr2[k / 2].x;
    }
}

// ---- MFMA form of the row-block kernel (round 2) -------------------------------------------------------------------
// Same outputs as mmd_small_kernel<D, ROWS = true> (every ordered pair of the launch's rows against all B columns), with
// the two inner products of a pair moved from the vector ALU to v_mfma_f32_16x16x4_f32:
//   * d2 of a 16 x 16 tile of (partner m, owner n) pairs = x'_m . y'_n with the augmented vectors
//       x' = [-2 x, |x|^2, 1],  y' = [y, 1, |y|^2]      (K = D + 2 <= 12: three MFMAs);
//     the result lands in the accumulator layout: lane (n = l % 16, kg = l / 16) holds partners m = 4 kg + r, r < 4 -
//     four independent pairs per lane, exactly what rbf19<4> wants.  |x|^2 + |y|^2 - 2 x.y cancels for near points, so
//     a tile in which any pair has d2 < 2 (the diagonal tile, near-coincident cells) recomputes its d2 from differences
//     on the vector ALU - far pairs only feed kernels whose derivative is small (dK = Kp dd2 with Kp ~ 0.1 at d2 ~ 20:
//     an absolute 1e-5 in d2 is 1e-7 of K);
//   * the gradient of the OWNER point, g_n = sum_m coef_mn (y_n - x_m) = y_n S_n - T_n with
//       [T_n ; S_n] = sum_m [x_m ; 1] coef_mn:
//     the accumulator registers of the tile ARE the B operand of that product (register r of lane (n, kg) = coef of
//     partner 4 kg + r, i.e. B[k = kg][n] of MFMA r), A = [x ; 1]^T from LDS - four MFMAs per tile whose output
//     accumulates over ALL partner tiles of the workgroup's range in four registers: no per-pair gradient FMAs, no
//     parked coefficients, and one slab per partner range (8 - 16 instead of 128).
// A wave owns 16 points (lanes replicate them over the four partner groups), a workgroup 64, and walks its partner
// range in staged chunks of 256 (x', [x ; 1] and the conditions in LDS).  The weight Wt[c_n][c_m] is fetched once per
// run of equal partner conditions (condition-sorted minibatches: one load per ~25 tiles); tiles that straddle two runs
// look their four weights up individually.
typedef float mm_f32x4 __attribute__((ext_vector_type(4)));
constexpr int MM_CH = 256;      // partners per staged chunk
constexpr int MM_LD = 12;       // floats per staged partner row (features 0..9, then the two augmentation slots)

template <int DMAX>
__global__ __launch_bounds__(256) void mmd_mfma_kernel(const float* __restrict__ z, int ldz,
                                                       const int32_t* __restrict__ cond, int B, int D, int C,
                                                       const float* __restrict__ Wt, double* __restrict__ loss_part,
                                                       float* __restrict__ grad, long long slab_stride, int per_range,
                                                       int row_lo, int row_n) {
    static_assert(DMAX <= 10, "augmented vectors must fit K = 12");
    __shared__ __attribute__((aligned(16))) float xq[MM_CH * MM_LD];      // x' = [-2 x, |x|^2, 1]
    __shared__ __attribute__((aligned(16))) float xa[MM_CH * MM_LD];      // [x, 1, 0]
    __shared__ int cj[MM_CH];
    __shared__ double redd[16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n16 = lane & 15, kg = lane >> 4;
    // ---- the wave's 16 owner points
    const int il = (blockIdx.x * 4 + w) * 16 + n16;            // row within the launch's row range
    const bool act = il < row_n;
    const int io = row_lo + min(il, row_n - 1);
    float y[DMAX];
    float n2 = 0.f;
#pragma unroll
    for (int k = 0; k < DMAX; ++k) {
        y[k] = k < D ? z[(long long)io * ldz + k] : 0.f;
        n2 = fmaf(y[k], y[k], n2);
    }
    const int co = act ? cond[io] : -1;
    const bool ok = co >= 0 && co < C;
    const float* wrow = Wt + (long long)(ok ? co : 0) * C;
    // B fragments of the d2 product: y'[n][4 s + kg], s = 0..2
    float yb[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int c = 4 * s + kg;
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < DMAX; ++k)
            if (k == c) v = y[k];
        if (c == 10) v = 1.0f;
        if (c == 11) v = n2;
        yb[s] = v;
    }
    mm_f32x4 gacc = {0.f, 0.f, 0.f, 0.f};      // [T ; S] rows 4 kg + r of this owner
    float gv[DMAX];                              // gradient share of near tiles (this lane's partners)
#pragma unroll
    for (int k = 0; k < DMAX; ++k) gv[k] = 0.f;
    float lsum = 0.f;
    int c_run = -2;                              // partner condition of the cached weight
    float w_run = 0.f;
    const int j_beg = blockIdx.y * per_range, j_end = min(B, j_beg + per_range);
    for (int j0 = j_beg; j0 < j_end; j0 += MM_CH) {
        const int nj = min(j_end - j0, MM_CH);
        __syncthreads();                         // previous chunk fully consumed
        {
            const int t = threadIdx.x;
            float v[DMAX], q = 0.f;
            const bool in = t < nj;
#pragma unroll
            for (int k = 0; k < DMAX; ++k) {
                v[k] = (in && k < D) ? z[(long long)(j0 + t) * ldz + k] : 0.f;
                q = fmaf(v[k], v[k], q);
            }
#pragma unroll
            for (int k = 0; k < 10; ++k) {
                xq[t * MM_LD + k] = k < DMAX ? -2.0f * v[k < DMAX ? k : 0] : 0.f;
                xa[t * MM_LD + k] = k < DMAX ? v[k < DMAX ? k : 0] : 0.f;
            }
            xq[t * MM_LD + 10] = q;  xq[t * MM_LD + 11] = 1.0f;
            xa[t * MM_LD + 10] = 1.0f; xa[t * MM_LD + 11] = 0.f;
            cj[t] = in ? cond[j0 + t] : -1;
        }
        __syncthreads();
        const int ntile = (nj + 15) >> 4;
        for (int tile = 0; tile < ntile; ++tile) {
            const int m0 = tile * 16;
            // ---- d2 tile: A[m = l % 16][k = l / 16] = x'[m0 + m][4 s + kg]
            mm_f32x4 dd = {0.f, 0.f, 0.f, 0.f};
            const float* xr = &xq[(m0 + n16) * MM_LD + kg];
#pragma unroll
            for (int s = 0; s < 3; ++s) dd = __builtin_amdgcn_mfma_f32_16x16x4f32(xr[4 * s], yb[s], dd, 0, 0, 0);
            float d2[4] = {fmaxf(dd[0], 0.f), fmaxf(dd[1], 0.f), fmaxf(dd[2], 0.f), fmaxf(dd[3], 0.f)};
            const bool near = d2[0] < 2.0f || d2[1] < 2.0f || d2[2] < 2.0f || d2[3] < 2.0f;
            const bool near_tile = __ballot(near) != 0ull;
            if (near_tile) {                     // exact differences for this tile (diagonal tile, coincident cells)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float* xm = &xa[(m0 + 4 * kg + r) * MM_LD];
                    float acc = 0.f;
#pragma unroll
                    for (int k = 0; k < DMAX; ++k) {
                        const float df = y[k] - xm[k];
                        acc = fmaf(df, df, acc);
                    }
                    d2[r] = acc;
                }
            }
            v2f KK[4];
            rbf19<4>(d2, KK);
            // ---- weights: one per run of equal partner conditions, else per pair
            const int c0 = cj[m0];
            const bool uni = __ballot(lane < 16 && cj[m0 + lane] != c0) == 0ull;
            float wq[4];
            if (uni) {
                if (c0 != c_run) {
                    c_run = c0;
                    w_run = (ok && c0 >= 0 && c0 < C) ? wrow[c0] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) wq[r] = w_run;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c2 = cj[m0 + 4 * kg + r];
                    wq[r] = (ok && c2 >= 0 && c2 < C) ? wrow[c2] : 0.f;
                }
            }
            mm_f32x4 coef;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                lsum = fmaf(wq[r], KK[r].x, lsum);
                coef[r] = -4.0f * wq[r] * KK[r].y;
            }
            if (near_tile) {
                // near pairs carry huge coefficients (narrow kernels) times tiny differences: y S - T would cancel
                // catastrophically (the pair of a point with itself must contribute exactly 0), so these tiles
                // accumulate coef (y - x) from differences on the vector ALU and stay out of the MFMA product
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float* xm = &xa[(m0 + 4 * kg + r) * MM_LD];
#pragma unroll
                    for (int k = 0; k < DMAX; ++k) gv[k] = fmaf(coef[r], y[k] - xm[k], gv[k]);
                    coef[r] = 0.f;
                }
            }
            // ---- [T ; S] += [x ; 1]^T coef: MFMA r contracts partners 4 k + r, A[i' = l % 16][k = l / 16]
            const float* ar = &xa[(m0 + 4 * kg) * MM_LD + min(n16, 11)];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                gacc = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[r * MM_LD], coef[r], gacc, 0, 0, 0);
        }
    }
    const double tot = block_sum_d((double)lsum, redd);
    if (threadIdx.x == 0) loss_part[blockIdx.y * gridDim.x + blockIdx.x] = tot;
    if (grad) {
        // g_n[f] = y_n[f] S_n - T_n[f];  S_n = row 10 = register 2 of the lane (n, kg = 2)
        const float S = __shfl(gacc[2], 32 + n16, 64);
#pragma unroll
        for (int k = 0; k < DMAX; ++k) {          // near-tile shares of the four partner groups of an owner
            gv[k] += __shfl_xor(gv[k], 16, 64);
            gv[k] += __shfl_xor(gv[k], 32, 64);
        }
        float* gs = grad + (long long)blockIdx.y * slab_stride + (long long)il * D;
        if (act) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 4 * kg + r;
                float yf = 0.f, gn = 0.f;
#pragma unroll
                for (int k = 0; k < DMAX; ++k)
                    if (k == f) { yf = y[k]; gn = gv[k]; }
                if (f < D) gs[f] = fmaf(yf, S, -gacc[r]) + gn;
            }
        }
    }
}

// Any feature dimension (MMD on the first decoder layer, mmd_on='y'): 16 x 16 pair tiles through LDS.
// A workgroup owns 16 rows i and a chunk of columns j; the gradient for its rows is accumulated in registers.
constexpr int MMD_GD = 64;    // feature chunk staged per pass
__global__ __launch_bounds__(256) void mmd_generic_kernel(const float* __restrict__ y, int ldy,
                                                          const int32_t* __restrict__ cond, int B, int D, int C,
                                                          const float* __restrict__ Wt, int jchunk,
                                                          double* __restrict__ loss_part,
                                                          float* __restrict__ grad_slabs, long long slab_stride,
                                                          int row_lo, int row_end) {
    extern __shared__ float sm[];          // yi [16][D] | yj [16][D] | coef [16][17]
    float* yi = sm;
    float* yj = sm + 16 * D;
    float* coef = sm + 32 * D;
    __shared__ double redd[16];
    const int t = threadIdx.x, ti = t >> 4, tj = t & 15;
    // rows [row_lo, row_end) against all columns: the whole minibatch, or one rank's row block of the gathered one
    const int i0 = row_lo + blockIdx.x * 16;
    const int jbeg = blockIdx.y * jchunk, jend = min(B, jbeg + jchunk);
    for (int q = t; q < 16 * D; q += 256) {
        const int r = q / D, k = q - r * D;
        yi[q] = (i0 + r < row_end) ? y[(long long)(i0 + r) * ldy + k] : 0.f;
    }
    const int gi = i0 + ti;
    const int ci = gi < row_end ? cond[gi] : -1;
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
        if (gi < row_end && gj < jend && ci >= 0 && ci < C) {
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
    if (gi < row_end && grad_slabs) {
        float* g = grad_slabs + blockIdx.y * slab_stride + (long long)(gi - row_lo) * D;
#pragma unroll
        for (int m = 0; m < MAXM; ++m) {
            const int k = tj + 16 * m;
            if (k < D) g[k] = gacc[m];
        }
    }
}

}  // namespace

// Column chunks per workgroup of the small-D kernel: about 1024 workgroups on or above the diagonal.
static int mmd_small_jr(int B) {
    const long long nb = (B + MMD_T - 1) / MMD_T, nbc = (B + MMD_TC - 1) / MMD_TC;
    long long jr = ((nb + 3) / 4) * nbc / 2 / 1024;
    static const int jr_env = [] { const char* v = getenv("SCA_MMD_JR"); return v ? atoi(v) : 0; }();
    if (jr_env > 0) jr = jr_env;
    if (jr < 1) jr = 1;
    if (jr > 32) jr = 32;
    return (int)jr;
}

extern "C" int sca_mmd_rows_num_slabs(int B, int row_n, int D);
extern "C" int sca_mmd_rows_num_partials(int B, int row_n, int D);
extern "C" int sca_mmd_rows_fwd_bwd(const float* feat, int ld, const int32_t* cond, int B, int D, int C, int boundary,
                                    float beta, float* Wt, int32_t* counts, int row_lo, int row_n, double* loss_part,
                                    float* grad_slabs, long long slab_stride, void* stream);
// Whole-minibatch MMD through the row-block instantiation (all ordered pairs, no transposed pass, no column atomics,
// evenly loaded workgroups) instead of the symmetric one (half the exps, triangular load): faster alone up to
// B = 8192 on MI355X (50 vs 62 us at 4096, 158 vs 187 at 8192, 498 vs 536 at 16384 with ~4096 workgroups) and inside
// the cfg2 step up to the 16384 cells measured (1.126 -> 1.094 ms at 12288, 1.569 -> 1.492 at 16384).
static bool mmd_full_as_rows(int B, int D) {
    // (inside the cfg2 step: 0.438 -> 0.411 ms at B = 4096, 0.735 -> 0.708 ms at B = 8192; SCA_MMD_ROWS_MAX=0 restores
    //  the symmetric kernel for A/B runs)
    static const int bmax = [] { const char* v = getenv("SCA_MMD_ROWS_MAX"); return v ? atoi(v) : 16384; }();
    return D <= 16 && B <= bmax;
}

// Number of gradient slabs / loss partials the main kernel will produce for (B, D).
extern "C" int sca_mmd_num_slabs(int B, int D) {
    if (mmd_full_as_rows(B, D)) return sca_mmd_rows_num_slabs(B, B, D);
    if (D <= 16) {                         // column-gradient accumulator + one row-gradient slab per column range
        const int nbc = (B + MMD_TC - 1) / MMD_TC, jr = mmd_small_jr(B);
        return 1 + (nbc + jr - 1) / jr;
    }
    // generic kernel: aim for >= 512 workgroups
    const int itiles = (B + 15) / 16;
    int chunks = (512 + itiles - 1) / itiles;
    const int jt = (B + 15) / 16;
    if (chunks > jt) chunks = jt;
    if (chunks < 1) chunks = 1;
    return chunks;
}
extern "C" int sca_mmd_num_partials(int B, int D) {
    if (mmd_full_as_rows(B, D)) return sca_mmd_rows_num_partials(B, B, D);
    if (D <= 16) {
        const int nb = (B + MMD_T - 1) / MMD_T, nbc = (B + MMD_TC - 1) / MMD_TC, jr = mmd_small_jr(B);
        return ((nb + 3) / 4) * ((nbc + jr - 1) / jr);
    }
    return sca_mmd_num_slabs(B, D) * ((B + 15) / 16);
}

// Wt: [C][C] floats; counts: [C] int32 (optional); loss_part: sca_mmd_num_partials doubles;
// grad_slabs: sca_mmd_num_slabs x [B][D] floats (slab_stride elements apart), may be null (forward only).
extern "C" int sca_mmd_fwd_bwd(const float* feat, int ld, const int32_t* cond, int B, int D, int C, int boundary,
                               float beta, float* Wt, int32_t* counts, double* loss_part, float* grad_slabs,
                               long long slab_stride, void* stream) {
    if (B <= 0 || C <= 0) return 0;
    if (C > 4096 || D > 1024) return -2;
    if (mmd_full_as_rows(B, D))
        return sca_mmd_rows_fwd_bwd(feat, ld, cond, B, D, C, boundary, beta, Wt, counts, 0, B, loss_part, grad_slabs,
                                    slab_stride, stream);
    hipStream_t st = (hipStream_t)stream;
    const bool small = D <= 16;
    const long long zero_n = (small && grad_slabs) ? (long long)B * D : 0;
    const int zero_wg = (int)((zero_n + 1023) / 1024);
    hipLaunchKernelGGL(mmd_prepare_kernel, dim3(1 + zero_wg), dim3(256), 2 * C * sizeof(int), st, cond, B, C, boundary,
                       beta, Wt, counts, grad_slabs, zero_n);
    const int ns = sca_mmd_num_slabs(B, D);
    if (small) {
        const int nb = (B + MMD_T - 1) / MMD_T, nbc = (B + MMD_TC - 1) / MMD_TC, jr = mmd_small_jr(B);
        dim3 grid((nb + 3) / 4, (nbc + jr - 1) / jr);
        if (D == 10)        // the default latent width: exact-size instantiation, all k < D guards fold away
            hipLaunchKernelGGL((mmd_small_kernel<10, false>), grid, dim3(256), 0, st, feat, ld, cond, B, D, C, Wt,
                               loss_part, grad_slabs, slab_stride, jr, 0, B);
        else
            hipLaunchKernelGGL((mmd_small_kernel<16, false>), grid, dim3(256), 0, st, feat, ld, cond, B, D, C, Wt,
                               loss_part, grad_slabs, slab_stride, jr, 0, B);
    } else {
        const int jt = (B + 15) / 16;
        const int jchunk = ((jt + ns - 1) / ns) * 16;
        dim3 grid((B + 15) / 16, ns);
        const size_t shm = (size_t)(32 * D + 16 * 17) * sizeof(float);
        if (shm > 48 * 1024)
            hipFuncSetAttribute((const void*)mmd_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        hipLaunchKernelGGL(mmd_generic_kernel, grid, dim3(256), shm, st, feat, ld, cond, B, D, C, Wt, jchunk, loss_part,
                           grad_slabs, slab_stride, 0, B);
    }
    SCA_CHECK_LAUNCH();
    return 0;
}

// ---- row-block form (data-parallel MMD over the all-gathered minibatch) ----------------------------------------
// Column chunks per workgroup: about 4096 workgroups over ceil(row_n / 256) x column ranges, and at least two chunks
// from 4096 columns on.  Short workgroups interleave better with the likelihood kernel that shares the chip with the
// MMD inside the step (cfg2 step at B = 8192: 0.711 ms with 8 chunks = 1024 workgroups, 0.703 / 0.684 / 0.673 / 0.693
// with 6 / 3 / 2 / 1; at B = 4096: 0.411 with 2, 0.423 with 1, 0.416 with 4; B = 2048: 1 and 2 equal).
// MFMA form (D <= 10): partners per workgroup range - about 2048 workgroups of 64 owners, whole staged chunks
// Off by default: measured on MI355X it is slower than the vector-ALU form (B = 8192: 175 vs 158 us alone, step 0.746 vs
// 0.684 ms).  PMC says why: both forms issue the same ~228 vector instructions per 256 pairs (rbf19<4> is ~170 of
// them and stays; the accumulator reads, clamps, weight selects and operand moves around the MFMAs cost what the
// difference / gradient FMAs cost before) at one vector instruction per ~4.4 cycles and SIMD - the kernel is bound by
// the instruction count, not by the FMA work the matrix unit took over.  sca_mmd_select_form(1) / SCA_MMD_MFMA=1 select
// it (tests, A/B runs); the choice is process-wide and must not change while launches are being sized.
static int g_mmd_form = [] { const char* v = getenv("SCA_MMD_MFMA"); return (v && atoi(v) != 0) ? 1 : 0; }();
extern "C" int sca_mmd_select_form(int form) {
    const int prev = g_mmd_form;
    if (form == 0 || form == 1) g_mmd_form = form;
    return prev;
}
// Bytes of LDS the vector-ALU row-block kernel reserves per workgroup without using them (see mmd_rows_launch): a
// scheduling knob - it caps the kernel's workgroups per CU - with no effect on results.  Process-wide; applies to the
// launches issued after the call.  Returns 0, or -1 for a size outside [0, 65536].
constexpr int MMD_PAD_MAX = 65536;
static std::atomic<int> g_mmd_pad{40000};
extern "C" int sca_mmd_reserve_lds(int bytes) {
    if (bytes < 0 || bytes > MMD_PAD_MAX) return -1;
    g_mmd_pad.store(bytes, std::memory_order_relaxed);
    return 0;
}
static bool mmd_use_mfma(int D) { return g_mmd_form == 1 && D <= 10; }
static int mmd_mfma_per_range(int B, int row_n) {
    const long long ob = (row_n + 63) / 64;
    static const int wg_env = [] { const char* v = getenv("SCA_MMD_MFMA_WG"); return v ? atoi(v) : 2048; }();
    long long S = (wg_env + ob - 1) / ob;
    const long long smax = (B + MM_CH - 1) / MM_CH;
    if (S > smax) S = smax;
    if (S > 32) S = 32;
    if (S < 1) S = 1;
    long long per = (B + S - 1) / S;
    per = ((per + MM_CH - 1) / MM_CH) * MM_CH;
    return (int)per;
}
static int mmd_rows_jr(int B, int row_n) {
    const long long nb = (row_n + MMD_T - 1) / MMD_T, nbc = (B + MMD_TC - 1) / MMD_TC;
    long long jr = ((nb + 3) / 4) * nbc / 4096;
    if (jr < 2 && nbc >= 128) jr = 2;
    static const int jr_env = [] { const char* v = getenv("SCA_MMD_ROWS_JR"); return v ? atoi(v) : 0; }();
    if (jr_env > 0) jr = jr_env;
    if (jr < 1) jr = 1;
    if (jr > 32) jr = 32;
    return (int)jr;
}
extern "C" int sca_mmd_rows_num_slabs(int B, int row_n, int D) {
    if (B <= 0 || row_n <= 0) return 0;
    if (mmd_use_mfma(D)) {
        const int per = mmd_mfma_per_range(B, row_n);
        return (B + per - 1) / per;
    }
    if (D <= 16) {
        const int nbc = (B + MMD_TC - 1) / MMD_TC, jr = mmd_rows_jr(B, row_n);
        return (nbc + jr - 1) / jr;
    }
    const int itiles = (row_n + 15) / 16, jt = (B + 15) / 16;
    int chunks = (512 + itiles - 1) / itiles;
    if (chunks > jt) chunks = jt;
    return chunks < 1 ? 1 : chunks;
}
extern "C" int sca_mmd_rows_num_partials(int B, int row_n, int D) {
    if (B <= 0 || row_n <= 0) return 0;
    const int ns = sca_mmd_rows_num_slabs(B, row_n, D);
    if (mmd_use_mfma(D)) return ((row_n + 63) / 64) * ns;
    if (D <= 16) return (((row_n + MMD_T - 1) / MMD_T + 3) / 4) * ns;
    return ns * ((row_n + 15) / 16);
}

// The weight table alone (Wt [C][C], counts [C] optional): one small latency-bound workgroup that only needs the
// minibatch's conditions - a caller that knows them at the start of the step issues it early, on a side stream, and
// passes prepared = 1 below so that the MMD launch does not wait for it.
extern "C" int sca_mmd_prepare(const int32_t* cond, int B, int C, int boundary, float beta, float* Wt, int32_t* counts,
                               void* stream) {
    if (B <= 0 || C <= 0) return 0;
    if (C > 4096) return -2;
    hipLaunchKernelGGL(mmd_prepare_kernel, dim3(1), dim3(256), 2 * C * sizeof(int), (hipStream_t)stream, cond, B, C,
                       boundary, beta, Wt, counts, (float*)nullptr, 0LL);
    SCA_CHECK_LAUNCH();
    return 0;
}

static int mmd_rows_launch(const float* feat, int ld, const int32_t* cond, int B, int D, int C, int boundary, float beta,
                           float* Wt, int32_t* counts, int row_lo, int row_n, double* loss_part, float* grad_slabs,
                           long long slab_stride, bool prepared, void* stream);

// MMD with the weight table already built by sca_mmd_prepare (same cond / B / C / boundary / beta): rows
// [row_lo, row_lo + row_n) of the minibatch against all B columns, outputs as for sca_mmd_rows_fwd_bwd (0, B = the whole
// minibatch; a caller may also cut the minibatch into row blocks and place the launches where the chip has room).
// Only for shapes the row-block kernels serve (sca_mmd_prepared_supported(B, D) == 1: D <= 16 up to 16384 cells).
extern "C" int sca_mmd_prepared_supported(int B, int D) { return mmd_full_as_rows(B, D) ? 1 : 0; }
extern "C" int sca_mmd_fwd_bwd_prepared(const float* feat, int ld, const int32_t* cond, int B, int D, int C,
                                        const float* Wt, int row_lo, int row_n, double* loss_part, float* grad_slabs,
                                        long long slab_stride, void* stream) {
    if (B <= 0 || C <= 0 || row_n <= 0) return 0;
    if (!mmd_full_as_rows(B, D)) return -2;
    return mmd_rows_launch(feat, ld, cond, B, D, C, -1, 0.f, const_cast<float*>(Wt), nullptr, row_lo, row_n, loss_part,
                           grad_slabs, slab_stride, true, stream);
}

// Rows [row_lo, row_lo + row_n) of the gathered minibatch `feat` [B][ld] against all B columns.  The weight table is
// built from ALL B conditions (rows with a condition outside [0, C) - padding - are ignored), so
//   sum over the ranks' row blocks of loss_part = the MMD of the whole minibatch, and
//   grad_slabs (sca_mmd_rows_num_slabs x [row_n][D], slab_stride elements apart, summed by the consumer) = d mmd / d
//   feat for the block's rows.  `beta` scales both linearly (a caller folds n_global / n_local in there so that a
//   gradient all-reduce weighted by n_local / n_global adds the MMD term with weight 1).
extern "C" int sca_mmd_rows_fwd_bwd(const float* feat, int ld, const int32_t* cond, int B, int D, int C, int boundary,
                                    float beta, float* Wt, int32_t* counts, int row_lo, int row_n, double* loss_part,
                                    float* grad_slabs, long long slab_stride, void* stream) {
    return mmd_rows_launch(feat, ld, cond, B, D, C, boundary, beta, Wt, counts, row_lo, row_n, loss_part, grad_slabs,
                           slab_stride, false, stream);
}

static int mmd_rows_launch(const float* feat, int ld, const int32_t* cond, int B, int D, int C, int boundary, float beta,
                           float* Wt, int32_t* counts, int row_lo, int row_n, double* loss_part, float* grad_slabs,
                           long long slab_stride, bool prepared, void* stream) {
    if (B <= 0 || C <= 0 || row_n <= 0) return 0;
    if (C > 4096 || D > 1024) return -2;
    if (row_lo < 0 || (long long)row_lo + row_n > B) return -1;
    if (grad_slabs && slab_stride < (long long)row_n * D) return -1;
    hipStream_t st = (hipStream_t)stream;
    if (!prepared)
        hipLaunchKernelGGL(mmd_prepare_kernel, dim3(1), dim3(256), 2 * C * sizeof(int), st, cond, B, C, boundary, beta,
                           Wt, counts, (float*)nullptr, 0LL);
    const int ns = sca_mmd_rows_num_slabs(B, row_n, D);
    if (mmd_use_mfma(D)) {
        dim3 grid((row_n + 63) / 64, ns);
        hipLaunchKernelGGL((mmd_mfma_kernel<10>), grid, dim3(256), 0, st, feat, ld, cond, B, D, C, Wt, loss_part,
                           grad_slabs, slab_stride, mmd_mfma_per_range(B, row_n), row_lo, row_n);
    } else if (D <= 16) {
        const int nb = (row_n + MMD_T - 1) / MMD_T, jr = mmd_rows_jr(B, row_n);
        dim3 grid((nb + 3) / 4, ns);
        // `pad` bytes of dynamic LDS are reserved per workgroup and never touched (default 40000): with its ~2 KB of
        // static LDS the kernel is then limited to THREE workgroups per CU (its registers allow four), which leaves a
        // CU's fourth slot - 34 KB of LDS, a wave per SIMD - to the kernels of the step's critical chain that run beside
        // it (output GEMM, likelihood): their workgroups otherwise trickle in behind the 4096 of this kernel.  Step at
        // 8192 cells with the MMD forked in front of the output GEMM: 0.654 -> 0.645 ms; 24 KB (still four per CU): no
        // change; 70 KB (two per CU): 0.695 ms.  sca_mmd_reserve_lds() sets it per launch sequence (the engine drops
        // the cap when it forks the MMD behind the likelihood kernel's launch); SCA_MMD_LDS_PAD overrides both.
        static const int env_pad = [] {
            const char* v = getenv("SCA_MMD_LDS_PAD");
            hipFuncSetAttribute((const void*)mmd_small_kernel<10, true>, hipFuncAttributeMaxDynamicSharedMemorySize, MMD_PAD_MAX);
            hipFuncSetAttribute((const void*)mmd_small_kernel<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, MMD_PAD_MAX);
            return v ? std::min(std::max(atoi(v), 0), MMD_PAD_MAX) : -1;
        }();
        const int pad = env_pad >= 0 ? env_pad : g_mmd_pad.load(std::memory_order_relaxed);
        if (D == 10)
            hipLaunchKernelGGL((mmd_small_kernel<10, true>), grid, dim3(256), pad, st, feat, ld, cond, B, D, C, Wt,
                               loss_part, grad_slabs, slab_stride, jr, row_lo, row_n);
        else
            hipLaunchKernelGGL((mmd_small_kernel<16, true>), grid, dim3(256), pad, st, feat, ld, cond, B, D, C, Wt,
                               loss_part, grad_slabs, slab_stride, jr, row_lo, row_n);
    } else {
        const int jt = (B + 15) / 16;
        const int jchunk = ((jt + ns - 1) / ns) * 16;
        dim3 grid((row_n + 15) / 16, ns);
        const size_t shm = (size_t)(32 * D + 16 * 17) * sizeof(float);
        if (shm > 48 * 1024)
            hipFuncSetAttribute((const void*)mmd_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        hipLaunchKernelGGL(mmd_generic_kernel, grid, dim3(256), shm, st, feat, ld, cond, B, D, C, Wt, jchunk, loss_part,
                           grad_slabs, slab_stride, row_lo, row_lo + row_n);
    }
    SCA_CHECK_LAUNCH();
    return 0;
}
