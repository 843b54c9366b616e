// Reconstruction losses of the conditional VAE, forward + backward in one pass over the decoder output
// (gfx950).  Restates, per cell row b with condition c_b and size factor s_b:
//     gamma = softmax(logits_b)                                  scarches/models/trvae/modules.py:181
//     mu    = gamma * s_b ; theta = exp(theta[:, c_b])           scarches/models/trvae/trvae.py:136-140
//     NB    log-lik                                              scarches/models/trvae/losses.py:93-100
//     ZINB  log-lik (softplus forms, x<eps / x>eps cases)        scarches/models/trvae/losses.py:144-162
//     recon = -sum_g ll / B                                      scarches/models/trvae/trvae.py:141
// and the MSE head (ReLU output, sum of squares)                 trvae.py:125-127, losses.py:27-42.
//
// Design: a workgroup holds one whole row of logits in registers, so softmax statistics, the likelihood,
// the softmax backward (needs sum_k mu_k dll/dmu_k) and the in-place dlogits store cost ONE read and ONE
// write of the B x G matrix.  Counts are read straight from CSR: at x = 0 every lgamma/digamma term cancels
// exactly, so the dense sweep is 1 exp + 1 log + 1 rcp per element and the lgamma/digamma work (fp64, for
// cancellation-free lgamma(x+theta)-lgamma(theta)) runs only on the row's nnz, one lane per stored count.
// theta gradients are kept in registers/LDS across the rows a workgroup owns and flushed with coalesced
// atomics once per condition run (minibatches arrive condition-sorted), the output-bias gradient likewise.
#include "common.h"
#include "nb_gamma.h"

namespace {

constexpr float NB_EPS = 1e-8f;

__device__ __forceinline__ float softplus_t(float x) {   // torch.nn.functional.softplus (beta 1, threshold 20)
    return x > 20.f ? x : log1pf(expf(x));
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// tab_theta[c][g] = exp(theta[g][c]),  tab_lte[c][g] = log(exp(theta[g][c]) + eps)
__global__ __launch_bounds__(256) void theta_prep_kernel(const float* __restrict__ theta, int G, int C,
                                                         float* __restrict__ tab_theta, float* __restrict__ tab_lte) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= G * C) return;
    const int c = i / G, g = i - c * G;
    const float th = expf(theta[(long long)g * C + c]);
    tab_theta[i] = th;
    tab_lte[i] = logf(th + NB_EPS);
}

// theta_prep_kernel + two zero fills (16-byte aligned, lengths in float4 units) in one launch: the prologue of a step
// was three launches on a side stream (tables, dthetaT = 0, flat gradient buffer = 0) and, at small batch sizes, ended
// after the encoder that the likelihood kernel waits behind.
__global__ __launch_bounds__(256) void theta_prep_zero_kernel(const float* __restrict__ theta, int G, int C,
                                                              float* __restrict__ tab_theta, float* __restrict__ tab_lte,
                                                              float4* __restrict__ z0, long long n0,
                                                              float4* __restrict__ z1, long long n1) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long k = i; k < n0; k += stride) z0[k] = zero;
    for (long long k = i; k < n1; k += stride) z1[k] = zero;
    for (long long k = i; k < (long long)G * C; k += stride) {
        const int c = (int)(k / G), g = (int)(k - (long long)c * G);
        const float th = expf(theta[(long long)g * C + c]);
        tab_theta[k] = th;
        tab_lte[k] = logf(th + NB_EPS);
    }
}

// dtheta[g][c] = scale * dthetaT[c][g]
__global__ __launch_bounds__(256) void transpose_scale_kernel(const float* __restrict__ src, int R, int Ccols,
                                                              float scale, float* __restrict__ dst) {
    // src [R][Ccols] -> dst [Ccols][R]
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= R * Ccols) return;
    const int cc = i / R, r = i - cc * R;       // dst index = cc*R + r
    dst[i] = scale * src[(long long)r * Ccols + cc];
}

struct ReconArgs {
    float* logits; int ldl;         // [B][G]: in = decoder linear output (bias included); out = d loss / d logits
    float* pi;                      // ZINB only: in = dropout logits, out = d loss / d pi
    const int64_t* indptr; const int32_t* indices; const float* data; const int64_t* row_ids;
    const float* sizefactor; const int32_t* cond;
    const float* tab_theta; const float* tab_lte;   // [C][G]
    int B, G, C, rows_per_block;
    float* row_ll;                  // [B] sum_g ll
    float* dthetaT;                 // [C][G] accumulates sum_b theta * dll/dtheta (unscaled), pre-zeroed
    float* dbias; float* dbias_pi;  // [G] accumulate column sums of dlogits / dpi, pre-zeroed (may be null)
    int compute_grad;
    const float* cond_off;          // optional [C][G]: per-condition logit offsets added on load (expiMap's
                                    // decoder.L0.cond_L, scarches/models/expimap/modules.py:83-84), transposed
    float* row_dsf;                 // optional [B]: d recon / d sizefactor_b (expiMap's library-size encoder makes the
                                    // size factor a function of the parameters, expimap.py:186-187)
    const float2* gam_tab;          // optional [C][G][17]: (lgam, dig) of integer counts 0..16 per (condition, gene),
                                    // sca_nb_gamma_table: replaces the per-count product loop of the fast path
};

// All loads of a register array issued before any of them is consumed: ONE asm statement names every element, so
// each load sits ahead of it in program order and the values are first touched behind it.  (hipcc turns
// `v = in_range ? p[i] : x` into a branch per element with the load and its wait inside - EPT serial memory latencies
// per row, which was most of this kernel's time - and sinks an unconditional load that feeds a select back into one.)
template <int N>
__device__ __forceinline__ void loads_in_flight(float (&v)[N]) {
    static_assert(N == 4 || N == 8 || N == 12, "instantiated element counts");
    if constexpr (N == 4)
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
    else if constexpr (N == 8)
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
    else
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                          "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]));
}

constexpr int QMAX = 1024;     // LDS queue of heavy entries per row (overflow is handled inline)

// (128 VGPRs: the cap spills the eight cached reciprocals of a thread - one scratch reload per element of the dense
//  sweep.  Without it - 148 VGPRs, no spills - the kernel alone runs 114 us instead of 130 at 8192 cells, and the step
//  0.643 ms instead of 0.596: beside three MMD waves per SIMD the wider workgroup is placed later still.)
template <int T, int EPT, bool ZINB, bool TAB = false, int WPS = 4>
__global__ __launch_bounds__(T, WPS) void recon_rows_kernel(const ReconArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* tx = lds;              // [G] per-row extras of t = mu dll/dmu at nnz (cleared by the owner when read)
    float* dx = lds + a.G;        // [G] theta-gradient extras, accumulated over the rows of a condition run
    float* px = lds + 2 * a.G;    // [G] ZINB: dpi extras (cleared when read)
    __shared__ float red[2][2][16];
    __shared__ int queue[QMAX];
    __shared__ int qn;
    const int G = a.G, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    constexpr int NW = T / 64;
    const int nlds = (ZINB ? 3 : 2) * G;
    for (int i = t; i < nlds; i += T) lds[i] = 0.f;
    if (t == 0) qn = 0;

    float acc_th[EPT], acc_b[EPT], acc_p[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) acc_th[j] = acc_b[j] = acc_p[j] = 0.f;

    const int b0 = blockIdx.x * a.rows_per_block, b1 = min(a.B, b0 + a.rows_per_block);
    const float invB = 1.0f / (float)a.B;
    int par = 0;                  // reduction scratch parity: alternating buffers need ONE barrier per reduction
    // Row scalars (condition, size factor, CSR extent) of up to RS rows are fetched in one go: the row id -> indptr ->
    // counts chain is paid once per RS rows instead of once per row
    constexpr int RS = 64;
    __shared__ long long s_ks[RS];
    __shared__ int s_nn[RS], s_c[RS];
    __shared__ float s_sf[RS];
    float thr[EPT], lter[EPT];    // per-gene table values of this thread, reloaded when the condition changes
    int ccur = -2;
    for (int b = b0; b < b1; ++b) {
        if (((b - b0) & (RS - 1)) == 0) {
            __syncthreads();
            if (t < RS && b + t < b1) {
                const int bb = b + t;
                const int64_t r = a.row_ids ? a.row_ids[bb] : (int64_t)bb;
                const int64_t k0 = a.indptr[r];
                s_ks[t] = k0;
                s_nn[t] = (int)(a.indptr[r + 1] - k0);
                s_c[t] = a.cond ? a.cond[bb] : 0;
                s_sf[t] = a.sizefactor[bb];
            }
            __syncthreads();
        }
        const int c = s_c[(b - b0) & (RS - 1)];
        const float s = s_sf[(b - b0) & (RS - 1)];
        const int64_t ks = s_ks[(b - b0) & (RS - 1)], ke = ks + s_nn[(b - b0) & (RS - 1)];
        float* lrow = a.logits + (long long)b * a.ldl;
        float* prow = ZINB ? a.pi + (long long)b * a.ldl : nullptr;
        const float* tth = a.tab_theta + (long long)c * G;
        const float* tlt = a.tab_lte + (long long)c * G;
        const float* coff = a.cond_off ? a.cond_off + (long long)c * G : nullptr;

        // ---- online softmax statistics: per-thread (max, sum exp), merged across the workgroup in one round
        float lg[EPT];
        float mx = -INFINITY;
        // the whole row (and its per-condition offsets) requested at once from clamped addresses
#pragma unroll
        for (int j = 0; j < EPT; ++j) lg[j] = lrow[min(t + T * j, G - 1)];
        if (coff) {
            float co[EPT];
#pragma unroll
            for (int j = 0; j < EPT; ++j) co[j] = coff[min(t + T * j, G - 1)];
            loads_in_flight(co);
            loads_in_flight(lg);
#pragma unroll
            for (int j = 0; j < EPT; ++j) lg[j] += co[j];
        } else {
            loads_in_flight(lg);
        }
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            if (t + T * j >= G) lg[j] = -INFINITY;
            mx = fmaxf(mx, lg[j]);
        }
        // the first stored count of this lane travels together with the logits (its two dependent reads overlap the
        // softmax reduction below)
        const bool has0 = ks + t < ke;
        float x0 = 0.f, th0 = 1.f, lv0 = 0.f;
        int g0 = 0;
        if (has0) {
            x0 = a.data[ks + t];
            g0 = a.indices[ks + t];
            th0 = tth[g0];
            lv0 = lrow[g0] + (coff ? coff[g0] : 0.f);
        }
        if (c != ccur) {
            ccur = c;
#pragma unroll
            for (int j = 0; j < EPT; ++j) thr[j] = tth[min(t + T * j, G - 1)];
            if (ZINB) {
#pragma unroll
                for (int j = 0; j < EPT; ++j) lter[j] = tlt[min(t + T * j, G - 1)];
                loads_in_flight(lter);
            }
            loads_in_flight(thr);
#pragma unroll
            for (int j = 0; j < EPT; ++j) {
                const bool in = t + T * j < G;
                if (!in) thr[j] = 1.f;
                // ZINB keeps log(theta + eps); NB keeps 1 / (theta + eps) (its dense sweep needs no other table value)
                if (ZINB) { if (!in) lter[j] = 0.f; }
                else lter[j] = __builtin_amdgcn_rcpf(thr[j] + NB_EPS);
            }
        }
        // lg[j] <- exp(logit - thread max): the ONE exp of a gene; gamma = lg[j] * fgam once the row statistics are in
        float se = 0.f;
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            lg[j] = (t + T * j < G) ? __expf(lg[j] - mx) : 0.f;
            se += lg[j];
        }
        const float mxt = mx;
        {
            float wm = wave_max(mx);
            se *= (mx == -INFINITY) ? 0.f : __expf(mx - wm);
            se = wave_sum(se);
            if (lane == 0) { red[par][0][wv] = wm; red[par][1][wv] = se; }
            __syncthreads();
            float M = red[par][0][0];
#pragma unroll
            for (int i = 1; i < NW; ++i) M = fmaxf(M, red[par][0][i]);
            float S = 0.f;
#pragma unroll
            for (int i = 0; i < NW; ++i) S += red[par][1][i] * __expf(red[par][0][i] - M);
            mx = M; se = S;
            par ^= 1;
        }
        const float inv = 1.0f / se;
        const float fgam = (mxt == -INFINITY) ? 0.f : __expf(mxt - mx) * inv, fmu = fgam * s;

        // ---- sparse phase: one lane per stored count of this row
        float ll = 0.f;
        // (TAB: a separate instantiation - carried as a run-time option the table costs the plain kernel six more spilled
        //  registers under its 128-VGPR cap: 125 -> 137 us alone at 8192 cells)
        const float2* gt = TAB ? a.gam_tab + (long long)c * G * 17 : nullptr;
        auto stored = [&](int64_t k, float x, int g, float th, float lv) {
            if (!(x > NB_EPS)) return;
            const float mu = __expf(lv - mx) * inv * s;
            const float den = th + mu + NB_EPS;
            const float L = fast_logf(den), rr = __builtin_amdgcn_rcpf(den);
            float lgam = 0.f, dig = 0.f;
            if (TAB && nb_gamma_fast_ok(x, th)) {
                // the same product / harmonic forms, evaluated once per (condition, gene, count) and step
                const float2 tv = gt[(long long)g * 17 + (int)x];
                lgam = tv.x;
                dig = tv.y;
            } else if (nb_gamma_fast_ok(x, th)) {
                // prod_k (theta+k) and sum_k 1/(theta+k) = num/den built without divisions (three full-rate ops per
                // term + the factorial alongside; theta < 128 keeps 16 factors inside fp32 range): ONE rcp per count
                const int n = min((int)x, 16);
                float num = 0.f, den = 1.f, fact = 1.f, f = th, fq = 1.f;
                for (int q = 0; q < n; ++q) {
                    num = fmaf(num, f, den);
                    den *= f;
                    fact *= fq;
                    f += 1.0f;
                    fq += 1.0f;
                }
                const float rden = __builtin_amdgcn_rcpf(den);
                dig = num * rden;
                lgam = -fast_logf(fact * rden);
            } else {
                const int slot = atomicAdd(&qn, 1);
                if (slot < QMAX) queue[slot] = (int)(k - ks);
                else nb_gamma_terms_heavy(x, th, &lgam, &dig);
            }
            float ex_ll = x * (fast_logf(mu + NB_EPS) - L) + lgam;
            float ex_t = mu * x * (__builtin_amdgcn_rcpf(mu + NB_EPS) - rr);
            float ex_th = th * (-x * rr + dig);
            if (ZINB) {
                // replace the zero-case (evaluated for every gene in the dense sweep) by the non-zero case
                const float pv = prow[g];
                const float base = th * (tlt[g] - L);
                const float ptl = -pv + base;
                const float sg = sigmoid_f(ptl);
                const float dbase_dth = tlt[g] - L + th / (th + NB_EPS) - th * rr;
                ex_ll += ptl - softplus_t(ptl);                      // (-sp + ptl + ..) - (softplus(ptl) - sp)
                ex_t += (1.0f - sg) * (-mu * th * rr);
                ex_th += th * (1.0f - sg) * dbase_dth;
                px[g] = -1.0f + sg;                                   // (sig(-pi) - 1) - (sig(-pi) - sg)
            }
            ll += ex_ll;
            tx[g] = ex_t;                 // joins t_g (and the row total T) in the dense sweep
            dx[g] += ex_th;
        };
        if (has0) stored(ks + t, x0, g0, th0, lv0);
        for (int64_t k = ks + t + T; k < ke; k += T) {
            const int g = a.indices[k];
            stored(k, a.data[k], g, tth[g], lrow[g] + (coff ? coff[g] : 0.f));
        }
        __syncthreads();
        // ---- heavy entries (large / non-integer counts), compacted so the fp64 code runs on full lanes
        {
            const int nq = min(qn, QMAX);
            for (int q = t; q < nq; q += T) {
                const int64_t k = ks + queue[q];
                const float x = a.data[k];
                const int g = a.indices[k];
                const float th = tth[g];
                float lgam, dig;
                nb_gamma_terms_heavy(x, th, &lgam, &dig);
                ll += lgam;
                dx[g] += th * dig;        // one entry per gene per row: no other lane touches dx[g] in this phase
            }
        }

        // ---- dense sweep (x = 0 terms for every gene)
        float Tsum = 0.f;
        float tj[EPT], pj[EPT];
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const int g = t + T * j;
            tj[j] = 0.f; pj[j] = 0.f;
            if (g < G) {
                const float th = thr[j];
                const float gam = lg[j] * fgam, mu = lg[j] * fmu;
                const float den = th + mu + NB_EPS;
                const float rr = __builtin_amdgcn_rcpf(den);
                float tv, dth;
                if (!ZINB) {
                    // theta (log(theta+eps) - log(theta+mu+eps)) as ONE log of the ratio; with u = -theta/den:
                    //   t = mu u,   theta d/dtheta = base + theta (theta/(theta+eps) + u)
                    const float r1 = lter[j];
                    const float base = -th * fast_logf(den * r1);
                    const float u = -th * rr;
                    ll += base;
                    tv = mu * u;
                    dth = fmaf(th, fmaf(th, r1, u), base);
                } else {
                    const float lte = lter[j];
                    const float L = fast_logf(den);
                    const float base = th * (lte - L);
                    const float dbase_dth = lte - L + th * (__builtin_amdgcn_rcpf(th + NB_EPS) - rr);
                    const float pv = prow[g];
                    const float ptl = -pv + base;
                    const float sg = sigmoid_f(ptl);
                    ll += softplus_t(ptl) - softplus_t(-pv);
                    tv = sg * (-mu * th * rr);
                    dth = th * sg * dbase_dth;
                    pj[j] = (sigmoid_f(-pv) - sg) + px[g];
                    px[g] = 0.f;
                }
                tv += tx[g];
                tx[g] = 0.f;
                tj[j] = tv;
                Tsum += tv;
                acc_th[j] += dth;
                lg[j] = gam;                                          // lg now holds gamma
            }
        }
        {
            ll = wave_sum(ll);
            Tsum = wave_sum(Tsum);
            if (lane == 0) { red[par][0][wv] = ll; red[par][1][wv] = Tsum; }
            __syncthreads();                  // also orders the queue / tx / dx traffic of this row before the next
            ll = 0.f; Tsum = 0.f;
#pragma unroll
            for (int i = 0; i < NW; ++i) { ll += red[par][0][i]; Tsum += red[par][1][i]; }
            par ^= 1;
        }
        if (t == 0) {
            a.row_ll[b] = ll;
            qn = 0;
            // recon = -sum ll / B and mu = gamma s: d recon / d s = -(1/B) sum_g gamma dll/dmu = -T / (s B)
            if (a.row_dsf) a.row_dsf[b] = s > 0.f ? -Tsum * invB / s : 0.f;
        }

        if (a.compute_grad) {
#pragma unroll
            for (int j = 0; j < EPT; ++j) {
                const int g = t + T * j;
                if (g < G) {
                    const float dl = (lg[j] * Tsum - tj[j]) * invB;
                    lrow[g] = dl;
                    acc_b[j] += dl;
                    if (ZINB) {
                        const float dp = -pj[j] * invB;
                        prow[g] = dp;
                        acc_p[j] += dp;
                    }
                }
            }
            const bool last = (b + 1 == b1);
            const int nxt = (b + 1 - b0) & (RS - 1);      // a refill of the row scalars also closes the run
            const int cn = (last || nxt == 0) ? -1 : s_c[nxt];
            if (last || cn != c) {
                float* dst = a.dthetaT + (long long)c * G;
#pragma unroll
                for (int j = 0; j < EPT; ++j) {
                    const int g = t + T * j;
                    if (g < G) {
                        atomicAdd(&dst[g], acc_th[j] + dx[g]);
                        dx[g] = 0.f;
                        acc_th[j] = 0.f;
                    }
                }
            }
        }
    }
    if (a.compute_grad) {
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const int g = t + T * j;
            if (g < G) {
                if (a.dbias) atomicAdd(&a.dbias[g], acc_b[j]);
                if (ZINB && a.dbias_pi) atomicAdd(&a.dbias_pi[g], acc_p[j]);
            }
        }
    }
}

// Output activations of the non-softmax heads (MaskedLinearDecoder.mean_decoder, models/expimap/modules.py:119-127;
// trVAE recon_decoder ReLU, models/trvae/modules.py:172-173): 0 identity, 1 relu, 2 softplus (beta 1, threshold 20), 3 exp.
__device__ __forceinline__ void head_act(int act, float l, float& o, float& d) {
    if (act == 1) { o = fmaxf(l, 0.f); d = l > 0.f ? 1.f : 0.f; }
    else if (act == 2) { o = softplus_t(l); d = l > 20.f ? 1.f : sigmoid_f(l); }
    else if (act == 3) { o = expf(l); d = o; }
    else { o = l; d = 1.f; }
}

// MSE head: out = act(lin (+ per-condition offset)) - relu for the trVAE recon_decoder, identity / relu / softplus / exp
// for expiMap's last layer (scarches/models/expimap/expimap.py:182-184) - loss_b = sum_g (out - x)^2,
// d lin = 2 (out - x) act'(lin) / B.
__global__ __launch_bounds__(256) void mse_rows_kernel(float* __restrict__ lin, int ldl, const float* __restrict__ x,
                                                       int ldx, int B, int G, int rows_per_block,
                                                       float* __restrict__ row_loss, float* __restrict__ dbias,
                                                       int compute_grad, int act, const int32_t* __restrict__ cond,
                                                       const float* __restrict__ cond_off) {
    __shared__ float red[16];
    const int b0 = blockIdx.x * rows_per_block, b1 = min(B, b0 + rows_per_block);
    const float invB = 1.0f / (float)B;
    for (int g0 = 0; g0 < G; g0 += 256) {
        const int g = g0 + threadIdx.x;
        float accb = 0.f;
        for (int b = b0; b < b1; ++b) {
            float sq = 0.f;
            if (g < G) {
                float l = lin[(long long)b * ldl + g];
                if (cond_off) l += cond_off[(long long)cond[b] * G + g];
                float o, dv;
                head_act(act, l, o, dv);
                const float d = o - x[(long long)b * ldx + g];
                sq = d * d;
                if (compute_grad) {
                    const float dl = 2.0f * d * dv * invB;
                    lin[(long long)b * ldl + g] = dl;
                    accb += dl;
                }
            }
            sq = block_sum(sq, red);
            if (threadIdx.x == 0) {
                if (g0 == 0) row_loss[b] = sq;
                else row_loss[b] += sq;
            }
        }
        if (compute_grad && dbias && g < G) atomicAdd(&dbias[g], accb);
    }
}

// NB likelihood on an ELEMENTWISE mean mu = act(lin) (expiMap with decoder_last_layer in {softplus, exp, relu,
// identity}: no softmax, no size factor - models/expimap/expimap.py:188-196, losses.py:63-102).  A workgroup walks
// rows_per_block rows; the row's counts are scattered from CSR into an LDS row, the dense sweep evaluates the full
// formula (lgamma / digamma only where x > 0), d lin = -(dll/dmu) act'(lin) / B is written in place and the theta
// gradient theta dll/dtheta is accumulated in LDS over a condition run and flushed with atomics.
__global__ __launch_bounds__(256) void nb_elem_rows_kernel(const ReconArgs a, int act) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xr = lds;             // [G] counts of the current row
    float* dth = lds + a.G;      // [G] theta gradient of the current condition run
    __shared__ float red[16];
    const int G = a.G, t = threadIdx.x;
    for (int i = t; i < 2 * G; i += 256) lds[i] = 0.f;
    const int b0 = blockIdx.x * a.rows_per_block, b1 = min(a.B, b0 + a.rows_per_block);
    const float invB = 1.0f / (float)a.B;
    __syncthreads();
    for (int b = b0; b < b1; ++b) {
        const int64_t r = a.row_ids ? a.row_ids[b] : (int64_t)b;
        const int64_t ks = a.indptr[r], ke = a.indptr[r + 1];
        const int c = a.cond ? a.cond[b] : 0;
        for (int64_t k = ks + t; k < ke; k += 256) xr[a.indices[k]] = a.data[k];
        __syncthreads();
        float* lrow = a.logits + (long long)b * a.ldl;
        const float* tth = a.tab_theta + (long long)c * G;
        const float* tlt = a.tab_lte + (long long)c * G;
        const float* coff = a.cond_off ? a.cond_off + (long long)c * G : nullptr;
        float ll = 0.f;
        for (int g = t; g < G; g += 256) {
            const float l = lrow[g] + (coff ? coff[g] : 0.f);
            float mu, dmu;
            head_act(act, l, mu, dmu);
            const float x = xr[g], th = tth[g], lte = tlt[g];
            xr[g] = 0.f;
            const float den = th + mu + NB_EPS;
            const float L = logf(den), rr = 1.0f / den;
            float v = th * (lte - L);
            float dll_dmu = -th * rr;
            float thd = th * (lte - L + th / (th + NB_EPS) - th * rr);     // theta dll/dtheta
            if (x > 0.f) {
                float lgam, dig;
                if (nb_gamma_fast_ok(x, th)) {
                    const int n = min((int)x, 16);
                    float num = 0.f, dn = 1.f, fact = 1.f, f = th, fq = 1.f;
                    for (int q = 0; q < n; ++q) { num = fmaf(num, f, dn); dn *= f; fact *= fq; f += 1.0f; fq += 1.0f; }
                    const float rden = 1.0f / dn;
                    dig = num * rden;
                    lgam = -logf(fact * rden);
                } else {
                    nb_gamma_terms_heavy(x, th, &lgam, &dig);
                }
                v += x * (logf(mu + NB_EPS) - L) + lgam;
                dll_dmu += x * (1.0f / (mu + NB_EPS) - rr);
                thd += th * (-x * rr + dig);
            }
            ll += v;
            if (a.compute_grad) {
                lrow[g] = -dll_dmu * dmu * invB;
                dth[g] += thd;
            }
        }
        ll = block_sum(ll, red);
        if (t == 0) a.row_ll[b] = ll;
        const bool flush = a.compute_grad && a.dthetaT && (b + 1 == b1 || (a.cond && a.cond[b + 1] != c));
        if (flush) {
            float* dst = a.dthetaT + (long long)c * G;
            for (int g = t; g < G; g += 256) {
                atomicAdd(&dst[g], dth[g]);
                dth[g] = 0.f;
            }
        }
        __syncthreads();
    }
}

// Poisson head of scPoli (models/scpoli/scpoli.py:189-191,607,640; models/trvae/losses.py:44-60): the softmax output is
// the Poisson rate itself (no size factor), loss_b = sum_g [rate - xlogy(x, rate) + lgamma(x + 1)]
//   = 1 - sum_nnz x log(gamma_g) + sum_nnz lgamma(x + 1),   d loss / d logit_g = (gamma_g S_b - x_bg) / B,  S_b = sum_g x.
// One workgroup walks rows_per_block rows; counts come straight from CSR.
__global__ __launch_bounds__(256) void poisson_rows_kernel(float* __restrict__ logits, int ldl,
                                                           const int64_t* __restrict__ indptr,
                                                           const int32_t* __restrict__ indices,
                                                           const float* __restrict__ data,
                                                           const int64_t* __restrict__ row_ids,
                                                           const float* __restrict__ sizefactor, int B, int G,
                                                           int rows_per_block, float* __restrict__ row_loss,
                                                           int compute_grad) {
    __shared__ float red[16];
    __shared__ double redd[16];
    const int t = threadIdx.x;
    const int b0 = blockIdx.x * rows_per_block, b1 = min(B, b0 + rows_per_block);
    const float invB = 1.0f / (float)B;
    for (int b = b0; b < b1; ++b) {
        float* lrow = logits + (long long)b * ldl;
        const int64_t r = row_ids ? row_ids[b] : (int64_t)b;
        const int64_t ks = indptr[r], ke = indptr[r + 1];
        float mx = -INFINITY;
        for (int g = t; g < G; g += 256) mx = fmaxf(mx, lrow[g]);
        mx = block_max(mx, red);
        float se = 0.f;
        for (int g = t; g < G; g += 256) se += expf(lrow[g] - mx);
        se = block_sum(se, red);
        const float lse = mx + logf(se);
        // stored counts: - x log gamma + lgamma(x + 1), in double (a few hundred terms per row)
        double acc = 0.0;
        for (int64_t k = ks + t; k < ke; k += 256) {
            const float x = data[k];
            if (x > 0.f) acc += -(double)x * (double)(lrow[indices[k]] - lse) + lgamma((double)x + 1.0);
        }
        acc = block_sum_d(acc, redd);
        if (t == 0) row_loss[b] = (float)(1.0 + acc);
        if (compute_grad) {
            const float S = sizefactor[b];
            __syncthreads();                       // every read of the logits is done
            for (int g = t; g < G; g += 256) lrow[g] = expf(lrow[g] - lse) * S * invB;
            __syncthreads();
            for (int64_t k = ks + t; k < ke; k += 256) lrow[indices[k]] -= data[k] * invB;   // canonical CSR: unique columns
        }
        __syncthreads();
    }
}

template <bool ZINB, bool TAB = false>
int launch_recon(const ReconArgs& a, hipStream_t st) {
    const int G = a.G;
    const size_t shm = (size_t)(ZINB ? 3 : 2) * G * sizeof(float);
    const int nblk = (a.B + a.rows_per_block - 1) / a.rows_per_block;
#define RL(TT, EE)                                                                                         \
    do {                                                                                                   \
        if (shm > 48 * 1024)                                                                               \
            hipFuncSetAttribute((const void*)recon_rows_kernel<TT, EE, ZINB, TAB>,                         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                     \
        hipLaunchKernelGGL((recon_rows_kernel<TT, EE, ZINB, TAB>), dim3(nblk), dim3(TT), shm, st, a);      \
    } while (0)
    if (G <= 256 * 4) RL(256, 4);
    else if (G <= 256 * 8) {
        // Three waves per SIMD, no register cap (148 VGPRs, nothing spilled; 114 us alone against 125 at 8192 cells):
        // measured inside the cfg2 step (round 3, 300-step runs) it LOSES up to 8192 cells per step - 0.566 -> 0.578 ms
        // at 8192, 0.425 -> 0.446 at 6144, 0.309 -> 0.311 at 4096: beside the MMD's waves the wider workgroups are
        // placed later - and wins above - 1.305 -> 1.270 ms at 16384, where the likelihood kernel has the chip to
        // itself for longer.  SCA_RECON_WIDE = 0 / 1 overrides the batch-size rule.
        static const int wide_env = [] { const char* e = getenv("SCA_RECON_WIDE"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
        const bool wide = wide_env >= 0 ? wide_env == 1 : a.B >= 12288;
        if (wide && !ZINB && !TAB) {
            if (shm > 48 * 1024)
                hipFuncSetAttribute((const void*)recon_rows_kernel<256, 8, false, false, 3>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
            hipLaunchKernelGGL((recon_rows_kernel<256, 8, false, false, 3>), dim3(nblk), dim3(256), shm, st, a);
        } else RL(256, 8);
    }
    else if (G <= 512 * 8) RL(512, 8);
    else if (G <= 1024 * 8) RL(1024, 8);
    else if (G <= 1024 * 12) RL(1024, 12);
    else return -2;     // > 12288 genes: LDS extras (2-3 x G floats) no longer fit next to useful occupancy
#undef RL
    return 0;
}

}  // namespace

extern "C" int sca_theta_prep(const float* theta, int G, int C, float* tab_theta, float* tab_lte, void* stream) {
    if (G * C <= 0) return 0;
    hipLaunchKernelGGL(theta_prep_kernel, dim3((G * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, theta, G, C,
                       tab_theta, tab_lte);
    SCA_CHECK_LAUNCH();
    return 0;
}

// sca_theta_prep plus two zero fills in the same launch (either may be NULL / 0 bytes; pointers 16-byte aligned, byte
// counts multiples of 16): the per-step prologue as ONE launch.
extern "C" int sca_theta_prep_zero(const float* theta, int G, int C, float* tab_theta, float* tab_lte, void* zero0,
                                   long long bytes0, void* zero1, long long bytes1, void* stream) {
    if ((((uintptr_t)zero0 | (uintptr_t)zero1) & 15) || ((bytes0 | bytes1) & 15) || bytes0 < 0 || bytes1 < 0) return -1;
    if (!zero0) bytes0 = 0;
    if (!zero1) bytes1 = 0;
    const long long work = (long long)G * C > (bytes0 + bytes1) / 16 ? (long long)G * C : (bytes0 + bytes1) / 16;
    if (work <= 0) return 0;
    long long nb = (work + 255) / 256;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(theta_prep_zero_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, theta, G, C,
                       tab_theta, tab_lte, (float4*)zero0, bytes0 / 16, (float4*)zero1, bytes1 / 16);
    SCA_CHECK_LAUNCH();
    return 0;
}

// dst[cols][rows] = scale * src[rows][cols]
extern "C" int sca_transpose_scale(const float* src, int rows, int cols, float scale, float* dst, void* stream) {
    if (rows * cols <= 0) return 0;
    hipLaunchKernelGGL(transpose_scale_kernel, dim3((rows * cols + 255) / 256), dim3(256), 0, (hipStream_t)stream, src,
                       rows, cols, scale, dst);
    SCA_CHECK_LAUNCH();
    return 0;
}

// sca_recon_rows with the gamma-term table of sca_nb_gamma_table (gam_tab [C][G][17][2], NULL = none): integer counts
// <= 16 read their (lgamma, digamma) differences instead of running the per-count product loop.
extern "C" int sca_recon_rows_tab(int zinb, float* logits, int ldl, float* pi, const int64_t* indptr,
                                  const int32_t* indices, const float* data, const int64_t* row_ids,
                                  const float* sizefactor, const int32_t* cond, const float* tab_theta,
                                  const float* tab_lte, int B, int G, int C, int rows_per_block, float* row_ll,
                                  float* dthetaT, float* dbias, float* dbias_pi, int compute_grad,
                                  const float* cond_off, float* row_dsf, const float* gam_tab, void* stream) {
    if (B <= 0) return 0;
    if (rows_per_block < 1 || ldl < G) return -1;
    if (cond_off && !cond) return -1;
    ReconArgs a{logits, ldl, pi, indptr, indices, data, row_ids, sizefactor, cond, tab_theta, tab_lte,
                B, G, C, rows_per_block, row_ll, dthetaT, dbias, dbias_pi, compute_grad, cond_off, row_dsf,
                reinterpret_cast<const float2*>(gam_tab)};
    const int rc = zinb ? launch_recon<true>(a, (hipStream_t)stream)
                        : (gam_tab ? launch_recon<false, true>(a, (hipStream_t)stream)
                                   : launch_recon<false>(a, (hipStream_t)stream));
    if (rc) return rc;
    SCA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sca_recon_rows(int zinb, float* logits, int ldl, float* pi, const int64_t* indptr,
                              const int32_t* indices, const float* data, const int64_t* row_ids,
                              const float* sizefactor, const int32_t* cond, const float* tab_theta,
                              const float* tab_lte, int B, int G, int C, int rows_per_block, float* row_ll,
                              float* dthetaT, float* dbias, float* dbias_pi, int compute_grad,
                              const float* cond_off, float* row_dsf, void* stream) {
    return sca_recon_rows_tab(zinb, logits, ldl, pi, indptr, indices, data, row_ids, sizefactor, cond, tab_theta,
                              tab_lte, B, G, C, rows_per_block, row_ll, dthetaT, dbias, dbias_pi, compute_grad,
                              cond_off, row_dsf, nullptr, stream);
}

extern "C" int sca_nb_elem_rows(int act, float* lin, int ldl, const int64_t* indptr, const int32_t* indices,
                                const float* data, const int64_t* row_ids, const int32_t* cond,
                                const float* tab_theta, const float* tab_lte, int B, int G, int C, int rows_per_block,
                                float* row_ll, float* dthetaT, int compute_grad, const float* cond_off, void* stream) {
    if (B <= 0) return 0;
    if (rows_per_block < 1 || ldl < G || act < 0 || act > 3) return -1;
    if (cond_off && !cond) return -1;
    const size_t shm = (size_t)2 * G * sizeof(float);
    if (shm > 150 * 1024) return -2;
    ReconArgs a{lin, ldl, nullptr, indptr, indices, data, row_ids, nullptr, cond, tab_theta, tab_lte,
                B, G, C, rows_per_block, row_ll, dthetaT, nullptr, nullptr, compute_grad, cond_off, nullptr, nullptr};
    if (shm > 48 * 1024)
        hipFuncSetAttribute((const void*)nb_elem_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipLaunchKernelGGL(nb_elem_rows_kernel, dim3((B + rows_per_block - 1) / rows_per_block), dim3(256), shm,
                       (hipStream_t)stream, a, act);
    SCA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sca_poisson_rows(float* logits, int ldl, const int64_t* indptr, const int32_t* indices,
                                const float* data, const int64_t* row_ids, const float* sizefactor, int B, int G,
                                int rows_per_block, float* row_loss, int compute_grad, void* stream) {
    if (B <= 0) return 0;
    if (rows_per_block < 1 || ldl < G) return -1;
    hipLaunchKernelGGL(poisson_rows_kernel, dim3((B + rows_per_block - 1) / rows_per_block), dim3(256), 0,
                       (hipStream_t)stream, logits, ldl, indptr, indices, data, row_ids, sizefactor, B, G,
                       rows_per_block, row_loss, compute_grad);
    SCA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sca_mse_rows(float* lin, int ldl, const float* x, int ldx, int B, int G, int rows_per_block,
                            float* row_loss, float* dbias, int compute_grad, int relu, const int32_t* cond,
                            const float* cond_off, void* stream) {
    if (B <= 0) return 0;
    if (cond_off && !cond) return -1;
    if (relu < 0 || relu > 3) return -1;
    const int nblk = (B + rows_per_block - 1) / rows_per_block;
    hipLaunchKernelGGL(mse_rows_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, lin, ldl, x, ldx, B, G,
                       rows_per_block, row_loss, dbias, compute_grad, relu, cond, cond_off);
    SCA_CHECK_LAUNCH();
    return 0;
}
