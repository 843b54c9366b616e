// Output layer of the trVAE decoder fused with the negative-binomial likelihood and its backward (gfx950).
//
// Replaces, for recon_loss = "nb" (scarches/models/trvae):
//     logits = h W^T + b ; gamma = softmax(logits)                 modules.py:179-181,207
//     mu = gamma * sizefactor ; theta = exp(theta[:, c_b])         trvae.py:136-140
//     recon_b = -sum_g nb(x, mu, theta)                            losses.py:63-102, trvae.py:141
// and the autograd of all of it down to d loss / d logits, d theta and d output-bias - i.e. the output-layer GEMM
// launch (gemm_f32.hip) plus recon_rows_kernel (recon_loss.hip) of the unfused path.  Results agree with that pair
// to fp32 rounding (tests/test_gpu_out_fused.py).
//
// Design.  A workgroup (512 threads = 8 waves) owns 16 cells and ALL genes:
//   1. logits [16 x G] by v_mfma_f32_16x16x4_f32 into REGISTERS: wave w owns up to 16 tiles of 16 genes, a lane holds
//      column n = lane % 16 of each tile for the four cells 4 (lane / 16) .. + 3.  The 16 x K activation tile sits in
//      LDS (row pitch K + 8: conflict-free ds_read_b128 fragments), W streams from L2 straight into fragment registers:
//      with the k order (16 s + 4 kg + j) both operands of four consecutive MFMAs are ONE 16-byte load per lane.
//   2. softmax statistics with a wave-local maximum (the online form: one exp per element, one barrier), the dense
//      x = 0 part of the likelihood and of its gradient from the registers - per-gene quantities (theta, its logs and
//      reciprocals) are computed once per tile and serve four cells.
//   3. the stored counts of the 16 cells straight from CSR, one lane per count, a wave per cell: they need the logit of
//      their (cell, gene), which lives in another lane's register - so the raw logits make one trip through the
//      d-logits buffer (L2 / Infinity Cache: written once, read only at the ~14 % stored positions).  The gamma-function
//      terms of integer counts <= 16 come from a per-step table [C][G][17] (sca_nb_gamma_table: the same product /
//      harmonic forms as recon_loss.hip, evaluated once per (condition, gene, count) instead of per stored count);
//      everything else takes the fp64 Stirling path through a per-wave compaction queue.
//   4. d loss / d logits = (gamma T - t) / B is written from the registers, then the stored positions are patched with
//      their extra term (kept per count in LDS; rows with more than ON_RS stored counts spill to a global buffer).
// theta / bias gradients are summed per workgroup in LDS and leave as one coalesced atomic per gene when the 16 cells
// share a condition (minibatches arrive condition-sorted); a workgroup that straddles conditions adds per cell.
// The matrix work of one workgroup overlaps the vector work of its neighbour on the same CU (two workgroups per CU).
#include "common.h"
#include "nb_gamma.h"

namespace {

constexpr float ON_EPS = 1e-8f;
constexpr int ON_ROWS = 16;          // cells per workgroup
constexpr int ON_NW = 16;            // waves per workgroup (one workgroup per CU: four waves per SIMD)
constexpr int ON_T = ON_NW * 64;
constexpr int ON_TPW = 8;            // tiles of 16 genes per wave (32 accumulator registers): G <= 8 * 16 * 16 = 2048
constexpr int ON_GMAX = ON_TPW * 16 * ON_NW;
constexpr int ON_RS = 512;           // LDS slots per cell for the stored counts' extras
constexpr int ON_QW = 128;           // heavy-path compaction queue per wave (two chunks)
constexpr int ON_NX = 17;            // tabulated counts 0 .. 16
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.69314718055994530942f;

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct OutNbArgs {
    const float* h; int ldh;             // [B][K] last decoder activation
    const float* W; int ldw;             // [G][K] output weight (nn.Linear layout)
    const float* bias;                   // [G]
    const int64_t* indptr; const int32_t* indices; const float* data; const int64_t* row_ids;
    const float* sizefactor; const int32_t* cond;
    const float* tab_theta;              // [C][G] exp(theta)
    const float2* gam_tab;               // [C][G][17] (lgam, dig) per integer count; lgam = NaN: not tabulated
    int B, G, K, C;
    float* dl; int ldl;                  // [B][G]: scratch for the logits, then d loss / d logits
    float* ext_ovf; int ld_ovf;          // [B][>= G] extras of stored counts beyond ON_RS per cell
    float* row_ll;                       // [B] sum_g ll
    float* dthetaT;                      // [C][G] accumulates sum_b theta dll/dtheta (pre-zeroed)
    float* dbias;                        // [G] accumulates column sums of d logits (pre-zeroed; may be null)
    int compute_grad;
    long long* stamps;                   // debug: [workgroup][12] s_memrealtime marks of thread 0 (NULL in production)
    int abl;                             // debug ablation mask (SCA_OUT_ABL): timing probes only, results are wrong
};

// (lgam, dig) of integer counts 1..16 per (condition, gene): the loop of recon_loss.hip's fast path, run once per
// table entry.  theta outside (1e-30, 128) is not tabulated (NaN): those counts take the fp64 path.
__global__ __launch_bounds__(256) void nb_gamma_table_kernel(const float* __restrict__ tab_theta, long long n,
                                                             float2* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float th = tab_theta[i];
    const bool ok = th < 128.0f && th > 1e-30f;
    float2* o = out + i * ON_NX;
    const float qnan = __int_as_float(0x7fc00000);
    o[0] = make_float2(ok ? 0.f : qnan, 0.f);
    float num = 0.f, den = 1.f, fact = 1.f, f = th, fq = 1.f;
    for (int q = 0; q < ON_NX - 1; ++q) {
        num = fmaf(num, f, den);
        den *= f;
        fact *= fq;
        f += 1.0f;
        fq += 1.0f;
        const float rden = __builtin_amdgcn_rcpf(den);
        o[q + 1] = make_float2(ok ? -fast_logf(fact * rden) : qnan, num * rden);
    }
}

__device__ __forceinline__ float ld_l2(const float* p) {      // L1-bypassing load (data another wave just stored)
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float xor16(float v) { return __shfl_xor(v, 16, 64); }
__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32, 64); }

template <int CPP>      // K / 32: chunks of two 16-deep k-steps per tile pair
__global__ __launch_bounds__(ON_T, 4) void out_nb_fused_kernel(const OutNbArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int G = a.G;
    constexpr int K = 32 * CPP, PITCH = K + 8;
    const int Gp = (G + 3) & ~3;
    float* hA = lds;                                  // [16][PITCH]
    float* dxs = hA + ON_ROWS * PITCH;                // [Gp] theta gradient of the workgroup (one condition)
    float* dbs = dxs + Gp;                            // [Gp] stored counts' share of the bias gradient, then the total
    float* ext = dbs + Gp;                            // [16][ON_RS]
    int* hq = reinterpret_cast<int*>(ext + ON_ROWS * ON_RS);     // [ON_NW][ON_QW]
    __shared__ float s_m[ON_NW][ON_ROWS], s_s[ON_NW][ON_ROWS], s_ll[ON_NW][ON_ROWS], s_T[ON_NW][ON_ROWS];
    __shared__ float s_xll[ON_ROWS], s_xT[ON_ROWS], s_sf[ON_ROWS];
    __shared__ long long s_ks[ON_ROWS];
    __shared__ int s_nn[ON_ROWS], s_c[ON_ROWS], s_hqn[ON_NW];

    const int t = threadIdx.x, lane = t & 63, w = t >> 6, n = lane & 15, kg = lane >> 4;
    const int r0 = blockIdx.x * ON_ROWS, nrows = min(ON_ROWS, a.B - r0);
    const float invB = 1.0f / (float)a.B;
#define ON_STAMP(i) do { if (a.stamps && t == 0) a.stamps[(long long)blockIdx.x * 12 + (i)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
    ON_STAMP(0);

    // ---- 0. row scalars, activation tile, cleared accumulators
    if (t < ON_ROWS) {
        long long k0 = 0; int nn = 0, c = 0; float sf = 0.f;
        if (t < nrows) {
            const int b = r0 + t;
            const int64_t r = a.row_ids ? a.row_ids[b] : (int64_t)b;
            k0 = a.indptr[r];
            nn = (int)(a.indptr[r + 1] - k0);
            c = a.cond ? a.cond[b] : 0;
            sf = a.sizefactor[b];
        }
        s_ks[t] = k0; s_nn[t] = nn; s_c[t] = c; s_sf[t] = sf;
        s_xll[t] = 0.f; s_xT[t] = 0.f;
    }
    if (t < ON_NW) s_hqn[t] = 0;
    for (int i = t; i < 2 * Gp; i += ON_T) dxs[i] = 0.f;
    {
        const int k4 = K >> 2;
        for (int i = t; i < ON_ROWS * k4; i += ON_T) {
            const int row = i / k4, c4 = i - row * k4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < nrows) v = *reinterpret_cast<const float4*>(a.h + (long long)(r0 + row) * a.ldh + 4 * c4);
            *reinterpret_cast<float4*>(hA + row * PITCH + 4 * c4) = v;
        }
    }
    __syncthreads();
    ON_STAMP(1);
    const int c0 = s_c[0];
    bool uni = true;
#pragma unroll
    for (int i = 1; i < ON_ROWS; ++i) uni = uni && (i >= nrows || s_c[i] == c0);

    // ---- 1. logits of this wave's tiles: acc[tile][r] = cell 4 kg + r, gene 16 (T0 + tile) + n
    const int NT = (G + 15) >> 4, tpw = (NT + ON_NW - 1) / ON_NW;
    const int T0 = w * tpw, ntw = max(0, min(tpw, NT - T0));
    f32x4 acc[ON_TPW];
    {
        // A chunk = two 16-deep k-steps (2 x 8 MFMAs) of a PAIR of tiles = four 16-byte W loads per lane.  The loads of
        // chunk i + 1 are issued before chunk i feeds the MFMAs (unconditional, fixed in number: counted vmcnt waits; a
        // chunk is 512 cycles of matrix work and four waves share the SIMD's pipe, so ~2000 cycles pass before the next
        // chunk is needed); tiles past the wave's last one read clamped rows, their accumulators are discarded below.
        const float* arow = hA + n * PITCH + 4 * kg;
        constexpr int NCH = (ON_TPW / 2) * CPP;
        unsigned woff[ON_TPW];           // byte offsets of this lane's W rows
#pragma unroll
        for (int tl = 0; tl < ON_TPW; ++tl) {
            const int gg = 16 * (T0 + tl) + n;
            woff[tl] = ((unsigned)min(gg, G - 1) * (unsigned)a.ldw + 4u * kg) * 4u;
            const float b = a.bias[min(gg, G - 1)];
            acc[tl] = f32x4{b, b, b, b};
        }
        float4 ring[2][4];
        auto issue = [&](int ci, float4 (&slot)[4]) {
            const int tp = ci / CPP, q = ci % CPP;
            const float* w0 = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.W) + (woff[2 * tp] + 128u * q));
            const float* w1 = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.W) + (woff[2 * tp + 1] + 128u * q));
            slot[0] = *reinterpret_cast<const float4*>(w0);
            slot[1] = *reinterpret_cast<const float4*>(w0 + 16);
            slot[2] = *reinterpret_cast<const float4*>(w1);
            slot[3] = *reinterpret_cast<const float4*>(w1 + 16);
        };
        auto mma8 = [&](f32x4& x0, f32x4& x1, const float4& av, const float4& p0, const float4& p1) {
            x0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, p0.x, x0, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, p1.x, x1, 0, 0, 0);
            x0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, p0.y, x0, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, p1.y, x1, 0, 0, 0);
            x0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, p0.z, x0, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, p1.z, x1, 0, 0, 0);
            x0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, p0.w, x0, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, p1.w, x1, 0, 0, 0);
        };
        if (!(a.abl & 1)) {
            issue(0, ring[0]);
#pragma unroll
            for (int ci = 0; ci < NCH; ++ci) {
                if (ci + 1 < NCH) issue(ci + 1, ring[(ci + 1) % 2]);
                const int tp = ci / CPP, q = ci % CPP;
                const float4 av0 = *reinterpret_cast<const float4*>(arow + 32 * q);
                const float4 av1 = *reinterpret_cast<const float4*>(arow + 32 * q + 16);
                mma8(acc[2 * tp], acc[2 * tp + 1], av0, ring[ci % 2][0], ring[ci % 2][2]);
                mma8(acc[2 * tp], acc[2 * tp + 1], av1, ring[ci % 2][1], ring[ci % 2][3]);
            }
        }
#pragma unroll
        for (int tl = 0; tl < ON_TPW; ++tl)
            if (!(tl < ntw && 16 * (T0 + tl) + n < G)) acc[tl] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    }
    ON_STAMP(2);
    // ---- 2. raw logits -> d-logits buffer (the stored counts read theirs back), wave-local softmax statistics
    // (32-bit byte offsets from the uniform base pointer: one address register per store instead of a 64-bit pair
    //  per (cell, tile) that the compiler would keep alive until the second pass; B * ldl < 2^30, checked by the host)
    unsigned rowb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rowb[r] = (unsigned)(r0 + min(4 * kg + r, nrows - 1)) * (unsigned)a.ldl * 4u;
    auto DL = [&](unsigned byteoff) -> float& { return *reinterpret_cast<float*>(reinterpret_cast<char*>(a.dl) + byteoff); };
    float mw[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int tl = 0; tl < ON_TPW; ++tl) {
        if (tl < ntw) {
            const int g = 16 * (T0 + tl) + n;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * kg + r;
                if (row < nrows && g < G && !(a.abl & 32)) DL(rowb[r] + 4u * (unsigned)g) = acc[tl][r];
                mw[r] = fmaxf(mw[r], acc[tl][r]);
            }
        }
    }
    float sw[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) mw[r] = row16_max(mw[r]);
#pragma unroll
    for (int tl = 0; tl < ON_TPW; ++tl) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // exp(logit - wave maximum); tiles / genes beyond the end hold -inf -> 0
            const float e = (tl < ntw && mw[r] > -INFINITY) ? __builtin_amdgcn_exp2f((acc[tl][r] - mw[r]) * LOG2E) : 0.f;
            acc[tl][r] = e;
            sw[r] += e;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        sw[r] = row16_sum(sw[r]);
        if (n == 0) { s_m[w][4 * kg + r] = mw[r]; s_s[w][4 * kg + r] = sw[r]; }
    }
    ON_STAMP(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's logits have reached L2
    __syncthreads();
    ON_STAMP(4);

    // row statistics M = max_w m_w, S = sum_w s_w exp(m_w - M)  (every lane, for any row it needs)
    auto row_stats = [&](int row, float& M, float& iS) {
        float m = s_m[0][row];
#pragma unroll
        for (int i = 1; i < ON_NW; ++i) m = fmaxf(m, s_m[i][row]);
        float S = 0.f;
#pragma unroll
        for (int i = 0; i < ON_NW; ++i) {
            const float mi = s_m[i][row];
            S += mi > -INFINITY ? s_s[i][row] * __builtin_amdgcn_exp2f((mi - m) * LOG2E) : 0.f;
        }
        M = m;
        iS = 1.0f / S;
    };
    float fgam[4], fmu[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float M, iS;
        row_stats(4 * kg + r, M, iS);
        fgam[r] = mw[r] > -INFINITY ? __builtin_amdgcn_exp2f((mw[r] - M) * LOG2E) * iS : 0.f;
        fmu[r] = fgam[r] * s_sf[4 * kg + r];
    }

    ON_STAMP(5);
    // ---- 3. stored counts: wave w takes cell w, one lane per count
    if (w < nrows && !(a.abl & 2)) {
        const int rr = w;
        const long long ks = s_ks[rr];
        const int nn = s_nn[rr], c = s_c[rr];
        const float s = s_sf[rr];
        float M, iS;
        row_stats(rr, M, iS);
        const float* tth = a.tab_theta + (long long)c * G;
        const float2* gt = a.gam_tab + (long long)c * G * ON_NX;
        float* lrow = a.dl + (long long)(r0 + rr) * a.ldl;
        float all = 0.f, aT = 0.f;
        // three-stage software pipeline over chunks of 64 counts: (A) count + gene index of chunk i + 2, (B) the loads
        // that depend on them - logit, theta, table entry - of chunk i + 1, (C) the arithmetic of chunk i.  Every load
        // is unconditional from a clamped address, so a chunk's dependent latencies overlap the previous chunks' work.
        struct StA { float x; int g; };
        struct StB { float x, lv, th; int g; float2 tv; bool isint; };
        auto loadA = [&](int base) {
            StA r;
            const int jj = min(base + lane, nn - 1);
            r.x = a.data[ks + jj];
            r.g = a.indices[ks + jj];
            return r;
        };
        auto loadB = [&](const StA& p) {
            StB r;
            r.x = p.x; r.g = p.g;
            r.isint = p.x <= 16.0f && p.x == floorf(p.x);
            r.lv = ld_l2(lrow + p.g);
            r.th = tth[p.g];
            r.tv = gt[(long long)p.g * ON_NX + (r.isint ? (int)p.x : 0)];
            return r;
        };
        if (nn > 0) {
            StA a1 = loadA(0);
            StA a2 = loadA(64);
            StB b0 = loadB(a1);
            for (int base = 0; base < nn; base += 64) {
                const StA a3 = loadA(base + 128);
                const StB b1 = loadB(a2);
                a2 = a3;
                const int j = base + lane;
                float ex_t = 0.f;
                if (j < nn && b0.x > ON_EPS) {
                    // (large / non-integer counts, theta outside the table: their gamma terms are added by the fp64
                    //  sweep at the end of the kernel - neither T nor d-logits depends on them)
                    const bool fast = b0.isint && b0.tv.x == b0.tv.x;
                    const float lgam = fast ? b0.tv.x : 0.f, dig = fast ? b0.tv.y : 0.f;
                    const float mu = __builtin_amdgcn_exp2f((b0.lv - M) * LOG2E) * iS * s;
                    const float den = b0.th + mu + ON_EPS;
                    const float L = fast_logf(den), rcp = __builtin_amdgcn_rcpf(den);
                    all += b0.x * (fast_logf(mu + ON_EPS) - L) + lgam;
                    ex_t = mu * b0.x * (__builtin_amdgcn_rcpf(mu + ON_EPS) - rcp);
                    aT += ex_t;
                    if (a.compute_grad) {
                        const float ex_th = b0.th * (-b0.x * rcp + dig);
                        if (uni) atomicAdd(&dxs[b0.g], ex_th);
                        else atomicAdd(&a.dthetaT[(long long)c * G + b0.g], ex_th);
                        atomicAdd(&dbs[b0.g], ex_t);
                    }
                }
                if (a.compute_grad && j < nn) {
                    if (j < ON_RS) ext[rr * ON_RS + j] = ex_t;
                    else a.ext_ovf[(long long)(r0 + rr) * a.ld_ovf + j] = ex_t;
                }
                b0 = b1;
            }
        }
        all = wave_sum(all);
        aT = wave_sum(aT);
        if (lane == 0) { s_xll[rr] = all; s_xT[rr] = aT; }
    }
    ON_STAMP(6);
    __syncthreads();
    ON_STAMP(7);

    // ---- 4. dense sweep (the x = 0 terms of every gene), first pass: row sums and the theta gradient
    float llr[4] = {0.f, 0.f, 0.f, 0.f}, Tr[4] = {0.f, 0.f, 0.f, 0.f};
    int cr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) cr[r] = s_c[4 * kg + r];
#pragma unroll
    for (int tl = 0; tl < ON_TPW; ++tl) {
        if (tl < ntw && !(a.abl & 4)) {
            const int g = 16 * (T0 + tl) + n, gi = min(g, G - 1);
            if (uni) {
                const float th = a.tab_theta[(long long)c0 * G + gi];
                // theta (log(theta + eps) - log(theta + mu + eps)) = -theta log(1 + mu / (theta + eps)): exactly 0 at mu = 0
                // (padded genes, cells past the end of the minibatch), absolute error ~ theta * 1e-7 as the ratio form
                const float the = th + ON_EPS, r1 = __builtin_amdgcn_rcpf(the);
                const float nthl = -th * LN2, q1 = th * r1;
                float dsum = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float mu = acc[tl][r] * fmu[r];
                    const float den = the + mu;
                    const float rcp = __builtin_amdgcn_rcpf(den);
                    const float base = nthl * __builtin_amdgcn_logf(fmaf(mu, r1, 1.0f));
                    const float u = -th * rcp;
                    llr[r] += base;
                    Tr[r] = fmaf(mu, u, Tr[r]);
                    dsum += fmaf(th, q1 + u, base);
                }
                if (a.compute_grad) {
                    dsum += xor16(dsum);
                    dsum += xor32(dsum);
                    if (kg == 0 && g < G) dxs[g] += dsum;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float th = a.tab_theta[(long long)cr[r] * G + gi];
                    const float the = th + ON_EPS, r1 = __builtin_amdgcn_rcpf(the);
                    const float mu = acc[tl][r] * fmu[r];
                    const float den = the + mu;
                    const float rcp = __builtin_amdgcn_rcpf(den);
                    const float base = -th * LN2 * __builtin_amdgcn_logf(fmaf(mu, r1, 1.0f));
                    const float u = -th * rcp;
                    llr[r] += base;
                    Tr[r] = fmaf(mu, u, Tr[r]);
                    const float dth = fmaf(th, fmaf(th, r1, u), base);
                    if (a.compute_grad && 4 * kg + r < nrows && g < G)
                        atomicAdd(&a.dthetaT[(long long)cr[r] * G + g], dth);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        llr[r] = row16_sum(llr[r]);
        Tr[r] = row16_sum(Tr[r]);
        if (n == 0) { s_ll[w][4 * kg + r] = llr[r]; s_T[w][4 * kg + r] = Tr[r]; }
    }
    __syncthreads();
    ON_STAMP(8);
    float Tt[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float v = s_xT[4 * kg + r];
#pragma unroll
        for (int i = 0; i < ON_NW; ++i) v += s_T[i][4 * kg + r];
        Tt[r] = v * invB;
    }

    // ---- 5. second pass: d loss / d logits = (gamma T - t) / B from the registers
#pragma unroll
    for (int tl = 0; tl < ON_TPW; ++tl) {
        if (tl < ntw && a.compute_grad && !(a.abl & 8)) {
            const int g = 16 * (T0 + tl) + n, gi = min(g, G - 1);
            float thu = 0.f;
            if (uni) thu = a.tab_theta[(long long)c0 * G + gi];
            float bsum = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float th = uni ? thu : a.tab_theta[(long long)cr[r] * G + gi];
                const float e = acc[tl][r];
                const float mu = e * fmu[r];
                const float rcp = __builtin_amdgcn_rcpf(th + ON_EPS + mu);
                const float tv = -(mu * th) * rcp * invB;
                const float dlv = fmaf(e * fgam[r], Tt[r], -tv);
                const int row = 4 * kg + r;
                if (row < nrows && g < G) {
                    DL(rowb[r] + 4u * (unsigned)g) = dlv;
                    bsum += dlv;
                }
            }
            bsum += xor16(bsum);
            bsum += xor32(bsum);
            if (kg == 0 && g < G) dbs[g] = bsum - invB * dbs[g];
        }
    }
    ON_STAMP(9);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ON_STAMP(10);

    // ---- 6. stored positions: subtract their extra term (same wave, same lane as in phase 3); gradient flush
    if (w < nrows && a.compute_grad && !(a.abl & 16)) {
        const int rr = w;
        const long long ks = s_ks[rr];
        const int nn = s_nn[rr];
        float* lrow = a.dl + (long long)(r0 + rr) * a.ldl;
        for (int j = lane; j < nn; j += 64) {
            const float ex_t = j < ON_RS ? ext[rr * ON_RS + j] : a.ext_ovf[(long long)(r0 + rr) * a.ld_ovf + j];
            if (ex_t != 0.f) {
                const int g = a.indices[ks + j];
                lrow[g] = ld_l2(lrow + g) - ex_t * invB;
            }
        }
    }
    // ---- 7. gamma terms of the counts the table does not hold (large / non-integer counts, theta out of range): the
    //         cell's counts are scanned once more, the hits compacted through a per-wave queue so that the fp64 code
    //         runs on full lanes.  Here - not in phase 3 - because the accumulator registers are dead by now.
    float heavy_ll = 0.f;
    if (w < nrows && !(a.abl & 2)) {
        const int rr = w;
        const long long ks = s_ks[rr];
        const int nn = s_nn[rr], c = s_c[rr];
        const float* tth = a.tab_theta + (long long)c * G;
        const float2* gt = a.gam_tab + (long long)c * G * ON_NX;
        int* q = hq + w * ON_QW;
        int nq = 0;                                      // wave-uniform
        auto drain = [&](int cnt) {                      // fp64 terms of q[0 .. cnt)
            if (lane < cnt) {
                const int j = q[lane];
                const float x = a.data[ks + j];
                const int g = a.indices[ks + j];
                const float th = tth[g];
                float lgam, dig;
                nb_gamma_terms_heavy(x, th, &lgam, &dig);
                heavy_ll += lgam;
                if (a.compute_grad) {
                    if (uni) atomicAdd(&dxs[g], th * dig);
                    else atomicAdd(&a.dthetaT[(long long)c * G + g], th * dig);
                }
            }
        };
        for (int base = 0; base < nn; base += 64) {
            const int j = base + lane;
            bool need = false;
            if (j < nn) {
                const float x = a.data[ks + j];
                if (x > ON_EPS) {
                    const bool isint = x <= 16.0f && x == floorf(x);
                    if (isint) {
                        const float lg = gt[(long long)a.indices[ks + j] * ON_NX + (int)x].x;
                        need = !(lg == lg);
                    } else {
                        need = true;
                    }
                }
            }
            const unsigned long long bal = __ballot(need);
            if (bal) {
                if (need) q[nq + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0))] = j;
                nq += __popcll(bal);
                __builtin_amdgcn_wave_barrier();
                if (nq >= 64) {
                    drain(64);
                    __builtin_amdgcn_wave_barrier();
                    if (lane < nq - 64) q[lane] = q[64 + lane];
                    nq -= 64;
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        drain(nq);
        heavy_ll = wave_sum(heavy_ll);
    }
    if (w < nrows && lane == 0) {
        float ll = s_xll[w] + heavy_ll;
#pragma unroll
        for (int i = 0; i < ON_NW; ++i) ll += s_ll[i][w];
        a.row_ll[r0 + w] = ll;
    }
    if (!a.compute_grad) return;
    __syncthreads();
    for (int g = t; g < G; g += ON_T) {
        if (uni) atomicAdd(&a.dthetaT[(long long)c0 * G + g], dxs[g]);
        if (a.dbias) atomicAdd(&a.dbias[g], dbs[g]);
    }
    ON_STAMP(11);
}

}  // namespace

static long long* g_on_stamps = nullptr;
// development aid: 12 s_memrealtime marks per workgroup of the next launches (NULL: off, the default)
extern "C" int sca_out_nb_fused_debug_stamps(long long* buf) { g_on_stamps = buf; return 0; }

extern "C" int sca_out_nb_fused_supported(int G, int K) {
    return (G >= 1 && G <= ON_GMAX && (K == 32 || K == 64 || K == 128 || K == 256)) ? 1 : 0;
}

// gam_tab [C][G][17][2] floats from tab_theta [C][G] (sca_theta_prep): once per step, before sca_out_nb_fused.
extern "C" int sca_nb_gamma_table(const float* tab_theta, int C, int G, float* gam_tab, void* stream) {
    if (C <= 0 || G <= 0) return 0;
    const long long n = (long long)C * G;
    hipLaunchKernelGGL(nb_gamma_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       tab_theta, n, reinterpret_cast<float2*>(gam_tab));
    SCA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sca_out_nb_fused(const float* h, int ldh, const float* W, int ldw, const float* bias,
                                const int64_t* indptr, const int32_t* indices, const float* data,
                                const int64_t* row_ids, const float* sizefactor, const int32_t* cond,
                                const float* tab_theta, const float* gam_tab, int B, int G, int K, int C, float* dl,
                                int ldl, float* ext_ovf, int ld_ovf, float* row_ll, float* dthetaT, float* dbias,
                                int compute_grad, void* stream) {
    if (B <= 0) return 0;
    if (!sca_out_nb_fused_supported(G, K)) return -2;
    if (C < 1 || ldh < K || ldw < K || ldl < G || !h || !W || !bias || !dl || !row_ll || !tab_theta || !gam_tab)
        return -1;
    if ((ldh & 3) || (ldw & 3) || ((uintptr_t)h & 15) || ((uintptr_t)W & 15)) return -2;
    if (compute_grad && (!dthetaT || !ext_ovf || ld_ovf < G)) return -1;
    if ((long long)B * ldl >= (1ll << 30) || (long long)G * ldw >= (1ll << 30)) return -2;      // 32-bit byte offsets
    OutNbArgs a{h, ldh, W, ldw, bias, indptr, indices, data, row_ids, sizefactor, cond, tab_theta,
                reinterpret_cast<const float2*>(gam_tab), B, G, K, C, dl, ldl, ext_ovf, ld_ovf, row_ll, dthetaT, dbias,
                compute_grad, g_on_stamps, [] { const char* v = getenv("SCA_OUT_ABL"); return v ? atoi(v) : 0; }()};
    const int Gp = (G + 3) & ~3;
    const size_t shm = sizeof(float) * ((size_t)ON_ROWS * (K + 8) + 2 * (size_t)Gp + (size_t)ON_ROWS * ON_RS +
                                        (size_t)ON_NW * ON_QW);
    if (shm > 96 * 1024) return -2;
    const dim3 grid((B + ON_ROWS - 1) / ON_ROWS), block(ON_T);
    hipStream_t st = (hipStream_t)stream;
#define ON_LAUNCH(CPP)                                                                                              \
    do {                                                                                                            \
        static const hipError_t attr = hipFuncSetAttribute((const void*)out_nb_fused_kernel<CPP>,                    \
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);  \
        if (attr != hipSuccess) return (int)attr;                                                                   \
        hipLaunchKernelGGL(out_nb_fused_kernel<CPP>, grid, block, shm, st, a);                                      \
    } while (0)
    if (K == 32) ON_LAUNCH(1);
    else if (K == 64) ON_LAUNCH(2);
    else if (K == 128) ON_LAUNCH(4);
    else ON_LAUNCH(8);
#undef ON_LAUNCH
    SCA_CHECK_LAUNCH();
    return 0;
}
