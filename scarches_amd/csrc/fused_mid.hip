// Fused middle of the conditional VAE (gfx950): everything between the two wide GEMMs of a step, one launch per
// direction.
//
// For G >> H the network is two wide layers (genes -> H, H -> genes) around a chain of narrow ones: the remaining
// encoder layers, the mean / log-var heads, the reparameterised sample + KL, and the decoder layers
// (scarches/models/trvae/modules.py:52-97 Encoder, :124-208 Decoder, models/base/_base.py:391-407 sampling,
// models/trvae/trvae.py:143-147 KL).  As separate GEMM + epilogue launches that chain is ~16 launches of a few us
// each in both directions - launch- and latency-bound, a third of the step at B = 4096.  Here a workgroup owns a tile of
// 16 cells and walks the whole chain with the activations in LDS: weights stream through LDS in 64-deep chunks, each
// wave owns two 16 x 16 v_mfma_f32_16x16x4_f32 output blocks, the row-wise epilogues (bias, condition column, LayerNorm,
// ReLU, dropout; their backward) are the ones of layers.hip.  (16 cells, not 32: the kernels are latency-bound chains,
// and at 8192 cells per step 512 workgroups put two independent chains on every SIMD where 256 left one.)
// Everything the backward pass and the weight-gradient GEMMs need (normalised pre-activations, rstd, activations,
// mean / log-var, noise, z, per-layer `da`) is still written to HBM; only the re-reads and the launches disappear.
#include "common.h"

#include "mid_args.h"

#ifdef SCA_MID_DEBUG
__device__ long long g_mid_dbg[64];
#define DBG(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { g_mid_dbg[i] = wall_clock64(); g_mid_dbg[32 + (i)] = clock64(); } } while (0)
#else
#define DBG(i)
#endif

namespace {

__device__ __attribute__((aligned(16))) float g_mid_blank[4];   // what an unconditional prefetch reads for an absent operand

#ifndef MID_WAVE_PRIO
#define MID_WAVE_PRIO 3
#endif
constexpr int TM = 16;          // cells per workgroup
constexpr int TPR = 256 / TM;   // threads per row in the row epilogues
constexpr int CG = 4 * TPR;     // columns covered by one 4-column group per thread of a row
constexpr int KC = 32;          // reduction chunk staged in LDS
constexpr int MS_LD = 129;      // leading dim of the staged weight chunk [KC][128] (+1: conflict-free transposed store)
constexpr int MAXW = 256;       // widest supported layer
constexpr int NST = KC * 128 / 256;   // staged weight values per thread and chunk

// out[r][j] = sum_i act[r][i] * M(i, j), r < TM, j < J, i < I.
//   FWD : M(i, j) = W[j*ldW + i]   (y = x W^T: i = input feature, j = output unit)
//   !FWD: M(i, j) = W[i*ldW + j]   (dx = da W: i = output unit, j = input feature)
// Every prefetch load is UNCONDITIONAL - out-of-range lanes read a clamped address - and its value stays RAW in its
// register until the consumer (stage_store, the row epilogues) drops or uses it: a fixed number of loads in flight
// lets the compiler wait for an older prefetch with an exact vmcnt, and no instruction between issue and consumer
// touches the value, so the wait really sits at the consumer.  (A conditional load, or a select / multiply right after
// it, puts the full L2 latency behind every issue: 3 - 7 us per weight chunk when 256 workgroups ask for the same lines.)
template <bool FWD>
__device__ __forceinline__ void stage_load(float (&v)[NST], const float* __restrict__ W, int ldW, int i0, int in_,
                                           int j0, int jn) {
    const int tid = threadIdx.x;
    if (FWD) {
        const int iic = min(tid & (KC - 1), in_ - 1);
#pragma unroll
        for (int q = 0; q < NST; ++q) {
            const int jjc = min(tid / KC + (256 / KC) * q, jn - 1);
            v[q] = W[(long long)(j0 + jjc) * ldW + i0 + iic];
        }
    } else {
        const int jjc = min(tid & 127, jn - 1);
#pragma unroll
        for (int q = 0; q < NST; ++q) {
            const int iic = min((tid >> 7) + 2 * q, in_ - 1);
            v[q] = W[(long long)(i0 + iic) * ldW + j0 + jjc];
        }
    }
}

// Chunk rows past in_ and columns past jn are stored as zeros (the raw registers hold clamped re-reads there).
template <bool FWD>
__device__ __forceinline__ void stage_store(const float (&v)[NST], float* __restrict__ Ms, int in_, int jn) {
    const int tid = threadIdx.x;
    if (FWD) {
        const int ii = tid & (KC - 1);
#pragma unroll
        for (int q = 0; q < NST; ++q) {
            const int jj = tid / KC + (256 / KC) * q;
            Ms[ii * MS_LD + jj] = (ii < in_ && jj < jn) ? v[q] : 0.f;
        }
    } else {
        const int jj = tid & 127;
#pragma unroll
        for (int q = 0; q < NST; ++q) {
            const int ii = (tid >> 7) + 2 * q;
            Ms[ii * MS_LD + jj] = (ii < in_ && jj < jn) ? v[q] : 0.f;
        }
    }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Two v_mfma_f32_16x16x4_f32 output blocks per wave: wave w owns columns [32w, 32w+32) of the current 128-column block
// as two 16-column halves, all TM = 16 rows.  Operand layout of the instruction: A lane (l16, kg) = act[l16][k + kg],
// B lane (l16, kg) = M(k + kg, l16); accumulator register r of lane (l16, kg) = out[4 kg + r][l16].
// First weight chunk of a tile_gemm, requested early: the caller issues it BEFORE the row epilogue (or tile load) that
// precedes the GEMM, so its L2 / HBM latency is hidden behind that phase instead of opening every GEMM.
template <bool FWD>
__device__ __forceinline__ void tile_gemm_prefetch(float (&nxt)[NST], const float* __restrict__ W, int ldW, int I, int J) {
    stage_load<FWD>(nxt, W, ldW, 0, min(KC, I), 0, min(128, J));
}

template <bool FWD>
__device__ __forceinline__ void tile_gemm(const float* __restrict__ act, int lda, int I, const float* __restrict__ W,
                                          int ldW, int J, float* __restrict__ out, int ldo, float* __restrict__ Ms,
                                          float (&nxt)[NST]) {
    // `nxt` arrives holding chunk (i0 = 0, j0 = 0) - tile_gemm_prefetch - and is consumed
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l16 = lane & 15, kg = lane >> 4;
    for (int j0 = 0; j0 < J; j0 += 128) {
        const int jn = min(128, J - j0);
        const bool has_cols = 32 * w < jn;          // narrow outputs (heads: 2 Z columns) leave whole waves without work
        f32x4 acc[2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[c][r] = 0.f;
        if (j0 > 0) stage_load<FWD>(nxt, W, ldW, 0, min(KC, I), j0, jn);
        for (int i0 = 0; i0 < I; i0 += KC) {
            __syncthreads();                         // previous chunk consumed (and `act` complete on first entry)
            stage_store<FWD>(nxt, Ms, min(KC, I - i0), jn);
            __syncthreads();
            if (i0 + KC < I) stage_load<FWD>(nxt, W, ldW, i0 + KC, min(KC, I - i0 - KC), j0, jn);
            if (!has_cols) continue;
            const float* ap = act + l16 * lda + i0 + kg;
            const float* bp = Ms + kg * MS_LD + 32 * w + l16;
            const int kn = (min(KC, I - i0) + 3) >> 2;      // k quads that hold data (z -> first decoder layer: 3 of 16)
            if (kn == KC / 4) {
                float av[KC / 4], bv[KC / 4][2];
#pragma unroll
                for (int s = 0; s < KC / 4; ++s) {
                    av[s] = ap[4 * s];
                    bv[s][0] = bp[4 * s * MS_LD];
                    bv[s][1] = bp[4 * s * MS_LD + 16];
                }
                // every fragment read issued before the first MFMA: left alone, hipcc interleaves them - read, wait for
                // the LDS round trip, two MFMAs, read, ... - and the 16 MFMAs of a chunk take 1250 cycles instead of 512
                static_assert(KC == 32, "the pin below names the 24 fragment registers of a 32-deep chunk");
                asm volatile("" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(av[4]), "+v"(av[5]), "+v"(av[6]),
                                  "+v"(av[7]), "+v"(bv[0][0]), "+v"(bv[0][1]), "+v"(bv[1][0]), "+v"(bv[1][1]), "+v"(bv[2][0]),
                                  "+v"(bv[2][1]), "+v"(bv[3][0]), "+v"(bv[3][1]), "+v"(bv[4][0]), "+v"(bv[4][1]),
                                  "+v"(bv[5][0]), "+v"(bv[5][1]), "+v"(bv[6][0]), "+v"(bv[6][1]), "+v"(bv[7][0]),
                                  "+v"(bv[7][1]));
#pragma unroll
                for (int s = 0; s < KC / 4; ++s) {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s][0], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s][1], acc[1], 0, 0, 0);
                }
            } else {
                // (columns of `act` and rows of the staged chunk past the data are zero: rounding kn up to 4 is harmless)
                for (int s4 = 0; s4 < kn; s4 += 4) {
                    float av[4], bv[4][2];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        av[s] = ap[4 * (s4 + s)];
                        bv[s][0] = bp[4 * (s4 + s) * MS_LD];
                        bv[s][1] = bp[4 * (s4 + s) * MS_LD + 16];
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s][0], acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s][1], acc[1], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int n = 32 * w + 16 * c + l16;
            if (n < jn) {
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(4 * kg + r) * ldo + j0 + n] = acc[c][r];
            }
        }
    }
    __syncthreads();
}

// act tiles are [TM][ld] with ld = round_up(width, KC) + 2: the A fragment of lane (l16, kg) sits at l16 * ld + kg, and
// ld = 2 (mod 32) puts the 32 lanes of a half-wave (kg = 0, 1) on 32 different banks (+1 left them 2-way conflicted); columns past the layer width must read as zero for the
// K-chunk loop above, so every producer of a tile zero-fills up to the next multiple of KC.

// 4 dropout keep decisions of columns col4 .. col4+3 (col4 % 4 == 0) from ONE Philox call - same bits as dropout_keep.
__device__ __forceinline__ void dropout_keep4(uint64_t seed, uint32_t stream, uint32_t row, uint32_t col4, float p,
                                              bool (&k)[4]) {
    const Philox4 r = philox4x32_10(row, col4 >> 2, stream, 0x5ca1ab1eu, (uint32_t)seed, (uint32_t)(seed >> 32));
    k[0] = u32_to_unit(r.x) > p; k[1] = u32_to_unit(r.y) > p; k[2] = u32_to_unit(r.z) > p; k[3] = u32_to_unit(r.w) > p;
}

__device__ __forceinline__ float row_sum(float v) {       // sum over the TPR consecutive lanes that share a row
    static_assert(TPR == 16, "one DPP row of 16 lanes per cell (row16_sum)");
    return row16_sum(v);                                    // four DPP steps, no ds_bpermute round trips
}

// Copy a [TM][width] row-major HBM tile into LDS (zero padded to Kp columns, rows past B read as zero); 16-byte loads
// when the row pitch allows it.  `S` slabs `stride` apart are summed on the way in.
__device__ __forceinline__ void load_tile(float* __restrict__ dst, int ld, const float* __restrict__ src, int S,
                                          long long stride, int r0, int B, int width, int Kp) {
    if ((width & 3) == 0 && ((uintptr_t)src & 15) == 0 && (stride & 3) == 0) {
        // NP positions of a thread x four slabs = 16 independent 16-byte loads in flight (a serial position / slab walk
        // pays one HBM latency per load: 11 us for the slabs of the first layer at B = 8192)
        constexpr int NP = 4;
        const int w4 = Kp >> 2;
        for (int t0 = threadIdx.x; t0 < TM * w4; t0 += 256 * NP) {
            const float* p[NP];
            float4 v[NP];
            int off[NP];
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int t = t0 + 256 * i;
                const int r = t / w4, k = (t - r * w4) * 4;
                const bool ok = t < TM * w4 && k < width && r0 + r < B;
                p[i] = ok ? src + (long long)(r0 + r) * width + k : nullptr;
                off[i] = t < TM * w4 ? r * ld + k : -1;
                v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            // slabs in groups of four (16 loads in flight), summed in slab order
            for (int q = 0; q < S; q += 4) {
                float4 u[NP][4];
#pragma unroll
                for (int i = 0; i < NP; ++i)
#pragma unroll
                    for (int h = 0; h < 4; ++h)
                        u[i][h] = (p[i] && q + h < S) ? *reinterpret_cast<const float4*>(p[i] + (q + h) * stride)
                                                      : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int i = 0; i < NP; ++i)
#pragma unroll
                    for (int h = 0; h < 4; ++h)
                        if (q + h < S) { v[i].x += u[i][h].x; v[i].y += u[i][h].y; v[i].z += u[i][h].z; v[i].w += u[i][h].w; }
            }
#pragma unroll
            for (int i = 0; i < NP; ++i)
                if (off[i] >= 0) {
                    float* d = dst + off[i];
                    d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
                }
        }
    } else {
        for (int t = threadIdx.x; t < TM * Kp; t += 256) {
            const int r = t / Kp, k = t - r * Kp;
            float v = 0.f;
            if (k < width && r0 + r < B)
                for (int q = 0; q < S; ++q) v += src[q * stride + (long long)(r0 + r) * width + k];
            dst[r * ld + k] = v;
        }
    }
}

// Row epilogue of one layer (forward): pre -> (+bias, +Wc[:, c]) -> LayerNorm -> ReLU -> dropout.
// TPR = 16 threads per row (all 16 rows of the tile in flight at once); thread t owns the 4-column groups 4 t + 64 q, so
// the threads of a row touch 256 contiguous bytes per group (16-byte HBM stores).  The row lives in registers from the
// first read on: the chain is latency-bound, not arithmetic-bound.
// The additive terms of a row epilogue (bias + condition column) depend on nothing the chain computes: they are
// requested before the GEMM whose output the epilogue consumes and stay raw until then (see stage_load).
template <int QM>
__device__ __forceinline__ void epilogue_fwd_prefetch(const sca_mid_args& a, const sca_mid_layer& L, int r0,
                                                      float (&vb)[QM][4], float (&vc)[QM][4]) {
    const int row = threadIdx.x / TPR, t8 = threadIdx.x % TPR;
    const int N = L.N;
    const int b = r0 + row;
    // absent terms read g_mid_blank and are dropped by the epilogue
    const int c = L.Wc ? a.cond[min(b, a.B - 1)] : 0;
#pragma unroll
    for (int q = 0; q < QM; ++q)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int hc = min(CG * q + 4 * t8 + u, N - 1);
            const float* pb = L.bias ? L.bias + hc : g_mid_blank;
            const float* pc = L.Wc ? L.Wc + (long long)hc * a.C + c : g_mid_blank;
            vb[q][u] = *pb;
            vc[q][u] = *pc;
        }
}

template <int QM>
__device__ __forceinline__ void layer_epilogue_fwd(const sca_mid_args& a, const sca_mid_layer& L, const float* pre,
                                                   float* nxt, int ld, int r0, const float (&vb)[QM][4],
                                                   const float (&vc)[QM][4]) {
    const int row = threadIdx.x / TPR, t8 = threadIdx.x % TPR;
    const int N = L.N, Npad = ((N + KC - 1) / KC) * KC;
    const int b = r0 + row;
    const bool live = b < a.B;
    const bool vec = (N & 3) == 0 && ((((uintptr_t)L.yn | (uintptr_t)L.h) & 15) == 0);
    const float* prow = pre + row * ld;
    float v[QM][4];
#pragma unroll
    for (int q = 0; q < QM; ++q)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float add = L.bias ? vb[q][u] : 0.f;
            if (L.Wc) add += vc[q][u];
            v[q][u] = add;
        }
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < QM; ++q)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int hh = CG * q + 4 * t8 + u;
            v[q][u] = (hh < N && live) ? v[q][u] + prow[hh] : 0.f;
            sum += v[q][u];
        }
    float mean = 0.f, rs = 1.f;
    if (a.norm == 1) {
        mean = row_sum(sum) / (float)N;
        float sq = 0.f;
#pragma unroll
        for (int q = 0; q < QM; ++q)
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (CG * q + 4 * t8 + u < N) {
                    const float d = v[q][u] - mean;
                    sq = fmaf(d, d, sq);
                }
        rs = 1.0f / sqrtf(row_sum(sq) / (float)N + 1e-5f);
    }
    if (t8 == 0 && live && L.rs) L.rs[b] = rs;
    const float keep_scale = a.p > 0.f ? 1.0f / (1.0f - a.p) : 1.f;
    const uint32_t stp = a.step ? a.step[0] : 0u;
    float* nrow = nxt + row * ld;
#pragma unroll
    for (int q = 0; q < QM; ++q) {
        const int h4 = CG * q + 4 * t8;
        if (h4 >= Npad) continue;
        bool keep[4] = {true, true, true, true};
        if (a.p > 0.f && live && !L.mask) dropout_keep4(a.seed + stp, L.rng_stream, (uint32_t)b, (uint32_t)h4, a.p, keep);
        float yv[4], ov[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int hh = h4 + u;
            yv[u] = ov[u] = 0.f;
            if (hh < N && live) {
                yv[u] = a.norm == 1 ? (v[q][u] - mean) * rs : v[q][u];
                float o = fmaxf(yv[u], 0.f);
                if (a.p > 0.f) {
                    const bool kp = L.mask ? (L.mask[(long long)b * N + hh] != 0) : keep[u];
                    o = kp ? o * keep_scale : 0.f;
                }
                ov[u] = o;
            }
            nrow[hh] = ov[u];
        }
        if (live && L.yn) {             // (eval-mode encoder passes keep nothing: yn == h == NULL)
            const long long g = (long long)b * N + h4;
            if (vec) {
                if (h4 < N) {
                    *reinterpret_cast<float4*>(L.yn + g) = make_float4(yv[0], yv[1], yv[2], yv[3]);
                    *reinterpret_cast<float4*>(L.h + g) = make_float4(ov[0], ov[1], ov[2], ov[3]);
                }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (h4 + u < N) { L.yn[g + u] = yv[u]; L.h[g + u] = ov[u]; }
            }
        }
    }
    __syncthreads();
}

// The argument block is ~1 KB (17 cache lines) and the chain reads it piecemeal, phase by phase: every first touch of a
// line is a scalar-cache miss that the single wave per SIMD waits out.  One burst of scalar loads up front - one
// per line, all in flight together - brings the block into the scalar cache before the chain starts.
__device__ __forceinline__ void warm_kernargs() {
    const uint32_t __attribute__((address_space(4)))* kp =
        (const uint32_t __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < (int)((sizeof(sca_mid_args) + 63) / 64); ++i) acc |= kp[16 * i];
    asm volatile("" ::"s"(acc));
}

template <int NS>
__global__ __launch_bounds__(256) void mid_fwd_kernel(const sca_mid_args a, int ld) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* A0 = smem;                  // [TM][ld]
    float* A1 = A0 + TM * ld;          // [TM][ld]
    float* Ms = A1 + TM * ld;          // [KC][MS_LD]
    const int r0 = blockIdx.x * TM;
    float* cur = A0;
    float* oth = A1;
    // latency-bound chain on the step's critical path: when it shares a SIMD with the waves of a GEMM that saturates the
    // matrix pipe (weight-gradient GEMMs on the side streams), its few MFMAs are issued first
    __builtin_amdgcn_s_setprio(MID_WAVE_PRIO);
    warm_kernargs();
    DBG(15);
    // ONE loop walks the chain - first-layer epilogue (g = -1), encoder layers, heads + latent block (g = n_enc), decoder
    // layers - so the GEMM and the row epilogue exist once in the code: a straight-line chain is ~100 KB of
    // instructions that every workgroup executes exactly once, more than the 64 KB instruction cache.
    // It is also a software pipeline: the first weight chunk of the NEXT GEMM and the additive terms of the current
    // epilogue are in flight while the current GEMM / epilogue run (one call site per prefetch, operands selected: a
    // prefetch inside a branch has no fixed load count either).
    const int Z = a.Z;
    const int ng = a.n_enc + 1 + a.n_dec;
    float wpre[NST], epb[NS][4], epc[NS][4];
    auto gemm_ref = [&](int g, const float*& W, int& I, int& J) {     // operands of GEMM g (FWD: ldW = I)
        if (g == a.n_enc) { W = a.Wh; I = a.Kh; J = 2 * Z; return; }
        const sca_mid_layer& L = g < a.n_enc ? a.enc[g] : a.dec[g - a.n_enc - 1];
        W = L.W; I = L.K; J = L.N;
    };
    {
        const float* W; int I, J;
        gemm_ref(0, W, I, J);
        tile_gemm_prefetch<true>(wpre, W, I, I, J);
    }
    if (!a.fuse_first)        // input tile (zero padded to a multiple of KC columns)
        load_tile(A0, ld, a.h_in, 1, 0, r0, a.B, a.K0, ((a.K0 + KC - 1) / KC) * KC);
    DBG(0);
    for (int g = a.fuse_first ? -1 : 0; g < ng; ++g) {
        const bool head = g == a.n_enc, first = g < 0;
        // the layer whose epilogue follows (heads: none - `first` stands in for the prefetch, which tolerates a blank layer)
        const sca_mid_layer& L = (first || head) ? a.first : (g < a.n_enc ? a.enc[g] : a.dec[g - a.n_enc - 1]);
        epilogue_fwd_prefetch<NS>(a, L, r0, epb, epc);
        if (first) {
            // first encoder layer: sum of the GEMM's split-K slabs (its epilogue below)
            load_tile(oth, ld, a.pre0, a.S0, a.pre0_stride, r0, a.B, a.K0, a.K0);
            __syncthreads();
        } else {
            const float* W; int I, J;
            gemm_ref(g, W, I, J);
            tile_gemm<true>(cur, ld, I, W, I, J, oth, ld, Ms, wpre);
        }
        {
            const float* W; int I, J;
            gemm_ref(min(g + 1, ng - 1), W, I, J);           // (after the last GEMM: a harmless re-read)
            tile_gemm_prefetch<true>(wpre, W, I, I, J);
        }
        DBG(1 + 2 * (g + 1));
        if (!head) {
            layer_epilogue_fwd<NS>(a, L, oth, cur, ld, r0, epb, epc);
        } else {
            // heads -> reparameterised sample + KL
            const int Zp = ((Z + KC - 1) / KC) * KC;
            const uint32_t stp = a.step ? a.step[0] : 0u;
            for (int t = threadIdx.x; t < TM * (Zp - Z); t += 256) {          // zero padding of the z tile
                const int r = t / (Zp - Z), k = Z + t - r * (Zp - Z);
                cur[r * ld + k] = 0.f;
            }
            for (int t = threadIdx.x; t < TM * Z; t += 256) {
                const int r = t / Z, k = t - r * Z, b = r0 + r;
                float zv = 0.f, kl = 0.f;
                if (b < a.B) {
                    const float mu = oth[r * ld + k] + a.bh[k], lv = oth[r * ld + Z + k] + a.bh[Z + k];
                    const float var = expf(lv) + 1e-4f, sd = sqrtf(var);
                    float e = 0.f;
                    if (a.sample) {
                        if (a.eps_in) {
                            e = a.eps_in[(long long)b * Z + k];
                        } else {
                            float n0, n1;
                            philox_normal2(a.seed + stp, a.rng_latent, (uint32_t)(b * ((Z + 1) / 2) + (k >> 1)), n0, n1);
                            e = (k & 1) ? n1 : n0;
                        }
                    }
                    const float vr = sd * sd;
                    kl = 0.5f * (vr + mu * mu - 1.0f - logf(vr));
                    a.mu_lv[(long long)b * 2 * Z + k] = mu;
                    a.mu_lv[(long long)b * 2 * Z + Z + k] = lv;
                    a.eps_out[(long long)b * Z + k] = e;
                    zv = fmaf(e, sd, mu);
                    a.z[(long long)b * Z + k] = zv;
                }
                cur[r * ld + k] = zv;
                oth[r * ld + 2 * Z + k] = kl;                 // parked next to the head outputs, summed below
            }
            __syncthreads();
            for (int r = threadIdx.x; r < TM; r += 256)
                if (r0 + r < a.B) {
                    float t = 0.f;
                    for (int k = 0; k < Z; ++k) t += oth[r * ld + 2 * Z + k];
                    a.kl_row[r0 + r] = t;
                }
            __syncthreads();
        }
        DBG(2 + 2 * (g + 1));
    }
}

// Row epilogue of one layer (backward): upstream gradient tile `g` -> dropout / ReLU / LayerNorm backward -> da
// (HBM + tile `dst`, zero padded).  Same thread layout as the forward epilogue; the saved normalised pre-activations
// are fetched from HBM up front.
// The saved normalised pre-activations (and rstd) an epilogue needs are requested one GEMM ahead.
template <int QM>
__device__ __forceinline__ void epilogue_bwd_prefetch(const sca_mid_args& a, const sca_mid_layer& L, bool none, int r0,
                                                      float (&y)[QM][4], float& rsv) {
    const int row = threadIdx.x / TPR, t8 = threadIdx.x % TPR;
    const int N = L.N;
    const int b = r0 + row;
    const bool live = b < a.B;
    const bool vec = (N & 3) == 0 && ((((uintptr_t)L.yn | (uintptr_t)L.da) & 15) == 0);
    // unconditional loads from clamped addresses (see stage_load); `none` (no epilogue follows) reads g_mid_blank
    const long long bc = min(b, a.B - 1);
    const bool have = !none && L.yn;
    const float* yb = have ? L.yn + bc * N : g_mid_blank;
#pragma unroll
    for (int q = 0; q < QM; ++q) {
        const int h4 = CG * q + 4 * t8;
        if (vec) {
            const float4 t = *reinterpret_cast<const float4*>(yb + (have ? min(h4, N - 4) : 0));
            y[q][0] = t.x; y[q][1] = t.y; y[q][2] = t.z; y[q][3] = t.w;
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) y[q][u] = yb[have ? min(h4 + u, N - 1) : 0];
        }
    }
    const bool has_rs = a.norm == 1 && !none && L.rs;
    rsv = *(has_rs ? L.rs + bc : g_mid_blank);
}

template <int QM>
__device__ __forceinline__ void layer_epilogue_bwd(const sca_mid_args& a, const sca_mid_layer& L, const float* g,
                                                   float* dst, int ld, int r0, const float (&y)[QM][4], float rsv) {
    const int row = threadIdx.x / TPR, t8 = threadIdx.x % TPR;
    const int N = L.N, Npad = ((N + KC - 1) / KC) * KC;
    const int b = r0 + row;
    const bool live = b < a.B;
    const bool vec = (N & 3) == 0 && ((((uintptr_t)L.yn | (uintptr_t)L.da) & 15) == 0);
    const float keep_scale = a.p > 0.f ? 1.0f / (1.0f - a.p) : 1.f;
    const uint32_t stp = a.step ? a.step[0] : 0u;
    const float* grow = g + row * ld;
    float* drow = dst + row * ld;
    float x[QM][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < QM; ++q) {
        const int h4 = CG * q + 4 * t8;
        bool keep[4] = {true, true, true, true};
        if (a.p > 0.f && live && !L.mask && h4 < N)
            dropout_keep4(a.seed + stp, L.rng_stream, (uint32_t)b, (uint32_t)h4, a.p, keep);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int hh = h4 + u;
            float xv = 0.f;
            if (hh < N && live) {
                xv = grow[hh];
                if (a.p > 0.f) {
                    const bool kp = L.mask ? (L.mask[(long long)b * N + hh] != 0) : keep[u];
                    xv = kp ? xv * keep_scale : 0.f;
                }
                if (!(y[q][u] > 0.f)) xv = 0.f;
            }
            x[q][u] = xv;
            s1 += xv;
            s2 = fmaf(xv, y[q][u], s2);
        }
    }
    float m1 = 0.f, m2 = 0.f;
    if (a.norm == 1) {
        m1 = row_sum(s1) / (float)N;
        m2 = row_sum(s2) / (float)N;
    }
#pragma unroll
    for (int q = 0; q < QM; ++q) {
        const int h4 = CG * q + 4 * t8;
        if (h4 >= Npad) continue;
        float ov[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ov[u] = 0.f;
            if (h4 + u < N && live) ov[u] = a.norm == 1 ? rsv * (x[q][u] - m1 - y[q][u] * m2) : x[q][u];
            drow[h4 + u] = ov[u];
        }
        if (live) {
            const long long o = (long long)b * N + h4;
            if (vec) {
                if (h4 < N) *reinterpret_cast<float4*>(L.da + o) = make_float4(ov[0], ov[1], ov[2], ov[3]);
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (h4 + u < N) L.da[o + u] = ov[u];
            }
        }
    }
    __syncthreads();
}

template <int NS>
__global__ __launch_bounds__(256) void mid_bwd_kernel(const sca_mid_args a, int ld) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* A0 = smem;
    float* A1 = A0 + TM * ld;
    float* Ms = A1 + TM * ld;
    const int r0 = blockIdx.x * TM;
    const int Z = a.Z;
    __builtin_amdgcn_s_setprio(MID_WAVE_PRIO);       // see mid_fwd_kernel
    warm_kernargs();
    // One loop walks the chain backwards (decoder layers, latent block + heads at g = n_enc, encoder layers, first-layer
    // epilogue at g = -1): one instance of the GEMM and of the epilogue, weights of the NEXT GEMM and the saved
    // activations of the NEXT epilogue requested one phase ahead - see mid_fwd_kernel.
    const int ng = a.n_enc + 1 + a.n_dec;
    const int gmin = a.fuse_first ? -1 : 0;
    float wpre[NST], ypre[NS][4], rspre = 0.f;
    // layer of chain position g (g != n_enc; only references into the kernel arguments: a local stand-in would push
    // the whole argument block through private memory)
#define MID_LAYER(g) ((g) < 0 ? a.first : ((g) < a.n_enc ? a.enc[(g)] : a.dec[(g) - a.n_enc - 1]))
    auto gemm_ref = [&](int g, const float*& W, int& ldW, int& I, int& J) {   // dX = dA W: I = outputs, J = inputs
        if (g == a.n_enc) { W = a.Wh; ldW = a.Kh; I = 2 * Z; J = a.Kh; return; }
        const sca_mid_layer& L = g < a.n_enc ? a.enc[g] : a.dec[g - a.n_enc - 1];
        W = L.W; ldW = L.K; I = L.N; J = L.K;
    };
    {
        const float* W; int ldW, I, J;
        gemm_ref(ng - 1, W, ldW, I, J);
        tile_gemm_prefetch<false>(wpre, W, ldW, I, J);
        const bool none = ng - 1 == a.n_enc;                 // (the heads have no epilogue)
        const sca_mid_layer& L = none ? a.first : a.dec[a.n_dec - 1];
        epilogue_bwd_prefetch<NS>(a, L, none, r0, ypre, rspre);
    }
    // upstream gradient tile: sum of the dX slabs of the output GEMM
    {
        const int N = a.n_dec ? a.dec[a.n_dec - 1].N : Z;
        load_tile(A0, ld, a.dh, a.S, a.dh_stride, r0, a.B, N, N);
        __syncthreads();
    }
    float* cur = A0;
    float* oth = A1;
    for (int g = ng - 1; g >= gmin; --g) {
        const bool head = g == a.n_enc;
        if (!head) {
            const sca_mid_layer& L = MID_LAYER(g);
            layer_epilogue_bwd<NS>(a, L, cur, oth, ld, r0, ypre, rspre);                // oth = da [TM][N]
        } else {
            // latent block: dz (+ dz2) + KL -> d[mean | log_var]
            const float akl = a.hyper[0] / (float)a.B;
            const int Hp = ((2 * Z + KC - 1) / KC) * KC;
            for (int t = threadIdx.x; t < TM * (Hp - 2 * Z); t += 256) {      // zero padding of the d-heads tile
                const int r = t / (Hp - 2 * Z), k = 2 * Z + t - r * (Hp - 2 * Z);
                oth[r * ld + k] = 0.f;
            }
            for (int t = threadIdx.x; t < TM * 2 * Z; t += 256) {
                const int r = t / (2 * Z), k = t - r * 2 * Z, b = r0 + r;
                float o = 0.f;
                if (b < a.B) {
                    const int kk = k < Z ? k : k - Z;
                    float gz = cur[r * ld + kk];
                    if (a.dz2) gz += a.dz2[(long long)b * Z + kk];
                    const float mu = a.mu_lv[(long long)b * 2 * Z + kk], lv = a.mu_lv[(long long)b * 2 * Z + Z + kk];
                    const float ev = expf(lv), var = ev + 1e-4f, sd = sqrtf(var);
                    o = k < Z ? gz + akl * mu
                              : gz * a.eps_out[(long long)b * Z + kk] * (0.5f * ev / sd) + akl * 0.5f * (1.0f - 1.0f / var) * ev;
                    a.dheads[(long long)b * 2 * Z + k] = o;
                }
                oth[r * ld + k] = o;
            }
            // tile_gemm starts with a barrier
        }
        if (g < 0) return;               // first encoder layer backward done: `oth` / da0 feed its dW GEMM
        // saved activations of the next epilogue (g - 1; none after the last GEMM or before the latent block)
        {
            const int gn = g - 1;
            const bool none = gn < gmin || gn == a.n_enc;
            const int gs = none ? -1 : gn;
            const sca_mid_layer& Ln = MID_LAYER(gs);
            epilogue_bwd_prefetch<NS>(a, Ln, none, r0, ypre, rspre);
        }
        {
            const float* W; int ldW, I, J;
            gemm_ref(g, W, ldW, I, J);
            tile_gemm<false>(oth, ld, I, W, ldW, J, cur, ld, Ms, wpre);       // cur = dX [TM][K]
            gemm_ref(max(g - 1, 0), W, ldW, I, J);                            // (after the last GEMM: a harmless re-read)
            tile_gemm_prefetch<false>(wpre, W, ldW, I, J);
        }
    }
#undef MID_LAYER
    if ((a.K0 & 3) == 0 && ((uintptr_t)a.dh_out & 15) == 0) {
        const int w4 = a.K0 >> 2;
        for (int t = threadIdx.x; t < TM * w4; t += 256) {
            const int r = t / w4, k = (t - r * w4) * 4;
            const float* p = cur + r * ld + k;
            if (r0 + r < a.B)
                *reinterpret_cast<float4*>(a.dh_out + (long long)(r0 + r) * a.K0 + k) = make_float4(p[0], p[1], p[2], p[3]);
        }
    } else {
        for (int t = threadIdx.x; t < TM * a.K0; t += 256) {
            const int r = t / a.K0, k = t - r * a.K0;
            if (r0 + r < a.B) a.dh_out[(long long)(r0 + r) * a.K0 + k] = cur[r * ld + k];
        }
    }
}

int check_and_ld(const sca_mid_args* a, int* ld, size_t* shm, int* wide) {
    if (a->B <= 0) return 1;
    if (a->n_enc < 0 || a->n_enc > 4 || a->n_dec < 0 || a->n_dec > 4) return -1;
    int maxw = a->K0 > a->Kh ? a->K0 : a->Kh;
    if (3 * a->Z > maxw) maxw = 3 * a->Z;
    for (int l = 0; l < a->n_enc; ++l) { if (a->enc[l].N > maxw) maxw = a->enc[l].N; if (a->enc[l].K > maxw) maxw = a->enc[l].K; }
    for (int l = 0; l < a->n_dec; ++l) { if (a->dec[l].N > maxw) maxw = a->dec[l].N; if (a->dec[l].K > maxw) maxw = a->dec[l].K; }
    if (maxw > MAXW) return -2;
    int maxn = 0;
    for (int l = 0; l < a->n_enc; ++l) if (a->enc[l].N > maxn) maxn = a->enc[l].N;
    for (int l = 0; l < a->n_dec; ++l) if (a->dec[l].N > maxn) maxn = a->dec[l].N;
    if (a->fuse_first) {
        if (a->first.N != a->K0 || a->S0 < 1 || !a->pre0) return -1;
        if (a->K0 > maxn) maxn = a->K0;
    }
    *wide = maxn > 128;
    *ld = ((maxw + KC - 1) / KC) * KC + 2;
    *shm = sizeof(float) * ((size_t)2 * TM * (*ld) + (size_t)KC * MS_LD);
    return 0;
}

}  // namespace


#ifdef SCA_MID_DEBUG
extern "C" int sca_mid_debug(long long* host64) {
    return (int)hipMemcpyFromSymbol(host64, HIP_SYMBOL(g_mid_dbg), sizeof(long long) * 64);
}
#endif

extern "C" int sca_mid_fwd_staged(const sca_mid_args* a, void* stream) {
    int ld, wide; size_t shm;
    const int rc = check_and_ld(a, &ld, &shm, &wide);
    if (rc) return rc > 0 ? 0 : rc;
    const dim3 grid((a->B + TM - 1) / TM);
    if (wide) {
        hipFuncSetAttribute((const void*)mid_fwd_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        hipLaunchKernelGGL(mid_fwd_kernel<4>, grid, dim3(256), shm, (hipStream_t)stream, *a, ld);
    } else {
        hipFuncSetAttribute((const void*)mid_fwd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        hipLaunchKernelGGL(mid_fwd_kernel<2>, grid, dim3(256), shm, (hipStream_t)stream, *a, ld);
    }
    SCA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sca_mid_bwd_staged(const sca_mid_args* a, void* stream) {
    int ld, wide; size_t shm;
    const int rc = check_and_ld(a, &ld, &shm, &wide);
    if (rc) return rc > 0 ? 0 : rc;
    if (!a->dh || !a->dheads || !a->hyper || (!a->fuse_first && !a->dh_out)) return -1;
    const dim3 grid((a->B + TM - 1) / TM);
    if (wide) {
        hipFuncSetAttribute((const void*)mid_bwd_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        hipLaunchKernelGGL(mid_bwd_kernel<4>, grid, dim3(256), shm, (hipStream_t)stream, *a, ld);
    } else {
        hipFuncSetAttribute((const void*)mid_bwd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        hipLaunchKernelGGL(mid_bwd_kernel<2>, grid, dim3(256), shm, (hipStream_t)stream, *a, ld);
    }
    SCA_CHECK_LAUNCH();
    return 0;
}
