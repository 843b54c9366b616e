// fp32-in / fp32-accumulate MFMA GEMM for the conditional-VAE layers (gfx950).
//
//   C[m][n] (+= over split-K slabs) = sum_k A(m,k) * B(k,n)
//
// Operand layouts (chosen per call so that forward, dX and dW all run without a transpose pass):
//   AMODE 0: A(m,k) = A[m*lda + k]   ("k-contiguous", activations [rows][features])
//   AMODE 1: A(m,k) = A[k*lda + m]   ("k-major": contraction over the rows of A, i.e. A^T . B)
//   BMODE 0: B(k,n) = B[n*ldb + k]   (torch Linear weight [out][in] used as x @ W^T)
//   BMODE 1: B(k,n) = B[k*ldb + n]
//
//   forward  y = x W^T        : AMODE 0 (x[M,K]),   BMODE 0 (W[N,K])
//   dX       = dY W           : AMODE 0 (dY[M,K']), BMODE 1 (W[K',N])
//   dW       = dY^T X         : AMODE 1 (dY[K',M]), BMODE 1 (X[K',N])
//
// Tile 128x128x16, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 v_mfma_f32_32x32x2_f32 tiles
// (exact fp32: bit-identical to an fmaf chain in k order, so parity with the CPU reference is limited only
// by summation order).  LDS tiles are k-major with a +4 pad: fragment reads are conflict-free ds_read_b32,
// transposing stores of the k-contiguous operands are 2-way (free on ds_write_b32).  Global->LDS staging is
// register-prefetched two k-steps ahead (two register sets), the k loop runs on a three-stage LDS ring with
// double-buffered fragment registers and is branch-free in its steady state (one barrier per k-step; see gemm_body).
// Measured and dropped: two accumulators per single-tile wave (even / odd MFMAs of a k-step) - slower, the
// accumulator chain is not what bounds the loop.
// Split-K (gridDim.z) writes per-split slabs; the consumer kernel sums them (deterministic, no atomics).
#include "common.h"
#include <type_traits>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

#ifndef GEMM_OCC
#define GEMM_OCC 2
#endif
#ifndef GEMM_BK
#define GEMM_BK 16
#endif
constexpr int BK = GEMM_BK;
constexpr int NSTAGE = 3;      // LDS stages of the k pipeline (see gemm_body)
// block tile = (64*WM) x (64*WN): 2x2 waves, each wave WM x WN MFMA tiles of 32x32.  LDS row stride = rows + 4.

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;   // optional [N], only applied when gridDim.z == 1
    int lda, ldb, ldc;
    int M, N, K;
    int k_chunk;         // multiple of BK (16)
    long long slab_stride;
    int tiles_m;
    int nsplit;          // number of split-K slabs of this GEMM (the bias is only added when it is 1)
    int vec;             // grouped launches: this job's operands allow 16-byte loads
    int tiles_n, n_fast; // tile order (see gemm_body)
};

// Load one (64*W)(rows) x BK(k) operand tile into W*BK/16 float4 registers per thread.
template <int MODE, bool VEC, int W>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, int ld, int R, int r0, int k0, int kend,
                                          float4 (&reg)[W * BK / 16]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < W * BK / 16; ++i) {
        const int idx = t + 256 * i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE == 0) {
            const int r = r0 + idx / (BK / 4), k = k0 + 4 * (idx % (BK / 4));
            if (r < R) {
                const float* p = P + (long long)r * ld + k;
                if (VEC && k + 3 < kend) {
                    v = *reinterpret_cast<const float4*>(p);
                } else {
                    if (k + 0 < kend) v.x = p[0];
                    if (k + 1 < kend) v.y = p[1];
                    if (k + 2 < kend) v.z = p[2];
                    if (k + 3 < kend) v.w = p[3];
                }
            }
        } else {
            const int k = k0 + idx / (16 * W), r = r0 + 4 * (idx % (16 * W));
            if (k < kend) {
                const float* p = P + (long long)k * ld + r;
                if (VEC && r + 3 < R) {
                    v = *reinterpret_cast<const float4*>(p);
                } else {
                    if (r + 0 < R) v.x = p[0];
                    if (r + 1 < R) v.y = p[1];
                    if (r + 2 < R) v.z = p[2];
                    if (r + 3 < R) v.w = p[3];
                }
            }
        }
        reg[i] = v;
    }
}

// Branch-free variant for tiles that are complete in k (every tile but possibly the last of a split): rows past
// the operand's end are clamped to a valid row instead of being zero-filled - such rows only feed output rows /
// columns that the epilogue never stores.  Needs 16-byte loads (VEC) and, for the k-major layout, R % 4 == 0.
// Straight-line code lets the compiler wait for exactly the tile it is about to store (counted vmcnt) while the
// next tile's loads stay in flight.
template <int MODE, int W>
__device__ __forceinline__ void load_tile_fast(const float* __restrict__ P, int ld, int R, int r0, int k0,
                                               float4 (&reg)[W * BK / 16]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < W * BK / 16; ++i) {
        const int idx = t + 256 * i;
        if (MODE == 0) {
            const int r = min(r0 + idx / (BK / 4), R - 1), k = k0 + 4 * (idx % (BK / 4));
            reg[i] = *reinterpret_cast<const float4*>(P + (long long)r * ld + k);
        } else {
            const int k = k0 + idx / (16 * W), r = min(r0 + 4 * (idx % (16 * W)), R - 4);
            reg[i] = *reinterpret_cast<const float4*>(P + (long long)k * ld + r);
        }
    }
}

template <int MODE, int W>
__device__ __forceinline__ void store_tile(float* __restrict__ T, const float4 (&reg)[W * BK / 16]) {
    constexpr int LDT = 64 * W + 4;
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < W * BK / 16; ++i) {
        const int idx = t + 256 * i;
        if (MODE == 0) {
            const int r = idx / (BK / 4), kc = 4 * (idx % (BK / 4));
            T[(kc + 0) * LDT + r] = reg[i].x;
            T[(kc + 1) * LDT + r] = reg[i].y;
            T[(kc + 2) * LDT + r] = reg[i].z;
            T[(kc + 3) * LDT + r] = reg[i].w;
        } else {
            const int k = idx / (16 * W), r = 4 * (idx % (16 * W));
            *reinterpret_cast<float4*>(&T[k * LDT + r]) = reg[i];
        }
    }
}

template <int AMODE, int BMODE, bool VEC, int WM, int WN>
__device__ __forceinline__ void gemm_body(const GemmArgs& a, int tile, int zs, float* __restrict__ lds) {
    constexpr int BM = 64 * WM, BN = 64 * WN, LDA = BM + 4, LDB = BN + 4;
    constexpr int ASZ = BK * LDA, BSZ = BK * LDB;

    // tile order: the dimension with more tiles varies fastest, so the few tiles that share the LARGE operand
    // (e.g. the two 64-row tiles of dW = da^T x that both stream x) are a multiple of 8 workgroups apart and land on
    // the same XCD - its L2 then serves the second read instead of the fabric
    const int tm = a.n_fast ? tile / a.tiles_n : tile % a.tiles_m;
    const int tn = a.n_fast ? tile % a.tiles_n : tile / a.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = zs * a.k_chunk;
    const int kend = min(a.K, kbeg + a.k_chunk);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wm = w >> 1, wn = w & 1, l31 = lane & 31, half = lane >> 5;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (kend - kbeg + BK - 1) / BK;
    // LDS image: three A stages then three B stages.  Stage s holds k-tile kt with kt % 3 == s.  In k-step kt
    //   * the MFMAs consume the fragment registers of tile kt (set kt & 1),
    //   * the fragments of tile kt+1 (in LDS since the previous barrier) are read into the other set BETWEEN them,
    //   * tile kt+2, requested from global memory one k-step ago (register set kt & 1), is stored to stage (kt+2) % 3,
    //   * tile kt+3 is requested into the other register set.
    // A global load so has two k-steps (>= 1024 MFMA cycles of this wave alone) to land; one barrier per k-step
    // suffices (the stage written in step kt was last read in step kt-2).
    float4 ra[2][WM * BK / 16], rb[2][WN * BK / 16];
    float* As = lds;
    float* Bs = lds + NSTAGE * ASZ;
    const int aoff = half * LDA + wm * (32 * WM) + l31, boff = half * LDB + wn * (32 * WN) + l31;
    float av[2][BK / 2][WM], bv[2][BK / 2][WN];
    // prologue: tiles 0 and 1 are requested together (ONE exposed memory latency, not two - a short-K GEMM such as the
    // output layer, K = 128, spends more time here than in its eight k-steps)
    if (nk > 0) {
        load_tile<AMODE, VEC, WM>(a.A, a.lda, a.M, m0, kbeg, kend, ra[0]);
        load_tile<BMODE, VEC, WN>(a.B, a.ldb, a.N, n0, kbeg, kend, rb[0]);
    }
    if (nk > 1) {
        load_tile<AMODE, VEC, WM>(a.A, a.lda, a.M, m0, kbeg + BK, kend, ra[1]);
        load_tile<BMODE, VEC, WN>(a.B, a.ldb, a.N, n0, kbeg + BK, kend, rb[1]);
    }
    if (nk > 0) {
        store_tile<AMODE, WM>(As, ra[0]);
        store_tile<BMODE, WN>(Bs, rb[0]);
    }
    if (nk > 1) {
        store_tile<AMODE, WM>(As + ASZ, ra[1]);
        store_tile<BMODE, WN>(Bs + BSZ, rb[1]);
    }
    if (nk > 2) {
        load_tile<AMODE, VEC, WM>(a.A, a.lda, a.M, m0, kbeg + 2 * BK, kend, ra[0]);
        load_tile<BMODE, VEC, WN>(a.B, a.ldb, a.N, n0, kbeg + 2 * BK, kend, rb[0]);
    }
    __syncthreads();
    if (nk > 0) {
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
#pragma unroll
            for (int i = 0; i < WM; ++i) av[0][kk][i] = As[aoff + (2 * kk) * LDA + 32 * i];
#pragma unroll
            for (int j = 0; j < WN; ++j) bv[0][kk][j] = Bs[boff + (2 * kk) * LDB + 32 * j];
        }
    }
    // steady state: tiles kt+1 .. kt+3 exist and are complete in k; no branches between the barriers
    auto kstep_fast = [&](auto PP, int kt) {
        constexpr int P = decltype(PP)::value;
        load_tile_fast<AMODE, WM>(a.A, a.lda, a.M, m0, kbeg + (kt + 3) * BK, ra[1 - P]);
        load_tile_fast<BMODE, WN>(a.B, a.ldb, a.N, n0, kbeg + (kt + 3) * BK, rb[1 - P]);
        const int s1 = (kt + 1) % NSTAGE;
        const float* An = As + s1 * ASZ + aoff;
        const float* Bn = Bs + s1 * BSZ + boff;
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[P][kk][i], bv[P][kk][j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < WM; ++i) av[1 - P][kk][i] = An[(2 * kk) * LDA + 32 * i];
#pragma unroll
            for (int j = 0; j < WN; ++j) bv[1 - P][kk][j] = Bn[(2 * kk) * LDB + 32 * j];
        }
        const int s2 = (kt + 2) % NSTAGE;
        store_tile<AMODE, WM>(As + s2 * ASZ, ra[P]);
        store_tile<BMODE, WN>(Bs + s2 * BSZ, rb[P]);
        __syncthreads();
    };
    // general k-step (first / last steps of a split, partial tiles, operands without 16-byte loads)
    auto kstep = [&](auto PP, int kt) {
        constexpr int P = decltype(PP)::value;
        const bool more3 = kt + 3 < nk, more2 = kt + 2 < nk, more1 = kt + 1 < nk;
        if (more3) {
            load_tile<AMODE, VEC, WM>(a.A, a.lda, a.M, m0, kbeg + (kt + 3) * BK, kend, ra[1 - P]);
            load_tile<BMODE, VEC, WN>(a.B, a.ldb, a.N, n0, kbeg + (kt + 3) * BK, kend, rb[1 - P]);
        }
        const int s1 = (kt + 1) % NSTAGE;
        const float* An = As + s1 * ASZ + aoff;
        const float* Bn = Bs + s1 * BSZ + boff;
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[P][kk][i], bv[P][kk][j], acc[i][j], 0, 0, 0);
            if (more1) {
#pragma unroll
                for (int i = 0; i < WM; ++i) av[1 - P][kk][i] = An[(2 * kk) * LDA + 32 * i];
#pragma unroll
                for (int j = 0; j < WN; ++j) bv[1 - P][kk][j] = Bn[(2 * kk) * LDB + 32 * j];
            }
        }
        if (more2) {
            const int s2 = (kt + 2) % NSTAGE;
            store_tile<AMODE, WM>(As + s2 * ASZ, ra[P]);
            store_tile<BMODE, WN>(Bs + s2 * BSZ, rb[P]);
        }
        __syncthreads();
    };
    // steps [0, nfast) run branch-free: tile kt+3 <= nk-2 is complete in k for every kt < nk-4
    const bool fast_ok = VEC && (AMODE == 0 || (a.M & 3) == 0) && (BMODE == 0 || (a.N & 3) == 0) &&
                         (AMODE == 1 || a.M >= 1) && (BMODE == 1 || a.N >= 1) && a.M >= 4 && a.N >= 4;
    const int nfast = fast_ok ? max(0, nk - 4) & ~1 : 0;
    int kt = 0;
    for (; kt < nfast; kt += 2) {
        kstep_fast(std::integral_constant<int, 0>{}, kt);
        kstep_fast(std::integral_constant<int, 1>{}, kt + 1);
    }
    for (; kt < nk; kt += 2) {
        kstep(std::integral_constant<int, 0>{}, kt);
        if (kt + 1 < nk) kstep(std::integral_constant<int, 1>{}, kt + 1);
    }

    float* C = a.C + (long long)zs * a.slab_stride;
    const bool add_bias = a.bias != nullptr && a.nsplit == 1;
    if constexpr (WM == 1 && WN == 1) {
        // 64 x 64 tiles leave through LDS: the accumulator layout gives every lane ONE column (4-byte stores, two
        // 128-byte row pieces per wave instruction); staged row-major, a thread stores 16 bytes and a wave instruction
        // four complete 256-byte row segments.  (The k loop ended with a barrier: the operand stages are free.)
        if ((a.ldc & 3) == 0 && (((uintptr_t)C) & 15) == 0 && (a.N & 3) == 0 && (a.slab_stride & 3) == 0) {
            constexpr int LDC = 64 + 4;
            float* Ct = lds;                   // 64 x 68 floats = 17 KB of the 26 KB operand ring
            const int nc = wn * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) Ct[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * LDC + nc] = acc[0][0][r];
            const int c4 = 4 * (threadIdx.x & 15), rr = threadIdx.x >> 4;
            float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (add_bias && n0 + c4 < a.N) b4 = *reinterpret_cast<const float4*>(a.bias + n0 + c4);
            __syncthreads();
            if (n0 + c4 < a.N) {
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    const int row = rr + 16 * pass, m = m0 + row;
                    if (m < a.M) {
                        float4 v = *reinterpret_cast<const float4*>(&Ct[row * LDC + c4]);
                        v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
                        *reinterpret_cast<float4*>(C + (long long)m * a.ldc + n0 + c4) = v;
                    }
                }
            }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int n = n0 + wn * (32 * WN) + j * 32 + l31;
            if (n >= a.N) continue;
            const float bv = add_bias ? a.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (32 * WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m < a.M) C[(long long)m * a.ldc + n] = acc[i][j][r] + bv;
            }
        }
}

template <int AMODE, int BMODE, bool VEC, int WM, int WN>
__global__ __launch_bounds__(256, GEMM_OCC) void gemm_f32_kernel(const GemmArgs a) {
    constexpr int LDSZ = NSTAGE * BK * (64 * WM + 4 + 64 * WN + 4);
    __shared__ __attribute__((aligned(16))) float lds[LDSZ];
    gemm_body<AMODE, BMODE, VEC, WM, WN>(a, blockIdx.x, blockIdx.z, lds);
}

// Several independent GEMMs in ONE launch (the weight-gradient GEMMs dY^T X of a step: same operand modes, 64x64
// tiles, each with its own split-K): a workgroup looks its job up from the cumulative workgroup counts.
constexpr int GROUP_MAX = 8;
struct GroupArgs {
    GemmArgs job[GROUP_MAX];
    int wg_begin[GROUP_MAX + 1];
    int tiles[GROUP_MAX];        // output tiles of a job (workgroups per split)
    int n;
};
__global__ __launch_bounds__(256, GEMM_OCC) void gemm_group_kernel(const GroupArgs g) {
    constexpr int LDSZ = NSTAGE * BK * (64 + 4 + 64 + 4);
    __shared__ __attribute__((aligned(16))) float lds[LDSZ];
    int j = 0;
    while (j + 1 < g.n && (int)blockIdx.x >= g.wg_begin[j + 1]) ++j;
    const int local = blockIdx.x - g.wg_begin[j];
    const int zs = local / g.tiles[j];
    if (g.job[j].vec) gemm_body<1, 1, true, 1, 1>(g.job[j], local - zs * g.tiles[j], zs, lds);
    else gemm_body<1, 1, false, 1, 1>(g.job[j], local - zs * g.tiles[j], zs, lds);
}

// ---- register-direct GEMM for two k-major operands (the weight-gradient family dW = dY^T X) ------------------------
//   C[m][n] = sum_k A[k*lda + m] * B[k*ldb + n]
// Both operands are contracted over their ROWS, so a wave can load MFMA fragments straight from global memory: the A
// fragment of v_mfma_f32_32x32x2_f32 is "lane l holds A(m_l, k = l/32)".  Which 32 output rows a fragment covers is
// free, so lane l loads a float4 at m = m0 + 4*(l%32) .. +3 of row k (lanes 0-31) / k+1 (lanes 32-63): ONE 16-byte
// load per lane yields the fragments of four interleaved row blocks (block c = rows m0 + 4 i + c) - two fully
// coalesced 512-byte segments per wave instruction; a float2 of B likewise yields two column blocks.  No LDS staging
// and no workgroup barrier in the k loop.  A wave owns a 128 x 64 output tile (8 MFMA tiles: 2 load instructions feed 8
// MFMAs = 512 matrix-pipe cycles) and keeps the fragments of the next TWO 16-row k blocks in flight in two further
// register sets (~3.4 us of matrix work: HBM latency) - one wave per SIMD, ~300 VGPRs.  The four waves of a workgroup
// take four consecutive k ranges of the SAME output tile (intra-workgroup split-K) and combine through LDS in a fixed
// order (deterministic), so a launch writes `splitk` slabs while contracting over 4 * splitk ranges.
// Needs lda % 4 == 0, M % 4 == 0, ldb % 2 == 0, N % 2 == 0 and 16- / 8-byte aligned bases (kk_shape_ok).
constexpr int KK_U = 8;                 // k pairs per register set (16 k rows)
constexpr int KK_BM = 128, KK_BN = 64;
struct KKArgs {
    const float* A; const float* B; float* C;
    int lda, ldb, ldc, M, N, K;
    int k_chunk;                        // k rows per WAVE (multiple of 16)
    int tiles_m, tiles_n, n_fast;
    long long slab_stride;
};

__device__ __forceinline__ void kk_body(const KKArgs& a, int tile, int zs, float* __restrict__ lds) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
    const int tm = a.n_fast ? tile / a.tiles_n : tile % a.tiles_m;
    const int tn = a.n_fast ? tile % a.tiles_n : tile / a.tiles_m;
    const int m0 = tm * KK_BM, n0 = tn * KK_BN;
    const int kbeg = min(a.K, (zs * 4 + w) * a.k_chunk), kend = min(a.K, kbeg + a.k_chunk);
    // columns past the operand's end are clamped: they only feed outputs that are never stored
    const float* pa = a.A + min(m0 + 4 * l31, a.M - 4);
    const float* pb = a.B + min(n0 + 2 * l31, a.N - 2);
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 ra[3][KK_U];
    float2 rb[3][KK_U];
    auto load = [&](auto SS, int k0) {
        constexpr int S = decltype(SS)::value;
#pragma unroll
        for (int u = 0; u < KK_U; ++u) {
            const long long k = k0 + 2 * u + half;
            ra[S][u] = *reinterpret_cast<const float4*>(pa + k * a.lda);
            rb[S][u] = *reinterpret_cast<const float2*>(pb + k * a.ldb);
        }
    };
    auto mma4 = [&](const float4& x, const float2& y) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, y.x, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, y.y, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, y.x, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, y.y, acc[1][1], 0, 0, 0);
        acc[2][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, y.x, acc[2][0], 0, 0, 0);
        acc[2][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, y.y, acc[2][1], 0, 0, 0);
        acc[3][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, y.x, acc[3][0], 0, 0, 0);
        acc[3][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, y.y, acc[3][1], 0, 0, 0);
    };
    auto mma = [&](auto SS) {
        constexpr int S = decltype(SS)::value;
#pragma unroll
        for (int u = 0; u < KK_U; ++u) mma4(ra[S][u], rb[S][u]);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    constexpr int KB = 2 * KK_U;
    const int nblk = (kend - kbeg) / KB;
    // block b lives in register set b % 3; blocks b+1 and b+2 are in flight while block b feeds the MFMAs
    if (nblk > 0) load(S0{}, kbeg);
    if (nblk > 1) load(S1{}, kbeg + KB);
    int b = 0;
    for (; b + 3 <= nblk; b += 3) {
        if (b + 2 < nblk) load(S2{}, kbeg + (b + 2) * KB);
        mma(S0{});
        if (b + 3 < nblk) load(S0{}, kbeg + (b + 3) * KB);
        mma(S1{});
        if (b + 4 < nblk) load(S1{}, kbeg + (b + 4) * KB);
        mma(S2{});
    }
    if (b < nblk) {                  // one or two blocks left (their loads are already in flight: sets 0 and 1)
        mma(S0{});
        if (b + 1 < nblk) mma(S1{});
    }
    // k tail (< 16 rows): rows past kend contribute zeros
    for (int k0 = kbeg + nblk * KB; k0 < kend; k0 += 2) {
        const int k = k0 + half;
        const bool ok = k < kend;
        const long long kc = ok ? k : kend - 1;
        float4 x = *reinterpret_cast<const float4*>(pa + kc * a.lda);
        float2 y = *reinterpret_cast<const float2*>(pb + kc * a.ldb);
        if (!ok) { x = make_float4(0.f, 0.f, 0.f, 0.f); y = make_float2(0.f, 0.f); }
        mma4(x, y);
    }
    // combine the four waves' partial tiles: (w0 + w2) + (w1 + w3), always in this order.
    // acc[c][j][r] is output (m0 + 4 * row(r) + c, n0 + 2 * l31 + j), row(r) = (r & 3) + 8 (r >> 2) + 4 half
    float* buf = lds + (w & 1) * (KK_BM * KK_BN);
    auto at = [&](int c, int j, int r) { return (4 * ((r & 3) + 8 * (r >> 2) + 4 * half) + c) * KK_BN + 2 * l31 + j; };
    if (w < 2) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                *reinterpret_cast<float2*>(&buf[at(c, 0, r)]) = make_float2(acc[c][0][r], acc[c][1][r]);
    }
    __syncthreads();
    if (w >= 2) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float2* p = reinterpret_cast<float2*>(&buf[at(c, 0, r)]);
                const float2 v = *p;
                *p = make_float2(v.x + acc[c][0][r], v.y + acc[c][1][r]);
            }
    }
    __syncthreads();
    float* C = a.C + (long long)zs * a.slab_stride;
    const bool vec = (a.ldc & 3) == 0 && (((uintptr_t)C) & 15) == 0 && n0 + KK_BN <= a.N;
#pragma unroll
    for (int q = 0; q < KK_BM / 16; ++q) {
        const int row = (threadIdx.x >> 4) + 16 * q, col = 4 * (threadIdx.x & 15);
        const int m = m0 + row;
        if (m >= a.M) continue;
        const float4 x = *reinterpret_cast<const float4*>(&lds[row * KK_BN + col]);
        const float4 y = *reinterpret_cast<const float4*>(&lds[KK_BM * KK_BN + row * KK_BN + col]);
        const float4 s = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
        float* dst = C + (long long)m * a.ldc + n0 + col;
        if (vec) {
            *reinterpret_cast<float4*>(dst) = s;
        } else {
            if (n0 + col + 0 < a.N) dst[0] = s.x;
            if (n0 + col + 1 < a.N) dst[1] = s.y;
            if (n0 + col + 2 < a.N) dst[2] = s.z;
            if (n0 + col + 3 < a.N) dst[3] = s.w;
        }
    }
}

__global__ __launch_bounds__(256, 1) void gemm_kk_kernel(const KKArgs a) {
    extern __shared__ __attribute__((aligned(16))) float kk_lds[];
    kk_body(a, blockIdx.x, blockIdx.z, kk_lds);
}

constexpr int KK_GROUP_MAX = 8;
struct KKGroupArgs {
    KKArgs job[KK_GROUP_MAX];
    int wg_begin[KK_GROUP_MAX + 1];
    int tiles[KK_GROUP_MAX];
    int n;
};
__global__ __launch_bounds__(256, 1) void gemm_kk_group_kernel(const KKGroupArgs g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int j = 0;
    while (j + 1 < g.n && (int)blockIdx.x >= g.wg_begin[j + 1]) ++j;
    const int local = blockIdx.x - g.wg_begin[j];
    const int zs = local / g.tiles[j];
    kk_body(g.job[j], local - zs * g.tiles[j], zs, lds);
}

// Sum S slabs into dst: dst[i] = sum_s slabs[s*stride + i].  4 independent accumulation chains keep several
// slab loads in flight per lane (the slabs are L2/Infinity-Cache resident right after the GEMM).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, int S, long long stride,
                                                            float* __restrict__ dst, long long n) {
    const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= n) return;
    if (i4 + 3 < n && (stride & 3) == 0) {
        float4 acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        int k = 0;
        for (; k + 3 < S; k += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 v = *reinterpret_cast<const float4*>(slabs + (k + u) * stride + i4);
                acc[u].x += v.x; acc[u].y += v.y; acc[u].z += v.z; acc[u].w += v.w;
            }
        }
        for (; k < S; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(slabs + k * stride + i4);
            acc[0].x += v.x; acc[0].y += v.y; acc[0].z += v.z; acc[0].w += v.w;
        }
        float4 s;
        s.x = (acc[0].x + acc[1].x) + (acc[2].x + acc[3].x);
        s.y = (acc[0].y + acc[1].y) + (acc[2].y + acc[3].y);
        s.z = (acc[0].z + acc[1].z) + (acc[2].z + acc[3].z);
        s.w = (acc[0].w + acc[1].w) + (acc[2].w + acc[3].w);
        *reinterpret_cast<float4*>(dst + i4) = s;
    } else {
        for (long long i = i4; i < n && i < i4 + 4; ++i) {
            float s = slabs[i];
            for (int k = 1; k < S; ++k) s += slabs[k * stride + i];
            dst[i] = s;
        }
    }
}

}  // namespace

// Tile + split-K plan (measured on MI355X, scripts/gemm_bench*.py): for the skinny shapes of this network
// (one dimension = 128 or less) 64x64 tiles with the smallest power-of-two split that yields >= 1024 workgroups
// (4 per CU, all resident at once: 76-90 VGPRs, 26 KB LDS) are fastest or tied - at B = 8192 the wide GEMMs take
// 45-49 us there against 53-59 us at 512 workgroups; 128x128 tiles only pay once the 64x64 grid exceeds ~4096 tiles
// (large square-ish problems).  SCA_GEMM_WG_TARGET overrides the workgroup target (tuning knob).
struct GemmPlan { int wm, wn, S; };
static GemmPlan plan_gemm(int M, int N, int K) {
    const int kt = (K + BK - 1) / BK;
    const long long t11 = (long long)((M + 63) / 64) * ((N + 63) / 64);
    GemmPlan p{1, 1, 1};
    long long tiles = t11;
    if (t11 >= 4096 && M > 64 && N > 64 && K >= 256) {     // short K: a 128x128 tile never amortises its prologue
        p.wm = p.wn = 2;
        tiles = (long long)((M + 127) / 128) * ((N + 127) / 128);
    }
    static const int wg_target = [] { const char* v = getenv("SCA_GEMM_WG_TARGET"); return v ? atoi(v) : 1024; }();
    while (tiles * p.S < wg_target && p.S < 64 && kt / (p.S * 2) >= 2) p.S *= 2;
    return p;
}

extern "C" int sca_gemm_splitk_heuristic(int M, int N, int K) { return plan_gemm(M, N, K).S; }

// Split plan of the register-direct k-major kernel (gemm_kk_kernel): `S` slabs, each summed over 4 wave-level k ranges.
// One workgroup (4 waves = one per SIMD, ~300 VGPRs, 64 KB LDS) per CU: target 256 workgroups; a wave keeps at least
// three 16-row register sets of work.
static int plan_kk(int M, int N, int K) {
    const long long tiles = (long long)((M + KK_BM - 1) / KK_BM) * ((N + KK_BN - 1) / KK_BN);
    static const int wg_target = [] { const char* v = getenv("SCA_GEMM_KK_WG"); return v ? atoi(v) : 256; }();
    long long S = (wg_target + tiles - 1) / tiles;
    const long long smax = K / (4 * 48);
    if (S > smax) S = smax;
    if (S > 64) S = 64;
    if (S < 1) S = 1;
    return (int)S;
}
extern "C" int sca_gemm_kk_splitk_heuristic(int M, int N, int K) { return plan_kk(M, N, K); }
static bool kk_shape_ok(const float* A, int lda, const float* B, int ldb, int M, int N) {
    static const bool on = [] { const char* v = getenv("SCA_GEMM_KK"); return !v || atoi(v) != 0; }();
    return on && (lda & 3) == 0 && (M & 3) == 0 && M >= 4 && (ldb & 1) == 0 && (N & 1) == 0 && N >= 2 &&
           ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 7) == 0;
}
extern "C" int sca_gemm_kk_supported(const float* A, int lda, const float* B, int ldb, int M, int N) {
    return kk_shape_ok(A, lda, B, ldb, M, N) ? 1 : 0;
}
constexpr size_t KK_LDS_BYTES = 2 * KK_BM * KK_BN * sizeof(float);
static void fill_kk(KKArgs& a, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                    int splitk, long long slab_stride) {
    a.A = A; a.B = B; a.C = C; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
    const int per = (K + 4 * splitk - 1) / (4 * splitk);
    a.k_chunk = ((per + 2 * KK_U - 1) / (2 * KK_U)) * (2 * KK_U);
    a.tiles_m = (M + KK_BM - 1) / KK_BM;
    a.tiles_n = (N + KK_BN - 1) / KK_BN;
    a.n_fast = a.tiles_n > a.tiles_m;
    a.slab_stride = slab_stride;
}

// C (+ splitk slabs) = A^T B for two k-major operands (amode == bmode == 1), register-direct kernel.
extern "C" int sca_gemm_kk_f32(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                               int splitk, long long slab_stride, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0 || splitk < 1 || lda < M || ldb < N || ldc < N) return -1;
    if (!kk_shape_ok(A, lda, B, ldb, M, N)) return -2;
    KKArgs a;
    fill_kk(a, A, lda, B, ldb, C, ldc, M, N, K, splitk, slab_stride);
    static const hipError_t attr = hipFuncSetAttribute((const void*)gemm_kk_kernel,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)KK_LDS_BYTES);
    if (attr != hipSuccess) return (int)attr;
    hipLaunchKernelGGL(gemm_kk_kernel, dim3(a.tiles_m * a.tiles_n, 1, splitk), dim3(256), KK_LDS_BYTES,
                       (hipStream_t)stream, a);
    SCA_CHECK_LAUNCH();
    return 0;
}

template <int AM, int BMD, int WM, int WN>
static void launch_gemm(const GemmArgs& a, dim3 grid, bool vec, hipStream_t st) {
    if (vec) hipLaunchKernelGGL((gemm_f32_kernel<AM, BMD, true, WM, WN>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((gemm_f32_kernel<AM, BMD, false, WM, WN>), grid, dim3(256), 0, st, a);
}
template <int AM, int BMD>
static void launch_gemm_tile(const GemmArgs& a, int wm, int wn, int tiles_n, int S, bool vec, hipStream_t st) {
    dim3 grid(a.tiles_m * tiles_n, 1, S);
    if (wm == 2 && wn == 2) launch_gemm<AM, BMD, 2, 2>(a, grid, vec, st);
    else if (wm == 1 && wn == 2) launch_gemm<AM, BMD, 1, 2>(a, grid, vec, st);
    else if (wm == 2 && wn == 1) launch_gemm<AM, BMD, 2, 1>(a, grid, vec, st);
    else launch_gemm<AM, BMD, 1, 1>(a, grid, vec, st);
}

// tile_wm / tile_wn in {1,2} select the block tile (64*wm) x (64*wn); 0 = choose automatically.
extern "C" int sca_gemm_f32_cfg(int amode, int bmode, const float* A, int lda, const float* B, int ldb, float* C,
                                int ldc, int M, int N, int K, int splitk, long long slab_stride, const float* bias,
                                int tile_wm, int tile_wn, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0 || splitk < 1 || (amode | bmode) & ~1) return -1;
    if (tile_wm < 0 || tile_wm > 2 || tile_wn < 0 || tile_wn > 2) return -1;
    if (tile_wm == 0 || tile_wn == 0) {
        const GemmPlan p = plan_gemm(M, N, K);
        tile_wm = p.wm; tile_wn = p.wn;
    }
    GemmArgs a;
    a.A = A; a.B = B; a.C = C; a.bias = bias;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.M = M; a.N = N; a.K = K;
    const int kt = (K + BK - 1) / BK;
    const int kt_per = (kt + splitk - 1) / splitk;
    a.k_chunk = kt_per * BK;
    a.slab_stride = slab_stride;
    a.nsplit = splitk;
    a.vec = 0;
    a.tiles_m = (M + 64 * tile_wm - 1) / (64 * tile_wm);
    const int tiles_n = (N + 64 * tile_wn - 1) / (64 * tile_wn);
    a.tiles_n = tiles_n;
    a.n_fast = tiles_n > a.tiles_m;
    const bool vec = (lda % 4 == 0) && (ldb % 4 == 0) && (((uintptr_t)A | (uintptr_t)B) % 16 == 0);
    hipStream_t st = (hipStream_t)stream;
    if (amode == 0 && bmode == 0) launch_gemm_tile<0, 0>(a, tile_wm, tile_wn, tiles_n, splitk, vec, st);
    else if (amode == 0 && bmode == 1) launch_gemm_tile<0, 1>(a, tile_wm, tile_wn, tiles_n, splitk, vec, st);
    else if (amode == 1 && bmode == 1) launch_gemm_tile<1, 1>(a, tile_wm, tile_wn, tiles_n, splitk, vec, st);
    else launch_gemm_tile<1, 0>(a, tile_wm, tile_wn, tiles_n, splitk, vec, st);
    SCA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sca_gemm_f32(int amode, int bmode, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                            int M, int N, int K, int splitk, long long slab_stride, const float* bias, void* stream) {
    // tile chosen for the caller's split count (the caller sized its slab workspace from the heuristic)
    GemmPlan p = plan_gemm(M, N, K);
    return sca_gemm_f32_cfg(amode, bmode, A, lda, B, ldb, C, ldc, M, N, K, splitk, slab_stride, bias, p.wm, p.wn,
                            stream);
}

extern "C" {
typedef struct {
    int32_t amode, bmode;        // must both be 1 (A^T B: contraction over the rows of both operands)
    const float* A; const float* B; float* C;
    int32_t lda, ldb, ldc, M, N, K, splitk, pad;
    long long slab_stride;
} sca_gemm_job;
}

extern "C" int sca_gemm_group_max_jobs(void) { return GROUP_MAX; }

extern "C" int sca_gemm_group_f32(const sca_gemm_job* jobs, int n, void* stream) {
    if (n <= 0) return 0;
    if (n > GROUP_MAX) return -2;
    GroupArgs g;
    int wg = 0;
    for (int j = 0; j < n; ++j) {
        const sca_gemm_job& J = jobs[j];
        if (J.amode != 1 || J.bmode != 1) return -2;
        if (J.M <= 0 || J.N <= 0 || J.K <= 0 || J.splitk < 1) return -1;
        GemmArgs& a = g.job[j];
        a.A = J.A; a.B = J.B; a.C = J.C; a.bias = nullptr;
        a.lda = J.lda; a.ldb = J.ldb; a.ldc = J.ldc;
        a.M = J.M; a.N = J.N; a.K = J.K;
        const int kt = (J.K + BK - 1) / BK;
        a.k_chunk = ((kt + J.splitk - 1) / J.splitk) * BK;
        a.slab_stride = J.slab_stride;
        a.nsplit = J.splitk;
        a.tiles_m = (J.M + 63) / 64;
        a.tiles_n = (J.N + 63) / 64;
        a.n_fast = a.tiles_n > a.tiles_m;
        g.tiles[j] = a.tiles_m * a.tiles_n;
        g.wg_begin[j] = wg;
        wg += g.tiles[j] * J.splitk;
        a.vec = (J.lda % 4 == 0) && (J.ldb % 4 == 0) && (((uintptr_t)J.A | (uintptr_t)J.B) % 16 == 0);
    }
    g.wg_begin[n] = wg;
    g.n = n;
    hipLaunchKernelGGL(gemm_group_kernel, dim3(wg), dim3(256), 0, (hipStream_t)stream, g);
    SCA_CHECK_LAUNCH();
    return 0;
}

// Many slabs of a short vector (the 128 MMD gradient slabs of [B][10]): one thread per (16-byte element, slab group) -
// 16 groups of a workgroup take every 16th slab each (all their loads in flight at once), their partial sums meet in
// LDS and are added in group order: a fixed summation order, ~16 x the memory parallelism of the kernel above.
__global__ __launch_bounds__(256) void slab_reduce_wide_kernel(const float* __restrict__ slabs, int S, long long stride,
                                                               float* __restrict__ dst, long long n4) {
    __shared__ float4 part[16][16];
    const int e = threadIdx.x & 15, g = threadIdx.x >> 4;
    const long long i4 = (long long)blockIdx.x * 16 + e;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i4 < n4) {
        const float4* p = reinterpret_cast<const float4*>(slabs) + i4;
        const long long st4 = stride >> 2;
        int k = g;
        for (; k + 48 < S; k += 64) {          // four slabs of this group per trip
            const float4 v0 = p[k * st4], v1 = p[(k + 16) * st4], v2 = p[(k + 32) * st4], v3 = p[(k + 48) * st4];
            acc.x += (v0.x + v1.x) + (v2.x + v3.x); acc.y += (v0.y + v1.y) + (v2.y + v3.y);
            acc.z += (v0.z + v1.z) + (v2.z + v3.z); acc.w += (v0.w + v1.w) + (v2.w + v3.w);
        }
        for (; k < S; k += 16) {
            const float4 v = p[k * st4];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    part[g][e] = acc;
    __syncthreads();
    if (g == 0 && i4 < n4) {
        float4 s = part[0][e];
#pragma unroll
        for (int q = 1; q < 16; ++q) {
            const float4 v = part[q][e];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        reinterpret_cast<float4*>(dst)[i4] = s;
    }
}

extern "C" int sca_splitk_reduce(const float* slabs, int S, long long slab_stride, float* dst, long long n,
                                 void* stream) {
    if (n <= 0) return 0;
    if (S >= 32 && (n & 3) == 0 && (slab_stride & 3) == 0 && (((uintptr_t)slabs | (uintptr_t)dst) & 15) == 0) {
        const long long n4 = n >> 2;
        hipLaunchKernelGGL(slab_reduce_wide_kernel, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, (hipStream_t)stream,
                           slabs, S, slab_stride, dst, n4);
        SCA_CHECK_LAUNCH();
        return 0;
    }
    const long long nthreads = (n + 3) / 4;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, slabs, S, slab_stride, dst, n);
    SCA_CHECK_LAUNCH();
    return 0;
}
