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
// register-prefetched one tile ahead (double-buffered LDS, one barrier per k-step).
// Split-K (gridDim.z) writes per-split slabs; the consumer kernel sums them (deterministic, no atomics).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BM = 128, BN = 128, BK = 16, LDT = 132;

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;   // optional [N], only applied when gridDim.z == 1
    int lda, ldb, ldc;
    int M, N, K;
    int k_chunk;         // multiple of BK
    long long slab_stride;
    int tiles_m;
};

// Load one 128(rows) x 16(k) operand tile into two float4 registers per thread.
template <int MODE, bool VEC>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, int ld, int R, int r0, int k0, int kend,
                                          float4 (&reg)[2]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = t + 256 * i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE == 0) {
            const int r = r0 + (idx >> 2), k = k0 + 4 * (idx & 3);
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
            const int k = k0 + (idx >> 5), r = r0 + 4 * (idx & 31);
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

template <int MODE>
__device__ __forceinline__ void store_tile(float* __restrict__ T, const float4 (&reg)[2]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = t + 256 * i;
        if (MODE == 0) {
            const int r = idx >> 2, kc = 4 * (idx & 3);
            T[(kc + 0) * LDT + r] = reg[i].x;
            T[(kc + 1) * LDT + r] = reg[i].y;
            T[(kc + 2) * LDT + r] = reg[i].z;
            T[(kc + 3) * LDT + r] = reg[i].w;
        } else {
            const int k = idx >> 5, r = 4 * (idx & 31);
            *reinterpret_cast<float4*>(&T[k * LDT + r]) = reg[i];
        }
    }
}

template <int AMODE, int BMODE, bool VEC>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const GemmArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * BK * LDT];
    constexpr int TSZ = BK * LDT;            // LDS image: [A0 | A1 | B0 | B1]

    const int tile = blockIdx.x;
    const int m0 = (tile % a.tiles_m) * BM, n0 = (tile / a.tiles_m) * BN;
    const int kbeg = blockIdx.z * a.k_chunk;
    const int kend = min(a.K, kbeg + a.k_chunk);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wm = w >> 1, wn = w & 1, l31 = lane & 31, half = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (kend - kbeg + BK - 1) / BK;
    float4 ra[2], rb[2];
    if (nk > 0) {
        load_tile<AMODE, VEC>(a.A, a.lda, a.M, m0, kbeg, kend, ra);
        load_tile<BMODE, VEC>(a.B, a.ldb, a.N, n0, kbeg, kend, rb);
        store_tile<AMODE>(lds, ra);
        store_tile<BMODE>(lds + 2 * TSZ, rb);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            load_tile<AMODE, VEC>(a.A, a.lda, a.M, m0, kbeg + (kt + 1) * BK, kend, ra);
            load_tile<BMODE, VEC>(a.B, a.ldb, a.N, n0, kbeg + (kt + 1) * BK, kend, rb);
        }
        const float* Ac = lds + cur * TSZ + half * LDT + wm * 64 + l31;
        const float* Bc = lds + (2 + cur) * TSZ + half * LDT + wn * 64 + l31;
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const float a0 = Ac[(2 * kk) * LDT], a1 = Ac[(2 * kk) * LDT + 32];
            const float b0 = Bc[(2 * kk) * LDT], b1 = Bc[(2 * kk) * LDT + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            store_tile<AMODE>(lds + (cur ^ 1) * TSZ, ra);
            store_tile<BMODE>(lds + (2 + (cur ^ 1)) * TSZ, rb);
        }
        __syncthreads();
    }

    float* C = a.C + (long long)blockIdx.z * a.slab_stride;
    const bool add_bias = a.bias != nullptr && gridDim.z == 1;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + l31;
            if (n >= a.N) continue;
            const float bv = add_bias ? a.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m < a.M) C[(long long)m * a.ldc + n] = acc[i][j][r] + bv;
            }
        }
}

// Sum S slabs (+ optional scale) into dst: dst[i] = sum_s slabs[s*stride + i].
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, int S, long long stride,
                                                            float* __restrict__ dst, long long n) {
    const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= n) return;
    if (i4 + 3 < n && (stride & 3) == 0) {
        float4 s = *reinterpret_cast<const float4*>(slabs + i4);
        for (int k = 1; k < S; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(slabs + k * stride + i4);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
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

extern "C" int sca_gemm_splitk_heuristic(int M, int N, int K) {
    const long long tiles = (long long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const int kt = (K + BK - 1) / BK;
    int S = 1;
    // aim for >= 512 workgroups (2 per CU) while keeping >= 4 k-steps per split
    while (tiles * S < 512 && S * 2 <= 64 && kt / (S * 2) >= 4) S *= 2;
    return S;
}

extern "C" int sca_gemm_f32(int amode, int bmode, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                            int M, int N, int K, int splitk, long long slab_stride, const float* bias, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0 || splitk < 1 || (amode | bmode) & ~1) return -1;
    GemmArgs a;
    a.A = A; a.B = B; a.C = C; a.bias = bias;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.M = M; a.N = N; a.K = K;
    const int kt = (K + BK - 1) / BK;
    const int kt_per = (kt + splitk - 1) / splitk;
    a.k_chunk = kt_per * BK;
    a.slab_stride = slab_stride;
    a.tiles_m = (M + BM - 1) / BM;
    const int tiles_n = (N + BN - 1) / BN;
    dim3 grid(a.tiles_m * tiles_n, 1, splitk), block(256);
    const bool vec = (lda % 4 == 0) && (ldb % 4 == 0) && (((uintptr_t)A | (uintptr_t)B) % 16 == 0);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(AM, BMD)                                                                  \
    do {                                                                                 \
        if (vec) hipLaunchKernelGGL((gemm_f32_kernel<AM, BMD, true>), grid, block, 0, st, a);  \
        else hipLaunchKernelGGL((gemm_f32_kernel<AM, BMD, false>), grid, block, 0, st, a);     \
    } while (0)
    if (amode == 0 && bmode == 0) LAUNCH(0, 0);
    else if (amode == 0 && bmode == 1) LAUNCH(0, 1);
    else if (amode == 1 && bmode == 1) LAUNCH(1, 1);
    else LAUNCH(1, 0);
#undef LAUNCH
    SCA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sca_splitk_reduce(const float* slabs, int S, long long slab_stride, float* dst, long long n,
                                 void* stream) {
    if (n <= 0) return 0;
    const long long nthreads = (n + 3) / 4;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, slabs, S, slab_stride, dst, n);
    SCA_CHECK_LAUNCH();
    return 0;
}
