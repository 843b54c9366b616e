// Masked gene-program decoder of expiMap as a tile-skipping MFMA GEMM family (gfx950).
//
// Reference: MaskedLinear / MaskedCondLayers (scarches/models/expimap/modules.py:9-25,62-85) compute
//     logits = z @ (W o M)^T          W, M [G, L(+ext)] - M binary, density O(1 %) for Reactome-like gene programs
// as a dense GEMM over a weight that is ~99 % zeros, plus a G x L multiply every forward; autograd does the same for
// dz = dlogits @ (W o M) and dW = dlogits^T z.
//
// Here the genes are cut into tiles of 32 (one MFMA tile) and, once at model build, every tile gets the sorted list
// of gene programs that are NOT fully masked for its 32 genes (`tile_cols`, padded to an even count with a pointer to
// an all-zero column).  The (32 genes x 2 programs) sub-tiles of W o M that are entirely masked are simply not in the
// list, so none of the three products issues an MFMA for them:
//     forward  logits[b, tile]      = sum_{j < K_t} z[b, cols[j]] * Wc[tile][j][:]           K_t instead of L k-steps
//     dX       dz[b, cols[j]]      += sum_{g in tile} dl[b, g] * Wc[tile][j][g]
//     dW       dWc[tile][j][g]      = sum_b z[b, cols[j]] * dl[b, g]
// Wc [NT][Kmax][32] is the compacted weight (packed per step from the live parameters: sca_masked_pack), z rows are
// staged in LDS (forward: gathered by column index from there; dX: the per-workgroup dz accumulator lives there), and
// the k loops are plain v_mfma_f32_32x32x2_f32 chains (exact fp32).
#include "common.h"

typedef float mg_f32x16 __attribute__((ext_vector_type(16)));

namespace {

struct MaskedArgs {
    const int32_t* tile_k;      // [NT]  active program count of a gene tile (even)
    const int32_t* tile_cols;   // [NT][Kmax] program (column) indices, padded with Zt (= the zero column)
    const float* Wc;            // [NT][Kmax][32] compacted weights
    int NT, Kmax, G, Zt, B;
};

// Wc[t][j][n] = Weff[32 t + n][cols[j]]  (0 for padding columns / genes past G)
__global__ __launch_bounds__(256) void masked_pack_kernel(const float* __restrict__ Weff, int ldw, MaskedArgs a,
                                                          float* __restrict__ Wc) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)a.NT * a.Kmax * 32;
    if (i >= total) return;
    const int n = (int)(i & 31);
    const long long tj = i >> 5;
    const int t = (int)(tj / a.Kmax), j = (int)(tj - (long long)t * a.Kmax);
    const int g = 32 * t + n;
    float v = 0.f;
    if (j < a.tile_k[t] && g < a.G) {
        const int c = a.tile_cols[(long long)t * a.Kmax + j];
        if (c < a.Zt) v = Weff[(long long)g * ldw + c];
    }
    Wc[i] = v;
}

// dWeff[32 t + n][cols[j]] = sum_s dWc[s][t][j][n]   (dWeff pre-zeroed; every (gene, column) pair occurs at most once)
__global__ __launch_bounds__(256) void masked_unpack_kernel(const float* __restrict__ dWc, int S, long long stride,
                                                            MaskedArgs a, float* __restrict__ dWeff, int ldw) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)a.NT * a.Kmax * 32;
    if (i >= total) return;
    const int n = (int)(i & 31);
    const long long tj = i >> 5;
    const int t = (int)(tj / a.Kmax), j = (int)(tj - (long long)t * a.Kmax);
    const int g = 32 * t + n;
    if (j >= a.tile_k[t] || g >= a.G) return;
    const int c = a.tile_cols[(long long)t * a.Kmax + j];
    if (c >= a.Zt) return;
    float v = 0.f;
    for (int s = 0; s < S; ++s) v += dWc[s * stride + i];
    dWeff[(long long)g * ldw + c] = v;
}

// ---- forward ----------------------------------------------------------------------------------------------------
// Workgroup = 64 rows of z (staged in LDS with an odd pitch and a trailing zero column) x a contiguous range of gene
// tiles.  Wave w: rows 32 (w & 1) .., tiles t0 + (w >> 1), + 2, ...  A fragment = LDS gather zs[row][cols[k]], B
// fragment = Wc[t][k][lane] straight from global / L2 (128-byte rows).
constexpr int MF_ROWS = 64;
__global__ __launch_bounds__(256) void masked_fwd_kernel(const float* __restrict__ z, int ldz, MaskedArgs a,
                                                         float* __restrict__ out, int ldo, int tiles_per_wg) {
    extern __shared__ __attribute__((aligned(16))) float zs[];
    const int ld = (a.Zt + 1) | 1;
    const int r0 = blockIdx.x * MF_ROWS;
    for (int i = threadIdx.x; i < MF_ROWS * ld; i += 256) {
        const int r = i / ld, c = i - r * ld;
        zs[i] = (r0 + r < a.B && c < a.Zt) ? z[(long long)(r0 + r) * ldz + c] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
    const int rb = 32 * (w & 1);
    const int t_beg = blockIdx.y * tiles_per_wg, t_end = min(a.NT, t_beg + tiles_per_wg);
    const float* zrow = zs + (rb + l31) * ld;
    for (int t = t_beg + (w >> 1); t < t_end; t += 2) {
        const int K = a.tile_k[t];
        const int32_t* cols = a.tile_cols + (long long)t * a.Kmax;
        const float* wc = a.Wc + (long long)t * a.Kmax * 32 + l31;
        mg_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        int k = 0;
        for (; k + 8 <= K; k += 8) {            // four MFMAs per trip: their operand loads are issued together
            float av[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kk = k + 2 * u + half;
                av[u] = zrow[cols[kk]];
                bv[u] = wc[(long long)kk * 32];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
        }
        for (; k < K; k += 2) {
            const int kk = k + half;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(zrow[cols[kk]], wc[(long long)kk * 32], acc, 0, 0, 0);
        }
        const int g = 32 * t + l31;
        if (g < a.G) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = r0 + rb + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < a.B) out[(long long)row * ldo + g] = acc[r];
            }
        }
    }
}

// ---- dX ----------------------------------------------------------------------------------------------------------
// Workgroup = 128 rows x a range of gene tiles; wave w owns rows 32 w .. exclusively (no races, fixed order) and keeps
// their dz accumulator in LDS [128][ld].  Per gene tile: P[32 rows x K_t] = dl[rows, tile] (A, 16 floats per lane via
// float4 loads, k pairs (i, 16 + i)) x Wc[tile][j][genes] (B, row j of the compact tile), scattered into the LDS
// accumulator at columns cols[j].  The launch writes one dz slab per gene-tile range (the consumer sums them).
constexpr int MX_ROWS = 128;
__global__ __launch_bounds__(256) void masked_dx_kernel(const float* __restrict__ dl, int ldl, MaskedArgs a,
                                                        float* __restrict__ dz, long long slab_stride,
                                                        int tiles_per_wg) {
    extern __shared__ __attribute__((aligned(16))) float acc_s[];
    const int ld = (a.Zt + 1) | 1;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
    const int r0 = blockIdx.x * MX_ROWS + 32 * w;
    float* mine = acc_s + (32 * w) * ld;
    for (int i = lane; i < 32 * ld; i += 64) mine[i] = 0.f;
    const int t_beg = blockIdx.y * tiles_per_wg, t_end = min(a.NT, t_beg + tiles_per_wg);
    const int row = min(r0 + l31, a.B - 1);
    const bool vec = (ldl & 3) == 0 && (((uintptr_t)dl) & 15) == 0;
    for (int t = t_beg; t < t_end; ++t) {
        const int K = a.tile_k[t];
        const int32_t* cols = a.tile_cols + (long long)t * a.Kmax;
        const int g0 = 32 * t + 16 * half;
        // A: dl[row][g0 .. g0 + 16) - genes past G read as zeros
        float av[16];
        const float* ap = dl + (long long)row * ldl + g0;
        if (vec && g0 + 16 <= a.G) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(ap + 4 * q);
                av[4 * q] = v.x; av[4 * q + 1] = v.y; av[4 * q + 2] = v.z; av[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) av[q] = g0 + q < a.G ? ap[q] : 0.f;
        }
        for (int nb = 0; nb < K; nb += 32) {
            // B: row j = nb + l31 of the compact tile, genes 16 half .. + 16 (rows past K are zero-padded by the pack
            // kernel up to Kmax; past Kmax they are clamped and their outputs dropped)
            const int j = min(nb + l31, a.Kmax - 1);
            const float* bp = a.Wc + ((long long)t * a.Kmax + j) * 32 + 16 * half;
            float bv[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(bp + 4 * q);
                bv[4 * q] = v.x; bv[4 * q + 1] = v.y; bv[4 * q + 2] = v.z; bv[4 * q + 3] = v.w;
            }
            mg_f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
            if (nb + l31 < K) {
                const int c = cols[nb + l31];           // (padding entries point at the zero column Zt: harmless)
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[((r & 3) + 8 * (r >> 2) + 4 * half) * ld + c] += acc[r];
            }
        }
    }
    // the wave's rows -> its slab
    float* dst = dz + (long long)blockIdx.y * slab_stride;
    for (int i = lane; i < 32 * a.Zt; i += 64) {
        const int r = i / a.Zt, c = i - r * a.Zt;
        if (r0 + r < a.B) dst[(long long)(r0 + r) * a.Zt + c] = mine[r * ld + c];
    }
}

// ---- dW ----------------------------------------------------------------------------------------------------------
// Workgroup = one gene tile x a range of rows; the four waves split the rows, contract over them with
// A(m = j, k = b) = z[b][cols[j]] (gathered from global / L1) and B(k = b, n = g) = dl[b][tile genes] (128-byte rows),
// and combine through LDS in a fixed order.  Output: dWc slab [NT][Kmax][32] of this row range.
constexpr int MW_MB_MAX = 8;   // at most 8 x 32 = 256 active programs per gene tile in registers
template <int MW_MB>           // accumulator blocks of 32 programs kept per wave (>= ceil(Kmax / 32))
__global__ __launch_bounds__(256) void masked_dw_kernel(const float* __restrict__ z, int ldz,
                                                        const float* __restrict__ dl, int ldl, MaskedArgs a,
                                                        float* __restrict__ dWc, long long slab_stride,
                                                        int rows_per_wg) {
    extern __shared__ __attribute__((aligned(16))) float red[];       // [3][Kpad][32]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
    const int t = blockIdx.x;
    const int K = a.tile_k[t];
    const int nmb = (K + 31) / 32;
    const int32_t* cols = a.tile_cols + (long long)t * a.Kmax;
    const int b_beg = blockIdx.y * rows_per_wg, b_end = min(a.B, b_beg + rows_per_wg);
    const int per = (((b_end - b_beg + 3) / 4) + 1) & ~1;             // rows per wave (even)
    const int wb = b_beg + w * per, we = min(b_end, wb + per);
    int col[MW_MB];
#pragma unroll
    for (int m = 0; m < MW_MB; ++m) {
        const int j = 32 * m + l31;
        col[m] = (m < nmb && j < K) ? min(cols[j], a.Zt - 1) : 0;      // the zero column reads as a valid one times 0
    }
    float colmask[MW_MB];
#pragma unroll
    for (int m = 0; m < MW_MB; ++m) {
        const int j = 32 * m + l31;
        colmask[m] = (m < nmb && j < K && cols[j] < a.Zt) ? 1.f : 0.f;
    }
    mg_f32x16 acc[MW_MB];
#pragma unroll
    for (int m = 0; m < MW_MB; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const int g = min(32 * t + l31, a.G - 1);
    const float gmask = 32 * t + l31 < a.G ? 1.f : 0.f;
    for (int b = wb; b < we; b += 2) {
        const int bb = b + half;
        const bool ok = bb < we;
        const long long br = ok ? bb : we - 1;
        const float bv = ok ? dl[br * ldl + g] * gmask : 0.f;
        const float* zr = z + br * ldz;
#pragma unroll
        for (int m = 0; m < MW_MB; ++m) {
            if (m < nmb) {
                const float av = ok ? zr[col[m]] * colmask[m] : 0.f;
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[m], 0, 0, 0);
            }
        }
    }
    // combine: waves 1..3 park their tiles, wave 0 adds them in order and stores
    const int Kpad = 32 * nmb;
    if (w > 0) {
        float* p = red + (long long)(w - 1) * Kpad * 32;
#pragma unroll
        for (int m = 0; m < MW_MB; ++m)
            if (m < nmb)
#pragma unroll
                for (int r = 0; r < 16; ++r) p[(32 * m + (r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = acc[m][r];
    }
    __syncthreads();
    if (w == 0) {
        float* dst = dWc + (long long)blockIdx.y * slab_stride + (long long)t * a.Kmax * 32;
#pragma unroll
        for (int m = 0; m < MW_MB; ++m)
            if (m < nmb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const int o = j * 32 + l31;
                    const float v = ((acc[m][r] + red[o]) + red[Kpad * 32 + o]) + red[2 * Kpad * 32 + o];
                    if (j < a.Kmax) dst[o] = v;
                }
    }
}

}  // namespace

extern "C" {

// Workgroup plan shared by the engine (workspace sizes) and the launchers.
int sca_masked_fwd_tiles_per_wg(int B, int NT) {
    const int row_tiles = (B + MF_ROWS - 1) / MF_ROWS;
    int chunks = (1024 + row_tiles - 1) / row_tiles;          // ~1024 workgroups
    if (chunks > (NT + 1) / 2) chunks = (NT + 1) / 2;          // a workgroup takes at least two tiles (two wave pairs)
    if (chunks < 1) chunks = 1;
    return (NT + chunks - 1) / chunks;
}
int sca_masked_dx_slabs(int B, int NT) {
    const int row_tiles = (B + MX_ROWS - 1) / MX_ROWS;
    int chunks = (256 + row_tiles - 1) / row_tiles;            // one 148 KB workgroup per CU
    if (chunks > NT) chunks = NT;
    if (chunks > 32) chunks = 32;
    if (chunks < 1) chunks = 1;
    const int per = (NT + chunks - 1) / chunks;
    return (NT + per - 1) / per;
}
int sca_masked_dw_slabs(int B, int NT) {
    int chunks = (768 + NT - 1) / NT;
    const int maxc = (B + 255) / 256;                           // a workgroup contracts over at least 256 rows
    if (chunks > maxc) chunks = maxc;
    if (chunks < 1) chunks = 1;
    const int per = (B + chunks - 1) / chunks;
    return (B + per - 1) / per;
}
int sca_masked_max_cols(void) { return 32 * MW_MB_MAX; }

static int masked_check(const int32_t* tile_k, const int32_t* tile_cols, int NT, int Kmax, int G, int Zt) {
    if (!tile_k || !tile_cols || NT != (G + 31) / 32 || Kmax < 2 || (Kmax & 1) || Zt < 1) return -1;
    if (Kmax > 32 * MW_MB_MAX) return -2;
    return 0;
}

int sca_masked_pack(const float* Weff, int ldw, const int32_t* tile_k, const int32_t* tile_cols, int NT, int Kmax,
                    int G, int Zt, float* Wc, void* stream) {
    const int rc = masked_check(tile_k, tile_cols, NT, Kmax, G, Zt);
    if (rc) return rc;
    MaskedArgs a{tile_k, tile_cols, nullptr, NT, Kmax, G, Zt, 0};
    const long long total = (long long)NT * Kmax * 32;
    hipLaunchKernelGGL(masked_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       Weff, ldw, a, Wc);
    SCA_CHECK_LAUNCH();
    return 0;
}

int sca_masked_unpack(const float* dWc, int S, long long slab_stride, const int32_t* tile_k, const int32_t* tile_cols,
                      int NT, int Kmax, int G, int Zt, float* dWeff, int ldw, void* stream) {
    const int rc = masked_check(tile_k, tile_cols, NT, Kmax, G, Zt);
    if (rc) return rc;
    MaskedArgs a{tile_k, tile_cols, nullptr, NT, Kmax, G, Zt, 0};
    const long long total = (long long)NT * Kmax * 32;
    hipLaunchKernelGGL(masked_unpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       dWc, S, slab_stride, a, dWeff, ldw);
    SCA_CHECK_LAUNCH();
    return 0;
}

int sca_masked_fwd(const float* z, int ldz, const float* Wc, const int32_t* tile_k, const int32_t* tile_cols, int NT,
                   int Kmax, int B, int G, int Zt, float* out, int ldo, void* stream) {
    if (B <= 0) return 0;
    const int rc = masked_check(tile_k, tile_cols, NT, Kmax, G, Zt);
    if (rc) return rc;
    if (ldz < Zt || ldo < G) return -1;
    MaskedArgs a{tile_k, tile_cols, Wc, NT, Kmax, G, Zt, B};
    const int per = sca_masked_fwd_tiles_per_wg(B, NT);
    const size_t shm = (size_t)MF_ROWS * ((Zt + 1) | 1) * sizeof(float);
    if (shm > 150 * 1024) return -2;
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute((const void*)masked_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr_done = true;
    }
    dim3 grid((B + MF_ROWS - 1) / MF_ROWS, (NT + per - 1) / per);
    hipLaunchKernelGGL(masked_fwd_kernel, grid, dim3(256), shm, (hipStream_t)stream, z, ldz, a, out, ldo, per);
    SCA_CHECK_LAUNCH();
    return 0;
}

int sca_masked_dx(const float* dl, int ldl, const float* Wc, const int32_t* tile_k, const int32_t* tile_cols, int NT,
                  int Kmax, int B, int G, int Zt, float* dz_slabs, long long slab_stride, void* stream) {
    if (B <= 0) return 0;
    const int rc = masked_check(tile_k, tile_cols, NT, Kmax, G, Zt);
    if (rc) return rc;
    if (ldl < G || slab_stride < (long long)B * Zt) return -1;
    MaskedArgs a{tile_k, tile_cols, Wc, NT, Kmax, G, Zt, B};
    const int S = sca_masked_dx_slabs(B, NT);
    const int per = (NT + S - 1) / S;
    const size_t shm = (size_t)MX_ROWS * ((Zt + 1) | 1) * sizeof(float);
    if (shm > 158 * 1024) return -2;
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute((const void*)masked_dx_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
        attr_done = true;
    }
    dim3 grid((B + MX_ROWS - 1) / MX_ROWS, S);
    hipLaunchKernelGGL(masked_dx_kernel, grid, dim3(256), shm, (hipStream_t)stream, dl, ldl, a, dz_slabs, slab_stride,
                       per);
    SCA_CHECK_LAUNCH();
    return 0;
}

int sca_masked_dw(const float* z, int ldz, const float* dl, int ldl, const int32_t* tile_k, const int32_t* tile_cols,
                  int NT, int Kmax, int B, int G, int Zt, float* dWc_slabs, long long slab_stride, void* stream) {
    if (B <= 0) return 0;
    const int rc = masked_check(tile_k, tile_cols, NT, Kmax, G, Zt);
    if (rc) return rc;
    if (ldz < Zt || ldl < G || slab_stride < (long long)NT * Kmax * 32) return -1;
    MaskedArgs a{tile_k, tile_cols, nullptr, NT, Kmax, G, Zt, B};
    const int S = sca_masked_dw_slabs(B, NT);
    const int per = (B + S - 1) / S;
    const size_t shm = (size_t)3 * (((Kmax + 31) / 32) * 32) * 32 * sizeof(float);
    if (shm > 100 * 1024) return -2;
    dim3 grid(NT, S);
    const int nmb = (Kmax + 31) / 32;
#define SCA_DW_CASE(NB)                                                                                              \
    {                                                                                                                \
        static bool attr_done = false;                                                                               \
        if (!attr_done) {                                                                                            \
            hipFuncSetAttribute((const void*)masked_dw_kernel<NB>, hipFuncAttributeMaxDynamicSharedMemorySize,       \
                                100 * 1024);                                                                         \
            attr_done = true;                                                                                        \
        }                                                                                                            \
        hipLaunchKernelGGL(masked_dw_kernel<NB>, grid, dim3(256), shm, (hipStream_t)stream, z, ldz, dl, ldl, a,      \
                           dWc_slabs, slab_stride, per);                                                             \
    }
    if (nmb <= 1) SCA_DW_CASE(1)
    else if (nmb <= 2) SCA_DW_CASE(2)
    else if (nmb <= 3) SCA_DW_CASE(3)
    else if (nmb <= 4) SCA_DW_CASE(4)
    else if (nmb <= 6) SCA_DW_CASE(6)
    else SCA_DW_CASE(8)
#undef SCA_DW_CASE
    SCA_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
