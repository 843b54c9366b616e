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

// Two copies of the compacted weight, n = NT * Kmax * 32 floats each:
//   Wc [t][j][g]            = Weff[32 t + g][cols[j]]   (0 for padding columns / genes past G)   - dX reads rows j
//   WcF[t][q][lane][u]      = Wc[t][8 q + 2 u + (lane >> 5)][lane & 31]                           - forward: the B
//                             fragments of four consecutive MFMAs of a lane in one 16-byte load
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
    const int q = j >> 3, u = (j & 7) >> 1, h = j & 1;
    Wc[total + (((long long)t * (a.Kmax >> 3) + q) * 64 + 32 * h + n) * 4 + u] = v;
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

// rows [r0, r0 + nrows) of a row-major matrix -> LDS tile [nrows][ld] with zeros in columns ncol .. ld and in rows past
// `rows_end`; wave w of 4 takes rows w, w + 4, ..., four of them per trip (their loads are issued together), a lane 16
// bytes at a time when the source allows it
__device__ __forceinline__ void stage_rows(const float* __restrict__ src, int lds_, long long r0, long long rows_end,
                                           int nrows, int ncol, float* dst, int ld, int w, int lane) {
    const bool vec = (lds_ & 3) == 0 && (ncol & 3) == 0 && (((uintptr_t)src) & 15) == 0;
    for (int rb = w; rb < nrows; rb += 16) {
        if (vec) {
            for (int c4 = lane; 4 * c4 < ncol; c4 += 64) {
                float4 v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = rb + 4 * i;
                    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (r < nrows && r0 + r < rows_end) v[i] = *reinterpret_cast<const float4*>(src + (r0 + r) * lds_ + 4 * c4);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = rb + 4 * i;
                    if (r < nrows) {
                        float* dp = dst + r * ld + 4 * c4;
                        dp[0] = v[i].x; dp[1] = v[i].y; dp[2] = v[i].z; dp[3] = v[i].w;
                    }
                }
            }
        } else {
            for (int c = lane; c < ncol; c += 64) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = rb + 4 * i;
                    if (r < nrows) dst[r * ld + c] = r0 + r < rows_end ? src[(r0 + r) * lds_ + c] : 0.f;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = rb + 4 * i;
            if (r < nrows)
                for (int c = ncol + lane; c < ld; c += 64) dst[r * ld + c] = 0.f;
        }
    }
}

// The active-column counts of a workgroup's tile range, fetched ONCE into registers (lane l of register j = tile
// t_beg + 64 j + l): the kernels' tile loops then read a count with v_readlane instead of a dependent memory load,
// which would drain every prefetch in flight.  Ranges are at most 256 tiles (the launchers see to that).
constexpr int MK_RANGE = 256;
struct TileK {
    int r[4];
};
__device__ __forceinline__ TileK load_tile_counts(const int32_t* __restrict__ tile_k, int t_beg, int t_end, int lane) {
    TileK k;
#pragma unroll
    for (int j = 0; j < 4; ++j) k.r[j] = t_beg + 64 * j + lane < t_end ? tile_k[t_beg + 64 * j + lane] : 0;
    return k;
}
__device__ __forceinline__ int tile_count(const TileK& k, int i) {      // i: wave-uniform index into the range
    const int l = i & 63, j = i >> 6;
    const int v0 = __builtin_amdgcn_readlane(k.r[0], l), v1 = __builtin_amdgcn_readlane(k.r[1], l);
    const int v2 = __builtin_amdgcn_readlane(k.r[2], l), v3 = __builtin_amdgcn_readlane(k.r[3], l);
    return j == 0 ? v0 : j == 1 ? v1 : j == 2 ? v2 : v3;
}

// ---- forward ----------------------------------------------------------------------------------------------------
// Workgroup = 64 rows of z (staged in LDS with an odd pitch and a trailing zero column) x a contiguous range of gene
// tiles.  Wave w: rows 32 (w & 1) .., tiles t0 + (w >> 1), + 2, ...  Per tile a lane holds the tile's column list in
// NCH registers (lane l of register ch = column 64 ch + l) and ALL its B fragments (one float4 of WcF per 8 columns);
// the A fragment of an MFMA is zs[row(lane)][column], the column broadcast out of the list register with v_readlane.
// The list and the B fragments of the wave's NEXT tile are loaded before the MFMAs of the current one.  Columns go in
// blocks of 16 (8 MFMAs); padding columns (index Zt) multiply the zero column by the zero rows that sca_masked_pack
// writes past tile_k.
constexpr int MF_ROWS = 64;

template <int NCH>      // 64-column chunks of a tile's list held in registers: NCH >= ceil(Kmax / 64)
__global__ __launch_bounds__(256) void masked_fwd_kernel(const float* __restrict__ z, int ldz, MaskedArgs a,
                                                         float* __restrict__ out, int ldo, int tiles_per_wg) {
    extern __shared__ __attribute__((aligned(16))) float zs[];
    const int ld = (a.Zt + 1) | 1;
    const int r0 = blockIdx.x * MF_ROWS;
    const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    stage_rows(z, ldz, r0, a.B, MF_ROWS, a.Zt, zs, ld, w, lane);
    __syncthreads();
    const int rb = 32 * (w & 1);
    const int t_beg = blockIdx.y * tiles_per_wg, t_end = min(a.NT, t_beg + tiles_per_wg);
    const float* zrow = zs + (rb + l31) * ld;
    const int NQ = a.Kmax >> 3;
    const float4* wf = reinterpret_cast<const float4*>(a.Wc + (long long)a.NT * a.Kmax * 32) + lane;
    const TileK counts = load_tile_counts(a.tile_k, t_beg, t_end, lane);

    // (prefetch loads are unconditional and fixed in number - the compiler can then wait for the older tile's
    //  registers with a partial vmcnt instead of draining the queue; out-of-range slots re-read a valid one)
    auto load_tile = [&](int t, float4 (&Bq)[NCH * 8], int (&Cq)[NCH]) {
        const int32_t* cols = a.tile_cols + (long long)t * a.Kmax;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) Cq[ch] = cols[min(64 * ch + lane, a.Kmax - 1)];
#pragma unroll
        for (int q = 0; q < NCH * 8; ++q) Bq[q] = wf[((long long)t * NQ + min(q, NQ - 1)) * 64];
    };
    auto compute_tile = [&](int t, const float4 (&Bq)[NCH * 8], const int (&Cq)[NCH]) {
        const int nq = (tile_count(counts, t - t_beg) + 7) >> 3;
        mg_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // A fragments of block p (16 columns = 8 MFMAs): columns j (half 0) and j + 1 (half 1), static lane indices
        auto gather = [&](int p, float (&av)[8]) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int j = 16 * p + 2 * i;
                const int c0 = __builtin_amdgcn_readlane(Cq[j >> 6], j & 63);
                const int c1 = __builtin_amdgcn_readlane(Cq[j >> 6], (j & 63) + 1);
                av[i] = zrow[half ? c1 : c0];
            }
        };
        float av[2][8];
        if (nq > 0) gather(0, av[0]);
#pragma unroll
        for (int p = 0; p < NCH * 4; ++p) {
            if (2 * p < nq) {
                if (p + 1 < NCH * 4 && 2 * (p + 1) < nq) gather(p + 1, av[(p + 1) & 1]);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float4 b = Bq[2 * p + (i >> 2)];
                    float bv = (i & 3) == 0 ? b.x : (i & 3) == 1 ? b.y : (i & 3) == 2 ? b.z : b.w;
                    // a block of 16 columns may reach one group of 8 past the tile's count: it contributes zeros
                    if (i >= 4 && 2 * p + 1 >= nq) bv = 0.f;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[p & 1][i], bv, acc, 0, 0, 0);
                }
            }
        }
        const int g = 32 * t + l31;
        if (g < a.G) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = r0 + rb + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < a.B) out[(long long)row * ldo + g] = acc[r];
            }
        }
    };

    float4 B0[NCH * 8], B1[NCH * 8];
    int C0[NCH], C1[NCH];
    int t = t_beg + (w >> 1);
    const int t_last = a.NT - 1;
    load_tile(min(t, t_last), B0, C0);
    while (t < t_end) {
        load_tile(min(t + 2, t_last), B1, C1);
        compute_tile(t, B0, C0);
        t += 2;
        if (t >= t_end) break;
        load_tile(min(t + 2, t_last), B0, C0);
        compute_tile(t, B1, C1);
        t += 2;
    }
}

// ---- dX ----------------------------------------------------------------------------------------------------------
// Workgroup = 128 rows x a range of gene tiles; wave w owns rows 32 w .. exclusively (no races, fixed order) and keeps
// their dz accumulator in LDS [128][ld].  Per (gene tile, block of 32 active columns): P[32 rows x 32 columns] =
// dl[rows, tile] (A, 16 floats per lane via float4 loads, k pairs (i, 16 + i)) x Wc[tile][j][genes] (B, row j of the
// compact tile), added into the LDS accumulator at columns cols[j].  All operands of a tile (A, the B fragments of
// every block, the column indices) sit in one register set; the set of the wave's next non-empty tile is loaded
// before the MFMAs of the current one.  One dz slab per gene-tile range (the consumer sums them).
constexpr int MX_ROWS = 128;

template <int NBM>
struct DxRegs {
    float a[16];
    float b[NBM][16];
    int c[NBM];
    int K;
};

template <int NBM>      // blocks of 32 columns per tile held in registers: NBM >= ceil(Kmax / 32)
__global__ __launch_bounds__(256) void masked_dx_kernel(const float* __restrict__ dl, int ldl, MaskedArgs a,
                                                        float* __restrict__ dz, long long slab_stride,
                                                        int tiles_per_wg) {
    extern __shared__ __attribute__((aligned(16))) float acc_s[];
    const int ld = (a.Zt + 1) | 1;
    const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r0 = blockIdx.x * MX_ROWS + 32 * w;
    float* mine = acc_s + (32 * w) * ld;
    for (int i = lane; i < 32 * ld; i += 64) mine[i] = 0.f;
    const int t_beg = blockIdx.y * tiles_per_wg, t_end = min(a.NT, t_beg + tiles_per_wg);
    const int row = min(r0 + l31, a.B - 1);
    const bool vec = (ldl & 3) == 0 && (((uintptr_t)dl) & 15) == 0;

    const TileK counts = load_tile_counts(a.tile_k, t_beg, t_end, lane);
    auto next_tile = [&](int t) {               // first tile >= t with active columns
        while (t < t_end && tile_count(counts, t - t_beg) <= 0) ++t;
        return t;
    };
    // (unconditional, fixed number of loads per tile - see the forward kernel.  Genes past G: the B rows of the pack
    //  kernel are zero there, so A only has to be finite and in bounds: offsets are clamped into the row pitch.)
    auto load_tile = [&](int t, DxRegs<NBM>& R) {
        R.K = t < t_end ? tile_count(counts, t - t_beg) : 0;
        const int g0 = 32 * t + 16 * half;
        const float* ap = dl + (long long)row * ldl;
        if (vec) {
            const int lim = ldl - 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(ap + min(g0 + 4 * q, lim));
                R.a[4 * q] = v.x; R.a[4 * q + 1] = v.y; R.a[4 * q + 2] = v.z; R.a[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) R.a[q] = ap[min(g0 + q, a.G - 1)];
        }
#pragma unroll
        for (int m = 0; m < NBM; ++m) {
            // row j of the compact tile, genes 16 half .. + 16 (rows past tile_k are zeros up to Kmax; past Kmax the
            // row is clamped and its outputs go to the dump column Zt)
            const int j = 32 * m + l31;
            const int jc = min(j, a.Kmax - 1);
            const float* bp = a.Wc + ((long long)t * a.Kmax + jc) * 32 + 16 * half;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(bp + 4 * q);
                R.b[m][4 * q] = v.x; R.b[m][4 * q + 1] = v.y; R.b[m][4 * q + 2] = v.z; R.b[m][4 * q + 3] = v.w;
            }
            const int c = a.tile_cols[(long long)t * a.Kmax + jc];
            R.c[m] = j < R.K ? c : a.Zt;
        }
    };
    // The scatter of a block is deferred: its 16 LDS reads are issued before the MFMAs of the NEXT block (of this or
    // the next tile) and its adds / writes after them.  The chain starts with zeros aimed at the dump column Zt.
    mg_f32x16 pend;
#pragma unroll
    for (int r = 0; r < 16; ++r) pend[r] = 0.f;
    int pend_c = a.Zt;
    auto compute_tile = [&](const DxRegs<NBM>& R) {
#pragma unroll
        for (int m = 0; m < NBM; ++m) {
            if (32 * m < R.K) {
                // column pend_c of the wave's 32 rows: the 16 cells of a lane are distinct rows
                float* cp = mine + (4 * half) * ld + pend_c;
                float old[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) old[r] = cp[((r & 3) + 8 * (r >> 2)) * ld];
                mg_f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int q = 0; q < 16; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(R.a[q], R.b[m][q], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) cp[((r & 3) + 8 * (r >> 2)) * ld] = old[r] + pend[r];
                pend = acc;
                pend_c = R.c[m];
            }
        }
    };
    auto flush_pending = [&]() {
        float* cp = mine + (4 * half) * ld + pend_c;
#pragma unroll
        for (int r = 0; r < 16; ++r) cp[((r & 3) + 8 * (r >> 2)) * ld] += pend[r];
    };

    DxRegs<NBM> R0, R1;
    const int t_last = a.NT - 1;
    int t = next_tile(t_beg);
    load_tile(min(t, t_last), R0);
    while (t < t_end) {
        int tn = next_tile(t + 1);
        load_tile(min(tn, t_last), R1);
        compute_tile(R0);
        t = tn;
        if (t >= t_end) break;
        tn = next_tile(t + 1);
        load_tile(min(tn, t_last), R0);
        compute_tile(R1);
        t = tn;
    }
    flush_pending();
    // the wave's rows -> its slab
    float* dst = dz + (long long)blockIdx.y * slab_stride;
    for (int r = 0; r < 32; ++r) {
        if (r0 + r >= a.B) break;
        for (int c = lane; c < a.Zt; c += 64) dst[(long long)(r0 + r) * a.Zt + c] = mine[r * ld + c];
    }
}

// ---- dW ----------------------------------------------------------------------------------------------------------
// Workgroup = FOUR adjacent gene tiles (one per wave) x a range of rows, walked in sub-chunks of 32 rows that are
// double-buffered in LDS: zs[32][ld] = the z rows (coalesced, with a zero column at Zt), dls[32][128] = dl[rows, the 128
// genes of the four tiles].  While a sub-chunk is contracted the next one is already in flight into registers.  Wave w
// contracts over the rows with A(m = j, k = row) = zs[row][cols_w[j]] (LDS gather; lane <-> column), B(k = row,
// n = gene) = dls[row][32 w + lane], one accumulator block per 32 active columns of ITS tile - no combine across
// waves.  The number of blocks NB is the wave's own (a wave-uniform switch picks the instantiation of dw_run, so the
// accumulate loop has no branches and its accumulators stay in place); every instantiation executes the same barriers.
// Output: the dWc slab of this row range.
constexpr int MW_MB_MAX = 8;   // at most 8 x 32 = 256 active programs per gene tile in registers
constexpr int MW_ROWS = 32;
constexpr int MW_ZV = 10;      // 16-byte slots per thread per z row: 8 threads x 10 x 4 = 320 columns >= Zt

template <int NB, bool VEC>
__device__ __forceinline__ void dw_run(const float* __restrict__ z, int ldz, const float* __restrict__ dl, int ldl,
                                       const MaskedArgs& a, float* __restrict__ dst, bool t_ok, int K, int t,
                                       float* smem, int b_beg, int b_end, int w, int lane) {
    const int ld = (a.Zt + 1) | 1;
    const int buf_floats = MW_ROWS * ld + MW_ROWS * 128;
    const int l31 = lane & 31, half = lane >> 5;
    int col[NB];
#pragma unroll
    for (int m = 0; m < NB; ++m) {
        const int j = 32 * m + l31;
        col[m] = (t_ok && j < K) ? a.tile_cols[(long long)t * a.Kmax + j] : a.Zt;     // Zt: the zero column
    }
    mg_f32x16 acc[NB];
#pragma unroll
    for (int m = 0; m < NB; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    // staging roles.  z: thread -> row tid >> 3 (32 rows), slots (tid & 7) + 8 i of 4 columns; dl: rows (tid >> 5) + 8 i,
    // slot tid & 31 of the 128 genes.  Every load is unconditional with a clamped address; rows past the range are
    // zeroed in dl only (their z values are finite and multiply zeros), columns past Zt are not stored.
    const int g_base = 128 * blockIdx.x;
    const int zr = threadIdx.x >> 3, zc = threadIdx.x & 7;
    const int dr = threadIdx.x >> 5, dc = threadIdx.x & 31;
    float zreg[MW_ZV][4], dreg[4][4];
    auto load_regs = [&](int b0) {
        const float* sp = z + (long long)min(b0 + zr, b_end - 1) * ldz;
#pragma unroll
        for (int i = 0; i < MW_ZV; ++i) {
            const int c = 4 * (zc + 8 * i);
            if (VEC) {
                const float4 v = *reinterpret_cast<const float4*>(sp + min(c, a.Zt - 4));
                zreg[i][0] = v.x; zreg[i][1] = v.y; zreg[i][2] = v.z; zreg[i][3] = v.w;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) zreg[i][k] = sp[min(c + k, a.Zt - 1)];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = dr + 8 * i;
            const float keep = b0 + r < b_end ? 1.f : 0.f;
            const float* dp = dl + (long long)min(b0 + r, b_end - 1) * ldl;
            const int g = g_base + 4 * dc;
            if (VEC) {
                const float4 v = *reinterpret_cast<const float4*>(dp + min(g, ldl - 4));
                dreg[i][0] = v.x * keep; dreg[i][1] = v.y * keep; dreg[i][2] = v.z * keep; dreg[i][3] = v.w * keep;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) dreg[i][k] = dp[min(g + k, a.G - 1)] * keep;
            }
        }
    };
    auto store_lds = [&](int buf) {
        float* zs = smem + buf * buf_floats;
        float* dls = zs + MW_ROWS * ld;
#pragma unroll
        for (int i = 0; i < MW_ZV; ++i) {
            const int c = 4 * (zc + 8 * i);
            float* dp = zs + zr * ld + c;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (c + k < a.Zt) dp[k] = zreg[i][k];              // (columns Zt .. ld are zeroed once, by the kernel)
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<float4*>(dls + (dr + 8 * i) * 128 + 4 * dc) =
                make_float4(dreg[i][0], dreg[i][1], dreg[i][2], dreg[i][3]);
    };
    const int nchunk = (b_end - b_beg + MW_ROWS - 1) / MW_ROWS;
    load_regs(b_beg);
    store_lds(0);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        load_regs(min(b_beg + (c + 1) * MW_ROWS, b_end - 1));      // (the last trip re-reads valid rows and drops them)
        const float* zs = smem + (c & 1) * buf_floats;
        const float* dls = zs + MW_ROWS * ld;
        float bv[MW_ROWS / 2];
#pragma unroll
        for (int ks = 0; ks < MW_ROWS / 2; ++ks) bv[ks] = dls[(2 * ks + half) * 128 + 32 * w + l31];
#pragma unroll
        for (int m = 0; m < NB; ++m) {
            const float* zp = zs + half * ld + col[m];
            float av[MW_ROWS / 2];
#pragma unroll
            for (int ks = 0; ks < MW_ROWS / 2; ++ks) av[ks] = zp[2 * ks * ld];
#pragma unroll
            for (int ks = 0; ks < MW_ROWS / 2; ++ks)
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks], bv[ks], acc[m], 0, 0, 0);
        }
        if (c + 1 < nchunk) store_lds((c + 1) & 1);
        __syncthreads();
    }
    if (t_ok && K > 0) {
#pragma unroll
        for (int m = 0; m < NB; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (j < a.Kmax) dst[j * 32 + l31] = acc[m][r];
            }
    }
}

template <int MW_MB, bool VEC>     // MW_MB >= ceil(Kmax / 32): the largest per-wave instantiation compiled in
__global__ __launch_bounds__(256) void masked_dw_kernel(const float* __restrict__ z, int ldz,
                                                        const float* __restrict__ dl, int ldl, MaskedArgs a,
                                                        float* __restrict__ dWc, long long slab_stride,
                                                        int rows_per_wg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int ld = (a.Zt + 1) | 1;
    const int buf_floats = MW_ROWS * ld + MW_ROWS * 128;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t = 4 * blockIdx.x + w;
    const bool t_ok = t < a.NT;
    const int K = t_ok ? a.tile_k[t] : 0;
    const int nmb = __builtin_amdgcn_readfirstlane((K + 31) / 32);
    const int b_beg = blockIdx.y * rows_per_wg, b_end = min(a.B, b_beg + rows_per_wg);
    // zero columns Zt .. ld of both buffers (never written again)
    for (int i = threadIdx.x; i < 2 * MW_ROWS * (ld - a.Zt); i += 256) {
        const int per = ld - a.Zt, buf = i / (MW_ROWS * per), k = i - buf * MW_ROWS * per;
        smem[buf * buf_floats + (k / per) * ld + a.Zt + k % per] = 0.f;
    }
    float* dst = dWc + (long long)blockIdx.y * slab_stride + (long long)(t_ok ? t : 0) * a.Kmax * 32;
#define SCA_DW_RUN(N)                                                                                                \
    if (MW_MB >= N && (nmb == N || (N == 1 && nmb == 0)))                                                            \
        dw_run<(N <= MW_MB ? N : 1), VEC>(z, ldz, dl, ldl, a, dst, t_ok, K, t, smem, b_beg, b_end, w, lane);
    SCA_DW_RUN(1) else SCA_DW_RUN(2) else SCA_DW_RUN(3) else SCA_DW_RUN(4) else SCA_DW_RUN(5) else SCA_DW_RUN(6)
    else SCA_DW_RUN(7) else SCA_DW_RUN(8)
#undef SCA_DW_RUN
}

}  // namespace

// every instantiation may use the whole LDS of a CU
template <typename Kern>
static void allow_lds(Kern k, bool& done) {
    if (!done) {
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        done = true;
    }
}

extern "C" {

// Workgroup plan shared by the engine (workspace sizes) and the launchers.
int sca_masked_fwd_tiles_per_wg(int B, int NT) {
    const int row_tiles = (B + MF_ROWS - 1) / MF_ROWS;
    int chunks = (512 + row_tiles - 1) / row_tiles;           // ~512 workgroups: two 72 KB tiles of z per CU
    if (chunks > (NT + 1) / 2) chunks = (NT + 1) / 2;          // a workgroup takes at least two tiles (two wave pairs)
    if (chunks < (NT + MK_RANGE - 1) / MK_RANGE) chunks = (NT + MK_RANGE - 1) / MK_RANGE;
    if (chunks < 1) chunks = 1;
    return (NT + chunks - 1) / chunks;
}
int sca_masked_dx_slabs(int B, int NT) {
    const int row_tiles = (B + MX_ROWS - 1) / MX_ROWS;
    int chunks = (256 + row_tiles - 1) / row_tiles;            // one 148 KB workgroup per CU
    if (chunks > NT) chunks = NT;
    if (chunks > 32) chunks = 32;
    if (chunks < (NT + MK_RANGE - 1) / MK_RANGE) chunks = (NT + MK_RANGE - 1) / MK_RANGE;   // <= 256 tiles per range
    if (chunks < 1) chunks = 1;
    const int per = (NT + chunks - 1) / chunks;
    return (NT + per - 1) / per;
}
int sca_masked_dw_slabs(int B, int NT) {
    const int quads = (NT + 3) / 4;                             // a workgroup = four gene tiles
    int chunks = 256 / quads;                                   // one double-buffered workgroup per CU
    const int maxc = (B + 127) / 128;                           // ... contracting over at least 128 rows
    if (chunks > maxc) chunks = maxc;
    if (chunks < 1) chunks = 1;
    const int per = (B + chunks - 1) / chunks;
    return (B + per - 1) / per;
}
int sca_masked_max_cols(void) { return 32 * MW_MB_MAX; }
int sca_masked_max_width(void) { return 304; }                  // Zt: the dX accumulator [128][Zt + 1 | 1] lives in LDS

static int masked_check(const int32_t* tile_k, const int32_t* tile_cols, int NT, int Kmax, int G, int Zt) {
    if (!tile_k || !tile_cols || NT != (G + 31) / 32 || Kmax < 8 || (Kmax & 7) || Zt < 1) return -1;
    if (Kmax > 32 * MW_MB_MAX || Zt > sca_masked_max_width()) return -2;
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
    if (ldz < Zt || ldo < G || (((uintptr_t)Wc) & 15)) return -1;
    MaskedArgs a{tile_k, tile_cols, Wc, NT, Kmax, G, Zt, B};
    const int per = sca_masked_fwd_tiles_per_wg(B, NT);
    const size_t shm = (size_t)MF_ROWS * ((Zt + 1) | 1) * sizeof(float);
    dim3 grid((B + MF_ROWS - 1) / MF_ROWS, (NT + per - 1) / per);
    const int nch = (Kmax + 63) / 64;
#define SCA_FWD_CASE(N)                                                                                              \
    {                                                                                                                \
        static bool done = false;                                                                                    \
        allow_lds(masked_fwd_kernel<N>, done);                                                                       \
        hipLaunchKernelGGL(masked_fwd_kernel<N>, grid, dim3(256), shm, (hipStream_t)stream, z, ldz, a, out, ldo,     \
                           per);                                                                                     \
    }
    if (nch <= 1) SCA_FWD_CASE(1)
    else if (nch == 2) SCA_FWD_CASE(2)
    else if (nch == 3) SCA_FWD_CASE(3)
    else SCA_FWD_CASE(4)
#undef SCA_FWD_CASE
    SCA_CHECK_LAUNCH();
    return 0;
}

int sca_masked_dx(const float* dl, int ldl, const float* Wc, const int32_t* tile_k, const int32_t* tile_cols, int NT,
                  int Kmax, int B, int G, int Zt, float* dz_slabs, long long slab_stride, void* stream) {
    if (B <= 0) return 0;
    const int rc = masked_check(tile_k, tile_cols, NT, Kmax, G, Zt);
    if (rc) return rc;
    if (ldl < G || slab_stride < (long long)B * Zt || (((uintptr_t)Wc) & 15)) return -1;
    MaskedArgs a{tile_k, tile_cols, Wc, NT, Kmax, G, Zt, B};
    const int S = sca_masked_dx_slabs(B, NT);
    const int per = (NT + S - 1) / S;
    const size_t shm = (size_t)MX_ROWS * ((Zt + 1) | 1) * sizeof(float);
    dim3 grid((B + MX_ROWS - 1) / MX_ROWS, S);
    const int nbm = (Kmax + 31) / 32;
#define SCA_DX_CASE(N)                                                                                               \
    {                                                                                                                \
        static bool done = false;                                                                                    \
        allow_lds(masked_dx_kernel<N>, done);                                                                        \
        hipLaunchKernelGGL(masked_dx_kernel<N>, grid, dim3(256), shm, (hipStream_t)stream, dl, ldl, a, dz_slabs,     \
                           slab_stride, per);                                                                        \
    }
    if (nbm <= 1) SCA_DX_CASE(1)
    else if (nbm == 2) SCA_DX_CASE(2)
    else if (nbm == 3) SCA_DX_CASE(3)
    else if (nbm == 4) SCA_DX_CASE(4)
    else if (nbm <= 6) SCA_DX_CASE(6)
    else SCA_DX_CASE(8)
#undef SCA_DX_CASE
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
    const int nmb = (Kmax + 31) / 32;
    // two buffers of [32][Zt + 1 | 1] z rows + [32][128] dl columns  (Zt <= 304: at most 113 KB; 10 float4 slots x 8
    // threads per row cover 320 columns)
    const size_t shm = (size_t)2 * (MW_ROWS * ((Zt + 1) | 1) + MW_ROWS * 128) * sizeof(float);
    dim3 grid((NT + 3) / 4, S);
    const bool vec = (ldz & 3) == 0 && (Zt & 3) == 0 && (ldl & 3) == 0 && ((((uintptr_t)z) | ((uintptr_t)dl)) & 15) == 0;
#define SCA_DW_LAUNCH(NB, V)                                                                                         \
    {                                                                                                                \
        static bool done = false;                                                                                    \
        allow_lds(masked_dw_kernel<NB, V>, done);                                                                    \
        hipLaunchKernelGGL((masked_dw_kernel<NB, V>), grid, dim3(256), shm, (hipStream_t)stream, z, ldz, dl, ldl, a, \
                           dWc_slabs, slab_stride, per);                                                             \
    }
#define SCA_DW_CASE(NB)                                                                                              \
    {                                                                                                                \
        if (vec) SCA_DW_LAUNCH(NB, true) else SCA_DW_LAUNCH(NB, false)                                               \
    }
    if (nmb <= 1) SCA_DW_CASE(1)
    else if (nmb <= 2) SCA_DW_CASE(2)
    else if (nmb <= 3) SCA_DW_CASE(3)
    else if (nmb <= 4) SCA_DW_CASE(4)
    else if (nmb <= 6) SCA_DW_CASE(6)
    else SCA_DW_CASE(8)
#undef SCA_DW_CASE
#undef SCA_DW_LAUNCH
    SCA_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
