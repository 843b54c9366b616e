// Weighted k-nearest-neighbour label transfer in the latent space (gfx950): brute-force euclidean top-k + the
// Gaussian-weighted vote.  Reference: scarches/utils/knn.py:10-139 - sklearn KNeighborsTransformer(algorithm="brute",
// metric="euclidean").kneighbors for the neighbours, then
//     std_i = std(d_i1..d_ik) ; w_ij = exp(-d_ij / (2 / std_i)^2) / sum_j(...) ; P_i(label) = sum_{j: y_j = label} w_ij
//     prediction = argmax_label P_i (first label in sorted order on ties), uncertainty = max(1 - P_i(pred), 0)
// evaluated there in a Python double loop over queries x label columns.
//
// Top-k: one thread per query keeps its current k best (squared distance, index) as an UNSORTED list in LDS (slot-major
// [k][threads]: conflict-free) together with the list's maximum; reference points stream through LDS in tiles and are
// read as broadcasts, so the inner loop is D FMAs + one compare per pair and an insertion (replace the maximum, rescan
// k slots) happens only ~k ln(N/k) times per query.  The reference set is split over blockIdx.y so that small query
// sets still fill the chip; a second kernel merges the partial lists and sorts them ascending.
// Algorithmic work per query: 3 D Nr flop, 4 D Nr bytes of reference read from L2/LDS (HBM: once per workgroup row).
#include "common.h"

namespace {

constexpr int KNN_T = 128;       // queries (threads) per workgroup
constexpr int KNN_RT = 256;      // reference points per LDS tile
constexpr int KNN_KMAX = 64;

// replace the list maximum by (d2, id) and find the new maximum
__device__ __forceinline__ void knn_insert(float* ld2, int* lid, int k, float d2, int id, float& worst, int& wslot) {
    const int t = threadIdx.x;
    ld2[wslot * KNN_T + t] = d2;
    lid[wslot * KNN_T + t] = id;
    float w = -1.f;
    int ws = 0;
    for (int s0 = 0; s0 < k; s0 += 8) {          // eight independent LDS reads in flight, then the compares
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = s0 + u < k ? ld2[(s0 + u) * KNN_T + t] : -1.f;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (v[u] > w) { w = v[u]; ws = s0 + u; }
    }
    worst = w;
    wslot = ws;
}

template <int DMAX>
__global__ __launch_bounds__(KNN_T) void knn_scan_kernel(const float* __restrict__ ref, int Nr,
                                                         const float* __restrict__ qry, int Nq, int D, int k, int chunk,
                                                         float* __restrict__ pd2, int32_t* __restrict__ pidx) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ld2 = lds;                                   // [k][KNN_T]
    int* lid = reinterpret_cast<int*>(lds + k * KNN_T); // [k][KNN_T]
    float* tile = lds + 2 * k * KNN_T;                  // [KNN_RT][DMAX]
    const int t = threadIdx.x;
    const int qi = blockIdx.x * KNN_T + t;
    const bool act = qi < Nq;
    float q[DMAX];
#pragma unroll
    for (int d = 0; d < DMAX; ++d) q[d] = (act && d < D) ? qry[(long long)qi * D + d] : 0.f;
    for (int s = 0; s < k; ++s) {
        ld2[s * KNN_T + t] = INFINITY;
        lid[s * KNN_T + t] = -1;
    }
    float worst = INFINITY;
    int wslot = 0;
    const int r_lo = blockIdx.y * chunk, r_hi = min(Nr, r_lo + chunk);
    for (int r0 = r_lo; r0 < r_hi; r0 += KNN_RT) {
        const int nr = min(KNN_RT, r_hi - r0);
        __syncthreads();
        for (int i = t; i < nr * DMAX; i += KNN_T) {
            const int r = i / DMAX, d = i - r * DMAX;
            tile[i] = d < D ? ref[(long long)(r0 + r) * D + d] : 0.f;
        }
        __syncthreads();
        if (act) {
            for (int r = 0; r < nr; ++r) {
                const float4* tr = reinterpret_cast<const float4*>(tile + r * DMAX);
                float d2 = 0.f;
#pragma unroll
                for (int d4 = 0; d4 < DMAX / 4; ++d4) {
                    const float4 v = tr[d4];
                    const float a0 = q[4 * d4] - v.x, a1 = q[4 * d4 + 1] - v.y, a2 = q[4 * d4 + 2] - v.z,
                                a3 = q[4 * d4 + 3] - v.w;
                    d2 = fmaf(a0, a0, d2); d2 = fmaf(a1, a1, d2); d2 = fmaf(a2, a2, d2); d2 = fmaf(a3, a3, d2);
                }
                if (d2 < worst) knn_insert(ld2, lid, k, d2, r0 + r, worst, wslot);
            }
        }
    }
    if (act) {
        float* od = pd2 + ((long long)blockIdx.y * Nq + qi) * k;
        int32_t* oi = pidx + ((long long)blockIdx.y * Nq + qi) * k;
        for (int s = 0; s < k; ++s) {
            od[s] = ld2[s * KNN_T + t];
            oi[s] = lid[s * KNN_T + t];
        }
    }
}

// k best of the `chunks` partial lists of each query, sorted ascending by (distance, index); distances leave as sqrt.
__global__ __launch_bounds__(KNN_T) void knn_merge_kernel(const float* __restrict__ pd2, const int32_t* __restrict__ pidx,
                                                          int Nq, int k, int chunks, float* __restrict__ dist,
                                                          int32_t* __restrict__ idx) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ld2 = lds;
    int* lid = reinterpret_cast<int*>(lds + k * KNN_T);
    const int t = threadIdx.x;
    const int qi = blockIdx.x * KNN_T + t;
    if (qi >= Nq) return;
    for (int s = 0; s < k; ++s) {
        ld2[s * KNN_T + t] = INFINITY;
        lid[s * KNN_T + t] = -1;
    }
    float worst = INFINITY;
    int wslot = 0;
    for (int c = 0; c < chunks; ++c) {
        const float* cd = pd2 + ((long long)c * Nq + qi) * k;
        const int32_t* ci = pidx + ((long long)c * Nq + qi) * k;
        for (int s = 0; s < k; ++s) {
            const float d2 = cd[s];
            const int id = ci[s];
            if (id >= 0 && d2 < worst) knn_insert(ld2, lid, k, d2, id, worst, wslot);
        }
    }
    // selection sort by (d2, index): k <= 64
    for (int o = 0; o < k; ++o) {
        float bd = ld2[o * KNN_T + t];
        int bi = lid[o * KNN_T + t], bs = o;
        for (int s = o + 1; s < k; ++s) {
            const float v = ld2[s * KNN_T + t];
            const int id = lid[s * KNN_T + t];
            if (v < bd || (v == bd && (unsigned)id < (unsigned)bi)) { bd = v; bi = id; bs = s; }
        }
        if (bs != o) {
            ld2[bs * KNN_T + t] = ld2[o * KNN_T + t];
            lid[bs * KNN_T + t] = lid[o * KNN_T + t];
            ld2[o * KNN_T + t] = bd;
            lid[o * KNN_T + t] = bi;
        }
        dist[(long long)qi * k + o] = sqrtf(bd);
        idx[(long long)qi * k + o] = bi;
    }
}

// Gaussian-weighted vote of one label column: pred[q] = winning label code, prob[q] = its probability (fp64).
__global__ __launch_bounds__(KNN_T) void knn_vote_kernel(const float* __restrict__ dist, const int32_t* __restrict__ idx,
                                                         int Nq, int k, const int32_t* __restrict__ labels,
                                                         int32_t* __restrict__ pred, double* __restrict__ prob) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* lw = lds;                                    // [k][KNN_T] distance of the neighbour
    int* ll = reinterpret_cast<int*>(lds + k * KNN_T);  // [k][KNN_T] label code of the neighbour
    const int t = threadIdx.x;
    const int qi = blockIdx.x * KNN_T + t;
    if (qi >= Nq) return;
    double mean = 0.0;
    for (int s = 0; s < k; ++s) {
        const float d = dist[(long long)qi * k + s];
        const int id = idx[(long long)qi * k + s];
        lw[s * KNN_T + t] = d;
        ll[s * KNN_T + t] = id >= 0 ? labels[id] : -1;
        mean += (double)d;
    }
    mean /= k;
    double var = 0.0;
    for (int s = 0; s < k; ++s) {
        const double c = (double)lw[s * KNN_T + t] - mean;
        var += c * c;
    }
    var /= k;                                            // np.std: population variance
    // d / (2 / std)^2 = d * var / 4 ; std == 0 -> (2 / 0)^2 = inf -> every weight exp(-0) = 1
    double wsum = 0.0;
    for (int s = 0; s < k; ++s) {
        wsum += var > 0.0 ? exp(-(double)lw[s * KNN_T + t] * var * 0.25) : 1.0;   // weights are re-derived in the vote
    }
    double best = 0.0;
    int best_label = -1;
    for (int i = 0; i < k; ++i) {
        const int li = ll[i * KNN_T + t];
        if (li < 0) continue;
        bool seen = false;                               // evaluate each distinct label once
        for (int j = 0; j < i; ++j) seen |= ll[j * KNN_T + t] == li;
        if (seen) continue;
        double p = 0.0;
        for (int j = 0; j < k; ++j)
            if (ll[j * KNN_T + t] == li)
                p += var > 0.0 ? exp(-(double)lw[j * KNN_T + t] * var * 0.25) : 1.0;
        p /= wsum;
        // reference: candidates in sorted label order, strict '<' keeps the first maximum = the smallest label
        if (p > best || (p == best && best_label >= 0 && li < best_label)) { best = p; best_label = li; }
    }
    pred[qi] = best_label;
    prob[qi] = best;
}

}  // namespace

extern "C" int sca_knn_max_k(void) { return KNN_KMAX; }

// Number of reference chunks (blockIdx.y) sca_knn_topk uses; the caller sizes the partial workspace
//   pd2 / pidx : chunks * Nq * k floats / int32 each.
extern "C" int sca_knn_num_chunks(int Nr, int Nq) {
    const long long qblocks = (Nq + KNN_T - 1) / KNN_T;
    long long chunks = (1024 + qblocks - 1) / qblocks;
    const long long maxc = (Nr + 4 * KNN_RT - 1) / (4 * KNN_RT);      // at least four tiles per chunk
    if (chunks > maxc) chunks = maxc;
    if (chunks < 1) chunks = 1;
    return (int)chunks;
}

extern "C" int sca_knn_topk(const float* ref, int Nr, const float* qry, int Nq, int D, int k, float* pd2,
                            int32_t* pidx, float* dist, int32_t* idx, void* stream) {
    if (Nq <= 0) return 0;
    if (Nr <= 0 || D <= 0 || k <= 0) return -1;
    if (k > KNN_KMAX || k > Nr || D > 64) return -2;
    hipStream_t st = (hipStream_t)stream;
    const int chunks = sca_knn_num_chunks(Nr, Nq);
    int chunk = (Nr + chunks - 1) / chunks;
    chunk = ((chunk + KNN_RT - 1) / KNN_RT) * KNN_RT;
    const dim3 grid((Nq + KNN_T - 1) / KNN_T, (Nr + chunk - 1) / chunk);
    const int dmax = D <= 12 ? 12 : (D <= 16 ? 16 : (D <= 32 ? 32 : 64));
    const size_t shm = sizeof(float) * ((size_t)2 * k * KNN_T + (size_t)KNN_RT * dmax);
#define KNN_SCAN(DM)                                                                                                \
    do {                                                                                                            \
        hipFuncSetAttribute((const void*)knn_scan_kernel<DM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
        hipLaunchKernelGGL((knn_scan_kernel<DM>), grid, dim3(KNN_T), shm, st, ref, Nr, qry, Nq, D, k, chunk, pd2,   \
                           pidx);                                                                                   \
    } while (0)
    if (dmax == 12) KNN_SCAN(12);
    else if (dmax == 16) KNN_SCAN(16);
    else if (dmax == 32) KNN_SCAN(32);
    else KNN_SCAN(64);
#undef KNN_SCAN
    const size_t shm2 = sizeof(float) * (size_t)2 * k * KNN_T;
    hipFuncSetAttribute((const void*)knn_merge_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm2);
    hipLaunchKernelGGL(knn_merge_kernel, dim3(grid.x), dim3(KNN_T), shm2, st, pd2, pidx, Nq, k, (int)grid.y, dist, idx);
    SCA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sca_knn_vote(const float* dist, const int32_t* idx, int Nq, int k, const int32_t* labels, int32_t* pred,
                            double* prob, void* stream) {
    if (Nq <= 0) return 0;
    if (k <= 0 || k > KNN_KMAX) return -2;
    const size_t shm = sizeof(float) * (size_t)2 * k * KNN_T;
    hipFuncSetAttribute((const void*)knn_vote_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipLaunchKernelGGL(knn_vote_kernel, dim3((Nq + KNN_T - 1) / KNN_T), dim3(KNN_T), shm, (hipStream_t)stream, dist, idx,
                       Nq, k, labels, pred, prob);
    SCA_CHECK_LAUNCH();
    return 0;
}
