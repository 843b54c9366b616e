// Weighted k-nearest-neighbour label transfer in the latent space (gfx950): brute-force euclidean top-k + the
// Gaussian-weighted vote.  Reference: scarches/utils/knn.py:10-139 - sklearn KNeighborsTransformer(algorithm="brute",
// metric="euclidean").kneighbors for the neighbours, then
//     std_i = std(d_i1..d_ik) ; w_ij = exp(-d_ij / (2 / std_i)^2) / sum_j(...) ; P_i(label) = sum_{j: y_j = label} w_ij
//     prediction = argmax_label P_i (first label in sorted order on ties), uncertainty = max(1 - P_i(pred), 0)
// evaluated there in a Python double loop over queries x label columns.
//
// Top-k: see knn_scan_kernel (a wave per group of queries, a lane per reference point, the k best of a query one per
// lane in registers).  The reference set is split over blockIdx.y so that small query sets still fill the chip; a
// second kernel merges the partial lists and sorts them ascending.
// Algorithmic work per query: 3 D Nr flop; the reference is read from HBM/L2 once per 32 queries.
#include "common.h"
#include <mutex>

namespace {

constexpr int KNN_T = 128;       // queries (threads) per workgroup of the merge / vote kernels
constexpr int KNN_RT = 256;      // reference points per LDS tile
constexpr int KNN_KMAX = 64;     // one list slot per lane
constexpr int KNN_QWG = 32;      // queries per workgroup of the scan kernel at D <= 12 (4 waves x 8)

// replace the list maximum by (d2, id) and find the new maximum (merge kernel: one thread per query, list in LDS)
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

__device__ __forceinline__ float knn_uniform(float v) {   // a wave-uniform value, held in a scalar register
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

// Scan: a WAVE owns QW queries (their coordinates are wave-uniform registers) and every lane one reference point of
// the current 64-point sub-tile, so one LDS read of the point serves QW distance evaluations and nothing diverges.
// The k best of a query live one per lane (k <= 64: lane l < k holds list slot l, in registers): a candidate below
// the list maximum is found with one ballot per 64 points, taken lane by lane through a scalar loop, and inserted by
// the lane that holds the maximum (ballot + one wave max) - O(k ln(N/k)) insertions per query, each ~30 instructions.
template <int DMAX, int QW>
__global__ __launch_bounds__(256) void knn_scan_kernel(const float* __restrict__ ref, int Nr,
                                                       const float* __restrict__ qry, int Nq, int D, int k, int chunk,
                                                       float* __restrict__ pd2, int32_t* __restrict__ pidx) {
    constexpr int LDR = DMAX + 1;                       // odd pitch: lane l reads point l without bank conflicts
    __shared__ float tile[KNN_RT * LDR];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int q0 = (blockIdx.x * 4 + w) * QW;
    float q[QW][DMAX];
#pragma unroll
    for (int qi = 0; qi < QW; ++qi)
#pragma unroll
        for (int d = 0; d < DMAX; ++d)
        {   // (kept in vector registers on purpose: 8 x 12 coordinates do not fit the scalar register file)
            q[qi][d] = (q0 + qi < Nq && d < D) ? qry[(long long)(q0 + qi) * D + d] : 0.f;
            asm volatile("" : "+v"(q[qi][d]));
        }
    float ld2[QW], worst[QW];
    int lid[QW];
#pragma unroll
    for (int qi = 0; qi < QW; ++qi) {
        ld2[qi] = lane < k ? INFINITY : -1.f;           // lanes past k never hold the maximum
        lid[qi] = -1;
        worst[qi] = INFINITY;
    }
    const int r_lo = blockIdx.y * chunk, r_hi = min(Nr, r_lo + chunk);
    for (int r0 = r_lo; r0 < r_hi; r0 += KNN_RT) {
        const int nr = min(KNN_RT, r_hi - r0);
        __syncthreads();
        const float* src = ref + (long long)r0 * D;
        for (int e = t; e < nr * D; e += 256) {
            const int rr = e / D, d = e - rr * D;
            tile[rr * LDR + d] = src[e];
        }
        __syncthreads();
        for (int sub = 0; sub < nr; sub += 64) {
            const int rl = sub + lane;
            const bool valid = rl < nr;
            const float* tr = tile + min(rl, nr - 1) * LDR;
            float r[DMAX];
#pragma unroll
            for (int d = 0; d < DMAX; ++d) r[d] = d < D ? tr[d] : 0.f;
#pragma unroll
            for (int qi = 0; qi < QW; ++qi) {
                float d2 = 0.f;
#pragma unroll
                for (int d = 0; d < DMAX; ++d) {
                    const float a = r[d] - q[qi][d];
                    d2 = fmaf(a, a, d2);
                }
                unsigned long long m = __ballot(valid && d2 < worst[qi]);
                while (m) {
                    const int l = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const float cd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d2), l));
                    if (cd < worst[qi]) {
                        const int wl = __ffsll((long long)__ballot(ld2[qi] == worst[qi])) - 1;
                        if (lane == wl) { ld2[qi] = cd; lid[qi] = r0 + sub + l; }
                        worst[qi] = knn_uniform(wave_max(ld2[qi]));
                    }
                }
            }
        }
    }
#pragma unroll
    for (int qi = 0; qi < QW; ++qi)
        if (q0 + qi < Nq && lane < k) {
            const long long o = ((long long)blockIdx.y * Nq + q0 + qi) * k + lane;
            pd2[o] = ld2[qi];
            pidx[o] = lid[qi];
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
    const long long qblocks = (Nq + KNN_QWG - 1) / KNN_QWG;
    long long chunks = (1024 + qblocks - 1) / qblocks;
    const long long maxc = (Nr + 8191) / 8192;           // a chunk scans at least 8192 reference points
    if (chunks > maxc) chunks = maxc;
    if (chunks < 1) chunks = 1;
    return (int)chunks;
}

// Once per device, outside any stream capture: the merge / vote kernels' dynamic-LDS limit for the largest k (a
// hipFuncSetAttribute is not a stream operation and must not run while a stream of the thread captures).
namespace {
std::mutex g_knn_init_mu;
bool g_knn_init_done[64] = {};
int knn_init_current_device() {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= 64) return -4;
    std::lock_guard<std::mutex> lock(g_knn_init_mu);
    if (!g_knn_init_done[dev]) {
        const int shm = (int)(sizeof(float) * (size_t)2 * KNN_KMAX * KNN_T);
        e = hipFuncSetAttribute((const void*)knn_merge_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, shm);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute((const void*)knn_vote_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, shm);
        if (e != hipSuccess) return (int)e;
        g_knn_init_done[dev] = true;
    }
    return 0;
}
}  // namespace

// Shared launcher of the two top-k entry points: scan over reference chunks into the partial lists, then merge.
static int knn_topk_launch(const float* ref, int Nr, const float* qry, int Nq, int D, int k, float* pd2,
                           int32_t* pidx, float* dist, int32_t* idx, hipStream_t st) {
    const int chunks = sca_knn_num_chunks(Nr, Nq);
    int chunk = (Nr + chunks - 1) / chunks;
    chunk = ((chunk + KNN_RT - 1) / KNN_RT) * KNN_RT;
    const int ny = (Nr + chunk - 1) / chunk;
#define KNN_SCAN(DM, QW)                                                                                            \
    hipLaunchKernelGGL((knn_scan_kernel<DM, QW>), dim3((Nq + 4 * QW - 1) / (4 * QW), ny), dim3(256), 0, st, ref, Nr, \
                       qry, Nq, D, k, chunk, pd2, pidx)
    if (D <= 12) KNN_SCAN(12, 8);
    else if (D <= 16) KNN_SCAN(16, 6);
    else if (D <= 32) KNN_SCAN(32, 3);
    else KNN_SCAN(64, 1);
#undef KNN_SCAN
    const size_t shm2 = sizeof(float) * (size_t)2 * k * KNN_T;
    hipLaunchKernelGGL(knn_merge_kernel, dim3((Nq + KNN_T - 1) / KNN_T), dim3(KNN_T), shm2, st, pd2, pidx, Nq, k, ny,
                       dist, idx);
    SCA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sca_knn_topk(const float* ref, int Nr, const float* qry, int Nq, int D, int k, float* pd2,
                            int32_t* pidx, float* dist, int32_t* idx, void* stream) {
    if (Nq <= 0) return 0;
    if (Nr <= 0 || D <= 0 || k <= 0) return -1;
    if (k > KNN_KMAX || k > Nr || D > 64) return -2;
    if (const int rc = knn_init_current_device()) return rc;
    return knn_topk_launch(ref, Nr, qry, Nq, D, k, pd2, pidx, dist, idx, (hipStream_t)stream);
}

// ---- sca_knn_topk without a caller workspace and without any allocation ------------------------------------------
// For callers that cannot allocate (code under HIP graph capture, serving loops, harnesses that forbid hipMalloc): the
// partial top-k lists live in a file-scope device array of the library, SCA_KNN_STATIC_ENTRIES (distance, index) pairs
// = 16 MiB per GPU the library runs on, held until the process ends.  The array's name is only a host-side handle:
// its device address is looked up once per device with hipGetSymbolAddress (each GPU has its own copy at its own
// address) and only that pointer is handed to the kernels.  Nothing has to be cleared between calls: the scan kernel
// writes every partial entry the merge kernel reads.  Calls that overlap on one GPU (different streams) share the
// array and race; issue them on one stream.
constexpr long long SCA_KNN_STATIC_ENTRIES = 2ll << 20;
static __device__ __align__(256) unsigned char scratch[SCA_KNN_STATIC_ENTRIES * 8];

namespace {
constexpr int KNN_MAX_DEVICES = 64;
std::mutex g_scratch_mu;
unsigned char* g_scratch_ptr[KNN_MAX_DEVICES] = {};      // per device: address of `scratch` on that GPU

int knn_scratch_for_current_device(unsigned char** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= KNN_MAX_DEVICES) return -4;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (g_scratch_ptr[dev] == nullptr) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));     // with device `dev` current: this GPU's copy
        if (e != hipSuccess) return (int)e;
        if (p == nullptr) return -4;
        g_scratch_ptr[dev] = static_cast<unsigned char*>(p);
    }
    *out = g_scratch_ptr[dev];
    return 0;
}
}  // namespace

extern "C" long long sca_knn_static_max_entries(void) { return SCA_KNN_STATIC_ENTRIES; }

extern "C" int sca_knn_topk_static(const float* ref, int Nr, const float* qry, int Nq, int D, int k, float* dist,
                                   int32_t* idx, void* stream) {
    if (Nq <= 0) {
        // WARM-UP call (Nq = 0, other arguments ignored): resolves this device's address of the static array and sets
        // the kernels' attributes - the two things below that are not stream operations.  A caller that will use the
        // entry point under HIP graph capture makes this call once per device BEFORE capturing.
        unsigned char* base = nullptr;
        if (const int rc = knn_init_current_device()) return rc;
        return knn_scratch_for_current_device(&base);
    }
    if (Nr <= 0 || D <= 0 || k <= 0) return -1;
    if (k > KNN_KMAX || k > Nr || D > 64) return -2;
    if (const int rc0 = knn_init_current_device()) return rc0;
    const long long entries = (long long)sca_knn_num_chunks(Nr, Nq) * Nq * k;
    if (entries > SCA_KNN_STATIC_ENTRIES) return -3;          // beyond the static workspace: use sca_knn_topk
    unsigned char* base = nullptr;
    const int rc = knn_scratch_for_current_device(&base);
    if (rc) return rc;
    float* pd2 = reinterpret_cast<float*>(base);
    int32_t* pidx = reinterpret_cast<int32_t*>(base + SCA_KNN_STATIC_ENTRIES * 4);
    return knn_topk_launch(ref, Nr, qry, Nq, D, k, pd2, pidx, dist, idx, (hipStream_t)stream);
}

extern "C" int sca_knn_vote(const float* dist, const int32_t* idx, int Nq, int k, const int32_t* labels, int32_t* pred,
                            double* prob, void* stream) {
    if (Nq <= 0) return 0;
    if (k <= 0 || k > KNN_KMAX) return -2;
    if (const int rc = knn_init_current_device()) return rc;
    const size_t shm = sizeof(float) * (size_t)2 * k * KNN_T;
    hipLaunchKernelGGL(knn_vote_kernel, dim3((Nq + KNN_T - 1) / KNN_T), dim3(KNN_T), shm, (hipStream_t)stream, dist, idx,
                       Nq, k, labels, pred, prob);
    SCA_CHECK_LAUNCH();
    return 0;
}
