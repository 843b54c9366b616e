// In-kernel CSR-row gather: builds the dense fp32 minibatch in HBM straight from the resident CSR matrix.
// Replaces the reference's per-cell `self.data[index].toarray()` + torch.stack on the host
// (scarches/dataset/trvae/anndata.py:74-81, trainers/trvae/_utils.py:208-240) and fuses
//   x_log = log(1 + x)                      (models/trvae/trvae.py:117)
//   sizefactor = sum_g x                    (dataset/trvae/anndata.py:44)
// One workgroup per cell: the row is assembled in LDS (zero fill + scatter of the stored counts, which are
// log1p-transformed on the way: log1p(0) = 0, so only the nnz pay for the transform) and copied out with coalesced
// 16-byte stores, so every output byte is written exactly once.  The first stored counts of a row are requested before
// the LDS row is cleared.  128 threads per cell at G = 2000 (16 cells in flight per CU): 18 us at B = 8192 = 4.6 TB/s of
// algorithmic bytes (64 threads: 22 us, 256: 19 us; the first version - dense log1p sweep, 256 threads - took 28 us at
// B = 4096).
// Algorithmic bytes per cell: 8*nnz (indices+data) + 16 (indptr) + 8 (row id) read, 4*G + 4 written.
#include "common.h"
#include <stdlib.h>

namespace {

template <bool LDS_ROW, int T>
__global__ __launch_bounds__(T) void csr_gather_kernel(const int64_t* __restrict__ indptr,
                                                         const int32_t* __restrict__ indices,
                                                         const float* __restrict__ data,
                                                         const int64_t* __restrict__ row_ids, int B, int G,
                                                         int apply_log1p, float* __restrict__ xout, int ldx,
                                                         float* __restrict__ sizefactor) {
    extern __shared__ __attribute__((aligned(16))) float row[];
    __shared__ float red[16];
    const int b = blockIdx.x;
    if (b >= B) return;
    const int64_t r = row_ids ? row_ids[b] : (int64_t)b;
    const int64_t s = indptr[r], e = indptr[r + 1];
    float* out = xout + (long long)b * ldx;
    float sum = 0.f;
    if (LDS_ROW) {
        // the first stored counts of the row travel while the LDS row is cleared
        const int64_t k0 = s + threadIdx.x;
        float v0 = 0.f;
        int g0 = -1;
        if (k0 < e) { v0 = data[k0]; g0 = indices[k0]; }
        const int g4 = G >> 2;
        for (int i = threadIdx.x; i < g4; i += T) *reinterpret_cast<float4*>(&row[4 * i]) = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int g = 4 * g4 + threadIdx.x; g < G; g += T) row[g] = 0.f;
        __syncthreads();
        // log1p(0) = 0: only the stored counts are transformed (canonical CSR - one entry per column and row - is a
        // host-side precondition of the whole library, see include/scarches_hip.h)
        if (g0 >= 0) {
            sum = v0;
            row[g0] = apply_log1p ? log1pf(v0) : v0;
        }
        for (int64_t k = k0 + T; k < e; k += T) {
            const float v = data[k];
            sum += v;
            row[indices[k]] = apply_log1p ? log1pf(v) : v;
        }
        __syncthreads();
        const bool vec = ((ldx & 3) == 0) && ((((uintptr_t)xout) & 15) == 0);
        if (vec) {
            for (int i = threadIdx.x; i < g4; i += T)
                *reinterpret_cast<float4*>(out + 4 * i) = *reinterpret_cast<const float4*>(&row[4 * i]);
            for (int g = 4 * g4 + threadIdx.x; g < G; g += T) out[g] = row[g];
        } else {
            for (int g = threadIdx.x; g < G; g += T) out[g] = row[g];
        }
    } else {
        for (int g = threadIdx.x; g < G; g += T) out[g] = 0.f;
        __syncthreads();
        for (int64_t k = s + threadIdx.x; k < e; k += T) {
            const float v = data[k];
            sum += v;
            atomicAdd(&out[indices[k]], v);
        }
        if (apply_log1p) {
            __syncthreads();
            for (int64_t k = s + threadIdx.x; k < e; k += T) {
                const int g = indices[k];
                out[g] = log1pf(out[g]);
            }
        }
    }
    if (T == 64) {
        sum = wave_sum(sum);
    } else {
        sum = block_sum(sum, red);
    }
    if (threadIdx.x == 0 && sizefactor) sizefactor[b] = sum;
}

// Dense input path (adata.X given as a dense matrix): row gather + log1p + row sum.
__global__ __launch_bounds__(256) void dense_gather_kernel(const float* __restrict__ X, int ldX,
                                                           const int64_t* __restrict__ row_ids, int B, int G,
                                                           int apply_log1p, float* __restrict__ xout, int ldx,
                                                           float* __restrict__ sizefactor) {
    __shared__ float red[16];
    const int b = blockIdx.x;
    if (b >= B) return;
    const int64_t r = row_ids ? row_ids[b] : (int64_t)b;
    const float* in = X + r * (long long)ldX;
    float* out = xout + (long long)b * ldx;
    float sum = 0.f;
    for (int g = threadIdx.x; g < G; g += 256) {
        const float v = in[g];
        sum += v;
        out[g] = apply_log1p ? log1pf(v) : v;
    }
    sum = block_sum(sum, red);
    if (threadIdx.x == 0 && sizefactor) sizefactor[b] = sum;
}

}  // namespace

extern "C" int sca_csr_gather_rows(const int64_t* indptr, const int32_t* indices, const float* data,
                                   const int64_t* row_ids, int B, int G, int apply_log1p, float* xout, int ldx,
                                   float* sizefactor, void* stream) {
    if (B <= 0) return 0;
    if (G <= 0 || ldx < G) return -1;
    hipStream_t st = (hipStream_t)stream;
    if ((size_t)G * sizeof(float) + 64 <= 65536) {      // the LDS row + the static reduction scratch fit 64 KB
        // threads per cell (tuning knob, SCA_GATHER_T = 64 | 128 | 256): fewer threads = more cells in flight per CU
        static const int tsel = [] { const char* v = getenv("SCA_GATHER_T"); return v ? atoi(v) : 0; }();
        const int T = tsel ? tsel : (G <= 1024 ? 64 : G <= 4096 ? 128 : 256);
        const size_t shm = (size_t)G * sizeof(float);
        if (T == 64)
            hipLaunchKernelGGL((csr_gather_kernel<true, 64>), dim3(B), dim3(64), shm, st, indptr, indices, data,
                               row_ids, B, G, apply_log1p, xout, ldx, sizefactor);
        else if (T == 128)
            hipLaunchKernelGGL((csr_gather_kernel<true, 128>), dim3(B), dim3(128), shm, st, indptr, indices, data,
                               row_ids, B, G, apply_log1p, xout, ldx, sizefactor);
        else
            hipLaunchKernelGGL((csr_gather_kernel<true, 256>), dim3(B), dim3(256), shm, st, indptr, indices, data,
                               row_ids, B, G, apply_log1p, xout, ldx, sizefactor);
    } else {
        hipLaunchKernelGGL((csr_gather_kernel<false, 256>), dim3(B), dim3(256), 0, st, indptr, indices, data, row_ids,
                           B, G, apply_log1p, xout, ldx, sizefactor);
    }
    SCA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sca_dense_gather_rows(const float* X, int ldX, const int64_t* row_ids, int B, int G, int apply_log1p,
                                     float* xout, int ldx, float* sizefactor, void* stream) {
    if (B <= 0) return 0;
    if (G <= 0 || ldx < G || ldX < G) return -1;
    hipLaunchKernelGGL(dense_gather_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, X, ldX, row_ids, B, G,
                       apply_log1p, xout, ldx, sizefactor);
    SCA_CHECK_LAUNCH();
    return 0;
}
