// In-kernel CSR-row gather: builds the dense fp32 minibatch in HBM straight from the resident CSR matrix.
// Replaces the reference's per-cell `self.data[index].toarray()` + torch.stack on the host
// (scarches/dataset/trvae/anndata.py:74-81, trainers/trvae/_utils.py:208-240) and fuses
//   x_log = log(1 + x)                      (models/trvae/trvae.py:117)
//   sizefactor = sum_g x                    (dataset/trvae/anndata.py:44)
// One workgroup per cell: the row is assembled in LDS (zero fill + scatter of the nnz) and written out with
// coalesced 16-byte stores, so every output byte is written exactly once.  A cell costs three dependent HBM round
// trips (row id -> indptr -> indices/data); 256-thread workgroups (8 cells in flight per CU) measured faster than
// single-wave ones (28 vs 36 us at B = 4096, G = 2000: the dense log1p + store phase wants the lanes).
// Algorithmic bytes per cell: 8*nnz (indices+data) + 16 (indptr) + 8 (row id) read, 4*G + 4 written.
#include "common.h"

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
        for (int g = threadIdx.x; g < G; g += T) row[g] = 0.f;
        __syncthreads();
        for (int64_t k = s + threadIdx.x; k < e; k += T) {
            const float v = data[k];
            sum += v;
            // canonical CSR (unique column per row) is a host-side precondition; atomics keep duplicates exact anyway
            atomicAdd(&row[indices[k]], v);
        }
        __syncthreads();
        const bool vec = ((ldx & 3) == 0) && ((((uintptr_t)xout) & 15) == 0);
        if (vec) {
            const int g4 = G >> 2;
            for (int i = threadIdx.x; i < g4; i += T) {
                float4 v = *reinterpret_cast<const float4*>(&row[4 * i]);
                if (apply_log1p) {
                    v.x = log1pf(v.x); v.y = log1pf(v.y); v.z = log1pf(v.z); v.w = log1pf(v.w);
                }
                *reinterpret_cast<float4*>(out + 4 * i) = v;
            }
            for (int g = 4 * g4 + threadIdx.x; g < G; g += T) out[g] = apply_log1p ? log1pf(row[g]) : row[g];
        } else {
            for (int g = threadIdx.x; g < G; g += T) out[g] = apply_log1p ? log1pf(row[g]) : row[g];
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
                // duplicates (non-canonical CSR) would be transformed twice: the host asserts canonical form
                const int g = indices[k];
                out[g] = log1pf(out[g]);
            }
        }
    }
    sum = block_sum(sum, red);
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
    if (G <= 16384) {
        hipLaunchKernelGGL((csr_gather_kernel<true, 256>), dim3(B), dim3(256), (size_t)G * sizeof(float), st, indptr,
                           indices, data, row_ids, B, G, apply_log1p, xout, ldx, sizefactor);
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
