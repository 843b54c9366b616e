// Gradient tail: every batch reduction that only feeds Adam, in ONE launch (gfx950).
//
// After the backward dX chain the step still owes (a) the sums of the split-K slabs of each weight-gradient GEMM and
// (b) the column reductions autograd derives from bias broadcasts and one-hot condition GEMMs
//   db[h] = sum_b da[b][h],   dWc[h][c] = sum_{b: cond[b] = c} da[b][h]
// (scarches/models/trvae/modules.py:21-27 CondLayers, nn.Linear biases).  As separate kernels these were ~20 tiny
// launches per step, each paying the ~5 us launch floor; here they are jobs of one kernel.  Column-sum jobs split
// the batch into 256-row chunks; with more than one chunk the partial sums are added with float atomics, so the
// caller must have zeroed db / dWc (the engine zeroes the whole flat gradient buffer at the start of a step).  The job table travels by
// value in the kernel arguments (no device-side table, nothing to upload).
#include "common.h"

extern "C" {
typedef struct {
    int32_t kind;          // 0: dst[i] = sum_s src[s*stride + i], i < n      1: column sums of da [B][H]
    int32_t S;             // kind 0: number of slabs
    long long stride;      // kind 0: slab stride (elements)
    const float* src;      // kind 0: slabs                                   kind 1: da
    float* dst;            // kind 0: destination                             kind 1: db [H] or NULL
    long long n;           // kind 0: elements
    float* dWc;            // kind 1: [H][C] or NULL
    const int32_t* cond;   // kind 1: [B] (with dWc)
    int32_t B, H, C, pad;
} sca_tail_job;
}

namespace {

constexpr int TAIL_MAX_JOBS = 24;
constexpr int TAIL_COLS = 16;          // columns per workgroup of a column-sum job (16 x 16 thread layout)
constexpr int TAIL_MAX_C = 512;        // conditions whose segment table fits LDS: 512 * 16 * 4 B = 32 KB
constexpr int TAIL_ROWS = 256;         // batch rows per workgroup of a column-sum job (more chunks -> float atomics)

struct TailArgs {
    sca_tail_job job[TAIL_MAX_JOBS];
    int32_t wg_begin[TAIL_MAX_JOBS + 1];
    int32_t n_jobs;
};

__global__ __launch_bounds__(256) void grad_tail_kernel(const TailArgs a) {
    extern __shared__ float seg[];      // [max(C,1)][TAIL_COLS] of the column-sum job this workgroup serves
    int j = 0;
    while (j + 1 < a.n_jobs && (int)blockIdx.x >= a.wg_begin[j + 1]) ++j;
    const sca_tail_job& J = a.job[j];
    const int wg = blockIdx.x - a.wg_begin[j];
    if (J.kind == 0) {
        const long long i0 = ((long long)wg * 256 + threadIdx.x) * 4;
        if (i0 >= J.n) return;
        if (i0 + 3 < J.n && (J.stride & 3) == 0 && (((uintptr_t)J.src | (uintptr_t)J.dst) & 15) == 0) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            const float* p = J.src + i0;
            int s = 0;
            for (; s + 8 <= J.S; s += 8) {          // 8 independent 16-byte loads in flight per thread
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(p + (s + u) * J.stride);
#pragma unroll
                for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
            }
            for (; s < J.S; ++s) {
                const float4 v = *reinterpret_cast<const float4*>(p + s * J.stride);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            *reinterpret_cast<float4*>(J.dst + i0) = acc;
        } else {
            for (long long i = i0; i < min(i0 + 4, J.n); ++i) {
                float acc = 0.f;
                for (int s = 0; s < J.S; ++s) acc += J.src[s * J.stride + i];
                J.dst[i] = acc;
            }
        }
        return;
    }
    // ---- column sums: this workgroup owns TAIL_COLS columns, 16 row groups stride over the batch
    const int cx = threadIdx.x & (TAIL_COLS - 1), ry = threadIdx.x / TAIL_COLS;
    const int R = J.pad;                 // row chunks (set by the host wrapper)
    const int stripe = wg / R, chunk = wg - stripe * R;
    const int h = stripe * TAIL_COLS + cx;
    const int r_lo = chunk * TAIL_ROWS, r_hi = min(J.B, r_lo + TAIL_ROWS);
    const int Cs = J.dWc ? J.C : 1;
    for (int i = threadIdx.x; i < Cs * TAIL_COLS; i += 256) seg[i] = 0.f;
    __syncthreads();
    if (h < J.H) {
        // condition-sorted minibatches: a register accumulates the current run, one LDS atomic per run change
        constexpr int RG = 256 / TAIL_COLS, U = 4;
        int cur = -1;
        float acc = 0.f;
        for (int r = r_lo + ry; r < r_hi; r += RG * U) {
            float v[U];
            int c[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {           // all loads of the group in flight together
                const int rr = r + u * RG;
                const bool ok = rr < r_hi;
                v[u] = ok ? J.src[(long long)rr * J.H + h] : 0.f;
                c[u] = ok ? (J.dWc ? J.cond[rr] : 0) : cur;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (c[u] != cur) {
                    if (cur >= 0) atomicAdd(&seg[cur * TAIL_COLS + cx], acc);
                    acc = 0.f;
                    cur = c[u];
                }
                acc += v[u];
            }
        }
        if (cur >= 0) atomicAdd(&seg[cur * TAIL_COLS + cx], acc);
    }
    __syncthreads();
    if (h < J.H && ry == 0) {
        float tot = 0.f;
        for (int c = 0; c < Cs; ++c) {
            const float v = seg[c * TAIL_COLS + cx];
            tot += v;
            if (J.dWc) {
                if (R == 1) J.dWc[(long long)h * J.C + c] = v;
                else if (v != 0.f) atomicAdd(&J.dWc[(long long)h * J.C + c], v);   // targets pre-zeroed by the caller
            }
        }
        if (J.dst) {
            if (R == 1) J.dst[h] = tot;
            else atomicAdd(&J.dst[h], tot);
        }
    }
}

}  // namespace

extern "C" int sca_grad_tail_max_jobs(void) { return TAIL_MAX_JOBS; }

extern "C" int sca_grad_tail(const sca_tail_job* jobs, int n_jobs, void* stream) {
    if (n_jobs <= 0) return 0;
    if (n_jobs > TAIL_MAX_JOBS) return -2;
    TailArgs a;
    int wg = 0, maxC = 1;
    for (int j = 0; j < n_jobs; ++j) {
        const sca_tail_job& J = jobs[j];
        a.job[j] = J;
        a.wg_begin[j] = wg;
        if (J.kind == 0) {
            if (J.n <= 0 || J.S < 1) return -1;
            wg += (int)((J.n + 1023) / 1024);
        } else if (J.kind == 1) {
            if (J.B <= 0 || J.H <= 0) return -1;
            if (J.dWc) {
                if (J.C < 1 || J.C > TAIL_MAX_C || !J.cond) return -2;
                if (J.C > maxC) maxC = J.C;
            }
            a.job[j].pad = (J.B + TAIL_ROWS - 1) / TAIL_ROWS;
            wg += ((J.H + TAIL_COLS - 1) / TAIL_COLS) * a.job[j].pad;
        } else {
            return -1;
        }
    }
    a.wg_begin[n_jobs] = wg;
    a.n_jobs = n_jobs;
    const size_t shm = (size_t)maxC * TAIL_COLS * sizeof(float);
    hipLaunchKernelGGL(grad_tail_kernel, dim3(wg), dim3(256), shm, (hipStream_t)stream, a);
    SCA_CHECK_LAUNCH();
    return 0;
}
