// Dev tool: issue rate of the two fp32 MFMA shapes on gfx950, one wave per SIMD, independent accumulators.
// build: hipcc --offload-arch=gfx950 -O3 scripts/mfma_rate.hip -o scripts/mfma_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ void k16(float* out, long long* cyc, int iters) {
    f32x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][3];
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int CHAINS>
__global__ void k32(float* out, long long* cyc, int iters) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][15];
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <typename K>
static void run(const char* name, K kern, int chains, int flops_per) {
    float* out; long long* cyc; long long h = 0;
    hipMalloc(&out, 256 * 1024 * sizeof(float)); hipMalloc(&cyc, 8);
    const int iters = 4096;
    for (int waves = 1; waves <= 2; ++waves) {       // waves per SIMD (workgroup of 256 * waves threads on one CU)
        hipLaunchKernelGGL(kern, dim3(1), dim3(256 * waves), 0, 0, out, cyc, iters);
        hipLaunchKernelGGL(kern, dim3(1), dim3(256 * waves), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        const double per = (double)h / ((double)iters * chains);
        printf("%s chains=%d waves/SIMD=%d: %.1f shader cycles per MFMA per wave (clock64 ticks; %d flop each)\n", name,
               chains, waves, per, flops_per);
    }
    hipFree(out); hipFree(cyc);
}
int main() {
    run("v_mfma_f32_16x16x4_f32", k16<1>, 1, 2048);
    run("v_mfma_f32_16x16x4_f32", k16<2>, 2, 2048);
    run("v_mfma_f32_16x16x4_f32", k16<4>, 4, 2048);
    run("v_mfma_f32_32x32x2_f32", k32<1>, 1, 4096);
    run("v_mfma_f32_32x32x2_f32", k32<2>, 2, 4096);
    return 0;
}
