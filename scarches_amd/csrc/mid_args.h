// Argument block of the fused middle kernels (sca_mid_args of include/scarches_hip.h), shared by the two kernel
// families: fused_mid.hip (weights staged through LDS) and fused_mid_direct.hip (weight fragments straight from L2).
#pragma once
#include <stdint.h>

extern "C" {
typedef struct {
    const float* W;        // [N][K]
    const float* bias;     // [N] or NULL
    const float* Wc;       // [N][C] one-hot condition weights or NULL
    const uint8_t* mask;   // optional injected dropout keep-mask [B][N]
    float* yn;             // [B][N] normalised pre-activation (saved by forward, read by backward)
    float* rs;             // [B]
    float* h;              // [B][N] layer output
    float* da;             // [B][N] backward: gradient w.r.t. the Linear output
    int32_t N, K;
    uint32_t rng_stream;
    int32_t pad;
} sca_mid_layer;

typedef struct {
    sca_mid_layer enc[4];  // encoder layers AFTER the first one
    sca_mid_layer dec[4];  // decoder hidden layers (first one carries Wc)
    int32_t n_enc, n_dec;
    const float* Wh;       // fused heads [2Z][Kh] = [mean | log_var]
    const float* bh;       // [2Z]
    const float* eps_in;   // optional injected noise [B][Z]
    float* mu_lv;          // [B][2Z]
    float* eps_out;        // [B][Z]
    float* z;              // [B][Z]
    float* kl_row;         // [B]
    int32_t Z, Kh;
    uint32_t rng_latent;
    int32_t sample;
    int32_t B, C;
    const int32_t* cond;   // [B]
    int32_t norm;          // 0 none, 1 LayerNorm
    float p;               // dropout rate
    uint64_t seed;
    const uint32_t* step;
    const float* h_in;     // [B][K0] output of the first encoder layer
    int32_t K0, S;
    // backward only
    const float* dh;       // [S][B][N_last] slabs of d loss / d (last decoder layer output)
    long long dh_stride;
    const float* dz2;      // optional extra gradient w.r.t. z [B][Z] (MMD)
    const float* hyper;    // hyper[0] = alpha_kl
    float* dheads;         // [B][2Z]
    float* dh_out;         // [B][K0] gradient w.r.t. h_in
    // optional: the first encoder layer's epilogue fused in (its GEMM stays a separate launch)
    int32_t fuse_first, S0;
    const float* pre0;     // [S0][B][K0] split-K slabs of x W0^T (forward)
    long long pre0_stride;
    sca_mid_layer first;   // bias / Wc / mask / yn / rs / h / da of the first layer (W, K unused); N == K0
} sca_mid_args;
}
