/* scarches_hip.h - C ABI of the MI355X-native conditional-VAE train/map step (libscarches_hip.so).
 *
 * The reference (theislab/scarches v0.6.1) is pure Python on PyTorch and has no FFI layer; its narrowest seam is
 * `nn.Module.forward(...) -> (recon, kl, mmd)` + `loss.backward()` + `optimizer.step()`
 * (scarches/trainers/trvae/unsupervised.py:62-64, trainers/trvae/trainer.py:272-280).  Each entry point below
 * replaces the ATen op sequence the reference runs for one piece of that seam; the citation names the reference
 * lines whose result it reproduces.  All pointers are DEVICE pointers unless stated; all matrices are row-major
 * fp32; `stream` is a hipStream_t passed as void* (NULL = default stream).  Every function only enqueues work
 * (no allocation, no synchronisation) and returns 0 on success, a negative value for an unsupported shape /
 * argument, or a positive hipError_t.
 *
 * Reference-side binding: see INTEGRATION.md (ctypes stubs a maintainer would add next to the reference modules).
 */
#ifndef SCARCHES_HIP_H
#define SCARCHES_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Library / device sanity: returns the ABI version (3; bumped on every incompatible change of a signature or of the
 * meaning of an argument - callers assert it after dlopen); fills name[0..len) with the HIP device name (host buffer). */
int sca_abi_version(void);
int sca_device_name(char* name, int len);

/* ---- CSR minibatch loader ------------------------------------------------------------------------------------
 * Replaces AnnotatedDataset.__getitem__ + custom_collate (scarches/dataset/trvae/anndata.py:74-92,
 * trainers/trvae/_utils.py:208-240) and the log1p of trVAE.forward (models/trvae/trvae.py:117):
 * xout[b,:] = (apply_log1p ? log(1+x) : x) of CSR row row_ids[b] (row_ids NULL: rows 0..B-1),
 * sizefactor[b] = sum of the raw row (dataset/trvae/anndata.py:44).  CSR must be canonical (unique, any order). */
int sca_csr_gather_rows(const int64_t* indptr, const int32_t* indices, const float* data, const int64_t* row_ids,
                        int B, int G, int apply_log1p, float* xout, int ldx, float* sizefactor, void* stream);
int sca_dense_gather_rows(const float* X, int ldX, const int64_t* row_ids, int B, int G, int apply_log1p,
                          float* xout, int ldx, float* sizefactor, void* stream);

/* ---- fp32 MFMA GEMM ------------------------------------------------------------------------------------------
 * Replaces the nn.Linear / one-hot GEMMs of CondLayers, Encoder, Decoder (models/trvae/modules.py:21-27,77,86-87,
 * 159,173-181) and their autograd backward.  C[m][n] = sum_k A(m,k) B(k,n);
 *   amode 0: A(m,k)=A[m*lda+k]   amode 1: A(m,k)=A[k*lda+m]
 *   bmode 0: B(k,n)=B[n*ldb+k]   bmode 1: B(k,n)=B[k*ldb+n]
 * splitk > 1 writes `splitk` partial slabs C + s*slab_stride (consumers sum them); bias[n] (optional) is added
 * only when splitk == 1. */
int sca_gemm_splitk_heuristic(int M, int N, int K);
int sca_gemm_f32(int amode, int bmode, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M,
                 int N, int K, int splitk, long long slab_stride, const float* bias, void* stream);
/* Several weight-gradient GEMMs dW = dY^T X in ONE launch (torch autograd issues one mm per nn.Linear weight in
 * CondLayers/Encoder/Decoder backward, scarches/models/trvae/modules.py:21-27,52-97,124-208).  Every job must have
 * amode == bmode == 1; 64x64 tiles; each job has its own split-K count and slab stride (sum the slabs as for
 * sca_gemm_f32).  At most sca_gemm_group_max_jobs() jobs. */
typedef struct {
    int32_t amode, bmode;
    const float* A; const float* B; float* C;
    int32_t lda, ldb, ldc, M, N, K, splitk, pad;
    long long slab_stride;
} sca_gemm_job;
int sca_gemm_group_max_jobs(void);
int sca_gemm_group_f32(const sca_gemm_job* jobs, int n, void* stream);

/* Register-direct kernel for two k-major operands, C (+ slabs) = A^T B with A(m,k) = A[k*lda+m], B(k,n) = B[k*ldb+n]
 * (the amode == bmode == 1 case of sca_gemm_f32: the weight-gradient products dW = dY^T X that torch autograd issues
 * per nn.Linear, scarches/models/trvae/modules.py:21-27,77,159,173-181).  MFMA fragments are loaded straight from
 * global memory (no LDS staging, no barrier in the k loop); a launch writes `splitk` slabs, each already summed over
 * 4 wave-level k ranges.  Requires sca_gemm_kk_supported(...) == 1 (lda % 4 == 0, M % 4 == 0, ldb % 2 == 0,
 * N % 2 == 0, 16- / 8-byte aligned bases); returns -2 otherwise.  sca_gemm_kk_splitk_heuristic: the slab count the
 * engine sizes its workspaces for. */
int sca_gemm_kk_splitk_heuristic(int M, int N, int K);
int sca_gemm_kk_supported(const float* A, int lda, const float* B, int ldb, int M, int N);
int sca_gemm_kk_f32(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                    int splitk, long long slab_stride, void* stream);

/* Same with an explicit block tile (64*tile_wm) x (64*tile_wn), tile_w* in {1,2}; 0 = automatic (tuning hook). */
int sca_gemm_f32_cfg(int amode, int bmode, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M,
                     int N, int K, int splitk, long long slab_stride, const float* bias, int tile_wm, int tile_wn,
                     void* stream);
int sca_splitk_reduce(const float* slabs, int S, long long slab_stride, float* dst, long long n, void* stream);

/* ---- layer epilogues -----------------------------------------------------------------------------------------
 * sca_layer_fwd: sum of S GEMM slabs + bias + condition column Wc[:,cond[b]] (or dense condition vector
 * cvec[b,:E] . Wc[h,:E], scPoli embeddings: models/scpoli/scpoli.py:137-150) -> LayerNorm(no affine, eps 1e-5)
 * (norm=1) -> ReLU -> Dropout(p) (models/trvae/modules.py:78-84).  mask: optional injected keep-mask [B][H] (parity
 * tests); otherwise Philox(seed + *step, rng_stream).  Writes ynorm (pre-ReLU, needed by backward), rstd, h.
 * sca_layer_bwd: gradient through Dropout, ReLU, LayerNorm: da = d loss / d (Linear output).
 * sca_cond_colsum: dWc[h][c] = sum_{b: cond[b]=c} da[b][h]; db[h] = sum_b da[b][h] (autograd of the one-hot GEMM and
 * the bias broadcast).  segT: scratch of max(C,1)*H floats. */
int sca_layer_fwd(const float* slabs, int S, long long slab_stride, int B, int H, const float* bias, const float* Wc,
                  int C, const int32_t* cond, const float* cvec, int E, int norm, int relu, float p,
                  const uint8_t* mask, uint64_t seed, uint32_t rng_stream, const uint32_t* step, float* ynorm,
                  float* rstd, float* h, void* stream);
int sca_layer_bwd(const float* dh, int S, long long slab_stride, const float* extra, int B, int H, const float* ynorm,
                  const float* rstd, int norm, int relu, float p, const uint8_t* mask, uint64_t seed,
                  uint32_t rng_stream, const uint32_t* step, float* da, void* stream);
int sca_cond_colsum(const float* da, int B, int H, const int32_t* cond, int C, float* segT, float* dWc, float* db,
                    void* stream);

/* ---- latent block --------------------------------------------------------------------------------------------
 * heads = [mean | log_var] GEMM slabs.  z = mu + eps*sqrt(exp(logvar)+1e-4) (models/base/_base.py:391-407),
 * kl_row[b] = sum_z KL(N(mu,sigma)||N(0,1)) (models/trvae/trvae.py:143-147).  eps NULL: Philox noise.
 * sample=0 gives z = mu (get_latent(mean=True), models/base/_base.py:409-432).
 * sca_latent_bwd: d/d[mean|log_var] of  (upstream dz + dz2) + hyper[0]/B * KL. */
int sca_latent_fwd(const float* heads, int S, long long slab_stride, const float* bias, int B, int Z, const float* eps,
                   uint64_t seed, uint32_t rng_stream, const uint32_t* step, int sample, float* mu_lv, float* eps_out,
                   float* z, float* kl_row, void* stream);
int sca_latent_bwd(const float* dz, int S, long long slab_stride, const float* dz2, int S2, long long slab_stride2,
                   const float* mu_lv, const float* eps, const float* hyper, int B, int Z, float* dheads, void* stream);

/* ---- reconstruction losses -----------------------------------------------------------------------------------
 * sca_theta_prep: tab_theta[c][g] = exp(theta[g][c]), tab_lte[c][g] = log(tab_theta + 1e-8) (trvae.py:139-140).
 * sca_recon_rows: softmax over genes, x size factor, NB (zinb=0, models/trvae/losses.py:63-102) or ZINB (zinb=1,
 * losses.py:109-163) log-likelihood with counts read straight from CSR, and - when compute_grad - the gradient of
 * recon = -sum/B written IN PLACE over logits (and pi), theta gradients accumulated (unscaled sum_b theta*dll/dtheta)
 * into dthetaT[C][G] and the output-bias gradients into dbias[G] (both must be zeroed by the caller).
 * cond_off (optional, [C][G]): per-condition offsets added to the logits on load - expiMap's single-layer decoder adds
 * cond_L.weight[:, c_b] right before the softmax (models/expimap/modules.py:83-84, 142).
 * row_dsf (optional, [B]): d recon / d sizefactor_b - expiMap's library-size encoder makes the size factor a function of
 * the parameters (models/expimap/expimap.py:164-171,186-187).
 * sca_mse_rows: sum of squares behind an output activation `relu` = 0 identity (expimap.py:182-184), 1 ReLU
 * (trvae.py:125-127), 2 softplus, 3 exp (MaskedLinearDecoder last layers, models/expimap/modules.py:119-127).
 * sca_nb_elem_rows: NB log-likelihood on an elementwise mean mu = act(lin (+ cond_off)) without softmax / size factor
 * (expiMap with decoder_last_layer in {identity, relu, softplus, exp}: expimap.py:188-196, losses.py:63-102), counts from
 * CSR; d recon / d lin in place, theta gradients accumulated into dthetaT like sca_recon_rows.
 * sca_transpose_scale: dst[cols][rows] = scale * src[rows][cols] (dthetaT -> theta.grad layout [G][C]). */
int sca_theta_prep(const float* theta, int G, int C, float* tab_theta, float* tab_lte, void* stream);
/* sca_theta_prep plus two zero fills (dthetaT, the flat gradient buffer = optimizer.zero_grad(), trainer.py:272) in the
 * same launch: the per-step prologue as one kernel.  zero0 / zero1 may be NULL; pointers 16-byte aligned, byte counts
 * multiples of 16 (-1 otherwise).  G * C == 0 skips the tables. */
int sca_theta_prep_zero(const float* theta, int G, int C, float* tab_theta, float* tab_lte, void* zero0,
                        long long bytes0, void* zero1, long long bytes1, void* stream);
int sca_recon_rows(int zinb, float* logits, int ldl, float* pi, const int64_t* indptr, const int32_t* indices,
                   const float* data, const int64_t* row_ids, const float* sizefactor, const int32_t* cond,
                   const float* tab_theta, const float* tab_lte, int B, int G, int C, int rows_per_block, float* row_ll,
                   float* dthetaT, float* dbias, float* dbias_pi, int compute_grad, const float* cond_off,
                   float* row_dsf, void* stream);
/* sca_recon_rows with the per-step gamma-term table of sca_nb_gamma_table (gam_tab [C][G][17][2], NULL = none, then
 * identical to sca_recon_rows): integer counts <= 16 look their lgamma / digamma differences up instead of running the
 * per-count product loop (the table holds the results of that same loop, so values do not change). */
int sca_recon_rows_tab(int zinb, float* logits, int ldl, float* pi, const int64_t* indptr, const int32_t* indices,
                       const float* data, const int64_t* row_ids, const float* sizefactor, const int32_t* cond,
                       const float* tab_theta, const float* tab_lte, int B, int G, int C, int rows_per_block,
                       float* row_ll, float* dthetaT, float* dbias, float* dbias_pi, int compute_grad,
                       const float* cond_off, float* row_dsf, const float* gam_tab, void* stream);

/* ---- output layer fused with the NB likelihood (round 3) -------------------------------------------------------
 * sca_out_nb_fused = sca_gemm_f32 (logits = h W^T + bias; models/trvae/modules.py:179-181,207) + sca_recon_rows
 * (zinb = 0: softmax, mu = gamma * sizefactor, theta = exp(theta[:, c]), NB log-likelihood and its backward;
 * trvae.py:136-141, losses.py:63-102) in ONE launch: the logits stay in MFMA accumulator registers, only d loss /
 * d logits is written (dl [B][ldl]; it also serves as scratch for the raw logits while the kernel runs).  Outputs as
 * sca_recon_rows: row_ll [B], dthetaT [C][G] (accumulated, pre-zeroed), dbias [G] (accumulated, pre-zeroed, may be
 * NULL).  gam_tab [C][G][17][2]: (lgamma, digamma) differences of integer counts 0..16 per (condition, gene), built by
 * sca_nb_gamma_table from tab_theta once per step; ext_ovf [B][ld_ovf >= G]: scratch touched only by cells with more
 * than 512 stored counts.  h / W: 16-byte aligned, ldh / ldw multiples of 4.  Shapes: sca_out_nb_fused_supported(G, K)
 * (G <= 2048, K in {32, 64, 128, 256}), else -2.
 * Measured on MI355X: not yet faster than the pair it replaces - the engine uses it only with
 * SCA_OUT_FUSED=1. */
int sca_out_nb_fused_supported(int G, int K);
int sca_nb_gamma_table(const float* tab_theta, int C, int G, float* gam_tab, void* stream);
/* development aid: 12 s_memrealtime marks per workgroup of the next sca_out_nb_fused launches (NULL: off, the default) */
int sca_out_nb_fused_debug_stamps(long long* buf);
int sca_out_nb_fused(const float* h, int ldh, const float* W, int ldw, const float* bias, const int64_t* indptr,
                     const int32_t* indices, const float* data, const int64_t* row_ids, const float* sizefactor,
                     const int32_t* cond, const float* tab_theta, const float* gam_tab, int B, int G, int K, int C,
                     float* dl, int ldl, float* ext_ovf, int ld_ovf, float* row_ll, float* dthetaT, float* dbias,
                     int compute_grad, void* stream);
/* sca_poisson_rows: scPoli's 'poisson' head (models/scpoli/scpoli.py:189-191,607,640; models/trvae/losses.py:44-60): the
 * softmax output is the Poisson rate, row_loss[b] = sum_g [rate - xlogy(x, rate) + lgamma(x + 1)] with counts from CSR,
 * d (sum_b row_loss / B) / d logits written in place; sizefactor[b] = sum_g x_bg. */
int sca_poisson_rows(float* logits, int ldl, const int64_t* indptr, const int32_t* indices, const float* data,
                     const int64_t* row_ids, const float* sizefactor, int B, int G, int rows_per_block, float* row_loss,
                     int compute_grad, void* stream);
int sca_nb_elem_rows(int act, float* lin, int ldl, const int64_t* indptr, const int32_t* indices, const float* data,
                     const int64_t* row_ids, const int32_t* cond, const float* tab_theta, const float* tab_lte, int B,
                     int G, int C, int rows_per_block, float* row_ll, float* dthetaT, int compute_grad,
                     const float* cond_off, void* stream);
int sca_mse_rows(float* lin, int ldl, const float* x, int ldx, int B, int G, int rows_per_block, float* row_loss,
                 float* dbias, int compute_grad, int relu, const int32_t* cond, const float* cond_off, void* stream);
int sca_transpose_scale(const float* src, int rows, int cols, float scale, float* dst, void* stream);
/* Elementwise helpers of expiMap's library-size encoder path (models/expimap/expimap.py:186-187): the size factor is
 * exp(sampled library latent) (sca_exp_f32) and its gradient chains through a product (sca_mul_f32: dst = a * b). */
int sca_exp_f32(const float* src, float* dst, long long n, void* stream);
int sca_mul_f32(const float* a, const float* b, float* dst, long long n, void* stream);

/* ---- MMD -----------------------------------------------------------------------------------------------------
 * Multi-scale RBF MMD between condition blocks (models/trvae/losses.py:166-270, _utils.py:18-24), forward value as
 * double partial sums (sum them) and gradient w.r.t. feat as sca_mmd_num_slabs(B,D) slabs [B][D] (sum them).
 * boundary < 0: all pairs.  Wt: scratch [C][C]; counts: optional [C]. */
int sca_mmd_num_slabs(int B, int D);
int sca_mmd_num_partials(int B, int D);
int sca_mmd_fwd_bwd(const float* feat, int ld, const int32_t* cond, int B, int D, int C, int boundary, float beta,
                    float* Wt, int32_t* counts, double* loss_part, float* grad_slabs, long long slab_stride,
                    void* stream);
/* The same in two steps, for a caller that knows the minibatch's conditions before its features exist (the weight
 * table = the partition of the batch by condition, losses.py:255-268): sca_mmd_prepare builds Wt (one small
 * latency-bound workgroup, issued early on a side stream), sca_mmd_fwd_bwd_prepared evaluates rows [row_lo, row_lo +
 * row_n) of the minibatch against all B columns with it (outputs as for sca_mmd_rows_fwd_bwd below: loss partials =
 * the block's share, gradient slabs [row_n][D]; row_lo = 0, row_n = B gives value and gradient of sca_mmd_fwd_bwd).
 * Only where sca_mmd_prepared_supported(B, D) == 1 (-2 otherwise). */
int sca_mmd_prepare(const int32_t* cond, int B, int C, int boundary, float beta, float* Wt, int32_t* counts,
                    void* stream);
int sca_mmd_prepared_supported(int B, int D);
int sca_mmd_fwd_bwd_prepared(const float* feat, int ld, const int32_t* cond, int B, int D, int C, const float* Wt,
                             int row_lo, int row_n, double* loss_part, float* grad_slabs, long long slab_stride,
                             void* stream);
/* Tuning hook, process-wide: 0 = vector-ALU row-block kernel (default), 1 = the form that computes the pair distances
 * and the owner gradients with v_mfma_f32_16x16x4_f32 (D <= 10; same outputs, different slab / partial counts - call
 * before sizing workspaces with sca_mmd_*_num_slabs / _num_partials).  Returns the previous selection. */
int sca_mmd_select_form(int form);
/* Scheduling hook, process-wide, no effect on results: bytes of LDS (0..65536, default 40000) each workgroup of the
 * vector-ALU row-block kernel reserves without touching them, which caps its workgroups per CU (40000 -> three) and so
 * leaves room for the kernels launched beside it.  Applies to launches issued after the call; the environment variable
 * SCA_MMD_LDS_PAD overrides it.  Returns 0, -1 for a size out of range.  No counterpart in the reference (a stream-
 * schedule knob of this implementation). */
int sca_mmd_reserve_lds(int bytes);
/* Row-block form for data-parallel training (same reference lines; SURVEY 8(e) exception 1: the reference's MMD
 * couples every pair of cells of the global minibatch).  feat / cond hold the all-gathered minibatch [B]; the launch
 * evaluates rows [row_lo, row_lo + row_n) against all B columns: loss partials = the block's share of the MMD (the
 * shares of all blocks add up to the whole-minibatch value), gradient slabs [row_n][D] for the block's rows.
 * Conditions outside [0, C) mark padding rows and are ignored.  beta scales value and gradient linearly. */
int sca_mmd_rows_num_slabs(int B, int row_n, int D);
int sca_mmd_rows_num_partials(int B, int row_n, int D);
int sca_mmd_rows_fwd_bwd(const float* feat, int ld, const int32_t* cond, int B, int D, int C, int boundary, float beta,
                         float* Wt, int32_t* counts, int row_lo, int row_n, double* loss_part, float* grad_slabs,
                         long long slab_stride, void* stream);

/* ---- optimiser / bookkeeping ---------------------------------------------------------------------------------
 * hyper (device, float[8]): [0] alpha_kl  [1] lr  [2] 1-beta1^t  [3] sqrt(1-beta2^t)  [4] adam eps  [5] clip value
 * [6] weight of this minibatch in the epoch-log accumulators (1; n_local / n_global on a data-parallel rank; 0 reads
 *     as 1, so a zero-initialised block keeps the un-weighted behaviour).
 * sca_adam_step: torch.optim.Adam semantics (trainers/trvae/trainer.py:221-223, clip: :277-278) on a flat range.
 * sca_finalize_losses: losses[0..3] = recon (= recon_sign * sum(row_recon)/B), kl, mmd, recon + alpha*kl + mmd
 * (trainers/trvae/unsupervised.py:62-70); accum[0..3] += hyper[6] * those, accum[4] += hyper[6] (per-epoch log means
 * without host syncs; trainers/trvae/trainer.py:284-285). */
int sca_adam_step(float* p, const float* g, float* m, float* v, long long n, const float* hyper, float beta1,
                  float beta2, float weight_decay, void* stream);
/* sca_begin_step: hyper[0..5] <- the per-step scalars (values of trainer.py:193-209 annealing, param_groups lr, Adam
 * bias corrections for step t), step_ctr += 1 (Philox stream of this step's dropout / z noise).
 * sca_fill_zero: zero `bytes` bytes (optimizer.zero_grad() for the accumulating gradient targets). */
int sca_begin_step(float* hyper, uint32_t* step_ctr, float alpha_kl, float lr, float bias_corr1,
                   float bias_corr2_sqrt, float eps, float clip, void* stream);
/* sca_begin_step plus the copy of the step's row ids [B] (int64) and conditions [B] (int32) into the static
 * buffers that a recorded launch plan / HIP graph reads: the reference hands each minibatch to the model as fresh
 * tensors (trainers/trvae/trainer.py:305-312 on_iteration), a replayed plan needs them at fixed addresses. */
int sca_begin_step_batch(float* hyper, uint32_t* step_ctr, float alpha_kl, float lr, float bias_corr1,
                         float bias_corr2_sqrt, float eps, float clip, const int64_t* rows_src, int64_t* rows_dst,
                         const int32_t* cond_src, int32_t* cond_dst, int B, void* stream);
/* ... plus the row ids [B_next] of the NEXT minibatch into next_dst: the engine's training step gathers minibatch
 * n + 1 into the dense input buffer beside the gradient tail and Adam of step n (the role the reference gives its
 * DataLoader workers, trainers/trvae/trainer.py:171-181 n_workers; here the loader is a kernel, so the
 * overlap is a stream schedule).  B_next = 0: same as sca_begin_step_batch. */
int sca_begin_step_batch_next(float* hyper, uint32_t* step_ctr, float alpha_kl, float lr, float bias_corr1,
                              float bias_corr2_sqrt, float eps, float clip, const int64_t* rows_src,
                              int64_t* rows_dst, const int32_t* cond_src, int32_t* cond_dst, int B,
                              const int64_t* next_src, int64_t* next_dst, int B_next, void* stream);
int sca_fill_zero(void* p, long long bytes, void* stream);
int sca_finalize_losses(const float* row_recon, float recon_sign, const float* row_kl, int B, const double* mmd_part,
                        int n_part, const float* hyper, float* losses, double* accum, void* stream);

/* ---- gradient tail ---------------------------------------------------------------------------------------------
 * Every batch reduction that only feeds Adam, as jobs of ONE launch: kind 0 sums the split-K slabs of a weight-gradient
 * GEMM (as sca_splitk_reduce), kind 1 is the column reduction of sca_cond_colsum (db[h] = sum_b da[b][h], dWc[h][c] =
 * sum over rows with cond == c; autograd of the bias broadcast / one-hot GEMM of CondLayers, modules.py:21-27).
 * `jobs` is a HOST array (copied into the kernel arguments); at most sca_grad_tail_max_jobs() jobs, C <= 512.
 * For B > 256 the kind-1 targets (db, dWc) are accumulated with atomics and must be zeroed by the caller. */
typedef struct {
    int32_t kind;          /* 0: dst[i] = sum_s src[s*stride + i], i < n;  1: column sums of da [B][H] */
    int32_t S;
    long long stride;
    const float* src;      /* kind 0: slabs; kind 1: da */
    float* dst;            /* kind 0: destination; kind 1: db [H] or NULL */
    long long n;
    float* dWc;            /* kind 1: [H][C] or NULL */
    const int32_t* cond;   /* kind 1: [B] (with dWc) */
    int32_t B, H, C, pad;
} sca_tail_job;
int sca_grad_tail_max_jobs(void);
int sca_grad_tail(const sca_tail_job* jobs, int n_jobs, void* stream);

/* ---- fused middle of the network ------------------------------------------------------------------------------
 * Everything between the two wide GEMMs of a step in ONE launch per direction: the encoder layers after the first
 * (modules.py:52-97), the fused mean / log-var heads, sampling + KL (_base.py:391-407, trvae.py:143-147) and the decoder
 * hidden layers (modules.py:124-208), each with bias / one-hot condition column / LayerNorm / ReLU / dropout.  A
 * workgroup walks the chain for a tile of 16 cells with activations in LDS.  sca_mid_fwd reads h_in [B][K0] (output of
 * the first encoder layer) and writes every tensor the unfused kernels write (yn, rs, h per layer, mu_lv, eps, z,
 * kl_row); sca_mid_bwd consumes the dX slabs of the output GEMM (dh, S, dh_stride) and optional dz2 [B][Z], writes the
 * per-layer da, dheads and dh_out [B][K0] (gradient w.r.t. h_in).  Layer widths <= sca_mid_max_width().  `a` is a HOST
 * struct (copied into the kernel arguments).  Encoder-only forward pass (get_latent, _base.py:285-340): n_dec = 0 with
 * dec[0] still naming the first decoder layer's weights (the kernels prefetch them before they return), and
 * yn = rs = h = NULL in every layer = nothing is kept for a backward pass. */
typedef struct {
    const float* W; const float* bias; const float* Wc; const uint8_t* mask;
    float* yn; float* rs; float* h; float* da;
    int32_t N, K; uint32_t rng_stream; int32_t pad;
} sca_mid_layer;
typedef struct {
    sca_mid_layer enc[4]; sca_mid_layer dec[4];
    int32_t n_enc, n_dec;
    const float* Wh; const float* bh; const float* eps_in; float* mu_lv; float* eps_out; float* z; float* kl_row;
    int32_t Z, Kh; uint32_t rng_latent; int32_t sample;
    int32_t B, C; const int32_t* cond;
    int32_t norm; float p; uint64_t seed; const uint32_t* step;
    const float* h_in; int32_t K0, S;
    const float* dh; long long dh_stride; const float* dz2; const float* hyper; float* dheads; float* dh_out;
    /* optional: the first encoder layer's epilogue fused in (its GEMM stays a separate launch) */
    int32_t fuse_first, S0;
    const float* pre0;     /* [S0][B][K0] split-K slabs of x W0^T (forward) */
    long long pre0_stride;
    sca_mid_layer first;   /* bias / Wc / mask / yn / rs / h / da of the first layer (W, K unused); N == K0 */
} sca_mid_args;
int sca_mid_max_width(void);
int sca_mid_fwd(const sca_mid_args* a, void* stream);
int sca_mid_bwd(const sca_mid_args* a, void* stream);
/* The same two passes through the first kernel family (weights staged through LDS in 32-deep chunks, 4 waves per
 * 16-cell tile; fused_mid.hip).  sca_mid_fwd / sca_mid_bwd run the second family (fused_mid_direct.hip: MFMA weight
 * fragments read straight from L2 with a re-labelled reduction index, 8 waves per tile, no barrier inside a GEMM) and
 * route here when SCA_MID_DIRECT=0 is set in the environment; same arguments, same outputs (summation order aside). */
int sca_mid_fwd_staged(const sca_mid_args* a, void* stream);
int sca_mid_bwd_staged(const sca_mid_args* a, void* stream);

/* ---- BatchNorm1d variant of the layer block (use_bn=True) -------------------------------------------------------
 * nn.BatchNorm1d(affine=True) between Linear and ReLU (models/trvae/modules.py:79,146,161).  sca_bn_fwd takes the
 * pre-activation a [B][H] (sca_layer_fwd with norm=0, relu=0), overwrites it with xhat and writes
 * h = dropout(relu(xhat*gamma + beta)); use_batch_stats=1: biased batch statistics (train mode) and, with
 * update_running, running <- (1-momentum) running + momentum * (mean, unbiased var); use_batch_stats=0: running
 * statistics (eval mode).  stats [2][H] = (mean, rstd) is kept for backward; acc: scratch of 2*H doubles.
 * sca_bn_bwd: da = d loss / d a, dgamma, dbeta (either may be NULL) from the upstream slabs dh (+ extra). */
int sca_bn_fwd(float* a, int B, int H, const float* gamma, const float* beta, float* running_mean, float* running_var,
               int use_batch_stats, int update_running, float momentum, float p, const uint8_t* mask, uint64_t seed,
               uint32_t rng_stream, const uint32_t* step, double* acc, float* stats, float* h, void* stream);
int sca_bn_bwd(const float* dh, int S, long long slab_stride, const float* extra, int B, int H, const float* xhat,
               const float* stats, const float* gamma, const float* beta, int batch_stats, float p,
               const uint8_t* mask, uint64_t seed, uint32_t rng_stream, const uint32_t* step, double* acc, float* da,
               float* dgamma, float* dbeta, void* stream);

/* ---- expiMap ---------------------------------------------------------------------------------------------------
 * sca_masked_copy: dst[r*ldd+c] = src[r*lds+c] * (mask ? mask[r*ldm+c] : 1).  Packs MaskedLinear weights
 * (weight*mask, models/expimap/modules.py:24-25) of expr_L / ext_L_m / ext_L into ONE decoder matrix
 * [G][L+n_ext_m+n_ext] (column order of MaskedCondLayers.forward, modules.py:62-85) and scatters the packed weight
 * gradient back through the masks (autograd of weight*mask).
 * sca_prox_l1 / sca_prox_group_lasso: ProxL1 / ProxGroupLasso (trainers/expimap/regularized.py:35-58, 5-32) in place
 * on a [rows][cols] weight; I (optional) has the weight's shape, entries != 0 are left untouched; omega: [cols].
 * sca_prox_group_lasso keeps its column sums of squares (rows > 256, cols <= 16384) in a 64 KB device array of the
 * library: two calls that overlap on one GPU (different streams) race on it; issue them on one stream.
 * sca_hsic_fwd_bwd: hsic(z[:, side==0], z[:, side==1]) (models/expimap/losses.py:23-31) with RBF bandwidths sigma_a /
 * sigma_b (losses.py:15-21, computed by the caller); writes (accumulate=0) or adds (accumulate=1) weight*hsic into
 * loss_weighted[0] (double) and hsic into loss_raw[0], and ADDS weight * d hsic / d z into dz (may be NULL).
 * work: scratch of sca_hsic_workspace_floats(B) floats. */
int sca_masked_copy(const float* src, int lds, const float* mask, int ldm, float* dst, int ldd, int rows, int cols,
                    void* stream);
int sca_prox_l1(float* W, long long n, float alpha, const float* I, void* stream);
int sca_prox_group_lasso(float* W, int rows, int cols, float alpha, const float* omega, void* stream);
/* The same operator with the column sums of squares in a caller-owned workspace of `cols` floats (any contents on
 * entry) instead of the library's file-scope array: no state in the library, so calls on different streams may overlap
 * when their workspaces differ.  -1 without a workspace. */
int sca_prox_group_lasso_ws(float* W, int rows, int cols, float alpha, const float* omega, float* colsq_ws,
                            void* stream);
long long sca_hsic_workspace_floats(int B);
int sca_hsic_fwd_bwd(const float* z, int ld, int B, int D, const uint8_t* side, float sigma_a, float sigma_b,
                     float weight, float* work, double* loss_weighted, float* loss_raw, int accumulate, float* dz,
                     int lddz, void* stream);

/* Tile-skipping masked decoder (csrc/masked_gemm.hip): the three products of MaskedLinear (modules.py:9-25) /
 * MaskedCondLayers.forward (modules.py:62-85) and of their autograd, issued only over the (32 genes x 2 programs)
 * sub-tiles of W o M that are not fully masked.  Plan, built once by the caller from the mask's nonzero pattern:
 * NT = ceil(G/32) gene tiles; tile_k[NT] = number of active columns of a tile, rounded up to even, <= Kmax (a multiple
 * of 8, <= sca_masked_max_cols()); tile_cols[NT][Kmax] = their sorted indices, padded with Zt (a column of zeros).
 *   sca_masked_pack   Wc[t][j][n] = Weff[32t+n][tile_cols[t][j]]  (Weff [G][ldw] = weight * mask, dense).  Wc holds
 *                     2 * NT*Kmax*32 floats: that array, then the same values in the forward kernel's fragment order;
 *                     16-byte aligned
 *   sca_masked_fwd    out[b][g] = sum_j z[b][tile_cols[t][j]] * Wc[t][j][g - 32t]            (out [B][ldo], z [B][ldz])
 *   sca_masked_dx     dz slabs [S][B][Zt], S = sca_masked_dx_slabs(B, NT), summed by the consumer: dz = dl @ Weff
 *   sca_masked_dw     dWc slabs [S][NT][Kmax][32], S = sca_masked_dw_slabs(B, NT): dWc[t][j][n] = sum_b z[b][col j] dl[b][32t+n]
 *   sca_masked_unpack dWeff[32t+n][tile_cols[t][j]] = sum_s dWc[s][t][j][n]  (dWeff pre-zeroed by the caller; the entries
 *                     of masked-out weights inside an active column are removed by the mask multiply of sca_masked_copy)
 * Return -2 when Kmax > sca_masked_max_cols() or Zt > sca_masked_max_width(), the limits of what the kernels keep in
 * registers / LDS (the caller uses sca_gemm_f32 then). */
int sca_masked_max_cols(void);
int sca_masked_max_width(void);
int sca_masked_fwd_tiles_per_wg(int B, int NT);
int sca_masked_dx_slabs(int B, int NT);
int sca_masked_dw_slabs(int B, int NT);
int sca_masked_pack(const float* Weff, int ldw, const int32_t* tile_k, const int32_t* tile_cols, int NT, int Kmax,
                    int G, int Zt, float* Wc, void* stream);
int sca_masked_unpack(const float* dWc, int S, long long slab_stride, const int32_t* tile_k, const int32_t* tile_cols,
                      int NT, int Kmax, int G, int Zt, float* dWeff, int ldw, void* stream);
int sca_masked_fwd(const float* z, int ldz, const float* Wc, const int32_t* tile_k, const int32_t* tile_cols, int NT,
                   int Kmax, int B, int G, int Zt, float* out, int ldo, void* stream);
int sca_masked_dx(const float* dl, int ldl, const float* Wc, const int32_t* tile_k, const int32_t* tile_cols, int NT,
                  int Kmax, int B, int G, int Zt, float* dz_slabs, long long slab_stride, void* stream);
int sca_masked_dw(const float* z, int ldz, const float* dl, int ldl, const int32_t* tile_k, const int32_t* tile_cols,
                  int NT, int Kmax, int B, int G, int Zt, float* dWc_slabs, long long slab_stride, void* stream);

/* ---- scPoli ----------------------------------------------------------------------------------------------------
 * sca_embed_lookup: out[b][0..E) = W[idx[b*stride]][0..E) after renormalising the touched rows of W [n][E] in place
 * to L2 norm <= max_norm (nn.Embedding(max_norm), models/scpoli/scpoli.py:99-101,137; max_norm <= 0: no renorm).
 * flags: zeroed scratch of n bytes (left zeroed).  sca_embed_scatter_add: dW[idx[b*stride]][:] += ga[b][:] + gb[b][:]
 * (autograd of the lookup; dW pre-zeroed by the caller, ga / gb may be NULL).
 * sca_proto_loss: prototype_labeled_loss (labels [B][n_levels], -1 = unknown, rows with labeled[b] == 0 skipped) and
 * prototype_unlabeled_loss (trainers/scpoli/trainer.py:721-773, p = 2): loss_weighted[0] = weight*labeled +
 * weight_u*unlabeled (double), raw[0] / raw[1] (+)= the two unweighted values, dz [B][Z] = d(loss_weighted)/dz
 * (overwritten), dPu (+)= d unlabeled / d Pu (pre-zeroed, may be NULL).  P or Pu may be NULL.  work: scratch of
 * sca_proto_workspace_bytes(B, K, Ku) bytes.
 * sca_class_moments: fp64 per-class count / sum z / sum z z^T over the (row, level) entries of labels (rows with
 * sel[b] == 0 skipped), ACCUMULATED into pre-zeroed cnt[K], sum[K][Z], outer[K][Z][Z] - the quantities DP ranks
 * all-reduce.  sca_class_moments_finish: mean / unbiased covariance (trainers/scpoli/trainer.py:667-719,
 * _utils.py:237-282) for classes with cnt > 0 and update[v] != 0 (update NULL = all), else the previous values. */
int sca_embed_lookup(float* W, int n, int E, const int32_t* idx, int stride, int B, float max_norm, uint8_t* flags,
                     float* out, int ldo, void* stream);
int sca_embed_scatter_add(const float* ga, const float* gb, int ldg, int B, int E, int n, const int32_t* idx,
                          int stride, float* dW, void* stream);
long long sca_proto_workspace_bytes(int B, int K, int Ku);
int sca_proto_loss(const float* z, int ld, int B, int Z, const float* P, int K, const int32_t* labels, int n_levels,
                   const uint8_t* labeled, float weight, const float* Pu, int Ku, float weight_u, void* work,
                   double* loss_weighted, float* raw, int accumulate_raw, float* dz, int lddz, float* dPu,
                   void* stream);
/* sca_proto_classify: scpoli.classify (models/scpoli/scpoli.py:270-315, p = 2) for one hierarchy level.  classes [Kc]
 * = prototype rows of the level (every entry in [0, K): the caller checks); dists [B][Kc] = euclidean distances
 * (get_prob = 0) or softmax(-distance) over the level (get_prob = 1); pred [B] = classes[argmin distance] /
 * classes[argmax probability] (first one on ties, as torch.min / torch.max); uncert [B] = that minimum / maximum.
 * -2 when Kc * Z floats exceed 48 KB. */
int sca_proto_classify(const float* z, int ld, int B, int Z, const float* P, int K, const int32_t* classes, int Kc,
                       int get_prob, float* dists, int32_t* pred, float* uncert, void* stream);
int sca_class_moments(const float* z, int ld, int N, int Z, const int32_t* labels, int n_levels, const uint8_t* sel,
                      int K, double* cnt, double* sum, double* outer, void* stream);
int sca_class_moments_finish(const double* cnt, const double* sum, const double* outer, int K, int Z,
                             const float* prev_mean, const float* prev_cov, const uint8_t* update, float* mean,
                             float* cov, void* stream);

/* ---- weighted kNN label transfer in the latent space (scarches/utils/knn.py:10-139) --------------------------
 * sca_knn_topk replaces KNeighborsTransformer(algorithm="brute", metric="euclidean").fit(ref).kneighbors(qry)
 * (knn.py:28-41,102): dist [Nq][k] ascending euclidean distances, idx [Nq][k] reference indices.  pd2 / pidx are
 * workspaces of sca_knn_num_chunks(Nr, Nq) * Nq * k floats / int32.  k <= sca_knn_max_k(), k <= Nr, D <= 64.
 * sca_knn_vote replaces the weight computation and the query x label double loop (knn.py:104-135) for ONE label
 * column: labels [Nr] int32 codes in sorted label order; pred [Nq] winning code (first in sorted order on ties),
 * prob [Nq] its probability in fp64 (uncertainty = max(1 - prob, 0)). */
int sca_knn_max_k(void);
int sca_knn_num_chunks(int Nr, int Nq);
int sca_knn_topk(const float* ref, int Nr, const float* qry, int Nq, int D, int k, float* pd2, int32_t* pidx,
                 float* dist, int32_t* idx, void* stream);
int sca_knn_vote(const float* dist, const int32_t* idx, int Nq, int k, const int32_t* labels, int32_t* pred,
                 double* prob, void* stream);

/* sca_knn_topk_static: sca_knn_topk WITHOUT a caller workspace and without any allocation.  It has no counterpart in
 * the reference; it is for callers that cannot allocate device memory - code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  Same kernels, same results (bit for bit) as sca_knn_topk; the partial
 * top-k lists live in a 16 MiB file-scope device array of the library (one copy per GPU the library runs on, held
 * until the process ends), whose per-device address is looked up once with hipGetSymbolAddress.  Limit:
 * sca_knn_num_chunks(Nr, Nq) * Nq * k <= sca_knn_static_max_entries() (= 2^21; e.g. 32768 queries at k = 50 against
 * any reference size); larger problems return -3 and belong to sca_knn_topk.  The array keeps the previous call's
 * contents; nothing needs clearing (every partial entry that is read is written first).  Calls that overlap on one
 * GPU (different streams) share the array and race: issue them on one stream.  Calls no hipMalloc / hipFree.
 * Warm-up: a call with Nq = 0 (other arguments ignored) resolves this device's address of the array and sets the
 * kernels' attributes, the only steps that are not stream operations; callers that use the entry point under HIP graph
 * capture make that call once per device before the capture begins.  The 16 MiB array is part of the code object:
 * every GPU that loads the library holds it, whether or not this entry point is ever called. */
long long sca_knn_static_max_entries(void);
int sca_knn_topk_static(const float* ref, int Nr, const float* qry, int Nq, int D, int k, float* dist, int32_t* idx,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif
