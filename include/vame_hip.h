/* libvame_hip.so -- C ABI of the MI355X (gfx950) kernels behind VAME's RNN-VAE train + embed path.
 *
 * VAME has no FFI of its own: the reference reaches its numerics through PyTorch modules.  Each
 * entry point below therefore names the reference code (file:line under /root/reference) whose
 * arithmetic it replaces; INTEGRATION.md shows the ctypes stub a VAME maintainer would add.
 *
 * Conventions: all pointers are DEVICE pointers owned by the caller (the library never allocates,
 * frees, keeps or synchronises); all tensors fp32 row-major; sizes/strides in ELEMENTS; `stream`
 * is a hipStream_t passed as void*.  Every function returns 0 on success or a negative VAME_E_*
 * code; vame_last_error() returns a thread-local description.  Re-entrant across streams/threads.
 */
#ifndef VAME_HIP_H
#define VAME_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define VAME_OK 0
#define VAME_E_BADARG (-1)
#define VAME_E_SHAPE (-2)
#define VAME_E_HIP (-3)
#define VAME_E_UNSUPPORTED (-4)

int vame_version(void);
/* sha256 of the kernel sources this library was built from (Makefile SRC_ID; "unidentified" for other builds): the key that
 * binds committed rocprofv3 counter summaries to a build (profiles/, bench.py roofline.traffic). */
const char* vame_source_id(void);
const char* vame_last_error(void);
/* Measurement aid (bench.py): nblocks workgroups each write {s_memtime (shader-clock ticks), s_memrealtime (100 MHz ticks), XCC id,
 * HW_ID} to out[4 * block].  Two launches around a region give the average shader clock over it per XCD:
 * (d memtime / d realtime) x 100 MHz -- the box's sustained clock under the measured kernels, reported next to every roofline
 * fraction (`clock_mhz`, `frac_at_clock`). */
int vame_clock_stamp(int64_t* out, int nblocks, void* stream);

/* Sliding-window batcher: out[b,l,f] = X[f*N + start_b + l]  (B,L,F).
 * Replaces SEQUENCE_DATASET.__getitem__ + default collate + permute(0,2,1)
 * (vame/model/dataloader.py:45-56, vame/model/rnn_vae.py:108-115) and the per-window slicing of
 * embedd_latent_vectors (vame/analysis/pose_segmentation.py:89-90).
 * starts == NULL means start_b = start0 + b (the stride-1 embedding sweep). */
int vame_window_gather_f32(const float* X, int64_t N, int F, const int64_t* starts, int64_t start0,
                           int B, int L, float* out, void* stream);

/* Generic fp32 MFMA GEMM  C[M,N] (+)= opA(A) opB(B) (+ bias[N]).
 *   a_kmajor = 0: A is (M,K) row-major, lda = row stride;  1: A is (K,M) row-major (A^T stored).
 *   b_kmajor = 0: B is (N,K) row-major (a torch Linear weight); 1: B is (K,N) row-major.
 *   Row i of an operand lives at (seg ? (i/seg)*seg_stride + (i%seg)*ld : i*ld): two-level
 *   addressing for (batch,time) rows of padded sequences and (ld = 0) time-constant GRU inputs.
 *   accumulate != 0: C += result.  splitk > 1 needs ws of splitk*M*N floats (reduced internally).
 *   a_gap != 0 (k-major A only): column m >= a_gap_at of A is read at m + a_gap, i.e. a block of a wider
 *   row is skipped -- dW_hh reads [da_r|da_z|dgh_n] out of dG = [da_r|da_z|dgi_n|dgh_n] in one call.
 * Replaces the nn.Linear / nn.GRU input-projection and all weight-gradient contractions that
 * torch autograd performs for vame/model/rnn_model.py:34-35,56-57,91-97,125-131. */
int vame_gemm_f32(int M, int N, int K, const float* A, int64_t lda, int a_kmajor, int64_t a_seg,
                  int64_t a_seg_stride, const float* B, int64_t ldb, int b_kmajor, int64_t b_seg,
                  int64_t b_seg_stride, const float* bias, float* C, int64_t ldc, int accumulate,
                  int splitk, float* ws, int a_gap_at, int a_gap, void* stream);

/* `count` (1..8) problems of identical shape / layout in one launch (split-K >= 8, N > 64): C[g] (+)= op(A[g]) op(B[g]).  A, B, C are
 * HOST arrays of device pointers; ws holds count * splitk * M * N floats.  Used for the weight gradients of a backward pass (the six
 * dW_hh contractions of equal shape run as one launch with a third of the partial sums and no kernel boundary between them).  * If every C[g] is the same pointer the problems are summed into it: C (+)= sum_g A_g B_g. */
int vame_gemm_group_f32(int count, int M, int N, int K, const float* const* A, int64_t lda, int a_kmajor, int64_t a_seg,
                        int64_t a_seg_stride, const float* const* B, int64_t ldb, int b_kmajor, int64_t b_seg, int64_t b_seg_stride,
                        float* const* C, int64_t ldc, int accumulate, int splitk, float* ws, int a_gap_at, int a_gap, void* stream);

/* vame_gemm_group_f32 for two k-major operands (C[g] (+)= A[g]^T B[g]: the weight gradients of vame/model/rnn_vae.py:141-143,
 * loss.backward() -> dW_hh / dW_ih over K = batch x time), evaluated on the bf16 matrix cores with an ERROR-COMPENSATED SPLIT: every
 * fp32 operand value is split exactly into three bf16 planes (x = x1 + x2 + x3) and the six plane products (1,1) (1,2) (2,1) (2,2)
 * (1,3) (3,1) are accumulated in fp32 -- fp32-grade results at 6/16 of the f32-input matrix cores' time.  OPT-IN: the default path of
 * the library stays on true fp32 matrix instructions.  Same arguments and semantics as vame_gemm_group_f32 (both operands k-major);
 * additional requirements: M, N, lda, ldb, the segment strides, a_gap_at and a_gap even, operands 8-byte aligned, a slab of K / splitk
 * rows spans < 1 GiB.  Inf / NaN inputs give NaN.  opt: bits 0-1 = accumulators per output (0 = default 2: the leading product on its
 * own accumulator; 1 = one for all six products); any other bit is refused. */
int vame_gemm_group_bf16x6_f32(int count, int M, int N, int K, const float* const* A, int64_t lda, int64_t a_seg, int64_t a_seg_stride,
                               const float* const* B, int64_t ldb, int64_t b_seg, int64_t b_seg_stride, float* const* C, int64_t ldc,
                               int accumulate, int splitk, float* ws, int a_gap_at, int a_gap, int opt, void* stream);

/* The same split contraction for the two large contractions of a step whose K is a layer width instead of batch x time: a ROW-major A
 * (M = batch x time activation rows with vame_gemm_f32's two-level addressing, K contiguous) times a plain weight matrix B, row-major
 * (b_kmajor = 0: C = A B^T + bias, the input projection of the second encoder layer, vame/model/rnn_model.py:35 nn.GRU(..., num_layers=2)
 * -> gi = y W_ih^T + b) or k-major (b_kmajor = 1: C (+)= A B, its data gradient dY = dG W_ih in loss.backward(), rnn_vae.py:141-143).
 * No split-K.  OPT-IN like the grouped form; same error level (opt bits 0-1 as there).  Requirements: K a multiple of 32; lda, a_seg_stride
 * multiples of 4 and A 16-byte aligned; B row-major: ldb a multiple of 4, 16-byte aligned -- B k-major: ldb and N even, 8-byte aligned;
 * 128 rows of A span < 1 GiB. */
int vame_gemm_bf16x6_f32(int M, int N, int K, const float* A, int64_t lda, int64_t a_seg, int64_t a_seg_stride, const float* B, int64_t ldb,
                         int b_kmajor, const float* bias, float* C, int64_t ldc, int accumulate, int opt, void* stream);

/* Pack one GRU layer-direction's recurrent weights for the sequence kernels.
 *   W_hh (3H,H), b_ih/b_hh (3H) -> wp_fwd (3H*H, MFMA B-fragment order for h W_hh^T),
 *   wp_bwd (3H*H, fragment order for dgh W_hh), bias_gi (3H) = b_ih + [b_hr, b_hz, 0], b_hn (H). */
int vame_gru_pack_f32(const float* W_hh, const float* b_ih, const float* b_hh, int H,
                      float* wp_fwd, float* wp_bwd, float* bias_gi, float* b_hn, void* stream);
/* The same for up to VAME_GRU_PACK_MAX layer-directions in one launch (all GRUs of the model after an optimizer step):
 * items = n x VAME_GRU_PACK_FIELDS int64 on the HOST.  GP_W_IH != 0 also packs that item's layer-0 input projection
 * (vame_gru_pack_x_f32: W_ih (3H,F), F <= 32 -> GP_WPX). */
#define VAME_GRU_PACK_MAX 16
enum vame_gru_pack_field { GP_W_HH = 0, GP_B_IH, GP_B_HH, GP_H, GP_WP_FWD, GP_WP_BWD, GP_BIAS_GI, GP_B_HN, GP_W_IH, GP_F, GP_WPX,
                           VAME_GRU_PACK_FIELDS };
int vame_gru_pack_batch_f32(const int64_t* items, int n, void* stream);


/* GRU sequence forward for up to 8 independent (layer,direction) streams in one launch.
 * desc = nstreams x VAME_GRU_FWD_FIELDS int64 (see vame_gru_fwd_field).  Cell = torch.nn.GRU
 * (gate order r,z,n) as instantiated at vame/model/rnn_model.py:34-35 (encoder), :91-92 (decoder),
 * :125-126 (future decoder); the input projection gi = x W_ih^T + bias_gi is supplied by the caller.
 * Returns bytes of stash needed per stream via vame_gru_stash_floats(). */
enum vame_gru_fwd_field {
    GF_GI = 0, GF_GI_ROW, GF_GI_T,          /* gi (B,T,3H): ptr, row stride, time stride (0 = constant in time) */
    GF_WP, GF_BHN,                          /* packed W_hh (vame_gru_pack_f32), b_hn (H) */
    GF_H0, GF_H0_ROW,                       /* initial state rows (0 = zeros) */
    GF_Y, GF_Y_ROW, GF_Y_T,                 /* output sequence h_t (0 = not written) */
    GF_HN, GF_HN_ROW,                       /* final state (0 = not written) */
    GF_STASH,                               /* r,u,n,gh_n stash for backward (0 = inference) */
    GF_T, GF_REVERSE, GF_PAD,
    GF_WPX, GF_BGI, GF_XF,                  /* fused input projection: packed W_ih (vame_gru_pack_x_f32), bias_gi (3H), F (0 = off);
                                               GF_GI/_ROW/_T then describe x (B,T,F) instead of gi */
    GF_OPT,                                 /* launch options, read from stream 0 only (see VAME_GRU_OPT_*; 0 = defaults) */
    VAME_GRU_FWD_FIELDS
};
/* Launch options of the GRU sequence kernels: a bit field in GF_OPT / GB_OPT of stream 0.  Kernel selection and tuning are
 * ARGUMENTS of a call -- the library reads no process-global state (no environment variables) outside its tuning builds, so two
 * callers in one process cannot influence each other.
 *   bits 0..3   kernel: VAME_GRU_KERNEL_AUTO (the measured default per hidden size), _LOCKSTEP (all waves of a workgroup do the
 *               same thing per phase), _WS (BPTT, wave-specialised: MFMA waves + memory waves), _SKEWED (forward: the two waves of a
 *               SIMD run half a step apart, bit-identical to _LOCKSTEP); a kernel that is not instantiated for H is refused
 *   bits 8..15  pace_cp + 1, bits 16..23  pace_ld + 1: pacing of the wave-specialised kernels' memory waves in units of 256
 *               cycles per request group (0 = the measured default for the hidden size); _SKEWED: pace_cp = priority mode
 *               (0 none, 1 = the wave in its gate-math half raises its priority; default 1) */
enum vame_gru_kernel { VAME_GRU_KERNEL_AUTO = 0, VAME_GRU_KERNEL_LOCKSTEP = 1, VAME_GRU_KERNEL_WS = 2, VAME_GRU_KERNEL_SKEWED = 3 };
#define VAME_GRU_OPT(kernel, pace_cp, pace_ld) \
    ((int64_t)(kernel) | ((int64_t)((pace_cp) < 0 ? 0 : (pace_cp) + 1) << 8) | ((int64_t)((pace_ld) < 0 ? 0 : (pace_ld) + 1) << 16))
int64_t vame_gru_stash_floats(int B, int T, int H);
/* W_ih (3H,F), F <= 32 -> wpx (3H*32): zero-padded K = 32 input projection in MFMA B-fragment order (encoder layer 0:
 * the projection x_t W_ih^T is then computed inside the sequence kernel instead of by vame_gemm_f32). */
int vame_gru_pack_x_f32(const float* W_ih, int F, int H, float* wpx, void* stream);
int vame_gru_seq_fwd_f32(const int64_t* desc, int nstreams, int B, int H, void* stream);
/* 1 if vame_gru_seq_fwd_f32 has `kernel` (enum vame_gru_kernel) for hidden size H */
int vame_gru_seq_fwd_has_kernel(int H, int kernel);

/* GRU sequence backward (BPTT) for the same streams.  Writes dG (B,T,4H) = [da_r|da_z|dgi_n|dgh_n]
 * for the weight-gradient GEMMs, per-tile bias-gradient partials and optional dh0. */
enum vame_gru_bwd_field {
    GB_STASH = 0, GB_Y, GB_Y_ROW, GB_Y_T,   /* forward stash and h sequence */
    GB_H0, GB_H0_ROW,
    GB_WPT,                                 /* packed W_hh (bwd order) */
    GB_DY, GB_DY_ROW, GB_DY_T,              /* grad wrt output sequence (0 = none) */
    GB_DHN, GB_DHN_ROW,                     /* grad wrt final state (0 = none) */
    GB_DG,                                  /* out (B,T,4H) contiguous */
    GB_DH0, GB_DH0_ROW,                     /* out grad wrt initial state (0 = none) */
    GB_DBIAS,                               /* out (ntiles,4H) per-tile column sums of dG over rows and time */
    GB_OPT,                                 /* launch options, read from stream 0 only (VAME_GRU_OPT(...); 0 = defaults) */
    GB_T, GB_REVERSE, GB_PAD,
    VAME_GRU_BWD_FIELDS
};
int vame_gru_seq_bwd_f32(const int64_t* desc, int nstreams, int B, int H, void* stream);
/* 1 if vame_gru_seq_bwd_f32 has `kernel` (enum vame_gru_kernel) for hidden size H; asking for one it has not is VAME_E_UNSUPPORTED */
int vame_gru_seq_bwd_has_kernel(int H, int kernel);

/* Per-step GRU cell for hidden sizes beyond the persistent sequence kernels (H > 256, BASELINE config 4): the gate GEMM
 * gh = h_{t-1} W_hh^T is a vame_gemm_f32 call (a real dense contraction at batch x hidden = 8192 x 512) and these
 * kernels apply the gate math of torch.nn.GRU (vame/model/rnn_model.py:34-35).  gi includes b_ih + [b_hr,b_hz,0];
 * stash row = [cA|cB|u|r|gh_n] (5H); backward: dh in/out (B,H), dG row = [da_r|da_z|dgi_n|dgh_n], dgh = [da_r|da_z|dgh_n]. */
int vame_gru_cell_fwd_f32(const float* gi, int64_t gi_row, const float* gh, const float* bhn, const float* hprev, int64_t hp_row,
                          float* hout, int64_t ho_row, float* stash, int64_t st_row, int B, int H, void* stream);
int vame_gru_cell_bwd_f32(const float* stash, int64_t st_row, float* dh, const float* dy, int64_t dy_row, float* dG,
                          int64_t dg_row, float* dgh, int B, int H, void* stream);

/* Lambda reparameterisation + KL partials (vame/model/rnn_model.py:63-76, vame/model/rnn_vae.py:53-60).
 *   mu, lv_raw (B,Z) -> logvar (softplus optional), z = eps*exp(0.5*logvar)+mu (training) or mu;
 *   kl_out[0] += sum(1 + logvar - mu^2 - exp(logvar))   (the caller's buffer is zero before the step: vame_loss_finish_f32 leaves it so).
 *   rng == NULL: eps (B,Z) is an INPUT (parity tests inject the reference's draw).  rng != NULL (device, 4 x uint64: {seed, step, ticket = 0,
 *   unused}) in training mode: eps is an OUTPUT -- N(0,1) from Philox4x32-10 keyed by seed, counter = (element, step), Box-Muller -- and the
 *   launch advances `step` by one on the device (the reference's `torch.randn_like`, rnn_model.py:71-74, with no host op and no argument
 *   that changes between launches). */
int vame_latent_fwd_f32(const float* mu, const float* lv_raw, float* eps, int B, int Z, int softplus,
                        int training, float* logvar, float* z, float* kl_out, uint64_t* rng, void* stream);
/* Loss bookkeeping of one step (vame/model/rnn_vae.py:129-150) in one launch: raw (device, 8 floats: the sums left by the loss kernels in slots
 * 0..3 = rec, fut, KL sum, kmeans) -> out (device, 5 floats) = the four terms x scale[i] (fut = 0 unless with_fut) and their weighted total
 * sum_i weights[i] term[i]; acc (device, 6 doubles, may be NULL): += total, rec, fut, kl, kmeans; acc[5] = this step's total.  raw is
 * zeroed.  scale / weights: HOST arrays of 4 floats. */
int vame_loss_finish_f32(float* raw, const float* scale, const float* weights, int with_fut, float* out, double* acc, void* stream);
int vame_latent_bwd_f32(const float* dz, const float* mu, const float* logvar, const float* lv_raw,
                        const float* eps, int B, int Z, int softplus, float ckl, float* dmu, float* dlv, void* stream);

/* MSE (vame/model/rnn_vae.py:35-43): loss_out[0] += sum((pred-target)^2) ; dpred = gscale*(pred-target).
 * target rows are (T,F) windows inside a (B, tgt_row) buffer starting at column offset 0. */
int vame_mse_fwd_bwd_f32(const float* pred, const float* target, int64_t tgt_row, int B, int TF,
                         float gscale, float* dpred, float* loss_out, void* stream);

/* Output head of a decoder in the training step as ONE pass over the decoder's states (vame/model/rnn_model.py:107-108,139-140
 * hidden_to_output; rnn_vae.py:35-43 MSE with reduction "sum"; rnn_vae.py:141-143 loss.backward() through both): for the M = B*T rows
 * m = (b,t), b = m / y_seg,
 *   y_m = Y + b*y_seg_stride + t*y_ld (K floats);  pred[m,f] = y_m . W[f,:] + bias[f];
 *   e = pred - tgt[b*tgt_row + tgt_off + t*F + f];  loss[0] += sum e^2;  dpred[m,f] = gscale*e;
 *   dY[m*dy_ld + n] = sum_f dpred[m,f] W[f,n], n < K;   dW[f,n] = sum_m dpred[m,f] y_m[n]  (overwritten; deterministic order).
 * pred may be null.  Replaces vame_gemm_f32 (N = F) + vame_mse_fwd_bwd_f32 + vame_gemm_f32 (K = F) + the split-K weight-gradient
 * vame_gemm_f32 (M = F, K = B*T) and its reduction on the training path; the stand-alone decoder calls keep the GEMM.
 * Shapes: 1 <= F <= 32, K a multiple of 64 up to 512 or a multiple of 128 up to 1024; state rows, W and dY rows 16-byte aligned.  ws: vame_head_stream_ws_floats(M, F, K)
 * floats of scratch (per-workgroup dW sums); that function returns -1 for a shape the kernel does not cover. */
int64_t vame_head_stream_ws_floats(int M, int F, int K);
int vame_head_stream_f32(const float* Y, int64_t y_ld, int64_t y_seg, int64_t y_seg_stride, int M, int F, int K, const float* W,
                         const float* bias, const float* tgt, int64_t tgt_row, int64_t tgt_off, float gscale, float* pred,
                         float* dpred, float* dY, int64_t dy_ld, float* loss, float* dW, float* ws, void* stream);

/* cluster_loss (vame/model/rnn_vae.py:45-50) from the (Z,Z) Gram G = z^T z (un-normalised, from vame_gemm_f32):
 *   loss_out[0] = lmbda * sum_{i<k} sqrt(eig_i(G/bsize)),  Minv (Z,Z) = gscale*(lmbda/bsize) V_k S_k^-1 V_k^T
 * so that gscale * d loss/dz = z Minv (gscale = the KL-annealing weight).  k is clipped to
 * min(kloss, Z, nrows) like sv_2[:kloss] of the reference's (B,B) SVD.  One workgroup, parallel
 * cyclic Jacobi in fp64.  vstate (optional, ((Z+1)&~1)^2 doubles, zeroed before first use) carries the
 * eigenvectors from call to call as a warm start (1-2 sweeps instead of 6-8 during training).
 * 64 < Z <= 512: the matrices leave the LDS for the state buffer, which is then REQUIRED and 3 Z'^2 doubles
 * (vame_nuclear_state_doubles; same zero-on-first-use rule); one 1024-thread workgroup, milliseconds instead of ~0.1 ms. */
int64_t vame_nuclear_state_doubles(int Z);
int vame_nuclear_f32(const float* G, int Z, int kloss, int nrows, float lmbda, float bsize, float gscale,
                     float* loss_out, float* Minv, double* vstate, void* stream);

/* k-means E-step over the latent vectors (vame/analysis/pose_segmentation.py:141,179 run sklearn KMeans on
 * the embedding): labels[i] = argmin_k |X_i - C_k|^2 (ties -> lowest k), mind2[i] = that squared distance (optional),
 * onehot (N,Kp) rows (optional) so the M-step sums = onehot^T X is a vame_gemm_f32 call. */
int vame_kmeans_assign_f32(const float* X, int64_t N, int D, const float* C, int K, int* labels, float* mind2,
                           float* onehot, int Kp, void* stream);

/* out[b,c] = sum_t in[(b*T+t)*ld + c], c < C (C % 4 == 0): sum over time of a (B,T,ld) sequence -- the
 * gradient wrt the time-constant decoder input z (vame/model/rnn_model.py:169-170). */
int vame_timesum_f32(const float* in, int B, int T, int C, int64_t ld, float* out, void* stream);

/* out[c] (+)= sum_r in[r*ld + c]  (bias gradients: column sums of dG / dpred / dmu).  Deterministic
 * two-pass reduction; ws must hold vame_colsum_ws_floats(R, C) floats. */
int64_t vame_colsum_ws_floats(int64_t R, int C);
int vame_colsum_f32(const float* in, int64_t R, int C, int64_t ld, float* out, int accumulate, float* ws, void* stream);
/* njobs (<= 32) independent column sums out_i[c] = sum_r in_i[r*ld_i + c] in one launch; desc = njobs x {in, R, C, ld, out}. */
int vame_colsum_batch_f32(const int64_t* desc, int njobs, void* stream);

/* Fused Adam with AMSGrad over a flat parameter buffer (torch.optim.Adam(amsgrad=True), rnn_vae.py:332,143).
 * gscale multiplies the gradient first (1/world_size after an all-reduce SUM).  abort_flag (optional device word): when it is
 * non-zero at execution time the launch changes nothing -- the status word of the cooperative GRU launches goes here, so a step
 * whose gradients are undefined never reaches the weights (the host raises when it next reads the word).  Any non-zero 32-bit
 * pattern aborts, so the word may also be a float that an all-reduce SUM left > 0 (the multi-rank case: one rank's failure drops
 * the step on every rank).  dropped (optional device counter) is incremented by each launch that was aborted, so the host can
 * keep its bias-correction step count equal to the number of updates actually applied.
 * state (optional, device, 4 x int32 = {learning rate as a float, updates applied so far, 0, unused}): when given, `lr` and `step` are ignored --
 * the launch reads both from the device, and counts itself there unless it was aborted: no argument changes from step to step (a captured
 * hipGraph replays the launch; an LR scheduler writes state[0]). */
int vame_adam_amsgrad_f32(float* p, const float* g, float* m, float* v, float* vmax, int64_t n, float lr,
                          float beta1, float beta2, float eps, int step, float gscale, const int* abort_flag, int* dropped,
                          int* state, void* stream);

/* dst[dst_idx[i]] = src[src_idx[i]], i < n (element indices, no duplicates in dst_idx).  torch.nn.GRU accepts any hidden_size
 * (rnn_model.py:34,91,125) while the GRU kernels tile hidden units by 32: for other sizes the model keeps its parameters in the
 * reference's shapes (state_dict / optimizer) and this kernel copies them into a zero-padded image the kernels run on -- padded
 * units stay exactly 0 through the GRU recurrence -- and the padded gradients back (vame_amd/padding.py). */
int vame_index_copy_f32(float* dst, const int64_t* dst_idx, const float* src, const int64_t* src_idx, int64_t n, void* stream);

/* Encoder inter-layer dropout (torch.nn.GRU(dropout=p) at rnn_model.py:34-35, training only): out[r][c] = x[row(r)][c] * mask[r][c] * scale
 * over R x C, mask in {0,1}, scale = 1/(1-p).  x rows: seg = 0 -> r*ld + off, else (r/seg)*seg_stride + (r%seg)*ld + off (the padded
 * sequence layout); mask / out dense (R, C); C % 4 == 0; in-place on a dense x is allowed (the backward of the same op). */
int vame_mask_scale_f32(const float* x, int64_t off, int64_t ld, int64_t seg, int64_t seg_stride, const float* mask, float scale,
                        float* out, int64_t R, int C, void* stream);

/* `count` (1..8) Linear layers of ONE narrow input in one launch: C[g] (M, N[g]; row stride ldc[g]) = A (M, K; row stride lda) W[g]^T + bias[g],
 * W[g] (N[g], K) row-major contiguous, bias[g] (N[g]) or null, K <= 32.  W, bias, C, ldc, N are HOST arrays.  Replaces the six vame_gemm_f32
 * launches that apply the decoders' latent_to_hidden layers and GRU input weights to the time-constant input z
 * (vame/model/rnn_model.py:103-106, 136-140; nn.GRU's W_ih x_t with x_t = z for every t, rnn_model.py:169-170). */
int vame_linear_group_f32(int count, int M, int K, const float* A, int64_t lda, const float* const* W, const float* const* bias,
                          float* const* C, const int64_t* ldc, const int* N, void* stream);

/* y = a*x + y style helpers for the host orchestration */
int vame_axpy_f32(const float* x, float a, float* y, int64_t n, void* stream);

/* ---- column-split GRU forward for small batches (vame_amd/csrc/gru_coop.hip): same descriptor table, stash and sequence layout as
 * vame_gru_seq_fwd_f32, results equal to summation-order rounding (K = H is summed in two halves of 16 x 16 x 4 MFMAs; the same bits
 * for every form a launch can take).  A 32-row tile -- or each 16-row half of it, when twice the workgroups still get a CU each -- is
 * shared by H/32 workgroups that keep their slice of W_hh in LDS and hand their 32 columns of h_t to each other every step as
 * self-validating (value, tag) pairs (needs GF_Y, a precomputed gi, H = 128 or 256, and a grid that fits one workgroup per CU:
 * vame_gru_coop_supported).  GF_OPT / GB_OPT kernel: AUTO, or LOCKSTEP = 32-row groups even where 16-row groups fit.
 * flags: flag_ints >= vame_gru_coop_flag_ints() ints (flag words + the hand-off packets; the launch checks the size), initialised ONCE by the
 * caller to a value OLDER than the first epoch (e.g. epoch - 8: equal to no tag a launch will look for) and then only passed back.
 * epoch: device int[2] = {launch epoch, 0}, shared by every launch that shares `flags`: a launch reads its tag base from epoch[0] and its last
 * workgroup advances it by (steps + 2), so consecutive launches -- also replays of a captured hipGraph, whose arguments are frozen -- never see
 * each other's words.  *status is incremented if a bounded poll ever expires (results are then undefined, the launch still terminates). */
int64_t vame_gru_coop_flag_ints(int nstreams, int B, int H);
int vame_gru_coop_supported(int nstreams, int B, int H);   /* grid <= CUs and the runtime's occupancy query admits each kernel */
/* Poll budget of one hand-off wait (0 = default, about 0.3 s on the device); returns the previous value.  Process-wide;
 * for diagnostics.  polls < 0 = fault injection: every cooperative launch reports one timeout through *status although its
 * hand-offs complete (tests of the failure path: optimizer step dropped on the device, host exception). */
int vame_gru_coop_set_poll_limit(int polls);
int vame_gru_coop_fwd_f32(const int64_t* desc, int nstreams, int B, int H, int row0, int nrows, int* flags, int64_t flag_ints, int* epoch,
                          int* status, void* stream);   /* rows [row0, row0+nrows) of the batch, row0 % 32 == 0; nrows = 0: all rows */
/* BPTT counterpart (contract of vame_gru_seq_bwd_f32; results equal up to the summation order of the K = 3H contraction, which is
 * split by member; the same bits in 32- and 16-row groups): xbuf = vame_gru_coop_xbuf_floats() floats of scratch for the per-step
 * reduce-scatter of the dh partials (and, at the end of a launch in 16-row groups, the hand-over of the upper group's bias sums). */
int64_t vame_gru_coop_xbuf_floats(int nstreams, int B, int H);
int vame_gru_coop_bwd_f32(const int64_t* desc, int nstreams, int B, int H, int row0, int nrows, float* xbuf, int* flags, int64_t flag_ints,
                          int* epoch, int* status, void* stream);

/* ---- wide hidden sizes, 256 < H <= 512 (H % 64 == 0; vame_amd/csrc/gru_wide.hip): batch-tile-persistent forward with two 32-column
 * blocks per wave; descriptor table, sequence layout and fragment-order stash (NB = H/32 blocks) as vame_gru_seq_fwd_f32, no fused input
 * projection.  vame_gru_cell_bwd_frag_f32: one BPTT step's gate math (vame_gru_cell_bwd_f32) reading that fragment-order stash. */
int vame_gru_wide_supported(int H);
int vame_gru_wide_fwd_f32(const int64_t* desc, int nstreams, int B, int H, void* stream);
int vame_gru_wide_bwd_f32(const int64_t* desc, int nstreams, int B, int H, void* stream);   /* contract of vame_gru_seq_bwd_f32 */
int vame_gru_cell_bwd_frag_f32(const float* stash, int T, int t, float* dh, const float* dy, int64_t dy_row, float* dG, int64_t dg_row,
                               float* dgh, int B, int H, void* stream);

/* ---- Gaussian HMM over the latents (reference pose_segmentation.py:145-158 = hmmlearn GaussianHMM(covariance_type=
 * "full").fit / .predict; vame_amd/csrc/hmm.hip).  float64; X (N, D) float32 latents; K <= 32 states, D <= 64; L = frames per chunk of the
 * chunk-parallel recursions.  One EM iteration = emission -> forward -> backward -> stats; the M-step (K small matrices) is the host's.
 *   emission: logB[t,k] = logconst[k] - 0.5 |Linv_k (x_t - mean_k)|^2 (Linv_k = inverse lower Cholesky factor of Sigma_k, row-major),
 *             bexp = exp(logB - rowmax), rowmax[t] = max_k logB[t,k]
 *   forward : alpha-hat (N,K) and the per-frame normalisers cnorm (log-likelihood = sum_t log cnorm[t] + rowmax[t])
 *   backward: gamma (N,K) smoothed posteriors and R (N,K) with xi_t[i][j] = alpha_t[i] A[i][j] R_t[j]
 *   stats   : [post K | start K | sum_t alpha_t^T R_t (K*K, multiply by A for the expected transitions) | obs K*D | loglik 1 |
 *             sum_t gamma[t,k] x_t x_t^T (K*D*D)]; deterministic (fixed-order partial sums)
 *   viterbi : most likely state path (int32) and its log probability from logB and log start / transition probabilities */
int vame_hmm_emission_f64(const float* X, int64_t N, int D, const double* mean, const double* linv, const double* logconst, int K,
                          double* logB, double* bexp, double* rowmax, void* stream);
int64_t vame_hmm_ws_doubles(int64_t N, int K, int L);
int vame_hmm_forward_f64(const double* bexp, int64_t N, int K, const double* startprob, const double* transmat, int L, double* alpha,
                         double* cnorm, double* ws, void* stream);
int vame_hmm_backward_f64(const double* bexp, int64_t N, int K, const double* transmat, const double* alpha, int L, double* gamma, double* R,
                          double* ws, void* stream);
int64_t vame_hmm_stats_doubles(int K, int D);
int64_t vame_hmm_stats_ws_doubles(int K, int D);
int vame_hmm_stats_f64(const float* X, int64_t N, int D, int K, const double* alpha, const double* gamma, const double* R, const double* cnorm,
                       const double* rowmax, double* stats, double* ws, void* stream);
int64_t vame_hmm_viterbi_ws_bytes(int64_t N, int K, int L);
int vame_hmm_viterbi_f64(const double* logB, int64_t N, int K, const double* log_startprob, const double* log_transmat, int L, int* path,
                         double* logprob, double* ws, unsigned char* bws, void* stream);

/* ---- training-set preparation: the O(N*F) float64 passes of vame/model/create_training.py.
 * Arrays are (F, N) feature-major with a leading dimension (elements), like <file>-PE-seq.npy; results are bit-identical to
 * the reference's numpy / scipy arithmetic.  mean, sd, cutoff come from the host (np.mean / np.std / iqr_factor * scipy iqr). */

/* bytes of caller-provided scratch `ws` for the two reductions below (index pairs / partial sums per feature) */
int64_t vame_prep_ws_bytes(int F);
/* z[f, n] = (x[f, n] - mean) / sd; robust != 0: z > cutoff or z < -cutoff -> NaN   (create_training.py:112-143, 217-234) */
int vame_prep_zscore_mask_f64(const double* x, int F, int64_t N, int64_t ldx, double mean, double sd, double cutoff,
                              int robust, double* z, int64_t ldz, void* stream);
/* interpol() as the reference applies it to a whole aligned file (create_training.py:27-32,145): every NaN of feature f becomes
 * the last valid sample (in time) of feature f.  first_last[2f], [2f+1] = first / last valid value of feature f (NaN if none:
 * such features are left untouched for the caller to resolve). */
int vame_prep_fill_last_valid_f64(double* z, int F, int64_t N, int64_t ld, double* first_last, void* ws, void* stream);
/* interpol() per frame of a fixed (egocentric) file (create_training.py:236): np.interp across the feature index.
 * *n_empty += number of frames without any valid feature (left as NaN). */
int vame_prep_fill_across_features_f64(double* z, int F, int64_t N, int64_t ld, int* n_empty, void* stream);
/* np.mean / np.std over time per feature (create_training.py:153), deterministic two-pass. */
int vame_prep_rowstats_f64(const double* x, int F, int64_t N, int64_t ld, double* mean_out, double* std_out, void* ws, void* stream);
/* scipy.signal.savgol_filter along time, interior samples (create_training.py:181,245): w = reversed savgol_coeffs (L odd);
 * the first / last L/2 columns are copied through (mode='interp' refits them from the edge windows). */
int vame_prep_savgol_f64(const double* x, int F, int64_t N, int64_t ldx, const double* w, int L, double* y, int64_t ldy,
                         void* stream);

#ifdef __cplusplus
}
#endif
#endif
