/*
 * coma_unet.h -- C ABI of the MI355X (gfx950) CoMA-UNet training hot path.
 *
 * Drop-in boundary.  The reference (mborhi/CoMA-UNet) is pure Python: its hot
 * path is the nn.Module / criterion surface of
 *   attn_unet_data_parallel.py:503-693  (ContrastiveAttentionUNET_DP)
 *   attn_unet_data_parallel.py:779-912  (train_dp step body)
 *   criterions.py:124-211,485-644       (RoiMSE, GenerativeContrastiveLoss, RnC)
 * and it has no FFI of its own; every conv / norm / activation there is a
 * cuDNN/ATen kernel that PyTorch dispatches.  This library is what replaces
 * those dispatches: one entry point per row of SURVEY.md section 2.1.  The
 * Python host (coma_unet_amd/) mirrors the reference's module surface and calls
 * these functions through ctypes with raw device pointers.
 *
 * Conventions
 *  - All pointers are DEVICE pointers unless the name ends in _host.
 *  - Activations are channels-last volumes "NDHWC": element (b, z, y, x, c) of
 *    a coma_tensor t lives at  data + b*sb + ((z*H + y)*W + x)*ld + c  (in
 *    elements of t.dtype).  ld >= C lets a tensor be a channel slice of a wider
 *    buffer, which is how torch.cat((att, fromlower), dim=1)
 *    (attn_unet_data_parallel.py:229,651,654) is done without a copy.
 *  - Parameters, statistics and weight gradients are fp32.
 *  - No function allocates, synchronises or keeps global state.  `stream` is a
 *    hipStream_t passed as void*.  Return value 0 = ok, otherwise an error code;
 *    coma_last_error() returns a thread-local message.
 *  - Workspaces are caller-provided; the *_ws_bytes query says how much.
 */
#ifndef COMA_UNET_H
#define COMA_UNET_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 2): coma_conv_pick_algo / coma_conv_wgrad_algo may answer 3 (fp32 MFMA), algo 0 routes fp32 tensors to the
 * fp32 MFMA kernels and one-channel 1x1x1 layers to conv_point1 (fp32 kernel-layout weights) before the bf16 MFMA path;
 * new exports coma_last_kernel, coma_comm_*, coma_allreduce/reduce_scatter/allgather/broadcast.
 * 3 (round 3): normalisation statistics travel as caller-zeroed fp64 records (coma_norm_stats / coma_norm_act_fwd /
 * coma_norm_act_bwd / coma_conv_fwd_norm_stats: no mean / rstd tensors, no finalise launches); `zeroed` arguments on
 * coma_conv_fwd_ws / coma_conv_fwd_norm_stats / coma_conv_wgrad / coma_weight_prep_bwd. */
#define COMA_ABI_VERSION 3

enum { COMA_F32 = 0, COMA_BF16 = 1 };

/* `zeroed` argument of the entry points that merge partial results with atomics (weight gradients, split-K partial
 * tiles, routing gradients): what the caller guarantees to be all zeros on entry -- the callee then skips its own
 * memset.  A training step keeps ONE zeroed arena and hands out slices of it (~60 memset launches per step gone).
 * COMA_ZEROED_OUT: the output the call accumulates into (dwk / dr); COMA_ZEROED_WS: the first *_ws_bytes() bytes of
 * `ws`, which must then be private to this call (the callee leaves them dirty).  0 = the callee zeroes what it needs. */
enum { COMA_ZEROED_OUT = 1, COMA_ZEROED_WS = 2,
       /* coma_conv_fwd_ws only: y += conv(x) instead of y = conv(x) -- a data gradient added to the gradient another
        * consumer of the same activation has already written (no separate accumulation pass); valid where
        * coma_conv_accumulate_ok() answers 1 (the gather / pointwise kernel families)                                */
       COMA_ACCUMULATE = 4 };

/* activation after a normalisation (MONAI ADN "A" slot) */
enum {
  COMA_ACT_NONE = 0,
  COMA_ACT_RELU = 1,        /* attentionunet.ConvBlock / UpConv                         */
  COMA_ACT_PRELU = 2,       /* MONAI Convolution default, one shared slope             */
  COMA_ACT_LEAKY = 3,       /* StackedFusionConvLayers, attn_unet_data_parallel.py:487 */
  COMA_ACT_SIGMOID = 4,     /* AttentionBlock.psi                                       */
  COMA_ACT_PRELU_RELU = 5   /* final_pred_head PReLU followed by final_act ReLU (:654-656) */
};

enum { COMA_NORM_BATCH = 0, COMA_NORM_INSTANCE = 1 };

typedef struct coma_tensor {
  void*   data;
  int32_t dtype;          /* COMA_F32 | COMA_BF16 */
  int32_t B, D, H, W, C;
  int64_t ld;             /* elements between consecutive voxels (>= C)   */
  int64_t sb;             /* elements between consecutive samples          */
} coma_tensor;

/* Gather form of a (transposed) convolution with a cubic kernel.
 *   form 0 ("conv"):   in = out*stride - pad + tap       nn.Conv3d forward,
 *                                                         ConvTranspose3d data-gradient
 *   form 1 ("tconv"):  in = (out + pad - tap)/stride     nn.ConvTranspose3d forward,
 *                       (only when divisible)             Conv3d data-gradient
 * Kernel-layout weights are wk[b][tap][n][c] (c fastest), tap = (kz*k + ky)*k + kx. */
typedef struct coma_conv_desc {
  int32_t ksize;          /* 1 or 3 */
  int32_t stride;         /* 1 or 2 */
  int32_t pad;            /* 0 or 1 */
  int32_t form;           /* 0 conv, 1 tconv */
  int32_t per_sample_w;   /* 1: wk/bias have a leading B dim (CondConv) */
  int32_t algo;           /* 0 auto: MFMA wherever the shape allows (bf16 tensors: v_mfma_f32_32x32x16_bf16; fp32 tensors:
                             v_mfma_f32_32x32x2_f32, exact fp32), otherwise the direct kernels;
                             1 force the direct VALU fp32 kernels; 2 = 0 (kept for callers of ABI 1) */
} coma_conv_desc;

int         coma_abi_version(void);
const char* coma_last_error(void);
/* name of the convolution kernel variant the calling thread's most recent coma_conv_* call launched, spelled as
 * rocprofv3 prints it (e.g. "conv_mfma_halo2_k<2, 32, 1, 1>"): lets a host-side timer attribute its per-launch HIP-event
 * durations to the kernel names of a rocprofv3 --kernel-trace of the same command. */
const char* coma_last_kernel(void);

/* ---- CondConv expert mixing + weight re-layout  (replaces CondConv.CondConvolution's
 *      per-sample kernel synthesis, call sites attn_unet_data_parallel.py:126,285-306) ----
 * master: fp32 [E][..] with element (e, n, c, tap) at e*se + n*sn + c*sc + tap
 * r: fp32 [Bw][E] routing weights, or NULL (=> Bw = 1, E = 1, plain re-layout/cast)
 * out: [Bw][taps][N][C] in out_dtype                                              */
int coma_weight_prep(const float* master, const float* r, int32_t E, int32_t Bw,
                     int32_t N, int32_t C, int32_t taps, int64_t se, int64_t sn, int64_t sc,
                     void* out, int32_t out_dtype, void* stream);
/* 27-tap masters, both kernel layouts from one pass over the experts: master [E][A][B][27]
 * -> out_ab [Bw][27][A][B] and/or out_ba [Bw][27][B][A] (either may be NULL), each fp32 or bf16. */
int coma_weight_prep_pair(const float* master, const float* r, int32_t E, int32_t Bw, int32_t A,
                          int32_t B, void* out_ab, int32_t dtype_ab, void* out_ba, int32_t dtype_ba,
                          void* stream);
/* dwk: fp32 [Bw][taps][N][C]  ->  dmaster (=, fp32, master layout), dr [Bw][E] (=, fp32; accumulated with
 * atomics: zeroed here unless `zeroed` has COMA_ZEROED_OUT, i.e. the caller hands over a dr that is already zero) */
int coma_weight_prep_bwd(const float* dwk, const float* master, const float* r, int32_t E,
                         int32_t Bw, int32_t N, int32_t C, int32_t taps, int64_t se,
                         int64_t sn, int64_t sc, float* dmaster, float* dr, int32_t zeroed, void* stream);

/* ---- CondConv routing (call sites attn_unet_data_parallel.py:285-306):
 *      r[b][e] = sigmoid(cov[b] . Wr[e] + br[e]);  bias_mix[b][n] = sum_e r[b][e] * bias_e[e][n]
 * cov fp32 [B][NC], Wr [E][NC], br [E], bias_e [E][N] (or NULL with bias_mix NULL); B*E <= 64.   */
int coma_routing_fwd(const float* cov, int32_t B, int32_t NC, const float* Wr, const float* br,
                     int32_t E, const float* bias_e, int32_t N, float* r, float* bias_mix,
                     void* stream);
/* dr_w [B][E] (gradient reaching r through the mixed weights, or NULL), dbias_mix [B][N] (or
 * NULL)  ->  dWr [E][NC], dbr [E], dbias_e [E][N] (or NULL); all written with "=".             */
int coma_routing_bwd(const float* cov, int32_t B, int32_t NC, const float* r, int32_t E,
                     const float* bias_e, int32_t N, const float* dr_w, const float* dbias_mix,
                     float* dWr, float* dbr, float* dbias_e, void* stream);

/* ---- convolution (nn.Conv3d / nn.ConvTranspose3d and their data-gradients) ---- */
/* which kernel family algo==0 resolves to for this problem: 1 direct (wants fp32 wk), 2 bf16 MFMA
 * (wants bf16 wk), 3 fp32 MFMA (fp32 tensors, wants fp32 wk).  The host prepares the kernel-layout
 * weights accordingly.  coma_conv_wgrad_algo answers the same question for the weight gradient.   */
int coma_conv_pick_algo(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* y);
int coma_conv_fwd(const coma_conv_desc* d, const coma_tensor* x, const void* wk,
                  int32_t wk_dtype, const float* bias, const coma_tensor* y, void* stream);
/* coma_conv_fwd with a scratch buffer: layers with fewer output tiles than CUs (8^3 / 16^3 grids, 256-512 channels)
 * split their K loop (taps x channel chunks) over more blocks, merge fp32 partials in `ws` and convert once.
 * ws_bytes >= coma_conv_fwd_ws_bytes(...) enables it; a smaller or NULL ws runs the unsplit kernel.          */
size_t coma_conv_fwd_ws_bytes(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* y);
int coma_conv_accumulate_ok(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* y);
int coma_conv_fwd_ws(const coma_conv_desc* d, const coma_tensor* x, const void* wk, int32_t wk_dtype,
                     const float* bias, const coma_tensor* y, void* ws, size_t ws_bytes, int32_t zeroed, void* stream);
/* conv forward + statistics of the BatchNorm(train)/InstanceNorm that follows it (MONAI Convolution =
 * conv -> ADN): the conv epilogue ADDS its {sum, sumsq} to `sums` where the kernel supports it, so the conv
 * output is not re-read; otherwise coma_norm_stats runs behind the conv.  `sums`: see coma_norm_stats.       */
int coma_conv_fwd_norm_stats(const coma_conv_desc* d, const coma_tensor* x, const void* wk, int32_t wk_dtype,
                             const float* bias, const coma_tensor* y, int32_t mode, double* sums,
                             void* ws, size_t ws_bytes, int32_t zeroed, void* stream);
/* dwk[b][tap][n][c] (=) sum_m dy[m][n] * x[pos(m,tap)][c]; fp32; batch-summed when
 * !per_sample_w.  dbias[b][n] (=) sum_m dy[m][n] (may be NULL).                     */
int coma_conv_wgrad_algo(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* dy);
size_t coma_conv_wgrad_ws_bytes(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* dy);
/* the part of it that must be ZERO for the kernels' atomic merges (0 = none): with COMA_ZEROED_WS and dbias == NULL a
 * caller may pass a private pre-zeroed buffer of just this size as `ws`                                          */
size_t coma_conv_wgrad_zs_bytes(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* dy);
int coma_conv_wgrad(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* dy,
                    float* dwk, float* dbias, void* ws, size_t ws_bytes, int32_t zeroed, void* stream);

/* ---- BatchNorm3d (train) / InstanceNorm3d + activation  (MONAI ADN "N","A") ---- */
size_t coma_norm_ws_bytes(const coma_tensor* x);   /* scratch of coma_spatial_mean / the conv bias gradient */
/* Training statistics are fp64 records sums[COMA_STAT_REPLICAS][G][C][2] = {sum x, sum x^2}, G = 1 (batch) or B
 * (instance), replica stride COMA_NORM_RECORD_DOUBLES(G, C, 2), that the CALLER hands over ZEROED and the producers add
 * to with fp64 atomics (a block picks a replica by its index: short queues at the memory-side atomic units); every
 * consumer below sums the replicas and derives mean = sum / R and rstd = 1 / sqrt(sumsq / R - mean^2 + eps) itself
 * (R = voxels per group): there is no finalise launch and no (mean, rstd) tensor on the training path.             */
#define COMA_STAT_REPLICAS 8
/* doubles in ONE replica of a record with k values per (group, channel), rounded up to a 64-byte line */
#define COMA_NORM_RECORD_DOUBLES(G, C, k) ((((int64_t)(G) * (C) * (k)) + 7) & ~(int64_t)7)
int coma_norm_stats(const coma_tensor* x, int32_t mode, double* sums, void* stream);
/* y = act((x - mean)*rstd*gamma + beta); gamma/beta may be NULL; slope: device fp32[1].  Statistics: `sums`
 * (training) or, when sums == NULL, fp32 mean/rstd [G][C] (eval-mode BatchNorm: running statistics).
 * running_mean/var (BatchNorm training, may be NULL): r = (1-momentum)*r + momentum*stat, unbiased variance.  */
int coma_norm_act_fwd(const coma_tensor* x, int32_t mode, const double* sums, float eps, const float* mean,
                      const float* rstd, const float* gamma, const float* beta, int32_t act, const float* slope,
                      float* running_mean, float* running_var, float momentum, const coma_tensor* y, void* stream);
/* dx (=); dgamma/dbeta [C] (=); dslope [1] (=); any of the three may be NULL.  bsums: fp64
 * [COMA_STAT_REPLICAS][G][C][3] (replica stride COMA_NORM_RECORD_DOUBLES(G, C, 3)), ZEROED by the caller: the
 * backward's own partial sums {sum dz, sum dz*xhat, sum dy*dact/dslope}.                                         */
int coma_norm_act_bwd(const coma_tensor* x, const coma_tensor* dy, int32_t mode, const double* sums, float eps,
                      const float* gamma, const float* beta, int32_t act, const float* slope,
                      const coma_tensor* dx, float* dgamma, float* dbeta, float* dslope, double* bsums, void* stream);

/* ---- attention gate pieces (MONAI AttentionBlock, attn_unet_data_parallel.py:139-150) ---- */
/* out = relu(a + b);  da (=) db (=) dout * [out > 0] */
int coma_add_relu_fwd(const coma_tensor* a, const coma_tensor* b, const coma_tensor* out, void* stream);
int coma_add_relu_bwd(const coma_tensor* out, const coma_tensor* dout, const coma_tensor* da, void* stream);
/* out[v][c] = x[v][c] * psi[v]   (psi has C == 1) */
int coma_gate_mul_fwd(const coma_tensor* x, const coma_tensor* psi, const coma_tensor* out, void* stream);
int coma_gate_mul_bwd(const coma_tensor* x, const coma_tensor* psi, const coma_tensor* dout,
                      const coma_tensor* dx, int32_t accumulate_dx, const coma_tensor* dpsi, void* stream);

/* ---- the attention gate behind its W_g / W_x convolutions, fused (csrc/gate.hip; attn_unet_data_parallel.py:139-150):
 *      s = relu(BN_g(g1raw) + BN_x(x1raw));  psi_raw = w_psi . s + b_psi;  psi = sigmoid(BN_psi(psi_raw));  att = x * psi
 * All three BatchNorms in training mode on statistics records (coma_norm_stats; G = 1): sums_g / sums_x come out of the
 * W_g / W_x convolutions (coma_conv_fwd_norm_stats), sums_psi is a ZEROED record this call fills.  s and psi_raw are
 * written for the backward.  running_* may be NULL.                                                                    */
int coma_gate_mid_fwd(const coma_tensor* g1raw, const coma_tensor* x1raw, const double* sums_g, float eps_g,
                      const float* gamma_g, const float* beta_g, const double* sums_x, float eps_x,
                      const float* gamma_x, const float* beta_x, const float* w_psi, const float* b_psi,
                      float* rmean_g, float* rvar_g, float* rmean_x, float* rvar_x, float momentum,
                      const coma_tensor* s_out, const coma_tensor* psi_raw, double* sums_psi, void* stream);
/* psi = sigmoid(BN_psi(psi_raw)) (written: [B][V][1]) and att = x * psi (e.g. a channel slice of the concat buffer) */
int coma_gate_apply_fwd(const coma_tensor* x, const coma_tensor* psi_raw, const double* sums_psi, float eps,
                        const float* gamma_psi, const float* beta_psi, float* rmean_psi, float* rvar_psi,
                        float momentum, const coma_tensor* psi, const coma_tensor* att, void* stream);
/* dx (= or +=) d(att) * psi;  dz[v] = (sum_c d(att) x) psi (1 - psi);  bsums_psi: ZEROED record, k = 3            */
int coma_gate_apply_bwd(const coma_tensor* x, const coma_tensor* psi, const coma_tensor* psi_raw,
                        const coma_tensor* dout, const double* sums_psi, float eps, const float* gamma_psi,
                        const float* beta_psi, const coma_tensor* dx, int32_t accumulate_dx,
                        const coma_tensor* dz, double* bsums_psi, void* stream);
/* dz -> d(psi_raw) -> ds -> relu mask -> both BatchNorm backwards: dg1raw, dx1raw (=) and every parameter gradient (=;
 * any may be NULL).  rec: ZEROED scratch record [COMA_STAT_REPLICAS][COMA_NORM_RECORD_DOUBLES(1, F, 4)].            */
int coma_gate_mid_bwd(const coma_tensor* dz, const coma_tensor* psi_raw, const coma_tensor* s_in,
                      const coma_tensor* g1raw, const coma_tensor* x1raw, const double* sums_psi, float eps_psi,
                      const float* gamma_psi, const double* bsums_psi, const double* sums_g, float eps_g,
                      const float* gamma_g, const double* sums_x, float eps_x, const float* gamma_x,
                      const float* w_psi, double* rec, const coma_tensor* dg1raw, const coma_tensor* dx1raw,
                      float* dgamma_g, float* dbeta_g, float* dgamma_x, float* dbeta_x, float* dw_psi,
                      float* dgamma_psi, float* dbeta_psi, void* stream);

/* ---- generic strided element-wise helpers ---- */
/* dst = a (+ b).  b may be NULL.  a/b with B == 1 broadcast over dst's batch. */
int coma_add(const coma_tensor* a, const coma_tensor* b, const coma_tensor* dst, void* stream);
/* dst = dtype_of_dst(src): same grid and channels, any pitches, fp32 <-> bf16 (input staging into padded buffers) */
int coma_cast_copy(const coma_tensor* src, const coma_tensor* dst, void* stream);
/* base[off, off + len) = 0 for each of the n (off, len) int64 rows (elements) of the device table `ranges`; max_len = the
 * longest row (sizes the grid).  One launch for the sparse zero_grad of a flat gradient buffer.                         */
int coma_zero_ranges(float* base, const int64_t* ranges, int32_t n, int64_t max_len, void* stream);
/* dst[0] (=) sum_b src[b]  (gradient of a batch-broadcast parameter) */
int coma_batch_sum(const coma_tensor* src, const coma_tensor* dst, void* stream);
/* per-sample, per-channel mean over voxels -> fp32 [B][C] (AdaptiveAvgPool3d(1)) */
int coma_spatial_mean(const coma_tensor* x, float* out, void* ws, size_t ws_bytes, void* stream);

/* ---- ROI prior painting + prompt select (attn_unet_data_parallel.py:630-651) ----
 * roi: fp32 labels (C==1); x: the MRI (C==1); prior: fp32 [B][n_roi][2] (loc,std);
 * roi_ids: int32 [n_roi]; abeta: fp32 [B]; prompts: fp32, shape (1,D,H,W,1).
 * out3: (B,D,H,W,3) = cat(prompt_sel, saliency, suvr)  with the x < 1e-4 zeroing.   */
int coma_roi_paint_fwd(const coma_tensor* roi, const coma_tensor* x, const float* prior,
                       const int32_t* roi_ids, int32_t n_roi, const float* abeta,
                       const float* pos_prompt, const float* neg_prompt,
                       const coma_tensor* out3, void* stream);
/* dpos/dneg (+=, fp32 volumes): channel 0 of dout3 routed by abeta */
int coma_roi_paint_bwd(const coma_tensor* dout3, const float* abeta, float* dpos, float* dneg, void* stream);

/* ---- losses (criterions.py:181-211 RoiMSE voxel_wise=False; L1 = the MAE metric,
 *      attn_unet_data_parallel.py:1215) ----
 * loss[b] = mean_vox(mask_b) * mean_vox((pred_b - gt_b)^2), mask = weight LUT of roi */
size_t coma_loss_ws_bytes(const coma_tensor* pred);
int coma_roi_mse_fwd(const coma_tensor* pred, const coma_tensor* gt, const coma_tensor* roi,
                     const int32_t* roi_ids, const float* roi_w, int32_t n_roi,
                     float* loss, float* mask_mean, void* ws, size_t ws_bytes, void* stream);
/* dpred (=) gout[b] * mask_mean[b] * 2 (pred - gt) / V */
int coma_roi_mse_bwd(const coma_tensor* pred, const coma_tensor* gt, const float* gout,
                     const float* mask_mean, const coma_tensor* dpred, void* stream);
int coma_l1_fwd(const coma_tensor* pred, const coma_tensor* gt, float* loss, void* ws, size_t ws_bytes, void* stream);
int coma_l1_bwd(const coma_tensor* pred, const coma_tensor* gt, const float* gout,
                const coma_tensor* dpred, void* stream);

/* ---- evaluation statistics: one pass over (pred, gt, roi) giving, per sample and per
 *      bin (n_roi ROIs + the whole volume), the sums behind calc_roi_metrics (attn_unet_data_parallel.py:1361-1397),
 *      the global MAE/MAPE/RSE/RRMSE of contrastive_test (:1214-1231) and RoiCorrMetric (:49-60).
 *      stats: fp64 [B][n_roi+1][8] = {count, sum|d|, sum d^2, sum g, sum g^2, sum p, sum|d/g| (non-NaN), #non-NaN} ---- */
int coma_eval_stats(const coma_tensor* pred, const coma_tensor* gt, const coma_tensor* roi,
                    const int32_t* roi_ids, int32_t n_roi, double* stats, void* stream);

/* ---- AdamW over a flat fp32 buffer (torch.optim.AdamW defaults,
 *      attn_unet_data_parallel.py:736): p,g,m,v length n; step counted from 1.  When step_dev is
 *      non-NULL the step count is read from that device int32 instead (hipGraph replay). ---- */
int coma_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
               float beta2, float eps, float weight_decay, int32_t step, const int32_t* step_dev,
               void* stream);

/* ---- 3-D SSIM (MONAI SSIMMetric(spatial_dims=3), attn_unet_data_parallel.py:1176,1234) ----
 * x, y: single-channel volumes; window: `win` (<= 11) separable weights (host pointer); c1 = (k1 R)^2, c2 = (k2 R)^2.
 * Writes one fp64 partial sum of the SSIM map per 8^3 output tile: partial[b * ntiles + t]; the per-sample SSIM is
 * sum_t partial / ((D-win+1)(H-win+1)(W-win+1)).                                                              */
size_t coma_ssim_ws_bytes(const coma_tensor* x, int32_t win);
int coma_ssim_partial(const coma_tensor* x, const coma_tensor* y, const float* window, int32_t win, float c1,
                      float c2, double* partial, size_t partial_bytes, int32_t* ntiles_out, void* stream);

/* ---- input pipeline (replaces VolumeDataset_ADNI_A4_combined.py:95-133,:62) ----
 * Nearest-neighbour resample of a raw fp32 (z, y, x) volume with voxel spacing sp_* (mm) onto a Do x Ho x Wo grid with
 * spacing nsp_* (same origin and axes, identity transform, ITK rounding), optional nan_to_num, optional zeroing where
 * the dst-sized volume `zero_where` is 0 (the reference's `mri[roi == 0] = 0`).  Out-of-volume samples take
 * default_value (the reference passes the image's SimpleITK pixel-type id there).                              */
int coma_resample_nearest(const float* src, int32_t Dz, int32_t Hy, int32_t Wx, double sp_z, double sp_y,
                          double sp_x, float* dst, int32_t Do, int32_t Ho, int32_t Wo, double nsp_z,
                          double nsp_y, double nsp_x, float default_value, int32_t nan_to_num,
                          const float* zero_where, void* stream);

/* ---- data-parallel gradient exchange over RCCL / xGMI (SURVEY.md section 8(b) last row, 8(e)): replaces the
 *      torch.nn.DataParallel the reference imports but never applies (attn_unet_data_parallel.py:32,1554).
 * One communicator per process (= per GPU).  Rank 0 creates the 128-byte id, the host distributes it by any means
 * (the Python host uses its torch.distributed group), every rank calls coma_comm_init on ITS device.  Collectives are
 * enqueued on the stream passed in -- a side HIP stream, so that a bucket's exchange overlaps the rest of backward -- and
 * never synchronise the host; they may be captured into a hipGraph.  All buffers are fp32 device pointers, reduction
 * is SUM (the reference sums the per-sample losses, criterions.py:560).  RCCL is bound at run time: an instance already
 * loaded in the process (PyTorch's) is shared.                                                                     */
#define COMA_COMM_ID_BYTES 128
int coma_comm_unique_id(void* id_out /* host, COMA_COMM_ID_BYTES */);
int coma_comm_init(const void* id /* host */, int32_t rank, int32_t nranks, void** comm_out);
int coma_comm_destroy(void* comm);
int coma_allreduce_sum_f32(void* comm, float* buf, int64_t n, void* stream);                       /* in place */
/* recv[n_per_rank] = this rank's slice of the sum of every rank's send[nranks * n_per_rank]; recv may alias that slice */
int coma_reduce_scatter_sum_f32(void* comm, const float* send, float* recv, int64_t n_per_rank, void* stream);
/* recv[nranks * n_per_rank] = concatenation of every rank's send[n_per_rank]; send may alias its own slice of recv */
int coma_allgather_f32(void* comm, const float* send, float* recv, int64_t n_per_rank, void* stream);
int coma_broadcast_f32(void* comm, float* buf, int64_t n, int32_t root, void* stream);            /* in place */

/* External events (the data-parallel exchange of a graph-replayed step WITHOUT a collective inside the graph;
 * coma_unet_amd/data_parallel.py GraphBucketWatch).  coma_event_record_external called on a stream that is being captured
 * adds an event-record node to the graph (hipGraphAddEventRecordNode behind the capture's current dependencies) instead of an internal
 * dependency; after every launch of the graph a stream OUTSIDE it can wait for that node with coma_stream_wait_external
 * (hipStreamWaitEvent) and run a bucket's all-reduce beside the rest of the replayed backward.
 * Outside a capture both behave as plain record / wait.  Events are created without timing.                          */
int coma_event_create(void** event_out);
int coma_event_destroy(void* event);
int coma_event_record_external(void* event, void* stream);
int coma_stream_wait_external(void* stream, void* event);

#ifdef __cplusplus
}
#endif
#endif /* COMA_UNET_H */
