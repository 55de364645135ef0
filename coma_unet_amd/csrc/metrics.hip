// Evaluation statistics in ONE pass over (pred, gt, roi): per sample and per bin (36 ROIs + the whole
// volume) the sums every metric of the reference's evaluation loop is built from.  Replaces the 36 x
// (mask fill + 6 masked full-volume reductions) Python loop of calc_roi_metrics
// (attn_unet_data_parallel.py:1361-1397), the global MAE/MAPE/RSE/RRMSE reductions of contrastive_test
// (:1214-1231) and RoiCorrMetric.acc_roi_corr's masked means (:49-60).  HBM-bound: 3 volume reads.
#include "common.h"

// stats[b][bin][k], k: 0 count, 1 sum|d|, 2 sum d^2, 3 sum g, 4 sum g^2, 5 sum p,
//                      6 sum |d/g| over non-NaN entries, 7 count of non-NaN |d/g|
// For the whole-volume bin (index n_roi) slots 6/7 follow contrastive_test's nr_mape instead:
// |(g-p)/g| where |g| > 1e-8 (else NaN, skipped).
#define EV_K 8
template <typename T>
__global__ __launch_bounds__(256) void eval_stats_k(const T* pred, int64_t sbp, int64_t ldp, const T* gt, int64_t sbg, const float* roi,
                                                    int64_t sbr, const int32_t* ids_g, int n_roi, int64_t V, double* stats) {
  __shared__ int32_t ids[64];
  __shared__ double bins[64][EV_K];
  __shared__ signed char lut[COMA_ROI_LUT];
  const int b = blockIdx.y;
  if (threadIdx.x < n_roi) ids[threadIdx.x] = ids_g[threadIdx.x];
  __syncthreads();
  roi_lut_build(lut, ids, n_roi);
  for (int i = threadIdx.x; i < 64 * EV_K; i += 256) (&bins[0][0])[i] = 0.0;
  __syncthreads();
  double g0 = 0, g1 = 0, g2 = 0, g3 = 0, g4 = 0, g5 = 0, g6 = 0, g7 = 0;   // whole-volume bin in registers
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (int64_t)gridDim.x * 256) {
    const float p = ld_f(pred + b * sbp + v * ldp), g = ld_f(gt + b * sbg + v);
    const float d = p - g;
    const float ad = fabsf(d);
    g0 += 1.0; g1 += ad; g2 += (double)d * d; g3 += g; g4 += (double)g * g; g5 += p;
    if (fabsf(g) > 1e-8f) { g6 += (double)fabsf((g - p) / g); g7 += 1.0; }
    const float lab = roi[b * sbr + v];
    const int slot = roi_lut_slot(lut, ids, n_roi, lab);
    if (slot >= 0) {
      double* bn = bins[slot];
      atomicAdd(bn + 0, 1.0); atomicAdd(bn + 1, (double)ad); atomicAdd(bn + 2, (double)d * d);
      atomicAdd(bn + 3, (double)g); atomicAdd(bn + 4, (double)g * g); atomicAdd(bn + 5, (double)p);
      const float r = fabsf(d / g);            // raw_mape of :1217 -- inf where g == 0 and d != 0, NaN for 0/0
      if (!(r != r)) { atomicAdd(bn + 6, (double)r); atomicAdd(bn + 7, 1.0); }
    }
  }
  double* gb = bins[n_roi];
  atomicAdd(gb + 0, g0); atomicAdd(gb + 1, g1); atomicAdd(gb + 2, g2); atomicAdd(gb + 3, g3);
  atomicAdd(gb + 4, g4); atomicAdd(gb + 5, g5); atomicAdd(gb + 6, g6); atomicAdd(gb + 7, g7);
  __syncthreads();
  for (int i = threadIdx.x; i < (n_roi + 1) * EV_K; i += 256) {
    const double val = (&bins[0][0])[i];
    if (val != 0.0) atomicAdd(stats + (int64_t)b * (n_roi + 1) * EV_K + i, val);
  }
}

extern "C" int coma_eval_stats(const coma_tensor* pred, const coma_tensor* gt, const coma_tensor* roi, const int32_t* roi_ids,
                               int32_t n_roi, double* stats, void* stream) {
  COMA_CHECK(pred && gt && roi && pred->data && gt->data && roi->data && roi_ids && stats, "eval_stats: null argument");
  COMA_CHECK(t_same_grid(pred, gt) && t_same_grid(pred, roi) && pred->C == 1 && gt->C == 1 && roi->C == 1 &&
             gt->ld == 1 && roi->ld == 1 && pred->dtype == gt->dtype && roi->dtype == COMA_F32,
             "eval_stats: single-channel volumes of equal shape (gt, roi contiguous; roi fp32) expected");
  COMA_CHECK(n_roi > 0 && n_roi < 64, "eval_stats: n_roi=%d out of range", n_roi);
  hipStream_t s = (hipStream_t)stream;
  const int64_t V = t_vox(pred);
  if (hipMemsetAsync(stats, 0, sizeof(double) * pred->B * (n_roi + 1) * EV_K, s) != hipSuccess) {
    coma_set_error("eval_stats: memset failed"); return 2; }
  int nblk = (int)((V + 256 * 16 - 1) / (256 * 16));
  if (nblk > 1024) nblk = 1024;
  if (nblk < 1) nblk = 1;
  dim3 grid(nblk, pred->B);
  if (pred->dtype == COMA_F32)
    hipLaunchKernelGGL(eval_stats_k<float>, grid, dim3(256), 0, s, (const float*)pred->data, pred->sb, pred->ld, (const float*)gt->data,
                       gt->sb, (const float*)roi->data, roi->sb, roi_ids, n_roi, V, stats);
  else
    hipLaunchKernelGGL(eval_stats_k<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)pred->data, pred->sb, pred->ld,
                       (const bf16_t*)gt->data, gt->sb, (const float*)roi->data, roi->sb, roi_ids, n_roi, V, stats);
  COMA_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Input pipeline: nearest-neighbour resample of a raw (z, y, x) volume to the 2 mm training grid
// + nan_to_num + background masking in one pass.  Replaces the per-sample host pipeline of the reference's
// `load_volume_file` (VolumeDataset_ADNI_A4_combined.py:95-133: SimpleITK ResampleImageFilter with an identity transform,
// same origin / direction, sitkNearestNeighbor; torch.nan_to_num) and the `mri_tensor[roi_tensor == 0] = 0` of :62.
// Output voxel o samples the input at continuous index o * new_spacing / old_spacing per axis (computed in double like
// ITK's physical-point round trip), rounded half-up; outside [-0.5, size - 0.5) it takes `default_value`.
// HBM-bound: one scattered read + one coalesced write per output voxel.
// ---------------------------------------------------------------------------------------------------------------------
struct ResampleP {
  const float* src; int Dz, Hy, Wx;
  float* dst; int Do, Ho, Wo;
  double rz, ry, rx;              // new_spacing / old_spacing per axis
  float default_value;
  int nan_to_num;
  const float* zero_where;        // optional, dst-sized: output = 0 where this volume == 0
};

__device__ __forceinline__ int nn_index(int o, double ratio, int size) {
  const double c = (double)o * ratio;                 // continuous input index
  if (!(c >= -0.5 && c < (double)size - 0.5)) return -1;
  return (int)floor(c + 0.5);                         // itk::Math::RoundHalfIntegerUp
}

__global__ __launch_bounds__(256) void resample_nn_k(ResampleP p) {
  const int64_t total = (int64_t)p.Do * p.Ho * p.Wo;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int x = (int)(e % p.Wo), y = (int)((e / p.Wo) % p.Ho), z = (int)(e / ((int64_t)p.Wo * p.Ho));
    const int ix = nn_index(x, p.rx, p.Wx), iy = nn_index(y, p.ry, p.Hy), iz = nn_index(z, p.rz, p.Dz);
    float v = p.default_value;
    if (ix >= 0 && iy >= 0 && iz >= 0) v = p.src[((int64_t)iz * p.Hy + iy) * p.Wx + ix];
    if (p.nan_to_num) {                                // torch.nan_to_num defaults: nan -> 0, +-inf -> +-FLT_MAX
      if (v != v) v = 0.f;
      else if (v == INFINITY) v = 3.402823466e+38f;
      else if (v == -INFINITY) v = -3.402823466e+38f;
    }
    if (p.zero_where && p.zero_where[e] == 0.f) v = 0.f;
    p.dst[e] = v;
  }
}

extern "C" int coma_resample_nearest(const float* src, int32_t Dz, int32_t Hy, int32_t Wx, double sp_z, double sp_y, double sp_x,
                                     float* dst, int32_t Do, int32_t Ho, int32_t Wo, double nsp_z, double nsp_y, double nsp_x,
                                     float default_value, int32_t nan_to_num, const float* zero_where, void* stream) {
  COMA_CHECK(src && dst && Dz > 0 && Hy > 0 && Wx > 0 && Do > 0 && Ho > 0 && Wo > 0, "resample_nearest: bad argument");
  COMA_CHECK(sp_z > 0 && sp_y > 0 && sp_x > 0 && nsp_z > 0 && nsp_y > 0 && nsp_x > 0, "resample_nearest: spacings must be positive");
  ResampleP p;
  p.src = src; p.Dz = Dz; p.Hy = Hy; p.Wx = Wx; p.dst = dst; p.Do = Do; p.Ho = Ho; p.Wo = Wo;
  p.rz = nsp_z / sp_z; p.ry = nsp_y / sp_y; p.rx = nsp_x / sp_x;
  p.default_value = default_value; p.nan_to_num = nan_to_num; p.zero_where = zero_where;
  const int64_t total = (int64_t)Do * Ho * Wo;
  int64_t nb = (total + 255) / 256;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(resample_nn_k, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p);
  COMA_LAUNCH_CHECK();
  return 0;
}
