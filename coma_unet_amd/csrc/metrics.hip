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

// ---------------------------------------------------------------------------------------------------------------------
// 3-D SSIM (the reference's `SSIMMetric(spatial_dims=3, data_range=1)` from MONAI, attn_unet_data_parallel.py:1176,1234):
// windowed means of x, y, x^2, y^2, xy with a separable window ("valid" region, no padding), then
//   ssim = (2 mu_x mu_y + c1) / (mu_x^2 + mu_y^2 + c1) * (2 s_xy + c2) / (s_x + s_y + c2),
// averaged over the valid voxels of each sample.  One block = one 8^3 tile of outputs: its (8 + win - 1)^3 input cube is
// read once, the three separable passes run in LDS, and the block emits one fp64 partial sum.
// ---------------------------------------------------------------------------------------------------------------------
#define SSIM_T 8
#define SSIM_MAXW 11
struct SsimP {
  const void* x; int64_t ldx, sbx;
  const void* y; int64_t ldy, sby;
  int D, H, W, win;
  int ntx, nty, ntz;
  float c1, c2;
  float w[SSIM_MAXW];
  double* partial;      // [B][ntiles]
};

template <typename T>
__global__ __launch_bounds__(256) void ssim_partial_k(SsimP p) {
  constexpr int TO = SSIM_T, TI = SSIM_T + SSIM_MAXW - 1;       // outputs / inputs per axis (input extent used: TO + win - 1)
  __shared__ float in_[2][TI][TI][TI];                          // x, y cubes                        (2 * 18^3 * 4 = 46.7 KB)
  __shared__ float px[5][TI][TI][TO];                           // after the x pass: 5 quantities    (5 * 18*18*8 * 4 = 51.8 KB)
  __shared__ float py[5][TI][TO][TO];                           // after the y pass                  (23 KB)
  __shared__ double red[256];
  const int b = blockIdx.y;
  int t = blockIdx.x;
  const int tx = t % p.ntx; t /= p.ntx;
  const int ty = t % p.nty; const int tz = t / p.nty;
  const int x0 = tx * TO, y0 = ty * TO, z0 = tz * TO;
  const int ext = TO + p.win - 1;
  const int Do = p.D - p.win + 1, Ho = p.H - p.win + 1, Wo = p.W - p.win + 1;
  const T* xb = reinterpret_cast<const T*>(p.x) + (int64_t)b * p.sbx;
  const T* yb = reinterpret_cast<const T*>(p.y) + (int64_t)b * p.sby;
  const int tid = threadIdx.x;
  for (int i = tid; i < ext * ext * ext; i += 256) {
    const int ix = i % ext, iy = (i / ext) % ext, iz = i / (ext * ext);
    const int gx = x0 + ix, gy = y0 + iy, gz = z0 + iz;
    float a = 0.f, c = 0.f;
    if (gx < p.W && gy < p.H && gz < p.D) {
      const int64_t v = ((int64_t)gz * p.H + gy) * p.W + gx;
      a = ld_f(xb + v * p.ldx); c = ld_f(yb + v * p.ldy);
    }
    in_[0][iz][iy][ix] = a; in_[1][iz][iy][ix] = c;
  }
  __syncthreads();
  for (int i = tid; i < ext * ext * TO; i += 256) {             // x pass
    const int ox = i % TO, iy = (i / TO) % ext, iz = i / (TO * ext);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
    for (int k = 0; k < p.win; ++k) {
      const float a = in_[0][iz][iy][ox + k], c = in_[1][iz][iy][ox + k], w = p.w[k];
      s0 = fmaf(w, a, s0); s1 = fmaf(w, c, s1); s2 = fmaf(w, a * a, s2); s3 = fmaf(w, c * c, s3); s4 = fmaf(w, a * c, s4);
    }
    px[0][iz][iy][ox] = s0; px[1][iz][iy][ox] = s1; px[2][iz][iy][ox] = s2; px[3][iz][iy][ox] = s3; px[4][iz][iy][ox] = s4;
  }
  __syncthreads();
  for (int i = tid; i < 5 * ext * TO * TO; i += 256) {          // y pass
    const int ox = i % TO, oy = (i / TO) % TO, iz = (i / (TO * TO)) % ext, q = i / (TO * TO * ext);
    float s = 0.f;
    for (int k = 0; k < p.win; ++k) s = fmaf(p.w[k], px[q][iz][oy + k][ox], s);
    py[q][iz][oy][ox] = s;
  }
  __syncthreads();
  double acc = 0.0;
  for (int i = tid; i < TO * TO * TO; i += 256) {               // z pass + SSIM
    const int ox = i % TO, oy = (i / TO) % TO, oz = i / (TO * TO);
    if (x0 + ox < Wo && y0 + oy < Ho && z0 + oz < Do) {
      float m[5];
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        float s = 0.f;
        for (int k = 0; k < p.win; ++k) s = fmaf(p.w[k], py[q][oz + k][oy][ox], s);
        m[q] = s;
      }
      const float sx = m[2] - m[0] * m[0], sy = m[3] - m[1] * m[1], sxy = m[4] - m[0] * m[1];
      const float cs = (2.f * sxy + p.c2) / (sx + sy + p.c2);
      acc += (double)(((2.f * m[0] * m[1] + p.c1) / (m[0] * m[0] + m[1] * m[1] + p.c1)) * cs);
    }
  }
  red[tid] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  if (tid == 0) p.partial[(int64_t)b * gridDim.x + blockIdx.x] = red[0];
}

extern "C" size_t coma_ssim_ws_bytes(const coma_tensor* x, int32_t win) {
  if (win < 1 || x->D < win || x->H < win || x->W < win) return 0;
  const int64_t nt = (int64_t)((x->D - win + SSIM_T) / SSIM_T) * ((x->H - win + SSIM_T) / SSIM_T) * ((x->W - win + SSIM_T) / SSIM_T);
  return (size_t)x->B * nt * sizeof(double);
}

// partial sums [B][ntiles] (fp64) of the SSIM map over the valid region; *ntiles_out receives the tile count per sample.
extern "C" int coma_ssim_partial(const coma_tensor* x, const coma_tensor* y, const float* window, int32_t win, float c1, float c2,
                                 double* partial, size_t partial_bytes, int32_t* ntiles_out, void* stream) {
  COMA_CHECK(x && y && x->data && y->data && window && partial && ntiles_out, "ssim: null argument");
  COMA_CHECK(t_same_grid(x, y) && x->C == 1 && y->C == 1 && x->dtype == y->dtype, "ssim: single-channel volumes of equal shape expected");
  COMA_CHECK(win >= 1 && win <= SSIM_MAXW && x->D >= win && x->H >= win && x->W >= win, "ssim: window %d does not fit", win);
  COMA_CHECK(partial_bytes >= coma_ssim_ws_bytes(x, win), "ssim: partial buffer too small");
  SsimP p;
  p.x = x->data; p.ldx = x->ld; p.sbx = x->sb; p.y = y->data; p.ldy = y->ld; p.sby = y->sb;
  p.D = x->D; p.H = x->H; p.W = x->W; p.win = win; p.c1 = c1; p.c2 = c2;
  for (int k = 0; k < SSIM_MAXW; ++k) p.w[k] = k < win ? window[k] : 0.f;
  p.ntz = (x->D - win + SSIM_T) / SSIM_T; p.nty = (x->H - win + SSIM_T) / SSIM_T; p.ntx = (x->W - win + SSIM_T) / SSIM_T;
  p.partial = partial;
  const int nt = p.ntx * p.nty * p.ntz;
  *ntiles_out = nt;
  dim3 grid((unsigned)nt, (unsigned)x->B);
  if (x->dtype == COMA_F32) hipLaunchKernelGGL(ssim_partial_k<float>, grid, dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(ssim_partial_k<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, p);
  COMA_LAUNCH_CHECK();
  return 0;
}
