// BatchNorm3d (training statistics) / InstanceNorm3d + fused activation, forward and
// backward, for channels-last volumes.  Replaces the ATen norm + activation dispatches
// behind MONAI's ADN blocks ("N","A": BN+ReLU in ConvBlock/UpConv, IN+PReLU in the merge
// convs, IN+LeakyReLU in StackedFusionConvLayers attn_unet_data_parallel.py:480-501) and
// the bare BatchNorm3d layers of the attention gate (:141-144).  HBM-bound: one read for
// the statistics, one read + one write for the apply; statistics are accumulated in fp64
// and reduced in a fixed order (bitwise reproducible).
#include "common.h"

struct RowsP {            // a [G][R rows][C] view of a coma_tensor
  const void* x; int64_t ld, sb; int64_t V; int B; int C;
  int G;                  // groups: 1 (batch) or B (instance)
  int64_t R;              // rows per group: B*V or V
  int cv, cvp, ry;        // channel vectors, padded to pow2, rows per block step
  int64_t ch;             // rows per chunk
  int nchunks;
};

static RowsP make_rows(const coma_tensor* x, int mode, int vec) {
  RowsP p;
  p.x = x->data; p.ld = x->ld; p.sb = x->sb; p.V = t_vox(x); p.B = x->B; p.C = x->C;
  p.G = mode == COMA_NORM_INSTANCE ? x->B : 1;
  p.R = mode == COMA_NORM_INSTANCE ? p.V : p.V * x->B;
  p.cv = x->C / vec;
  int cvp = 1; while (cvp < p.cv) cvp <<= 1;
  if (cvp > 256) cvp = 256;
  p.cvp = cvp; p.ry = 256 / cvp;
  int64_t nch = (p.R + (int64_t)p.ry * 16 - 1) / ((int64_t)p.ry * 16);
  const int64_t cap = 1024 / p.G > 0 ? 1024 / p.G : 1;
  if (nch > cap) nch = cap;
  if (nch < 1) nch = 1;
  p.nchunks = (int)nch;
  p.ch = (p.R + nch - 1) / nch;
  return p;
}

__device__ __forceinline__ int64_t row_off(const RowsP& p, int g, int64_t r) {
  // group g, row r -> element offset of channel 0
  if (p.G == 1) { const int64_t b = r / p.V, v = r - b * p.V; return b * p.sb + v * p.ld; }
  return (int64_t)g * p.sb + r * p.ld;
}

// Rows r0 + ty, + ry, ... of group g without a 64-bit division per row: (sample, voxel) advance incrementally.
struct RowWalk {
  int64_t r, b, v;
  __device__ __forceinline__ RowWalk(const RowsP& p, int64_t r_) : r(r_) {
    if (p.G == 1) { b = r_ / p.V; v = r_ - b * p.V; } else { b = 0; v = r_; }
  }
  __device__ __forceinline__ int64_t off(const RowsP& p, int g, int64_t ld, int64_t sb) const {
    return p.G == 1 ? b * sb + v * ld : (int64_t)g * sb + v * ld;
  }
  __device__ __forceinline__ void step(const RowsP& p) {
    r += p.ry; v += p.ry;
    if (p.G == 1 && v >= p.V) { v -= p.V; ++b; }      // ry <= 256 << V
  }
};

// partial[(chunk*G + g)*C + c] = {sum, sumsq} over the chunk's rows
template <typename T, int VEC>
__global__ __launch_bounds__(256) void stats_partial_k(RowsP p, double2* partial) {
  __shared__ double sh[256][2 * VEC];
  const int tid = threadIdx.x, tx = tid % p.cvp, ty = tid / p.cvp;
  const int g = blockIdx.y;
  const int64_t r0 = (int64_t)blockIdx.x * p.ch;
  const int64_t r1 = r0 + p.ch < p.R ? r0 + p.ch : p.R;
  double s[VEC], q[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { s[j] = 0.0; q[j] = 0.0; }
  if (tx < p.cv) {
    const T* base = reinterpret_cast<const T*>(p.x);
    // bf16 volumes: fp32 sums over bursts of 8 rows, folded into the fp64 accumulators (an 8-term fp32 sum of
    // bf16-sized data loses nothing that matters; the long reduction stays fp64).  fp32 volumes: every term goes
    // straight to fp64 (burst of 1), as the exact-fp32 mode's gradient parity needs.
    constexpr int BURST = sizeof(T) == 2 ? 8 : 1;
    RowWalk w(p, r0 + ty);
    while (w.r < r1) {
      float fs[VEC], fq[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) { fs[j] = 0.f; fq[j] = 0.f; }
#pragma unroll
      for (int u = 0; u < BURST; ++u) {
        if (w.r < r1) {
          float v[VEC];
          vec_io<T, VEC>::load(base + w.off(p, g, p.ld, p.sb) + tx * VEC, v);
#pragma unroll
          for (int j = 0; j < VEC; ++j) { fs[j] += v[j]; fq[j] = fmaf(v[j], v[j], fq[j]); }
          w.step(p);
        }
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) { s[j] += (double)fs[j]; q[j] += (double)fq[j]; }
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) { sh[tid][2 * j] = s[j]; sh[tid][2 * j + 1] = q[j]; }
  __syncthreads();
  // rows of the block -> one: a tree over ty (ry = 256 / cvp is a power of two).  The serial walk this replaces cost
  // ry x 2 VEC LDS reads on cv threads -- a fifth of the kernel on the 16-channel volumes (ry = 128).
  for (int off = p.ry >> 1; off > 0; off >>= 1) {
    if (ty < off) {
#pragma unroll
      for (int j = 0; j < 2 * VEC; ++j) sh[tid][j] += sh[tid + off * p.cvp][j];
    }
    __syncthreads();
  }
  if (ty == 0 && tx < p.cv) {
#pragma unroll
    for (int j = 0; j < VEC; ++j)
      partial[((int64_t)blockIdx.x * p.G + g) * p.C + tx * VEC + j] = make_double2(sh[tid][2 * j], sh[tid][2 * j + 1]);
  }
}

__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// one wave per (group, channel): lanes stride over the chunk partials, fixed-order butterfly
__global__ __launch_bounds__(256) void stats_finalize_k(const double2* partial, int nchunks, int G, int C, int64_t R, float eps,
                                 float* mean, float* rstd, float* rmean, float* rvar, float momentum) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= G * C) return;
  double s = 0.0, q = 0.0;
  for (int k = lane; k < nchunks; k += 64) { const double2 v = partial[(int64_t)k * G * C + i]; s += v.x; q += v.y; }
  s = wave_sum(s); q = wave_sum(q);
  if (lane != 0) return;
  const double m = s / (double)R;
  double var = q / (double)R - m * m;
  if (var < 0.0) var = 0.0;
  mean[i] = (float)m;
  rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
  if (rmean && G == 1) {
    const double unb = R > 1 ? var * (double)R / (double)(R - 1) : var;
    rmean[i] = (1.f - momentum) * rmean[i] + momentum * (float)m;
    rvar[i] = (1.f - momentum) * rvar[i] + momentum * (float)unb;
  }
}

struct ApplyP {
  const void* x; int64_t ldx, sbx;
  void* y; int64_t ldy, sby;
  const void* dy; int64_t lddy, sbdy;
  int64_t V; int B, C, cv; int inst;
  const float *mean, *rstd, *gamma, *beta, *slope;
  int act;
};

template <typename T, int VEC>
__global__ __launch_bounds__(256) void norm_act_fwd_k(ApplyP p) {
  const int b = blockIdx.y;
  const int64_t total = p.V * p.cv;
  const float a = p.slope ? *p.slope : 0.25f;
  const T* xb = reinterpret_cast<const T*>(p.x) + (int64_t)b * p.sbx;
  T* yb = reinterpret_cast<T*>(p.y) + (int64_t)b * p.sby;
  const int g = p.inst ? b : 0;
  const int64_t stride = (int64_t)gridDim.x * 256, e0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (stride % p.cv == 0) {
    // the thread keeps its channel group for the whole sweep: scale / shift live in registers, no division in the loop
    const int c0 = (int)(e0 % p.cv) * VEC;
    float sc[VEC], sh[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float rs = p.rstd[g * p.C + c0 + j], ga = p.gamma ? p.gamma[c0 + j] : 1.f;
      sc[j] = rs * ga;
      sh[j] = (p.gamma ? p.beta[c0 + j] : 0.f) - p.mean[g * p.C + c0 + j] * rs * ga;
    }
    const int64_t dv = stride / p.cv;
    for (int64_t v = e0 / p.cv; v < p.V; v += dv) {
      float xv[VEC], yv[VEC];
      vec_io<T, VEC>::load(xb + v * p.ldx + c0, xv);
#pragma unroll
      for (int j = 0; j < VEC; ++j) yv[j] = act_fwd(p.act, fmaf(xv[j], sc[j], sh[j]), a);
      vec_io<T, VEC>::store(yb + v * p.ldy + c0, yv);
    }
    return;
  }
  for (int64_t e = e0; e < total; e += stride) {
    const int64_t v = e / p.cv; const int c0 = (int)(e - v * p.cv) * VEC;
    float xv[VEC], yv[VEC];
    vec_io<T, VEC>::load(xb + v * p.ldx + c0, xv);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int c = c0 + j;
      float z = (xv[j] - p.mean[g * p.C + c]) * p.rstd[g * p.C + c];
      if (p.gamma) z = z * p.gamma[c] + p.beta[c];
      yv[j] = act_fwd(p.act, z, a);
    }
    vec_io<T, VEC>::store(yb + v * p.ldy + c0, yv);
  }
}

// backward pass 1: per (g,c) partial {sum dz, sum dz*xhat, sum dy*dact/dslope}
template <typename T, int VEC>
__global__ __launch_bounds__(256) void norm_bwd_partial_k(RowsP p, const void* dyp, int64_t lddy, int64_t sbdy,
                                                          const float* mean, const float* rstd, const float* gamma,
                                                          const float* beta, int act, const float* slope,
                                                          double* partial /* [chunk][G][C][3] */) {
  __shared__ double sh[256][3 * VEC];
  const int tid = threadIdx.x, tx = tid % p.cvp, ty = tid / p.cvp;
  const int g = blockIdx.y;
  const int64_t r0 = (int64_t)blockIdx.x * p.ch;
  const int64_t r1 = r0 + p.ch < p.R ? r0 + p.ch : p.R;
  const float a = slope ? *slope : 0.25f;
  double s1[VEC], s2[VEC], s3[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { s1[j] = s2[j] = s3[j] = 0.0; }
  if (tx < p.cv) {
    const T* xb = reinterpret_cast<const T*>(p.x);
    const T* dyb = reinterpret_cast<const T*>(dyp);
    float mu[VEC], rs[VEC], ga[VEC], be[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int c = tx * VEC + j;
      mu[j] = mean[g * p.C + c]; rs[j] = rstd[g * p.C + c];
      ga[j] = gamma ? gamma[c] : 1.f; be[j] = gamma ? beta[c] : 0.f;
    }
    constexpr int BURST = sizeof(T) == 2 ? 8 : 1;
    RowWalk w(p, r0 + ty);
    while (w.r < r1) {             // bf16: fp32 bursts of 8 rows folded into fp64 (see stats_partial_k)
      float f1[VEC], f2[VEC], f3[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) { f1[j] = 0.f; f2[j] = 0.f; f3[j] = 0.f; }
#pragma unroll
      for (int u = 0; u < BURST; ++u) {
        if (w.r < r1) {
          float xv[VEC], dv[VEC];
          vec_io<T, VEC>::load(xb + w.off(p, g, p.ld, p.sb) + tx * VEC, xv);
          vec_io<T, VEC>::load(dyb + w.off(p, g, lddy, sbdy) + tx * VEC, dv);
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            const float xh = (xv[j] - mu[j]) * rs[j];
            const float z = xh * ga[j] + be[j];
            float ds; const float da = act_bwd(act, z, a, &ds);
            const float dz = dv[j] * da;
            f1[j] += dz; f2[j] = fmaf(dz, xh, f2[j]); f3[j] = fmaf(dv[j], ds, f3[j]);
          }
          w.step(p);
        }
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) { s1[j] += (double)f1[j]; s2[j] += (double)f2[j]; s3[j] += (double)f3[j]; }
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) { sh[tid][3 * j] = s1[j]; sh[tid][3 * j + 1] = s2[j]; sh[tid][3 * j + 2] = s3[j]; }
  __syncthreads();
  for (int off = p.ry >> 1; off > 0; off >>= 1) {        // tree over the block's rows (see stats_partial_k)
    if (ty < off) {
#pragma unroll
      for (int j = 0; j < 3 * VEC; ++j) sh[tid][j] += sh[tid + off * p.cvp][j];
    }
    __syncthreads();
  }
  if (ty == 0 && tx < p.cv) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      double* o = partial + (((int64_t)blockIdx.x * p.G + g) * p.C + tx * VEC + j) * 3;
      o[0] = sh[tid][3 * j]; o[1] = sh[tid][3 * j + 1]; o[2] = sh[tid][3 * j + 2];
    }
  }
}

// stage 1: one wave per channel reduces the chunk partials of each of its groups -> tot[(g*C+c)*3 .. +3] (fp64), the fp32
// means, and dgamma / dbeta[c] = sum over the groups (in group order: the same sum the separate finaliser made)
__global__ __launch_bounds__(256) void norm_bwd_reduce_k(const double* partial, int nchunks, int G, int C, int64_t R,
                                                         double* tot, float* sums, float* dgamma, float* dbeta) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (c >= C) return;
  double dg = 0.0, db = 0.0;
  for (int g = 0; g < G; ++g) {
    const int i = g * C + c;
    double a = 0.0, b = 0.0, s = 0.0;
    for (int k = lane; k < nchunks; k += 64) {
      const double* v = partial + ((int64_t)k * G * C + i) * 3;
      a += v[0]; b += v[1]; s += v[2];
    }
    a = wave_sum(a); b = wave_sum(b); s = wave_sum(s);
    if (lane == 0) {
      tot[i * 3] = a; tot[i * 3 + 1] = b; tot[i * 3 + 2] = s;
      sums[i * 2] = (float)(a / (double)R);
      sums[i * 2 + 1] = (float)(b / (double)R);
    }
    db += a; dg += b;
  }
  if (lane == 0) {
    if (dgamma) dgamma[c] = (float)dg;
    if (dbeta) dbeta[c] = (float)db;
  }
}
// stage 2: dgamma/dbeta[c] = sum over groups, dslope = sum over everything (<= 1024 numbers)
__global__ __launch_bounds__(256) void norm_bwd_finalize_k(const double* tot, int G, int C, float* dgamma, float* dbeta,
                                                           float* dslope) {
  __shared__ double ssl[256];
  double sl = 0.0;
  for (int c = threadIdx.x; c < C; c += 256) {
    double dg = 0.0, db = 0.0;
    for (int g = 0; g < G; ++g) { db += tot[(g * C + c) * 3]; dg += tot[(g * C + c) * 3 + 1]; sl += tot[(g * C + c) * 3 + 2]; }
    if (dgamma) dgamma[c] = (float)dg;
    if (dbeta) dbeta[c] = (float)db;
  }
  ssl[threadIdx.x] = sl;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) ssl[threadIdx.x] += ssl[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0 && dslope) *dslope = (float)ssl[0];
}

template <typename T, int VEC>
__global__ __launch_bounds__(256) void norm_act_bwd_apply_k(ApplyP p, const float* sums) {
  const int b = blockIdx.y;
  const int64_t total = p.V * p.cv;
  const float a = p.slope ? *p.slope : 0.25f;
  const T* xb = reinterpret_cast<const T*>(p.x) + (int64_t)b * p.sbx;
  const T* dyb = reinterpret_cast<const T*>(p.dy) + (int64_t)b * p.sbdy;
  T* ob = reinterpret_cast<T*>(p.y) + (int64_t)b * p.sby;
  const int g = p.inst ? b : 0;
  const int64_t stride = (int64_t)gridDim.x * 256, e0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (stride % p.cv == 0) {      // fixed channel group per thread: the per-channel constants live in registers
    const int c0 = (int)(e0 % p.cv) * VEC;
    float rs[VEC], mu[VEC], ga[VEC], be[VEC], s1[VEC], s2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int c = c0 + j;
      rs[j] = p.rstd[g * p.C + c]; mu[j] = p.mean[g * p.C + c];
      ga[j] = p.gamma ? p.gamma[c] : 1.f; be[j] = p.gamma ? p.beta[c] : 0.f;
      s1[j] = sums[(g * p.C + c) * 2]; s2[j] = sums[(g * p.C + c) * 2 + 1];
    }
    const int64_t dvx = stride / p.cv;
    for (int64_t v = e0 / p.cv; v < p.V; v += dvx) {
      float xv[VEC], dv[VEC], ov[VEC];
      vec_io<T, VEC>::load(xb + v * p.ldx + c0, xv);
      vec_io<T, VEC>::load(dyb + v * p.lddy + c0, dv);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float xh = (xv[j] - mu[j]) * rs[j];
        const float z = p.gamma ? xh * ga[j] + be[j] : xh;
        float ds; const float da = act_bwd(p.act, z, a, &ds);
        const float dz = dv[j] * da;
        ov[j] = rs[j] * ga[j] * (dz - s1[j] - xh * s2[j]);
      }
      vec_io<T, VEC>::store(ob + v * p.ldy + c0, ov);
    }
    return;
  }
  for (int64_t e = e0; e < total; e += stride) {
    const int64_t v = e / p.cv; const int c0 = (int)(e - v * p.cv) * VEC;
    float xv[VEC], dv[VEC], ov[VEC];
    vec_io<T, VEC>::load(xb + v * p.ldx + c0, xv);
    vec_io<T, VEC>::load(dyb + v * p.lddy + c0, dv);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int c = c0 + j;
      const float rs = p.rstd[g * p.C + c];
      const float xh = (xv[j] - p.mean[g * p.C + c]) * rs;
      const float ga = p.gamma ? p.gamma[c] : 1.f;
      const float z = p.gamma ? xh * ga + p.beta[c] : xh;
      float ds; const float da = act_bwd(p.act, z, a, &ds);
      const float dz = dv[j] * da;
      ov[j] = rs * ga * (dz - sums[(g * p.C + c) * 2] - xh * sums[(g * p.C + c) * 2 + 1]);
    }
    vec_io<T, VEC>::store(ob + v * p.ldy + c0, ov);
  }
}

// column sums of a [B][V][C] tensor -> fp32 out[B or 1][C]  (conv bias gradient, spatial mean)
__global__ __launch_bounds__(256) void colsum_finalize_k(const double2* partial, int nchunks, int G, int C, double scale, float* out) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= G * C) return;
  double s = 0.0;
  for (int k = lane; k < nchunks; k += 64) s += partial[(int64_t)k * G * C + i].x;
  s = wave_sum(s);
  if (lane == 0) out[i] = (float)(s * scale);
}

static int pick_vec(const coma_tensor* x) { return t_vec(x, 4) >= 4 ? 4 : 1; }
// 16 bytes per lane where the layout allows (bf16 volumes with C, pitch, base multiples of 8 channels)
static int pick_vec8(const coma_tensor* x) { return (x->dtype == COMA_BF16 && t_vec(x, 8) == 8) ? 8 : pick_vec(x); }

extern "C" size_t coma_norm_ws_bytes(const coma_tensor* x) {
  // partials: <= 1024 (chunk,group) pairs x C x 3 doubles, + sums [B][C][2] floats
  return (size_t)1024 * x->C * 3 * sizeof(double) + (size_t)x->B * x->C * (3 * sizeof(double) + 2 * sizeof(float)) + 256;
}

template <typename T>
static void launch_partial(const RowsP& p, int vec, double2* partial, hipStream_t s) {
  dim3 grid(p.nchunks, p.G);
  if (vec == 8) { if constexpr (sizeof(T) == 2) hipLaunchKernelGGL((stats_partial_k<T, 8>), grid, dim3(256), 0, s, p, partial); }
  else if (vec == 4) hipLaunchKernelGGL((stats_partial_k<T, 4>), grid, dim3(256), 0, s, p, partial);
  else hipLaunchKernelGGL((stats_partial_k<T, 1>), grid, dim3(256), 0, s, p, partial);
}

static int run_partial(const coma_tensor* x, int mode, RowsP& p, void* ws, size_t ws_bytes, hipStream_t s) {
  COMA_CHECK(x && x->data && ws, "norm: null argument");
  COMA_CHECK(ws_bytes >= coma_norm_ws_bytes(x), "norm: workspace too small (%zu < %zu)", ws_bytes, coma_norm_ws_bytes(x));
  const int vec = pick_vec8(x);
  p = make_rows(x, mode, vec);
  COMA_CHECK(p.cv <= 256, "norm: C=%d too large", x->C);
  if (x->dtype == COMA_F32) launch_partial<float>(p, vec, (double2*)ws, s);
  else launch_partial<bf16_t>(p, vec, (double2*)ws, s);
  COMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int coma_norm_stats(const coma_tensor* x, int32_t mode, float eps, float* mean, float* rstd,
                               float* running_mean, float* running_var, float momentum, void* ws,
                               size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  RowsP p;
  if (int rc = run_partial(x, mode, p, ws, ws_bytes, s)) return rc;
  const int n = p.G * p.C;
  hipLaunchKernelGGL(stats_finalize_k, dim3((n + 3) / 4), dim3(256), 0, s, (const double2*)ws, p.nchunks, p.G,
                     p.C, p.R, eps, mean, rstd, running_mean, running_var, momentum);
  COMA_LAUNCH_CHECK();
  return 0;
}

int norm_stats_finalize(const double2* partial, int nchunks, int G, int C, int64_t R, float eps, float* mean, float* rstd,
                        float* running_mean, float* running_var, float momentum, hipStream_t s) {
  const int n = G * C;
  hipLaunchKernelGGL(stats_finalize_k, dim3((n + 3) / 4), dim3(256), 0, s, partial, nchunks, G, C, R, eps, mean, rstd,
                     running_mean, running_var, momentum);
  COMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int coma_spatial_mean(const coma_tensor* x, float* out, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  RowsP p;
  if (int rc = run_partial(x, COMA_NORM_INSTANCE, p, ws, ws_bytes, s)) return rc;
  const int n = p.G * p.C;
  hipLaunchKernelGGL(colsum_finalize_k, dim3((n + 3) / 4), dim3(256), 0, s, (const double2*)ws, p.nchunks, p.G,
                     p.C, 1.0 / (double)p.R, out);
  COMA_LAUNCH_CHECK();
  return 0;
}

// used by coma_conv_wgrad for the bias gradient: out[B or 1][C] = sum over voxels (and batch)
int colsum(const coma_tensor* x, int per_sample, float* out, void* ws, size_t ws_bytes, hipStream_t s) {
  RowsP p;
  if (int rc = run_partial(x, per_sample ? COMA_NORM_INSTANCE : COMA_NORM_BATCH, p, ws, ws_bytes, s)) return rc;
  const int n = p.G * p.C;
  hipLaunchKernelGGL(colsum_finalize_k, dim3((n + 3) / 4), dim3(256), 0, s, (const double2*)ws, p.nchunks, p.G,
                     p.C, 1.0, out);
  COMA_LAUNCH_CHECK();
  return 0;
}

static ApplyP make_apply(const coma_tensor* x, const coma_tensor* y, int mode, int vec) {
  ApplyP p;
  p.x = x->data; p.ldx = x->ld; p.sbx = x->sb;
  p.y = y->data; p.ldy = y->ld; p.sby = y->sb;
  p.dy = nullptr; p.lddy = 0; p.sbdy = 0;
  p.V = t_vox(x); p.B = x->B; p.C = x->C; p.cv = x->C / vec; p.inst = mode == COMA_NORM_INSTANCE;
  return p;
}

static unsigned ew_blocks(int64_t total) {
  int64_t nb = (total + 255) / 256;
  if (nb > 4096) nb = 4096;
  return (unsigned)(nb < 1 ? 1 : nb);
}

extern "C" int coma_norm_act_fwd(const coma_tensor* x, int32_t mode, const float* mean, const float* rstd,
                                 const float* gamma, const float* beta, int32_t act, const float* slope,
                                 const coma_tensor* y, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  COMA_CHECK(x && y && x->data && y->data && mean && rstd, "norm_act_fwd: null argument");
  COMA_CHECK(t_same_grid(x, y) && x->C == y->C && x->dtype == y->dtype, "norm_act_fwd: shape/dtype mismatch");
  int vec = (pick_vec(x) == 4 && pick_vec(y) == 4) ? 4 : 1;
  if (x->dtype == COMA_BF16 && t_vec(x, 8) == 8 && t_vec(y, 8) == 8) vec = 8;     // 16 bytes per lane
  ApplyP p = make_apply(x, y, mode, vec);
  p.mean = mean; p.rstd = rstd; p.gamma = gamma; p.beta = beta; p.slope = slope; p.act = act;
  dim3 grid(ew_blocks(p.V * p.cv), x->B);
#define L(T, V) hipLaunchKernelGGL((norm_act_fwd_k<T, V>), grid, dim3(256), 0, s, p)
  if (x->dtype == COMA_F32) { if (vec == 4) L(float, 4); else L(float, 1); }
  else { if (vec == 8) L(bf16_t, 8); else if (vec == 4) L(bf16_t, 4); else L(bf16_t, 1); }
#undef L
  COMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int coma_norm_act_bwd(const coma_tensor* x, const coma_tensor* dy, int32_t mode, const float* mean,
                                 const float* rstd, const float* gamma, const float* beta, int32_t act,
                                 const float* slope, const coma_tensor* dx, float* dgamma, float* dbeta,
                                 float* dslope, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  COMA_CHECK(x && dy && dx && x->data && dy->data && dx->data && mean && rstd && ws, "norm_act_bwd: null argument");
  COMA_CHECK(t_same_grid(x, dy) && t_same_grid(x, dx) && x->C == dy->C && x->C == dx->C, "norm_act_bwd: shape mismatch");
  COMA_CHECK(x->dtype == dy->dtype && x->dtype == dx->dtype, "norm_act_bwd: dtype mismatch");
  COMA_CHECK(ws_bytes >= coma_norm_ws_bytes(x), "norm_act_bwd: workspace too small");
  const int vec = (pick_vec(x) == 4 && pick_vec(dy) == 4 && pick_vec(dx) == 4) ? 4 : 1;
  const int pvec = vec;     // (8-wide measured slower here: 24 fp64 accumulators per lane, 48 KB of LDS)
  RowsP rp = make_rows(x, mode, pvec);
  COMA_CHECK(rp.cv <= 256, "norm: C=%d too large", x->C);
  double* partial = (double*)ws;
  double* tot = (double*)((char*)ws + (size_t)1024 * x->C * 3 * sizeof(double));
  float* sums = (float*)(tot + (size_t)x->B * x->C * 3);
  dim3 pg(rp.nchunks, rp.G);
#define L(T, V) hipLaunchKernelGGL((norm_bwd_partial_k<T, V>), pg, dim3(256), 0, s, rp, dy->data, dy->ld, dy->sb, \
                                   mean, rstd, gamma, beta, act, slope, partial)
  if (x->dtype == COMA_F32) { if (pvec == 4) L(float, 4); else L(float, 1); }
  else { if (pvec == 8) L(bf16_t, 8); else if (pvec == 4) L(bf16_t, 4); else L(bf16_t, 1); }
#undef L
  COMA_LAUNCH_CHECK();
  hipLaunchKernelGGL(norm_bwd_reduce_k, dim3((rp.C + 3) / 4), dim3(256), 0, s, partial, rp.nchunks, rp.G, rp.C,
                     rp.R, tot, sums, dgamma, dbeta);
  COMA_LAUNCH_CHECK();
  if (dslope) {          // the PReLU slope is one number summed over every (group, channel): its own small launch
    hipLaunchKernelGGL(norm_bwd_finalize_k, dim3(1), dim3(256), 0, s, tot, rp.G, rp.C, (float*)nullptr, (float*)nullptr, dslope);
    COMA_LAUNCH_CHECK();
  }
  int avec = vec;
  if (x->dtype == COMA_BF16 && t_vec(x, 8) == 8 && t_vec(dy, 8) == 8 && t_vec(dx, 8) == 8) avec = 8;
  ApplyP p = make_apply(x, dx, mode, avec);
  p.dy = dy->data; p.lddy = dy->ld; p.sbdy = dy->sb;
  p.mean = mean; p.rstd = rstd; p.gamma = gamma; p.beta = beta; p.slope = slope; p.act = act;
  dim3 grid(ew_blocks(p.V * p.cv), x->B);
#define L(T, V) hipLaunchKernelGGL((norm_act_bwd_apply_k<T, V>), grid, dim3(256), 0, s, p, sums)
  if (x->dtype == COMA_F32) { if (avec == 4) L(float, 4); else L(float, 1); }
  else { if (avec == 8) L(bf16_t, 8); else if (avec == 4) L(bf16_t, 4); else L(bf16_t, 1); }
#undef L
  COMA_LAUNCH_CHECK();
  return 0;
}
