// BatchNorm3d (training statistics) / InstanceNorm3d + fused activation, forward and
// backward, for channels-last volumes.  Replaces the ATen norm + activation dispatches
// behind MONAI's ADN blocks ("N","A": BN+ReLU in ConvBlock/UpConv, IN+PReLU in the merge
// convs, IN+LeakyReLU in StackedFusionConvLayers attn_unet_data_parallel.py:480-501) and
// the bare BatchNorm3d layers of the attention gate (:141-144).  HBM-bound: one read for
// the statistics, one read + one write for the apply; statistics are accumulated in fp64
// and reduced in a fixed order (bitwise reproducible).
#include "common.h"

#include "norm_common.h"

// sums[(g*C + c)*2 .. +2] += {sum, sumsq} over the block's chunk of rows
template <typename T, int VEC>
__global__ __launch_bounds__(256) void stats_partial_k(RowsP p, double* sums, int64_t rs, double2* partial) {
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
    // straight to fp64, as the exact-fp32 mode's gradient parity needs.  Either way the burst's loads are issued
    // together (rows past the chunk re-read its last row and are masked): 4-8 requests of 16 bytes in flight per lane.
    constexpr int BURST = sizeof(T) == 2 ? 8 : 4;
    RowWalk w(p, r0 + ty);
    while (w.r < r1) {
      float v[BURST][VEC];
      bool ok[BURST];
      int64_t off = w.off(p, g, p.ld, p.sb);
#pragma unroll
      for (int u = 0; u < BURST; ++u) {
        ok[u] = w.r < r1;
        if (ok[u]) off = w.off(p, g, p.ld, p.sb);
        vec_io<T, VEC>::load(base + off + tx * VEC, v[u]);
        w.step(p);
      }
      if (sizeof(T) == 2) {
        float fs[VEC], fq[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) { fs[j] = 0.f; fq[j] = 0.f; }
#pragma unroll
        for (int u = 0; u < BURST; ++u)
#pragma unroll
          for (int j = 0; j < VEC; ++j) { const float t = ok[u] ? v[u][j] : 0.f; fs[j] += t; fq[j] = fmaf(t, t, fq[j]); }
#pragma unroll
        for (int j = 0; j < VEC; ++j) { s[j] += (double)fs[j]; q[j] += (double)fq[j]; }
      } else {
#pragma unroll
        for (int u = 0; u < BURST; ++u)
#pragma unroll
          for (int j = 0; j < VEC; ++j) { const float t = ok[u] ? v[u][j] : 0.f; s[j] += (double)t; q[j] += (double)(t * t); }
      }
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
  // row 0 of the tree holds the block's sums as sh[tx][2 VEC] = a contiguous [C][2] image
  if (sums) rec_add(sums, rs, (int64_t)g * p.C * 2, &sh[0][0], p.C * 2);
  else if (ty == 0 && tx < p.cv) {                     // (colsum / spatial mean: per-chunk partial rows)
#pragma unroll
    for (int j = 0; j < VEC; ++j)
      partial[((int64_t)blockIdx.x * p.G + g) * p.C + tx * VEC + j] = make_double2(sh[tid][2 * j], sh[tid][2 * j + 1]);
  }
}

struct ApplyP {
  const void* x; int64_t ldx, sbx;
  void* y; int64_t ldy, sby;
  const void* dy; int64_t lddy, sbdy;
  int64_t V; int B, C, cv; int inst;
  NormStat st;
  const float *gamma, *beta, *slope;
  int act;
  float *rmean, *rvar; float momentum;      // forward, BatchNorm(train): running statistics updated by block (0, 0)
  const double* bsums; int64_t brs;         // backward: [replica][G][C][3] {sum dz, sum dz*xhat, sum dy*dact/dslope}, replica stride
  float *dgamma, *dbeta, *dslope;           // backward: written by block (0, 0)
};

constexpr int NORM_UNR = 4;      // voxels per lane and loop trip of the apply kernels: their loads are issued together

template <typename T, int VEC>
__global__ __launch_bounds__(256) void norm_act_fwd_k(ApplyP p) {
  const int b = blockIdx.y;
  const int64_t total = p.V * p.cv;
  const float a = p.slope ? *p.slope : 0.25f;
  const T* xb = reinterpret_cast<const T*>(p.x) + (int64_t)b * p.sbx;
  T* yb = reinterpret_cast<T*>(p.y) + (int64_t)b * p.sby;
  const int g = p.inst ? b : 0;
  __shared__ float t_sc[NORM_TAB], t_sh[NORM_TAB];
  for (int c = threadIdx.x; c < p.C; c += 256) {
    float mu, rs; norm_mr(p.st, g * p.C + c, mu, rs);
    const float ga = p.gamma ? p.gamma[c] : 1.f;
    t_sc[c] = rs * ga;
    t_sh[c] = (p.gamma ? p.beta[c] : 0.f) - mu * rs * ga;
  }
  if (p.rmean && blockIdx.x == 0 && b == 0) {
    // BatchNorm3d(train): running_mean / running_var <- (1 - m) * old + m * {batch mean, unbiased batch variance}
    for (int c = threadIdx.x; c < p.C; c += 256) {
      const double m = rec_get(p.st.sums, p.st.rs, 2 * c) / p.st.R;
      double var = rec_get(p.st.sums, p.st.rs, 2 * c + 1) / p.st.R - m * m;
      if (var < 0.0) var = 0.0;
      const double unb = p.st.R > 1.0 ? var * p.st.R / (p.st.R - 1.0) : var;
      p.rmean[c] = (1.f - p.momentum) * p.rmean[c] + p.momentum * (float)m;
      p.rvar[c] = (1.f - p.momentum) * p.rvar[c] + p.momentum * (float)unb;
    }
  }
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * 256, e0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (stride % p.cv == 0) {
    // the thread keeps its channel group for the whole sweep: scale / shift live in registers, no division in the loop
    const int c0 = (int)(e0 % p.cv) * VEC;
    float sc[VEC], sh[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { sc[j] = t_sc[c0 + j]; sh[j] = t_sh[c0 + j]; }
    const int64_t dv = stride / p.cv;
    int64_t v = e0 / p.cv;
    for (; v + (NORM_UNR - 1) * dv < p.V; v += NORM_UNR * dv) {
      float xv[NORM_UNR][VEC];
#pragma unroll
      for (int u = 0; u < NORM_UNR; ++u) vec_io<T, VEC>::load(xb + (v + u * dv) * p.ldx + c0, xv[u]);
#pragma unroll
      for (int u = 0; u < NORM_UNR; ++u) {
        float yv[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) yv[j] = act_fwd(p.act, fmaf(xv[u][j], sc[j], sh[j]), a);
        vec_io<T, VEC>::store(yb + (v + u * dv) * p.ldy + c0, yv);
      }
    }
    for (; v < p.V; v += dv) {
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
    for (int j = 0; j < VEC; ++j) yv[j] = act_fwd(p.act, fmaf(xv[j], t_sc[c0 + j], t_sh[c0 + j]), a);
    vec_io<T, VEC>::store(yb + v * p.ldy + c0, yv);
  }
}

// backward pass 1: bsums[(g*C + c)*3 .. +3] += {sum dz, sum dz*xhat, sum dy*dact/dslope} over the block's rows
template <typename T, int VEC>
__global__ __launch_bounds__(256) void norm_bwd_partial_k(RowsP p, const void* dyp, int64_t lddy, int64_t sbdy,
                                                          NormStat st, const float* gamma, const float* beta, int act,
                                                          const float* slope, double* bsums, int64_t brs) {
  __shared__ double sh[256][3 * VEC];
  const int tid = threadIdx.x, tx = tid % p.cvp, ty = tid / p.cvp;
  const int g = blockIdx.y;
  const int64_t r0 = (int64_t)blockIdx.x * p.ch;
  const int64_t r1 = r0 + p.ch < p.R ? r0 + p.ch : p.R;
  const float a = slope ? *slope : 0.25f;
  double s1[VEC], s2[VEC], s3[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { s1[j] = s2[j] = s3[j] = 0.0; }
  __shared__ float t_mu[NORM_TAB], t_rs[NORM_TAB];
  norm_table(st, g, p.C, t_mu, t_rs);
  if (tx < p.cv) {
    const T* xb = reinterpret_cast<const T*>(p.x);
    const T* dyb = reinterpret_cast<const T*>(dyp);
    float mu[VEC], rs[VEC], ga[VEC], be[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int c = tx * VEC + j;
      mu[j] = t_mu[c]; rs[j] = t_rs[c];
      ga[j] = gamma ? gamma[c] : 1.f; be[j] = gamma ? beta[c] : 0.f;
    }
    // bf16: fp32 partial sums over bursts of 8 rows folded into fp64; fp32: every term straight to fp64 -- see
    // stats_partial_k.  Two rows (of both tensors) are in flight per lane: more costs registers, and with them the
    // occupancy a streaming kernel lives on (8 rows at once: 188 VGPRs, 2 waves per SIMD, half the bandwidth).
    constexpr int BURST = sizeof(T) == 2 ? 8 : 2, LD = 2;
    RowWalk w(p, r0 + ty);
    while (w.r < r1) {
      float f1[VEC], f2[VEC], f3[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) { f1[j] = 0.f; f2[j] = 0.f; f3[j] = 0.f; }
      // offsets of the last VALID row: a row past the chunk re-reads it (masked) -- computed here, where w.r < r1 holds; the
      // walk itself may step past the tensor (sample index B) and its offsets must never be dereferenced then
      int64_t lx = w.off(p, g, p.ld, p.sb), ldd = w.off(p, g, lddy, sbdy);
#pragma unroll 1
      for (int h = 0; h < BURST / LD; ++h) {      // (not unrolled: the compiler would hoist all 8 rows' loads to the top)
        float xv[LD][VEC], dv[LD][VEC];
        bool ok[LD];
#pragma unroll
        for (int u = 0; u < LD; ++u) {
          ok[u] = w.r < r1;
          if (ok[u]) { lx = w.off(p, g, p.ld, p.sb); ldd = w.off(p, g, lddy, sbdy); }
          vec_io<T, VEC>::load(xb + lx + tx * VEC, xv[u]);
          vec_io<T, VEC>::load(dyb + ldd + tx * VEC, dv[u]);
          w.step(p);
        }
#pragma unroll
        for (int u = 0; u < LD; ++u) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            const float xh = (xv[u][j] - mu[j]) * rs[j];
            const float z = xh * ga[j] + be[j];
            float ds; const float da = act_bwd(act, z, a, &ds);
            const float dvv = ok[u] ? dv[u][j] : 0.f;
            const float dz = dvv * da;
            if (sizeof(T) == 2) { f1[j] += dz; f2[j] = fmaf(dz, xh, f2[j]); f3[j] = fmaf(dvv, ds, f3[j]); }
            else { s1[j] += (double)dz; s2[j] += (double)(dz * xh); s3[j] += (double)(dvv * ds); }
          }
        }
      }
      if (sizeof(T) == 2) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) { s1[j] += (double)f1[j]; s2[j] += (double)f2[j]; s3[j] += (double)f3[j]; }
      }
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
  rec_add(bsums, brs, (int64_t)g * p.C * 3, &sh[0][0], p.C * 3);      // row 0 of the tree = a contiguous [C][3] image
}

// backward pass 2: dx = rstd * gamma * (dz - mean(dz) - xhat * mean(dz * xhat)); block (0, 0) also writes the parameter
// gradients dgamma / dbeta[c] = sum over the groups, dslope = sum over everything (<= 1024 numbers)
template <typename T, int VEC>
__global__ __launch_bounds__(256) void norm_act_bwd_apply_k(ApplyP p) {
  const int b = blockIdx.y;
  const int64_t total = p.V * p.cv;
  const float a = p.slope ? *p.slope : 0.25f;
  const T* xb = reinterpret_cast<const T*>(p.x) + (int64_t)b * p.sbx;
  const T* dyb = reinterpret_cast<const T*>(p.dy) + (int64_t)b * p.sbdy;
  T* ob = reinterpret_cast<T*>(p.y) + (int64_t)b * p.sby;
  const int g = p.inst ? b : 0;
  const int G = p.inst ? p.B : 1;
  __shared__ float t_mu[NORM_TAB], t_rs[NORM_TAB], t_s1[NORM_TAB], t_s2[NORM_TAB];
  for (int c = threadIdx.x; c < p.C; c += 256) {
    norm_mr(p.st, g * p.C + c, t_mu[c], t_rs[c]);
    t_s1[c] = (float)(rec_get(p.bsums, p.brs, ((int64_t)g * p.C + c) * 3) / p.st.R);
    t_s2[c] = (float)(rec_get(p.bsums, p.brs, ((int64_t)g * p.C + c) * 3 + 1) / p.st.R);
  }
  if (blockIdx.x == 0 && b == 0 && (p.dgamma || p.dbeta || p.dslope)) {
    __shared__ double ssl[256];
    double sl = 0.0;
    for (int c = threadIdx.x; c < p.C; c += 256) {
      double dg = 0.0, db = 0.0;
      for (int gg = 0; gg < G; ++gg) {
        const int64_t i = ((int64_t)gg * p.C + c) * 3;
        db += rec_get(p.bsums, p.brs, i); dg += rec_get(p.bsums, p.brs, i + 1); sl += rec_get(p.bsums, p.brs, i + 2);
      }
      if (p.dgamma) p.dgamma[c] = (float)dg;
      if (p.dbeta) p.dbeta[c] = (float)db;
    }
    if (p.dslope) {
      ssl[threadIdx.x] = sl;
      __syncthreads();
      for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) ssl[threadIdx.x] += ssl[threadIdx.x + o]; __syncthreads(); }
      if (threadIdx.x == 0) *p.dslope = (float)ssl[0];
    }
  }
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * 256, e0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (stride % p.cv == 0) {      // fixed channel group per thread: the per-channel constants live in registers
    const int c0 = (int)(e0 % p.cv) * VEC;
    float rs[VEC], mu[VEC], ga[VEC], be[VEC], s1[VEC], s2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int c = c0 + j;
      mu[j] = t_mu[c]; rs[j] = t_rs[c]; s1[j] = t_s1[c]; s2[j] = t_s2[c];
      ga[j] = p.gamma ? p.gamma[c] : 1.f; be[j] = p.gamma ? p.beta[c] : 0.f;
    }
    auto one = [&](const float* xv, const float* dv, float* ov) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float xh = (xv[j] - mu[j]) * rs[j];
        const float z = p.gamma ? xh * ga[j] + be[j] : xh;
        float ds; const float da = act_bwd(p.act, z, a, &ds);
        const float dz = dv[j] * da;
        ov[j] = rs[j] * ga[j] * (dz - s1[j] - xh * s2[j]);
      }
    };
    const int64_t dvx = stride / p.cv;
    int64_t v = e0 / p.cv;
    constexpr int U = NORM_UNR / 2 > 0 ? NORM_UNR / 2 : 1;       // two tensors are read: half the voxels per trip
    for (; v + (U - 1) * dvx < p.V; v += U * dvx) {
      float xv[U][VEC], dv[U][VEC];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        vec_io<T, VEC>::load(xb + (v + u * dvx) * p.ldx + c0, xv[u]);
        vec_io<T, VEC>::load(dyb + (v + u * dvx) * p.lddy + c0, dv[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float ov[VEC];
        one(xv[u], dv[u], ov);
        vec_io<T, VEC>::store(ob + (v + u * dvx) * p.ldy + c0, ov);
      }
    }
    for (; v < p.V; v += dvx) {
      float xv[VEC], dv[VEC], ov[VEC];
      vec_io<T, VEC>::load(xb + v * p.ldx + c0, xv);
      vec_io<T, VEC>::load(dyb + v * p.lddy + c0, dv);
      one(xv, dv, ov);
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
      const float rs = t_rs[c], xh = (xv[j] - t_mu[c]) * rs;
      const float ga = p.gamma ? p.gamma[c] : 1.f;
      const float z = p.gamma ? xh * ga + p.beta[c] : xh;
      float ds; const float da = act_bwd(p.act, z, a, &ds);
      const float dz = dv[j] * da;
      ov[j] = rs * ga * (dz - t_s1[c] - xh * t_s2[c]);
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
  // column sums (bias gradient, spatial mean): <= 1024 (chunk, group) partial rows of C {sum, sumsq} pairs
  return (size_t)1024 * x->C * sizeof(double2) + 256;
}

template <typename T>
static void launch_partial(const RowsP& p, int vec, double* sums, int64_t rs, double2* partial, hipStream_t s) {
  dim3 grid(p.nchunks, p.G);
  if (vec == 8) { if constexpr (sizeof(T) == 2) hipLaunchKernelGGL((stats_partial_k<T, 8>), grid, dim3(256), 0, s, p, sums, rs, partial); }
  else if (vec == 4) hipLaunchKernelGGL((stats_partial_k<T, 4>), grid, dim3(256), 0, s, p, sums, rs, partial);
  else hipLaunchKernelGGL((stats_partial_k<T, 1>), grid, dim3(256), 0, s, p, sums, rs, partial);
}

// sums != NULL: atomic adds into the caller's zeroed [G][C][2] record; else per-chunk partial rows in `ws`
static int run_partial(const coma_tensor* x, int mode, RowsP& p, double* sums, void* ws, size_t ws_bytes, hipStream_t s) {
  COMA_CHECK(x && x->data && (sums || ws), "norm: null argument");
  COMA_CHECK(sums || ws_bytes >= coma_norm_ws_bytes(x), "norm: workspace too small (%zu < %zu)", ws_bytes, coma_norm_ws_bytes(x));
  const int vec = pick_vec8(x);
  p = make_rows(x, mode, vec);
  COMA_CHECK(p.cv <= 256, "norm: C=%d too large", x->C);
  COMA_CHECK(x->C <= NORM_TAB, "norm: C=%d above the coefficient table size", x->C);
  const int64_t rs = COMA_NORM_RECORD_DOUBLES(p.G, p.C, 2);
  if (x->dtype == COMA_F32) launch_partial<float>(p, vec, sums, rs, (double2*)ws, s);
  else launch_partial<bf16_t>(p, vec, sums, rs, (double2*)ws, s);
  COMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int coma_norm_stats(const coma_tensor* x, int32_t mode, double* sums, void* stream) {
  COMA_CHECK(sums, "norm_stats: null sums record");
  RowsP p;
  return run_partial(x, mode, p, sums, nullptr, 0, (hipStream_t)stream);
}

extern "C" int coma_spatial_mean(const coma_tensor* x, float* out, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  RowsP p;
  if (int rc = run_partial(x, COMA_NORM_INSTANCE, p, nullptr, ws, ws_bytes, s)) return rc;
  const int n = p.G * p.C;
  hipLaunchKernelGGL(colsum_finalize_k, dim3((n + 3) / 4), dim3(256), 0, s, (const double2*)ws, p.nchunks, p.G,
                     p.C, 1.0 / (double)p.R, out);
  COMA_LAUNCH_CHECK();
  return 0;
}

// used by coma_conv_wgrad for the bias gradient: out[B or 1][C] = sum over voxels (and batch)
int colsum(const coma_tensor* x, int per_sample, float* out, void* ws, size_t ws_bytes, hipStream_t s) {
  RowsP p;
  if (int rc = run_partial(x, per_sample ? COMA_NORM_INSTANCE : COMA_NORM_BATCH, p, nullptr, ws, ws_bytes, s)) return rc;
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
  p.rmean = p.rvar = nullptr; p.momentum = 0.f; p.bsums = nullptr; p.brs = 0; p.dgamma = p.dbeta = p.dslope = nullptr;
  return p;
}

static NormStat make_stat(const coma_tensor* x, int mode, const double* sums, float eps, const float* mean, const float* rstd) {
  NormStat st;
  st.sums = sums; st.eps = eps; st.mean = mean; st.rstd = rstd;
  st.rs = COMA_NORM_RECORD_DOUBLES(mode == COMA_NORM_INSTANCE ? x->B : 1, x->C, 2);
  st.R = (double)(mode == COMA_NORM_INSTANCE ? t_vox(x) : t_vox(x) * x->B);
  return st;
}

static int norm_knob(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
static unsigned ew_blocks(int64_t total, int per_thread) {
  static const int cap = norm_knob("COMA_NORM_BLOCKS", 4096);
  int64_t nb = (total + 256 * (int64_t)per_thread - 1) / (256 * (int64_t)per_thread);
  if (nb > cap) nb = cap;
  return (unsigned)(nb < 1 ? 1 : nb);
}
// the apply kernels keep a thread on ONE channel group when the grid stride is a multiple of the groups per voxel
static unsigned apply_blocks(int64_t total, int cv, int per_thread) {
  unsigned nb = ew_blocks(total, per_thread);
  if (cv > 0 && cv <= 256 && (256 % cv) == 0) return nb;                    // 256 % cv == 0: any block count works
  while (nb > 1 && ((int64_t)nb * 256) % cv != 0) --nb;
  return nb;
}

extern "C" int coma_norm_act_fwd(const coma_tensor* x, int32_t mode, const double* sums, float eps, const float* mean,
                                 const float* rstd, const float* gamma, const float* beta, int32_t act, const float* slope,
                                 float* running_mean, float* running_var, float momentum, const coma_tensor* y, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  COMA_CHECK(x && y && x->data && y->data && (sums || (mean && rstd)), "norm_act_fwd: null argument");
  COMA_CHECK(t_same_grid(x, y) && x->C == y->C && x->dtype == y->dtype, "norm_act_fwd: shape/dtype mismatch");
  COMA_CHECK(x->C <= NORM_TAB, "norm: C=%d above the coefficient table size", x->C);
  COMA_CHECK(!running_mean || (sums && running_var && mode == COMA_NORM_BATCH), "norm_act_fwd: running statistics need a BatchNorm sums record");
  int vec = (pick_vec(x) == 4 && pick_vec(y) == 4) ? 4 : 1;
  if (x->dtype == COMA_BF16 && t_vec(x, 8) == 8 && t_vec(y, 8) == 8) vec = 8;     // 16 bytes per lane
  ApplyP p = make_apply(x, y, mode, vec);
  p.st = make_stat(x, mode, sums, eps, mean, rstd);
  p.gamma = gamma; p.beta = beta; p.slope = slope; p.act = act;
  p.rmean = running_mean; p.rvar = running_var; p.momentum = momentum;
  dim3 grid(apply_blocks(p.V * p.cv, p.cv, NORM_UNR), x->B);
#define L(T, V) hipLaunchKernelGGL((norm_act_fwd_k<T, V>), grid, dim3(256), 0, s, p)
  if (x->dtype == COMA_F32) { if (vec == 4) L(float, 4); else L(float, 1); }
  else { if (vec == 8) L(bf16_t, 8); else if (vec == 4) L(bf16_t, 4); else L(bf16_t, 1); }
#undef L
  COMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int coma_norm_act_bwd(const coma_tensor* x, const coma_tensor* dy, int32_t mode, const double* sums, float eps,
                                 const float* gamma, const float* beta, int32_t act, const float* slope,
                                 const coma_tensor* dx, float* dgamma, float* dbeta, float* dslope, double* bsums,
                                 void* stream) {
  hipStream_t s = (hipStream_t)stream;
  COMA_CHECK(x && dy && dx && x->data && dy->data && dx->data && sums && bsums, "norm_act_bwd: null argument");
  COMA_CHECK(t_same_grid(x, dy) && t_same_grid(x, dx) && x->C == dy->C && x->C == dx->C, "norm_act_bwd: shape mismatch");
  COMA_CHECK(x->dtype == dy->dtype && x->dtype == dx->dtype, "norm_act_bwd: dtype mismatch");
  const int vec = (pick_vec(x) == 4 && pick_vec(dy) == 4 && pick_vec(dx) == 4) ? 4 : 1;
  const int pvec = vec;     // (8-wide measured slower here: 24 fp64 accumulators per lane, 48 KB of LDS)
  RowsP rp = make_rows(x, mode, pvec);
  COMA_CHECK(rp.cv <= 256, "norm: C=%d too large", x->C);
  COMA_CHECK(x->C <= NORM_TAB, "norm: C=%d above the coefficient table size", x->C);
  const NormStat st = make_stat(x, mode, sums, eps, nullptr, nullptr);
  const int64_t brs = COMA_NORM_RECORD_DOUBLES(rp.G, rp.C, 3);
  dim3 pg(rp.nchunks, rp.G);
#define L(T, V) hipLaunchKernelGGL((norm_bwd_partial_k<T, V>), pg, dim3(256), 0, s, rp, dy->data, dy->ld, dy->sb, \
                                   st, gamma, beta, act, slope, bsums, brs)
  if (x->dtype == COMA_F32) { if (pvec == 4) L(float, 4); else L(float, 1); }
  else { if (pvec == 4) L(bf16_t, 4); else L(bf16_t, 1); }
#undef L
  COMA_LAUNCH_CHECK();
  int avec = vec;
  if (x->dtype == COMA_BF16 && t_vec(x, 8) == 8 && t_vec(dy, 8) == 8 && t_vec(dx, 8) == 8) avec = 8;
  ApplyP p = make_apply(x, dx, mode, avec);
  p.dy = dy->data; p.lddy = dy->ld; p.sbdy = dy->sb;
  p.st = st;
  p.gamma = gamma; p.beta = beta; p.slope = slope; p.act = act;
  p.bsums = bsums; p.brs = brs; p.dgamma = dgamma; p.dbeta = dbeta; p.dslope = dslope;
  dim3 grid(apply_blocks(p.V * p.cv, p.cv, NORM_UNR / 2), x->B);
#define L(T, V) hipLaunchKernelGGL((norm_act_bwd_apply_k<T, V>), grid, dim3(256), 0, s, p)
  if (x->dtype == COMA_F32) { if (avec == 4) L(float, 4); else L(float, 1); }
  else { if (avec == 8) L(bf16_t, 8); else if (avec == 4) L(bf16_t, 4); else L(bf16_t, 1); }
#undef L
  COMA_LAUNCH_CHECK();
  return 0;
}
