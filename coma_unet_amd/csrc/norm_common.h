// Shared by norm.hip and gate.hip: the row-walk of the statistics kernels and the fp64 statistics records
// (producers add with atomics, consumers derive mean / rstd themselves) -- see include/coma_unet.h, coma_norm_stats.
#pragma once
#include "common.h"

struct RowsP {            // a [G][R rows][C] view of a coma_tensor
  const void* x; int64_t ld, sb; int64_t V; int B; int C;
  int G;                  // groups: 1 (batch) or B (instance)
  int64_t R;              // rows per group: B*V or V
  int cv, cvp, ry;        // channel vectors, padded to pow2, rows per block step
  int64_t ch;             // rows per chunk
  int nchunks;
};

static inline RowsP make_rows(const coma_tensor* x, int mode, int vec) {
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

// ---- how a kernel obtains (mean, rstd) of (group g, channel c) -------------------------------------------------------
// Training statistics travel as fp64 {sum, sumsq} records `sums[G][C][2]` that the producers (the convolution epilogues,
// stats_partial_k) ADD to with global_atomic_add_f64 -- the caller hands over a zeroed record -- and every consumer derives
// mean / rstd from on the fly: no finalise launch between the statistics pass and the apply pass, no partial buffers.
// (fp64 sums of per-block fp64 partials: the order of the atomic adds moves the result by ~1e-16 relative, far below the
// fp32 mean / rstd every consumer rounds to.)  Eval-mode BatchNorm passes fp32 mean / rstd arrays instead.
struct NormStat {
  const double* sums; int64_t rs;      // record [COMA_STAT_REPLICAS][G][C][2], replica stride rs (doubles)
  double R; float eps;
  const float* mean; const float* rstd;
};
__device__ __forceinline__ double rec_get(const double* rec, int64_t rs, int64_t i) {      // value i summed over the replicas
  double v = 0.0;
#pragma unroll
  for (int r = 0; r < COMA_STAT_REPLICAS; ++r) v += rec[r * rs + i];
  return v;
}
__device__ __forceinline__ void norm_mr(const NormStat& q, int i, float& mu, float& rs) {
  if (q.sums) {
    const double m = rec_get(q.sums, q.rs, 2 * (int64_t)i) / q.R;
    double var = rec_get(q.sums, q.rs, 2 * (int64_t)i + 1) / q.R - m * m;
    if (var < 0.0) var = 0.0;
    mu = (float)m;
    rs = (float)(1.0 / sqrt(var + (double)q.eps));
  } else { mu = q.mean[i]; rs = q.rstd[i]; }
}
// Per-block coefficient tables in LDS: a consumer block derives the (mean, rstd) of its group's channels ONCE (16 L2
// reads per channel: 8 replicas x {sum, sumsq}) instead of once per thread.
#define NORM_TAB 1024
__device__ __forceinline__ void norm_table(const NormStat& q, int g, int C, float* t_mu, float* t_rs) {   // whole block
  for (int c = threadIdx.x; c < C; c += blockDim.x) norm_mr(q, g * C + c, t_mu[c], t_rs[c]);
  __syncthreads();
}

__device__ __forceinline__ void add_f64(double* p, double v) { unsafeAtomicAdd(p, v); }      // global_atomic_add_f64
// the block's reduced values rec_blk[0 .. n) -> replica (blockIdx.x & 7) of the record, lane i adding value i: one
// wave instruction covers 64 consecutive doubles = 8 lines, each touched once per block
__device__ __forceinline__ void rec_add(double* rec, int64_t rs, int64_t base, const double* rec_blk, int n) {
  double* dst = rec + (int64_t)(blockIdx.x & (COMA_STAT_REPLICAS - 1)) * rs + base;
  for (int t = threadIdx.x; t < n; t += blockDim.x) add_f64(dst + t, rec_blk[t]);
}

__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

