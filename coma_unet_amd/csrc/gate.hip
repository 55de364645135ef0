// The attention gate between its 1x1x1 convolutions and its output, fused (MONAI AttentionBlock as used by
// attn_unet_data_parallel.py:139-150):
//
//     g1 = BN_g(W_g g)   x1 = BN_x(W_x x)   s = relu(g1 + x1)   psi = sigmoid(BN_psi(w_psi . s + b_psi))   att = x * psi
//
// W_g / W_x stay MFMA convolutions with their BatchNorm statistics out of the epilogue (conv_mfma.hip); everything behind
// them is HBM-bound element-wise work that used to be eight launches forward (2 x BN apply, add-relu, psi dot product,
// psi statistics + finalise, sigmoid-BN apply, multiply) and fourteen backward, each re-reading what the previous one
// wrote.  Here: TWO launches forward, THREE backward.
//   gate_mid_fwd_k    g1raw, x1raw -> s (saved for backward), psi_raw, {sum, sumsq}(psi_raw)        1.5 F reads... per voxel
//   gate_apply_fwd_k  x, psi_raw   -> psi (saved), att = x * psi (written into the concat slice)
//   gate_apply_bwd_k  x, psi, psi_raw, d(att) -> dx (= or +=), dz = d(psi) psi (1 - psi), BN_psi backward sums
//   gate_mid_bwd_partial_k / _apply_k: dz -> d(psi_raw) -> ds -> relu mask -> the two BatchNorm backwards at once (they
//   share d(s)), d(w_psi) and every parameter gradient; only dg1raw / dx1raw are written.
// All BatchNorms are training-mode with statistics records (norm_common.h); R = B * V elements per channel.
#include "norm_common.h"

struct GateMidP {
  const void* g1; int64_t ldg, sbg;
  const void* x1; int64_t ldx, sbx;
  void* s; int64_t lds_, sbs;
  void* pr; int64_t ldp, sbp;          // psi_raw [B][V][1]
  const void* dz; int64_t lddz, sbdz;  // backward: d(BN_psi output pre-sigmoid ... ) see gate_apply_bwd_k
  void* dg1; int64_t lddg, sbdg;
  void* dx1; int64_t lddx, sbdx;
  int64_t V; int B, F, cv;
  NormStat stg, stx, stp;              // statistics of g1raw, x1raw, psi_raw
  const float *gam_g, *bet_g, *gam_x, *bet_x, *gam_p, *bet_p;
  const float* w; const float* bias;   // psi convolution: w[F], bias[1] or NULL
  double* psums; int64_t prs;          // forward: zeroed record of psi_raw [rep][1][1][2]
  const double* pbs; int64_t pbrs;     // backward: BN_psi backward sums [rep][1][1][3]
  double* rec; int64_t rrs;            // backward: zeroed record [rep][F][4] = {sum d, sum d ghat, sum d xhat, sum dpr s}
  float *rm_g, *rv_g, *rm_x, *rv_x; float momentum;
  float *dgam_g, *dbet_g, *dgam_x, *dbet_x, *dw, *dgam_p, *dbet_p;
};

// BatchNorm3d(train) running statistics from a record (one block calls this)
__device__ __forceinline__ void bn_running_update(const NormStat& st, int C, float* rm, float* rv, float momentum) {
  if (!rm) return;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const double m = rec_get(st.sums, st.rs, 2 * c) / st.R;
    double var = rec_get(st.sums, st.rs, 2 * c + 1) / st.R - m * m;
    if (var < 0.0) var = 0.0;
    const double unb = st.R > 1.0 ? var * st.R / (st.R - 1.0) : var;
    rm[c] = (1.f - momentum) * rm[c] + momentum * (float)m;
    rv[c] = (1.f - momentum) * rv[c] + momentum * (float)unb;
  }
}

#define GATE_TAB 512      // F <= 512 (the model: 16 .. 128)

template <typename T, int VEC>
__global__ __launch_bounds__(256) void gate_mid_fwd_k(GateMidP p) {
  __shared__ float t_scg[GATE_TAB], t_scx[GATE_TAB], t_sh[GATE_TAB], t_w[GATE_TAB];
  __shared__ double red[2][4];
  const int b = blockIdx.y, tid = threadIdx.x;
  for (int c = tid; c < p.F; c += 256) {
    float mg, rg, mx, rx;
    norm_mr(p.stg, c, mg, rg); norm_mr(p.stx, c, mx, rx);
    const float sg = rg * p.gam_g[c], sx = rx * p.gam_x[c];
    t_scg[c] = sg; t_scx[c] = sx;
    t_sh[c] = (p.bet_g[c] - mg * sg) + (p.bet_x[c] - mx * sx);
    t_w[c] = p.w[c];
  }
  if (blockIdx.x == 0 && b == 0) {
    bn_running_update(p.stg, p.F, p.rm_g, p.rv_g, p.momentum);
    bn_running_update(p.stx, p.F, p.rm_x, p.rv_x, p.momentum);
  }
  __syncthreads();
  const int cv = p.cv;                                   // lanes per voxel: a power of two <= 64
  const int c0 = (tid % cv) * VEC;
  float scg[VEC], scx[VEC], sh[VEC], w[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { scg[j] = t_scg[c0 + j]; scx[j] = t_scx[c0 + j]; sh[j] = t_sh[c0 + j]; w[j] = t_w[c0 + j]; }
  const float bias = p.bias ? *p.bias : 0.f;
  const T* gb = reinterpret_cast<const T*>(p.g1) + (int64_t)b * p.sbg;
  const T* xb = reinterpret_cast<const T*>(p.x1) + (int64_t)b * p.sbx;
  T* sb = reinterpret_cast<T*>(p.s) + (int64_t)b * p.sbs;
  T* pb = reinterpret_cast<T*>(p.pr) + (int64_t)b * p.sbp;
  const int vpb = 256 / cv;                              // voxels per block step
  float fs = 0.f, fq = 0.f;
  double ds = 0.0, dq = 0.0;
  int cnt = 0;
  const int64_t vstep = (int64_t)gridDim.x * vpb;
  int64_t v = (int64_t)blockIdx.x * vpb + tid / cv;
  constexpr int U = 2;
  auto one = [&](int64_t vv, const float* gv, const float* xv, bool live) {
    float sv[VEC], dot = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float t = fmaf(gv[j], scg[j], fmaf(xv[j], scx[j], sh[j]));
      sv[j] = t > 0.f ? t : 0.f;
    }
    if (live) vec_io<T, VEC>::store(sb + vv * p.lds_ + c0, sv);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      // the product the stand-alone psi convolution would form: the STORED (rounded) s times w
      float sr[1]; T tmp = static_cast<T>(sv[j]); sr[0] = static_cast<float>(tmp);
      dot = fmaf(sr[0], w[j], dot);
    }
    for (int o = cv >> 1; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
    if (live && c0 == 0) {
      const T q = static_cast<T>(dot + bias);
      pb[vv * p.ldp] = q;
      const float r = static_cast<float>(q);
      fs += r; fq = fmaf(r, r, fq);
    }
  };
  for (; v < p.V; v += U * vstep) {                      // (the shuffles need whole waves: every lane runs every trip)
    float gv[U][VEC], xv[U][VEC];
    bool live[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t vv = v + u * vstep;
      live[u] = vv < p.V;
      const int64_t va = live[u] ? vv : v;
      vec_io<T, VEC>::load(gb + va * p.ldg + c0, gv[u]);
      vec_io<T, VEC>::load(xb + va * p.ldx + c0, xv[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) one(v + u * vstep, gv[u], xv[u], live[u]);
    if (++cnt == 8) { ds += (double)fs; dq += (double)fq; fs = 0.f; fq = 0.f; cnt = 0; }
  }
  ds += (double)fs; dq += (double)fq;
  // block reduction of the two psi_raw sums (only the lanes with c0 == 0 hold any) -> one atomic pair per block
  ds = wave_sum(ds); dq = wave_sum(dq);
  if ((tid & 63) == 0) { red[0][tid >> 6] = ds; red[1][tid >> 6] = dq; }
  __syncthreads();
  if (tid < 2) {
    const double t = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
    add_f64(p.psums + (int64_t)(blockIdx.x & (COMA_STAT_REPLICAS - 1)) * p.prs + tid, t);
  }
}

struct GateApplyP {
  const void* x; int64_t ldx, sbx;
  const void* pr; int64_t ldp, sbp;        // psi_raw
  void* psi; int64_t ldq, sbq;             // psi (forward: out; backward: in)
  void* att; int64_t lda, sba;             // forward: x * psi
  const void* dout; int64_t ldo, sbo;      // backward: d(att)
  void* dx; int64_t lddx, sbdx; int acc;   // backward: dx (=) or (+=) dout * psi
  void* dz; int64_t lddz, sbdz;            // backward: dz [B][V][1]
  int64_t V; int B, C, cv;
  NormStat stp; const float *gam_p, *bet_p;
  float *rm_p, *rv_p; float momentum;
  double* pbs; int64_t pbrs;               // backward: zeroed record [rep][1][1][3]
};

template <typename T, int VEC>
__global__ __launch_bounds__(256) void gate_apply_fwd_k(GateApplyP p) {
  const int b = blockIdx.y, tid = threadIdx.x;
  float mu, rs; norm_mr(p.stp, 0, mu, rs);
  const float sc = rs * p.gam_p[0], sh = p.bet_p[0] - mu * sc;
  if (blockIdx.x == 0 && b == 0) bn_running_update(p.stp, 1, p.rm_p, p.rv_p, p.momentum);
  const int cv = p.cv, c0 = (tid % cv) * VEC;
  const T* xb = reinterpret_cast<const T*>(p.x) + (int64_t)b * p.sbx;
  const T* rb = reinterpret_cast<const T*>(p.pr) + (int64_t)b * p.sbp;
  T* qb = reinterpret_cast<T*>(p.psi) + (int64_t)b * p.sbq;
  T* ab = reinterpret_cast<T*>(p.att) + (int64_t)b * p.sba;
  const int64_t total = p.V * cv, stride = (int64_t)gridDim.x * 256;
  constexpr int U = 4;
  int64_t e = (int64_t)blockIdx.x * 256 + tid;
  for (; e + (U - 1) * stride < total; e += U * stride) {
    float xv[U][VEC], pr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t v = (e + u * stride) / cv;
      vec_io<T, VEC>::load(xb + v * p.ldx + c0, xv[u]);
      pr[u] = ld_f(rb + v * p.ldp);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t v = (e + u * stride) / cv;
      const T q = static_cast<T>(1.f / (1.f + expf(-fmaf(pr[u], sc, sh))));
      const float ps = static_cast<float>(q);            // the stored psi (what the stand-alone multiply would read)
      if (c0 == 0) qb[v * p.ldq] = q;
      float ov[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) ov[j] = xv[u][j] * ps;
      vec_io<T, VEC>::store(ab + v * p.lda + c0, ov);
    }
  }
  for (; e < total; e += stride) {
    const int64_t v = e / cv;
    float xv[VEC], ov[VEC];
    vec_io<T, VEC>::load(xb + v * p.ldx + c0, xv);
    const T q = static_cast<T>(1.f / (1.f + expf(-fmaf(ld_f(rb + v * p.ldp), sc, sh))));
    const float ps = static_cast<float>(q);
    if (c0 == 0) qb[v * p.ldq] = q;
#pragma unroll
    for (int j = 0; j < VEC; ++j) ov[j] = xv[j] * ps;
    vec_io<T, VEC>::store(ab + v * p.lda + c0, ov);
  }
}

// d(att) -> dx (=|+=) d(att) * psi;  dpsi = sum_c d(att) x;  dz = dpsi * psi (1 - psi) (the gradient at the sigmoid's
// input = BN_psi's output);  BN_psi backward sums {sum dz, sum dz psihat}
template <typename T, int VEC>
__global__ __launch_bounds__(256) void gate_apply_bwd_k(GateApplyP p) {
  __shared__ double red[2][4];
  const int b = blockIdx.y, tid = threadIdx.x;
  float mu, rs; norm_mr(p.stp, 0, mu, rs);
  const int cv = p.cv, c0 = (tid % cv) * VEC;
  const T* xb = reinterpret_cast<const T*>(p.x) + (int64_t)b * p.sbx;
  const T* rb = reinterpret_cast<const T*>(p.pr) + (int64_t)b * p.sbp;
  const T* qb = reinterpret_cast<const T*>(p.psi) + (int64_t)b * p.sbq;
  const T* gb = reinterpret_cast<const T*>(p.dout) + (int64_t)b * p.sbo;
  T* dxb = reinterpret_cast<T*>(p.dx) + (int64_t)b * p.sbdx;
  T* dzb = reinterpret_cast<T*>(p.dz) + (int64_t)b * p.sbdz;
  const int vpb = 256 / cv;
  const int64_t vstep = (int64_t)gridDim.x * vpb;
  float f1 = 0.f, f2 = 0.f;
  double s1 = 0.0, s2 = 0.0;
  int cnt = 0;
  constexpr int U = 2;
  for (int64_t v = (int64_t)blockIdx.x * vpb + tid / cv; v < p.V; v += U * vstep) {
    float xv[U][VEC], gv[U][VEC], ov[U][VEC], ps[U], pr[U];
    bool live[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t vv = v + u * vstep;
      live[u] = vv < p.V;
      const int64_t va = live[u] ? vv : v;
      vec_io<T, VEC>::load(xb + va * p.ldx + c0, xv[u]);
      vec_io<T, VEC>::load(gb + va * p.ldo + c0, gv[u]);
      if (p.acc) vec_io<T, VEC>::load(dxb + va * p.lddx + c0, ov[u]);
      ps[u] = ld_f(qb + va * p.ldq); pr[u] = ld_f(rb + va * p.ldp);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t vv = v + u * vstep;
      float dot = 0.f;
#pragma unroll
      for (int j = 0; j < VEC; ++j) { dot = fmaf(xv[u][j], gv[u][j], dot); ov[u][j] = (p.acc ? ov[u][j] : 0.f) + gv[u][j] * ps[u]; }
      if (live[u]) vec_io<T, VEC>::store(dxb + vv * p.lddx + c0, ov[u]);
      for (int o = cv >> 1; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
      if (live[u] && c0 == 0) {
        const T q = static_cast<T>(dot * ps[u] * (1.f - ps[u]));
        dzb[vv * p.lddz] = q;
        const float dz = static_cast<float>(q);
        f1 += dz; f2 = fmaf(dz, (pr[u] - mu) * rs, f2);
      }
    }
    if (++cnt == 8) { s1 += (double)f1; s2 += (double)f2; f1 = 0.f; f2 = 0.f; cnt = 0; }
  }
  s1 += (double)f1; s2 += (double)f2;
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  if ((tid & 63) == 0) { red[0][tid >> 6] = s1; red[1][tid >> 6] = s2; }
  __syncthreads();
  if (tid < 2) {
    const double t = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
    add_f64(p.pbs + (int64_t)(blockIdx.x & (COMA_STAT_REPLICAS - 1)) * p.pbrs + tid, t);
  }
}

// ---- backward of the middle: per-channel sums {sum d, sum d ghat, sum d xhat, sum dpr s}, d = dpr * w * [s > 0],
// dpr = d(psi_raw) = gamma_psi rstd_psi (dz - mean(dz) - psihat mean(dz psihat)) ----
template <typename T, int VEC>
__global__ __launch_bounds__(256) void gate_mid_bwd_partial_k(RowsP rp, GateMidP p) {
  __shared__ double sh[256][4 * VEC];
  __shared__ float t_mg[GATE_TAB], t_rg[GATE_TAB], t_mx[GATE_TAB], t_rx[GATE_TAB], t_w[GATE_TAB];
  const int tid = threadIdx.x, tx = tid % rp.cvp, ty = tid / rp.cvp;
  for (int c = tid; c < p.F; c += 256) {
    norm_mr(p.stg, c, t_mg[c], t_rg[c]); norm_mr(p.stx, c, t_mx[c], t_rx[c]);
    t_w[c] = p.w[c];
  }
  float mp, rsp; norm_mr(p.stp, 0, mp, rsp);
  const float cp = p.gam_p[0] * rsp;
  const float m1 = (float)(rec_get(p.pbs, p.pbrs, 0) / p.stp.R), m2 = (float)(rec_get(p.pbs, p.pbrs, 1) / p.stp.R);
  __syncthreads();
  const int64_t r0 = (int64_t)blockIdx.x * rp.ch;
  const int64_t r1 = r0 + rp.ch < rp.R ? r0 + rp.ch : rp.R;
  double acc[4 * VEC];
#pragma unroll
  for (int j = 0; j < 4 * VEC; ++j) acc[j] = 0.0;
  if (tx < rp.cv) {
    const T* gb = reinterpret_cast<const T*>(p.g1);
    const T* xb = reinterpret_cast<const T*>(p.x1);
    const T* sb = reinterpret_cast<const T*>(p.s);
    const T* zb = reinterpret_cast<const T*>(p.dz);
    const T* rb = reinterpret_cast<const T*>(p.pr);
    float mg[VEC], rg[VEC], mx[VEC], rx[VEC], w[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int c = tx * VEC + j;
      mg[j] = t_mg[c]; rg[j] = t_rg[c]; mx[j] = t_mx[c]; rx[j] = t_rx[c]; w[j] = t_w[c];
    }
    constexpr int BURST = 4;
    RowWalk wk(rp, r0 + ty);
    while (wk.r < r1) {
      float gv[BURST][VEC], xv[BURST][VEC], sv[BURST][VEC], dz[BURST], pr[BURST];
      bool ok[BURST];
      int64_t og = wk.off(rp, 0, p.ldg, p.sbg), ox = wk.off(rp, 0, p.ldx, p.sbx), os = wk.off(rp, 0, p.lds_, p.sbs),
              oz = wk.off(rp, 0, p.lddz, p.sbdz), opr = wk.off(rp, 0, p.ldp, p.sbp);
#pragma unroll
      for (int u = 0; u < BURST; ++u) {
        ok[u] = wk.r < r1;
        if (ok[u]) { og = wk.off(rp, 0, p.ldg, p.sbg); ox = wk.off(rp, 0, p.ldx, p.sbx); os = wk.off(rp, 0, p.lds_, p.sbs);
                     oz = wk.off(rp, 0, p.lddz, p.sbdz); opr = wk.off(rp, 0, p.ldp, p.sbp); }
        vec_io<T, VEC>::load(gb + og + tx * VEC, gv[u]);
        vec_io<T, VEC>::load(xb + ox + tx * VEC, xv[u]);
        vec_io<T, VEC>::load(sb + os + tx * VEC, sv[u]);
        dz[u] = ld_f(zb + oz); pr[u] = ld_f(rb + opr);
        wk.step(rp);
      }
      float f[4 * VEC];
#pragma unroll
      for (int j = 0; j < 4 * VEC; ++j) f[j] = 0.f;
#pragma unroll
      for (int u = 0; u < BURST; ++u) {
        const float dpr = ok[u] ? cp * (dz[u] - m1 - (pr[u] - mp) * rsp * m2) : 0.f;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float d = sv[u][j] > 0.f ? dpr * w[j] : 0.f;
          f[4 * j] += d;
          f[4 * j + 1] = fmaf(d, (gv[u][j] - mg[j]) * rg[j], f[4 * j + 1]);
          f[4 * j + 2] = fmaf(d, (xv[u][j] - mx[j]) * rx[j], f[4 * j + 2]);
          f[4 * j + 3] = fmaf(dpr, sv[u][j], f[4 * j + 3]);
        }
      }
#pragma unroll
      for (int j = 0; j < 4 * VEC; ++j) acc[j] += (double)f[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 4 * VEC; ++j) sh[tid][j] = acc[j];
  __syncthreads();
  for (int off = rp.ry >> 1; off > 0; off >>= 1) {
    if (ty < off) {
#pragma unroll
      for (int j = 0; j < 4 * VEC; ++j) sh[tid][j] += sh[tid + off * rp.cvp][j];
    }
    __syncthreads();
  }
  rec_add(p.rec, p.rrs, 0, &sh[0][0], p.F * 4);          // row 0 of the tree = a contiguous [F][4] image
}

template <typename T, int VEC>
__global__ __launch_bounds__(256) void gate_mid_bwd_apply_k(GateMidP p) {
  __shared__ float t_mg[GATE_TAB], t_rg[GATE_TAB], t_mx[GATE_TAB], t_rx[GATE_TAB], t_w[GATE_TAB];
  __shared__ float t_a[GATE_TAB], t_bg[GATE_TAB], t_bx[GATE_TAB], t_cg[GATE_TAB], t_cx[GATE_TAB];
  const int b = blockIdx.y, tid = threadIdx.x;
  const bool first = blockIdx.x == 0 && b == 0;
  for (int c = tid; c < p.F; c += 256) {
    norm_mr(p.stg, c, t_mg[c], t_rg[c]); norm_mr(p.stx, c, t_mx[c], t_rx[c]);
    t_w[c] = p.w[c];
    const double a = rec_get(p.rec, p.rrs, 4 * c), bg = rec_get(p.rec, p.rrs, 4 * c + 1), bx = rec_get(p.rec, p.rrs, 4 * c + 2);
    t_a[c] = (float)(a / p.stg.R); t_bg[c] = (float)(bg / p.stg.R); t_bx[c] = (float)(bx / p.stg.R);
    t_cg[c] = t_rg[c] * p.gam_g[c]; t_cx[c] = t_rx[c] * p.gam_x[c];
    if (first) {
      if (p.dgam_g) p.dgam_g[c] = (float)bg;
      if (p.dbet_g) p.dbet_g[c] = (float)a;
      if (p.dgam_x) p.dgam_x[c] = (float)bx;
      if (p.dbet_x) p.dbet_x[c] = (float)a;
      if (p.dw) p.dw[c] = (float)rec_get(p.rec, p.rrs, 4 * c + 3);
    }
  }
  if (first && tid == 0) {
    if (p.dbet_p) p.dbet_p[0] = (float)rec_get(p.pbs, p.pbrs, 0);
    if (p.dgam_p) p.dgam_p[0] = (float)rec_get(p.pbs, p.pbrs, 1);
  }
  float mp, rsp; norm_mr(p.stp, 0, mp, rsp);
  const float cp = p.gam_p[0] * rsp;
  const float m1 = (float)(rec_get(p.pbs, p.pbrs, 0) / p.stp.R), m2 = (float)(rec_get(p.pbs, p.pbrs, 1) / p.stp.R);
  __syncthreads();
  const int cv = p.cv, c0 = (tid % cv) * VEC;
  float mg[VEC], rg[VEC], mx[VEC], rx[VEC], w[VEC], a[VEC], bg[VEC], bx[VEC], cg[VEC], cx[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const int c = c0 + j;
    mg[j] = t_mg[c]; rg[j] = t_rg[c]; mx[j] = t_mx[c]; rx[j] = t_rx[c]; w[j] = t_w[c];
    a[j] = t_a[c]; bg[j] = t_bg[c]; bx[j] = t_bx[c]; cg[j] = t_cg[c]; cx[j] = t_cx[c];
  }
  const T* gb = reinterpret_cast<const T*>(p.g1) + (int64_t)b * p.sbg;
  const T* xb = reinterpret_cast<const T*>(p.x1) + (int64_t)b * p.sbx;
  const T* sb = reinterpret_cast<const T*>(p.s) + (int64_t)b * p.sbs;
  const T* zb = reinterpret_cast<const T*>(p.dz) + (int64_t)b * p.sbdz;
  const T* rb = reinterpret_cast<const T*>(p.pr) + (int64_t)b * p.sbp;
  T* dgb = reinterpret_cast<T*>(p.dg1) + (int64_t)b * p.sbdg;
  T* dxb = reinterpret_cast<T*>(p.dx1) + (int64_t)b * p.sbdx;
  const int64_t total = p.V * cv, stride = (int64_t)gridDim.x * 256;
  constexpr int U = 2;
  auto one = [&](int64_t v, const float* gv, const float* xv, const float* sv, float dz, float pr) {
    const float dpr = cp * (dz - m1 - (pr - mp) * rsp * m2);
    float og[VEC], ox[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float d = sv[j] > 0.f ? dpr * w[j] : 0.f;
      og[j] = cg[j] * (d - a[j] - (gv[j] - mg[j]) * rg[j] * bg[j]);
      ox[j] = cx[j] * (d - a[j] - (xv[j] - mx[j]) * rx[j] * bx[j]);
    }
    vec_io<T, VEC>::store(dgb + v * p.lddg + c0, og);
    vec_io<T, VEC>::store(dxb + v * p.lddx + c0, ox);
  };
  int64_t e = (int64_t)blockIdx.x * 256 + tid;
  for (; e + (U - 1) * stride < total; e += U * stride) {
    float gv[U][VEC], xv[U][VEC], sv[U][VEC], dz[U], pr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t v = (e + u * stride) / cv;
      vec_io<T, VEC>::load(gb + v * p.ldg + c0, gv[u]);
      vec_io<T, VEC>::load(xb + v * p.ldx + c0, xv[u]);
      vec_io<T, VEC>::load(sb + v * p.lds_ + c0, sv[u]);
      dz[u] = ld_f(zb + v * p.lddz); pr[u] = ld_f(rb + v * p.ldp);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) one((e + u * stride) / cv, gv[u], xv[u], sv[u], dz[u], pr[u]);
  }
  for (; e < total; e += stride) {
    const int64_t v = e / cv;
    float gv[VEC], xv[VEC], sv[VEC];
    vec_io<T, VEC>::load(gb + v * p.ldg + c0, gv);
    vec_io<T, VEC>::load(xb + v * p.ldx + c0, xv);
    vec_io<T, VEC>::load(sb + v * p.lds_ + c0, sv);
    one(v, gv, xv, sv, ld_f(zb + v * p.lddz), ld_f(rb + v * p.ldp));
  }
}

// ---------------------------------------------------------------------------------------------------------------------
static NormStat bn_stat(const double* sums, int C, double R, float eps) {
  NormStat st;
  st.sums = sums; st.rs = COMA_NORM_RECORD_DOUBLES(1, C, 2); st.R = R; st.eps = eps; st.mean = st.rstd = nullptr;
  return st;
}
static int pow2_le64(int v) { return v >= 1 && v <= 64 && (v & (v - 1)) == 0; }
static unsigned gate_blocks(int64_t items, int per_block, int cap) {
  int64_t nb = (items + per_block - 1) / per_block;
  if (nb > cap) nb = cap;
  return (unsigned)(nb < 1 ? 1 : nb);
}
// vector width all of the given [.., F] tensors allow (8 bf16 / 4 fp32 elements = 16 bytes, else 4, else 1)
static int gate_vec(const coma_tensor* const* ts, int n, int want) {
  int v = want;
  for (int i = 0; i < n; ++i) { const int t = t_vec(ts[i], want); if (t < v) v = t; }
  return v;
}
#define GATE_DISPATCH(KERNEL, DT, VEC, GRID, ...)                                                              \
  do {                                                                                                         \
    if ((DT) == COMA_F32) { if ((VEC) == 4) hipLaunchKernelGGL((KERNEL<float, 4>), GRID, dim3(256), 0, s, __VA_ARGS__);      \
                            else hipLaunchKernelGGL((KERNEL<float, 1>), GRID, dim3(256), 0, s, __VA_ARGS__); }               \
    else { if ((VEC) == 8) hipLaunchKernelGGL((KERNEL<bf16_t, 8>), GRID, dim3(256), 0, s, __VA_ARGS__);                     \
           else if ((VEC) == 4) hipLaunchKernelGGL((KERNEL<bf16_t, 4>), GRID, dim3(256), 0, s, __VA_ARGS__);                 \
           else hipLaunchKernelGGL((KERNEL<bf16_t, 1>), GRID, dim3(256), 0, s, __VA_ARGS__); }                               \
  } while (0)

static int gate_pick_vec(int dtype, const coma_tensor* const* ts, int n, int F, int max_lanes) {
  int vec = gate_vec(ts, n, dtype == COMA_BF16 ? 8 : 4);
  if (dtype == COMA_F32 && vec == 8) vec = 4;
  if (vec == 2) vec = 1;
  // lanes per voxel must be a power of two <= 64 (wave shuffles): widen nothing, but fall back to narrower vectors never
  // makes it worse -- F / vec too large means an unsupported F
  while (vec > 1 && !pow2_le64(F / vec)) vec = vec == 8 ? 4 : 1;
  (void)max_lanes;
  return vec;
}

extern "C" int coma_gate_mid_fwd(const coma_tensor* g1, const coma_tensor* x1, const double* sums_g, float eps_g,
                                 const float* gamma_g, const float* beta_g, const double* sums_x, float eps_x,
                                 const float* gamma_x, const float* beta_x, const float* w_psi, const float* b_psi,
                                 float* rmean_g, float* rvar_g, float* rmean_x, float* rvar_x, float momentum,
                                 const coma_tensor* s_out, const coma_tensor* psi_raw, double* sums_psi, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  COMA_CHECK(g1 && x1 && s_out && psi_raw && g1->data && x1->data && s_out->data && psi_raw->data && sums_g && sums_x && sums_psi &&
             gamma_g && beta_g && gamma_x && beta_x && w_psi, "gate_mid_fwd: null argument");
  COMA_CHECK(t_same_grid(g1, x1) && t_same_grid(g1, s_out) && t_same_grid(g1, psi_raw) && g1->C == x1->C && g1->C == s_out->C &&
             psi_raw->C == 1 && g1->dtype == x1->dtype && g1->dtype == s_out->dtype && g1->dtype == psi_raw->dtype,
             "gate_mid_fwd: shape/dtype mismatch");
  const int F = g1->C;
  COMA_CHECK(F <= GATE_TAB, "gate_mid_fwd: F=%d too large", F);
  const coma_tensor* ts[3] = {g1, x1, s_out};
  const int vec = gate_pick_vec(g1->dtype, ts, 3, F, 64);
  COMA_CHECK(pow2_le64(F / vec), "gate_mid_fwd: F=%d unsupported (F / vector width must be a power of two <= 64)", F);
  GateMidP p{};
  p.g1 = g1->data; p.ldg = g1->ld; p.sbg = g1->sb; p.x1 = x1->data; p.ldx = x1->ld; p.sbx = x1->sb;
  p.s = s_out->data; p.lds_ = s_out->ld; p.sbs = s_out->sb; p.pr = psi_raw->data; p.ldp = psi_raw->ld; p.sbp = psi_raw->sb;
  p.V = t_vox(g1); p.B = g1->B; p.F = F; p.cv = F / vec;
  const double R = (double)p.V * g1->B;
  p.stg = bn_stat(sums_g, F, R, eps_g); p.stx = bn_stat(sums_x, F, R, eps_x);
  p.gam_g = gamma_g; p.bet_g = beta_g; p.gam_x = gamma_x; p.bet_x = beta_x; p.w = w_psi; p.bias = b_psi;
  p.psums = sums_psi; p.prs = COMA_NORM_RECORD_DOUBLES(1, 1, 2);
  p.rm_g = rmean_g; p.rv_g = rvar_g; p.rm_x = rmean_x; p.rv_x = rvar_x; p.momentum = momentum;
  const int vpb = 256 / p.cv;
  int cap = 1024 / g1->B; if (cap < 1) cap = 1;
  dim3 grid(gate_blocks(p.V, vpb * 2, cap), g1->B);
  GATE_DISPATCH(gate_mid_fwd_k, g1->dtype, vec, grid, p);
  COMA_LAUNCH_CHECK();
  return 0;
}

static int gate_apply_setup(GateApplyP& p, const coma_tensor* x, const coma_tensor* psi_raw, const coma_tensor* psi,
                            const double* sums_psi, float eps, const float* gamma_p, const float* beta_p) {
  p.x = x->data; p.ldx = x->ld; p.sbx = x->sb; p.pr = psi_raw->data; p.ldp = psi_raw->ld; p.sbp = psi_raw->sb;
  p.psi = psi->data; p.ldq = psi->ld; p.sbq = psi->sb;
  p.V = t_vox(x); p.B = x->B; p.C = x->C;
  p.stp = bn_stat(sums_psi, 1, (double)p.V * x->B, eps);
  p.gam_p = gamma_p; p.bet_p = beta_p;
  return 0;
}

extern "C" int coma_gate_apply_fwd(const coma_tensor* x, const coma_tensor* psi_raw, const double* sums_psi, float eps,
                                   const float* gamma_psi, const float* beta_psi, float* rmean_psi, float* rvar_psi,
                                   float momentum, const coma_tensor* psi, const coma_tensor* att, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  COMA_CHECK(x && psi_raw && psi && att && x->data && psi_raw->data && psi->data && att->data && sums_psi && gamma_psi && beta_psi,
             "gate_apply_fwd: null argument");
  COMA_CHECK(t_same_grid(x, psi_raw) && t_same_grid(x, psi) && t_same_grid(x, att) && psi_raw->C == 1 && psi->C == 1 &&
             x->C == att->C && x->dtype == psi_raw->dtype && x->dtype == psi->dtype && x->dtype == att->dtype,
             "gate_apply_fwd: shape/dtype mismatch");
  const coma_tensor* ts[2] = {x, att};
  const int vec = gate_pick_vec(x->dtype, ts, 2, x->C, 64);
  COMA_CHECK(pow2_le64(x->C / vec), "gate_apply_fwd: C=%d unsupported", x->C);
  GateApplyP p{};
  gate_apply_setup(p, x, psi_raw, psi, sums_psi, eps, gamma_psi, beta_psi);
  p.cv = x->C / vec;
  p.att = att->data; p.lda = att->ld; p.sba = att->sb;
  p.rm_p = rmean_psi; p.rv_p = rvar_psi; p.momentum = momentum;
  dim3 grid(gate_blocks(p.V * p.cv, 256 * 4, 4096), x->B);
  GATE_DISPATCH(gate_apply_fwd_k, x->dtype, vec, grid, p);
  COMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int coma_gate_apply_bwd(const coma_tensor* x, const coma_tensor* psi, const coma_tensor* psi_raw,
                                   const coma_tensor* dout, const double* sums_psi, float eps, const float* gamma_psi,
                                   const float* beta_psi, const coma_tensor* dx, int32_t accumulate_dx,
                                   const coma_tensor* dz, double* bsums_psi, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  COMA_CHECK(x && psi && psi_raw && dout && dx && dz && x->data && psi->data && psi_raw->data && dout->data && dx->data &&
             dz->data && sums_psi && bsums_psi && gamma_psi, "gate_apply_bwd: null argument");
  COMA_CHECK(t_same_grid(x, psi) && t_same_grid(x, psi_raw) && t_same_grid(x, dout) && t_same_grid(x, dx) && t_same_grid(x, dz) &&
             psi->C == 1 && psi_raw->C == 1 && dz->C == 1 && x->C == dout->C && x->C == dx->C && x->dtype == dout->dtype &&
             x->dtype == dx->dtype && x->dtype == dz->dtype && x->dtype == psi->dtype, "gate_apply_bwd: shape/dtype mismatch");
  const coma_tensor* ts[3] = {x, dout, dx};
  const int vec = gate_pick_vec(x->dtype, ts, 3, x->C, 64);
  COMA_CHECK(pow2_le64(x->C / vec), "gate_apply_bwd: C=%d unsupported", x->C);
  GateApplyP p{};
  gate_apply_setup(p, x, psi_raw, psi, sums_psi, eps, gamma_psi, beta_psi);
  p.cv = x->C / vec;
  p.dout = dout->data; p.ldo = dout->ld; p.sbo = dout->sb;
  p.dx = dx->data; p.lddx = dx->ld; p.sbdx = dx->sb; p.acc = accumulate_dx;
  p.dz = dz->data; p.lddz = dz->ld; p.sbdz = dz->sb;
  p.pbs = bsums_psi; p.pbrs = COMA_NORM_RECORD_DOUBLES(1, 1, 3);
  int cap = 1024 / x->B; if (cap < 1) cap = 1;
  dim3 grid(gate_blocks(p.V, (256 / p.cv) * 2, cap), x->B);
  GATE_DISPATCH(gate_apply_bwd_k, x->dtype, vec, grid, p);
  COMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int coma_gate_mid_bwd(const coma_tensor* dz, const coma_tensor* psi_raw, const coma_tensor* s_in,
                                 const coma_tensor* g1, const coma_tensor* x1, const double* sums_psi, float eps_psi,
                                 const float* gamma_psi, const double* bsums_psi, const double* sums_g, float eps_g,
                                 const float* gamma_g, const double* sums_x, float eps_x, const float* gamma_x,
                                 const float* w_psi, double* rec, const coma_tensor* dg1, const coma_tensor* dx1,
                                 float* dgamma_g, float* dbeta_g, float* dgamma_x, float* dbeta_x, float* dw_psi,
                                 float* dgamma_psi, float* dbeta_psi, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  COMA_CHECK(dz && psi_raw && s_in && g1 && x1 && dg1 && dx1 && dz->data && psi_raw->data && s_in->data && g1->data && x1->data &&
             dg1->data && dx1->data && sums_psi && bsums_psi && sums_g && sums_x && gamma_psi && gamma_g && gamma_x && w_psi && rec,
             "gate_mid_bwd: null argument");
  COMA_CHECK(t_same_grid(g1, x1) && t_same_grid(g1, s_in) && t_same_grid(g1, dz) && t_same_grid(g1, psi_raw) && t_same_grid(g1, dg1) &&
             t_same_grid(g1, dx1) && g1->C == x1->C && g1->C == s_in->C && g1->C == dg1->C && g1->C == dx1->C && dz->C == 1 &&
             psi_raw->C == 1 && g1->dtype == x1->dtype && g1->dtype == s_in->dtype && g1->dtype == dz->dtype &&
             g1->dtype == dg1->dtype && g1->dtype == dx1->dtype && g1->dtype == psi_raw->dtype, "gate_mid_bwd: shape/dtype mismatch");
  const int F = g1->C;
  COMA_CHECK(F <= GATE_TAB, "gate_mid_bwd: F=%d too large", F);
  GateMidP p{};
  p.g1 = g1->data; p.ldg = g1->ld; p.sbg = g1->sb; p.x1 = x1->data; p.ldx = x1->ld; p.sbx = x1->sb;
  p.s = s_in->data; p.lds_ = s_in->ld; p.sbs = s_in->sb; p.pr = psi_raw->data; p.ldp = psi_raw->ld; p.sbp = psi_raw->sb;
  p.dz = dz->data; p.lddz = dz->ld; p.sbdz = dz->sb;
  p.dg1 = dg1->data; p.lddg = dg1->ld; p.sbdg = dg1->sb; p.dx1 = dx1->data; p.lddx = dx1->ld; p.sbdx = dx1->sb;
  p.V = t_vox(g1); p.B = g1->B; p.F = F;
  const double R = (double)p.V * g1->B;
  p.stg = bn_stat(sums_g, F, R, eps_g); p.stx = bn_stat(sums_x, F, R, eps_x); p.stp = bn_stat(sums_psi, 1, R, eps_psi);
  p.gam_g = gamma_g; p.gam_x = gamma_x; p.gam_p = gamma_psi; p.w = w_psi;
  p.pbs = bsums_psi; p.pbrs = COMA_NORM_RECORD_DOUBLES(1, 1, 3);
  p.rec = rec; p.rrs = COMA_NORM_RECORD_DOUBLES(1, F, 4);
  p.dgam_g = dgamma_g; p.dbet_g = dbeta_g; p.dgam_x = dgamma_x; p.dbet_x = dbeta_x; p.dw = dw_psi;
  p.dgam_p = dgamma_psi; p.dbet_p = dbeta_psi;
  // pass 1: per-channel sums (row walk of the statistics kernels, 4-wide: 16 fp64 accumulators per lane)
  {
    const coma_tensor* ts[3] = {g1, x1, s_in};
    const int pv = gate_vec(ts, 3, 4) >= 4 ? 4 : 1;
    RowsP rp = make_rows(g1, COMA_NORM_BATCH, pv);
    COMA_CHECK(rp.cv <= 256, "gate_mid_bwd: F=%d too large", F);
    dim3 pg(rp.nchunks, 1);
    if (g1->dtype == COMA_F32) { if (pv == 4) hipLaunchKernelGGL((gate_mid_bwd_partial_k<float, 4>), pg, dim3(256), 0, s, rp, p);
                                 else hipLaunchKernelGGL((gate_mid_bwd_partial_k<float, 1>), pg, dim3(256), 0, s, rp, p); }
    else { if (pv == 4) hipLaunchKernelGGL((gate_mid_bwd_partial_k<bf16_t, 4>), pg, dim3(256), 0, s, rp, p);
           else hipLaunchKernelGGL((gate_mid_bwd_partial_k<bf16_t, 1>), pg, dim3(256), 0, s, rp, p); }
    COMA_LAUNCH_CHECK();
  }
  // pass 2: dg1raw, dx1raw + the parameter gradients
  const coma_tensor* ts[5] = {g1, x1, s_in, dg1, dx1};
  const int vec = gate_pick_vec(g1->dtype, ts, 5, F, 64);
  COMA_CHECK(pow2_le64(F / vec), "gate_mid_bwd: F=%d unsupported", F);
  p.cv = F / vec;
  dim3 grid(gate_blocks(p.V * p.cv, 256 * 2, 4096), g1->B);
  GATE_DISPATCH(gate_mid_bwd_apply_k, g1->dtype, vec, grid, p);
  COMA_LAUNCH_CHECK();
  return 0;
}
