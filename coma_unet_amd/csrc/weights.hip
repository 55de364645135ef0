// CondConv expert mixing fused with the re-layout / cast of convolution weights into the
// kernel layout wk[b][tap][n][c].  Replaces the per-sample kernel synthesis of the missing
// upstream module CondConv.CondConvolution (call sites attn_unet_data_parallel.py:126,285-306;
// spec: DESIGN.md "CondConv spec"):  W_b = sum_e r[b][e] * W_e.
// HBM-bound: the E expert tensors are read once (coalesced 27-tap runs), B mixed copies written.
#include "common.h"

template <typename TO, int TAPS, int BB>
__global__ __launch_bounds__(256) void weight_prep_k(const float* master, const float* r, int E, int b0, int nb, int N,
                                                     int C, int64_t se, int64_t sn, int64_t sc, TO* out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)N * C) return;
  const int n = (int)(i / C), c = (int)(i - (int64_t)n * C);
  float acc[BB][TAPS];
#pragma unroll
  for (int b = 0; b < BB; ++b)
#pragma unroll
    for (int t = 0; t < TAPS; ++t) acc[b][t] = 0.f;
  for (int e = 0; e < E; ++e) {
    const float* mp = master + e * se + n * sn + c * sc;
    float w[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) w[t] = mp[t];
#pragma unroll
    for (int b = 0; b < BB; ++b) {
      if (b < nb) {
        const float rb = r ? r[(b0 + b) * E + e] : 1.f;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) acc[b][t] = fmaf(rb, w[t], acc[b][t]);
      }
    }
  }
#pragma unroll
  for (int b = 0; b < BB; ++b) {
    if (b < nb) {
      TO* op = out + ((int64_t)(b0 + b) * TAPS) * N * C + i;
#pragma unroll
      for (int t = 0; t < TAPS; ++t) st_f(op + (int64_t)t * N * C, acc[b][t]);
    }
  }
}

extern "C" int coma_weight_prep(const float* master, const float* r, int32_t E, int32_t Bw, int32_t N, int32_t C,
                                int32_t taps, int64_t se, int64_t sn, int64_t sc, void* out, int32_t out_dtype,
                                void* stream) {
  COMA_CHECK(master && out, "weight_prep: null argument");
  COMA_CHECK(taps == 27 || taps == 1, "weight_prep: taps=%d unsupported", taps);
  COMA_CHECK(E >= 1 && Bw >= 1 && (r || (E == 1)), "weight_prep: E=%d needs routing weights", E);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)(((int64_t)N * C + 255) / 256));
  for (int b0 = 0; b0 < Bw; b0 += 2) {
    const int nb = Bw - b0 < 2 ? Bw - b0 : 2;
#define L(TO, TP) hipLaunchKernelGGL((weight_prep_k<TO, TP, 2>), grid, dim3(256), 0, s, master, r, E, b0, nb, N, C, se, sn, sc, (TO*)out)
    if (out_dtype == COMA_F32) { if (taps == 27) L(float, 27); else L(float, 1); }
    else { if (taps == 27) L(bf16_t, 27); else L(bf16_t, 1); }
#undef L
    COMA_LAUNCH_CHECK();
  }
  return 0;
}

// dmaster[e] (=) sum_b r[b][e] * dwk[b]  (master layout);  dr[b][e] (=) <dwk[b], master[e]>
template <int TAPS, int BB>
__global__ __launch_bounds__(256) void weight_prep_bwd_k(const float* dwk, const float* master, const float* r, int E, int Bw,
                                                         int N, int C, int64_t se, int64_t sn, int64_t sc, float* dmaster,
                                                         float* dr) {
  __shared__ float red[256];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = i < (int64_t)N * C;
  const int n = live ? (int)(i / C) : 0, c = live ? (int)(i - (int64_t)n * C) : 0;
  float g[BB][TAPS];
#pragma unroll
  for (int b = 0; b < BB; ++b)
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
      g[b][t] = (live && b < Bw) ? dwk[((int64_t)b * TAPS + t) * N * C + i] : 0.f;
  for (int e = 0; e < E; ++e) {
    const int64_t off = e * se + n * sn + c * sc;
    float w[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) w[t] = live ? master[off + t] : 0.f;
    float dm[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) dm[t] = 0.f;
#pragma unroll
    for (int b = 0; b < BB; ++b) {
      if (b < Bw) {
        const float rb = r ? r[b * E + e] : 1.f;
        float dot = 0.f;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) { dm[t] = fmaf(rb, g[b][t], dm[t]); dot = fmaf(g[b][t], w[t], dot); }
        if (dr) {
          red[threadIdx.x] = dot;
          __syncthreads();
          for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
          if (threadIdx.x == 0) atomicAdd(dr + b * E + e, red[0]);
          __syncthreads();
        }
      }
    }
    if (live) {
#pragma unroll
      for (int t = 0; t < TAPS; ++t) dmaster[off + t] = dm[t];
    }
  }
}

extern "C" int coma_weight_prep_bwd(const float* dwk, const float* master, const float* r, int32_t E, int32_t Bw, int32_t N,
                                    int32_t C, int32_t taps, int64_t se, int64_t sn, int64_t sc, float* dmaster, float* dr,
                                    void* stream) {
  COMA_CHECK(dwk && master && dmaster, "weight_prep_bwd: null argument");
  COMA_CHECK(taps == 27 || taps == 1, "weight_prep_bwd: taps=%d unsupported", taps);
  COMA_CHECK(Bw >= 1 && Bw <= 8, "weight_prep_bwd: Bw=%d out of range (1..8)", Bw);
  hipStream_t s = (hipStream_t)stream;
  if (dr && hipMemsetAsync(dr, 0, sizeof(float) * Bw * E, s) != hipSuccess) { coma_set_error("weight_prep_bwd: memset failed"); return 2; }
  dim3 grid((unsigned)(((int64_t)N * C + 255) / 256));
#define L(TP, BBV) hipLaunchKernelGGL((weight_prep_bwd_k<TP, BBV>), grid, dim3(256), 0, s, dwk, master, r, E, Bw, N, C, se, sn, sc, dmaster, dr)
  if (taps == 27) { if (Bw <= 2) L(27, 2); else if (Bw <= 4) L(27, 4); else L(27, 8); }
  else { if (Bw <= 2) L(1, 2); else if (Bw <= 4) L(1, 4); else L(1, 8); }
#undef L
  COMA_LAUNCH_CHECK();
  return 0;
}
