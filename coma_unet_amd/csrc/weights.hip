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

// ---------------------------------------------------------------------------------------
// Tiled 27-tap variants.  The kernels above give each thread one (n, c) pair and its 27 contiguous taps: lanes are
// 108 bytes (or a whole weight row) apart, every load instruction touches ~54 cache lines, and the mix ran at a
// quarter of the HBM rate.  Here a block owns a 16 x 16 tile of (a, b) = (master dim 0, dim 1) pairs; each expert's
// tile is 16 runs of 432 contiguous floats read with consecutive lanes on consecutive addresses, the mixed tile goes
// through LDS once, and both kernel layouts ([tap][a][b] and [tap][b][a]) are written from the same pass in 16-element
// rows.  (The same tiling applied to the backward kernel measured 2x SLOWER than the per-pair form below -- 2.07 vs
// 1.10 ms per step -- and was dropped.)
// ---------------------------------------------------------------------------------------
#define WT_T 16
#define WT_ROW (WT_T * 27)          // floats of one a-row of the tile
#define WT_PITCH (WT_ROW + 1)       // LDS pitch of an a-row: odd, so column reads (fixed b, tap; a varies) are conflict-free
#define WT_ELEMS (WT_T * WT_ROW)    // 6912 = 27 per thread

__device__ __forceinline__ void wt_store(void* base, int dtype, int64_t idx, float v) {
  if (dtype == COMA_F32) reinterpret_cast<float*>(base)[idx] = v;
  else reinterpret_cast<bf16_t*>(base)[idx] = static_cast<bf16_t>(v);
}

template <int BB>
__global__ __launch_bounds__(256) void weight_prep_tiled_k(const float* __restrict__ master, const float* __restrict__ r, int E,
                                                           int bw0, int nb, int A, int Bd, int64_t se, void* out_ab, int dt_ab,
                                                           void* out_ba, int dt_ba) {
  __shared__ float tile[WT_T * WT_PITCH];
  const int tid = threadIdx.x;
  const int a0 = blockIdx.y * WT_T, b0 = blockIdx.x * WT_T;
  const int bspan = (Bd - b0 < WT_T ? Bd - b0 : WT_T) * 27;      // valid floats of an a-row of this tile
  float acc[BB][27];
#pragma unroll
  for (int b = 0; b < BB; ++b)
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[b][k] = 0.f;
  for (int e = 0; e < E; ++e) {
    float rb[BB];
#pragma unroll
    for (int b = 0; b < BB; ++b) rb[b] = (b < nb) ? (r ? r[(bw0 + b) * E + e] : 1.f) : 0.f;
    const float* mp = master + e * se + ((int64_t)a0 * Bd + b0) * 27;
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      const int idx = tid + 256 * k, al = idx / WT_ROW, rem = idx - al * WT_ROW;
      const float w = (a0 + al < A && rem < bspan) ? mp[(int64_t)al * Bd * 27 + rem] : 0.f;
#pragma unroll
      for (int b = 0; b < BB; ++b) acc[b][k] = fmaf(rb[b], w, acc[b][k]);
    }
  }
  const int64_t AB = (int64_t)A * Bd;
  const int col = tid & 15, rw = tid >> 4;        // output phase: 16 rows of 16 contiguous elements per sweep
#pragma unroll
  for (int b = 0; b < BB; ++b) {
    if (b < nb) {                                  // (block-uniform)
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 27; ++k) {
        const int idx = tid + 256 * k, al = idx / WT_ROW;
        tile[idx + al] = acc[b][k];                // al * WT_PITCH + rem
      }
      __syncthreads();
      const int64_t ob = (int64_t)(bw0 + b) * 27 * AB;
      if (out_ab) {
        for (int j = rw; j < 27 * WT_T; j += 16) {     // row j = (tap, al), column = bl
          const int tap = j >> 4, al = j & 15;
          if (a0 + al < A && b0 + col < Bd)
            wt_store(out_ab, dt_ab, ob + ((int64_t)tap * A + a0 + al) * Bd + b0 + col, tile[al * WT_PITCH + col * 27 + tap]);
        }
      }
      if (out_ba) {
        for (int j = rw; j < 27 * WT_T; j += 16) {     // row j = (tap, bl), column = al
          const int tap = j >> 4, bl = j & 15;
          if (a0 + col < A && b0 + bl < Bd)
            wt_store(out_ba, dt_ba, ob + ((int64_t)tap * Bd + b0 + bl) * A + a0 + col, tile[col * WT_PITCH + bl * 27 + tap]);
        }
      }
    }
  }
}

// master [E][A][B][27] fp32 (+ r [Bw][E] or NULL)  ->  out_ab [Bw][27][A][B] and / or out_ba [Bw][27][B][A]
extern "C" int coma_weight_prep_pair(const float* master, const float* r, int32_t E, int32_t Bw, int32_t A, int32_t B,
                                     void* out_ab, int32_t dtype_ab, void* out_ba, int32_t dtype_ba, void* stream) {
  COMA_CHECK(master && (out_ab || out_ba), "weight_prep_pair: null argument");
  COMA_CHECK(E >= 1 && Bw >= 1 && (r || E == 1), "weight_prep_pair: E=%d needs routing weights", E);
  COMA_CHECK((!out_ab || dtype_ab == COMA_F32 || dtype_ab == COMA_BF16) && (!out_ba || dtype_ba == COMA_F32 || dtype_ba == COMA_BF16),
             "weight_prep_pair: unsupported output dtype");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((B + WT_T - 1) / WT_T), (unsigned)((A + WT_T - 1) / WT_T));
  const int64_t se = (int64_t)A * B * 27;
  for (int b0 = 0; b0 < Bw; b0 += 2) {
    const int nb = Bw - b0 < 2 ? Bw - b0 : 2;
    hipLaunchKernelGGL((weight_prep_tiled_k<2>), grid, dim3(256), 0, s, master, r, E, b0, nb, A, B, se, out_ab, dtype_ab, out_ba, dtype_ba);
    COMA_LAUNCH_CHECK();
  }
  return 0;
}

// dmaster[e] (=) sum_b r[b][e] * dwk[b]  (master layout);  dr[b][e] (=) <dwk[b], master[e]>
template <int TAPS, int BB>
__global__ __launch_bounds__(256) void weight_prep_bwd_k(const float* dwk, const float* master, const float* r, int E, int Bw,
                                                         int N, int C, int64_t se, int64_t sn, int64_t sc, float* dmaster,
                                                         float* dr) {
  __shared__ float red[64];          // [b][e] block totals of the dr dot products (Bw * E <= 64)
  if (threadIdx.x < 64) red[threadIdx.x] = 0.f;
  __syncthreads();
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
    float w[TAPS];        // the expert's weights are only needed for dr; one guarded block so the 27 loads merge into dwordx4
#pragma unroll
    for (int t = 0; t < TAPS; ++t) w[t] = 0.f;
    if (live && dr) {
#pragma unroll
      for (int t = 0; t < TAPS; ++t) w[t] = master[off + t];
    }
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
        if (dr) {       // wave butterfly, one LDS atomic per wave: no block barrier inside the expert loop
          for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
          if ((threadIdx.x & 63) == 0) atomicAdd(&red[b * E + e], dot);
        }
      }
    }
    if (live) {
#pragma unroll
      for (int t = 0; t < TAPS; ++t) dmaster[off + t] = dm[t];
    }
  }
  if (dr) {
    __syncthreads();
    if (threadIdx.x < Bw * E) atomicAdd(dr + threadIdx.x, red[threadIdx.x]);
  }
}

// The same for 27-tap masters, with every master / dmaster access COALESCED: a block owns 256 consecutive (a, b) pairs of
// the master layout = one contiguous run of 256 x 27 floats per expert.  The kernel-layout gradient of those pairs is
// brought into master order ONCE (through LDS: lane = pair writes its 27 taps, lane = linear index reads them back), then
// each expert is 27 fully coalesced loads (for dr) and 27 fully coalesced stores per lane.  (The per-pair kernel above
// reads and writes 108-byte runs per lane: every dword instruction touches 64 different lines -- 2.7 TB/s; a first tiled
// version that transposed per expert was slower still.)  transposed: master [c][n][27] (ConvTranspose3d), else [n][c][27].
template <int BB>
__global__ __launch_bounds__(256) void weight_prep_bwd27_k(const float* __restrict__ dwk, const float* __restrict__ master,
                                                           const float* __restrict__ r, int E, int Bw, int N, int C, int64_t se,
                                                           int transposed, float* __restrict__ dmaster, float* __restrict__ dr) {
  __shared__ float T[BB][256 * 28];      // [sample][pair][27 taps, pitch 28]
  __shared__ float red[64];
  const int tid = threadIdx.x;
  if (tid < 64) red[tid] = 0.f;
  const int64_t NC = (int64_t)N * C;
  const int64_t j0 = (int64_t)blockIdx.x * 256;                 // first pair (master order) of this block
  const int64_t j = j0 + tid;
  if (j < NC) {
    // pair j of the master -> (n, c) of the kernel layout dwk[b][tap][n][c]
    const int64_t a = transposed ? j / N : j / C, bq = transposed ? j - a * N : j - a * C;
    const int64_t i = transposed ? bq * C + a : a * C + bq;
#pragma unroll
    for (int b = 0; b < BB; ++b)
#pragma unroll
      for (int t = 0; t < 27; ++t) T[b][tid * 28 + t] = b < Bw ? dwk[((int64_t)b * 27 + t) * NC + i] : 0.f;
  }
  __syncthreads();
  const int64_t nvalid = (NC - j0 < 256 ? NC - j0 : 256) * 27;   // floats of this block's run
  float g[BB][27];
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const int m = k * 256 + tid, pr = m / 27, tp = m - pr * 27;
#pragma unroll
    for (int b = 0; b < BB; ++b) g[b][k] = m < nvalid ? T[b][pr * 28 + tp] : 0.f;
  }
  for (int e = 0; e < E; ++e) {
    const float* mp = master + e * se + j0 * 27;
    float* dp = dmaster + e * se + j0 * 27;
    float rb[BB], dot[BB];
#pragma unroll
    for (int b = 0; b < BB; ++b) { rb[b] = b < Bw ? (r ? r[b * E + e] : 1.f) : 0.f; dot[b] = 0.f; }
    float w[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) { const int m = k * 256 + tid; w[k] = (dr && m < nvalid) ? mp[m] : 0.f; }
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      const int m = k * 256 + tid;
      float dm = 0.f;
#pragma unroll
      for (int b = 0; b < BB; ++b) { dm = fmaf(rb[b], g[b][k], dm); dot[b] = fmaf(g[b][k], w[k], dot[b]); }
      if (m < nvalid) dp[m] = dm;
    }
    if (dr) {
#pragma unroll
      for (int b = 0; b < BB; ++b) {
        if (b < Bw) {
          float d = dot[b];
          for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
          if ((tid & 63) == 0) atomicAdd(&red[b * E + e], d);
        }
      }
    }
  }
  if (dr) {
    __syncthreads();
    if (tid < Bw * E) atomicAdd(dr + tid, red[tid]);
  }
}

extern "C" int coma_weight_prep_bwd(const float* dwk, const float* master, const float* r, int32_t E, int32_t Bw, int32_t N,
                                    int32_t C, int32_t taps, int64_t se, int64_t sn, int64_t sc, float* dmaster, float* dr,
                                    int32_t zeroed, void* stream) {
  COMA_CHECK(dwk && master && dmaster, "weight_prep_bwd: null argument");
  COMA_CHECK(taps == 27 || taps == 1, "weight_prep_bwd: taps=%d unsupported", taps);
  COMA_CHECK(Bw >= 1 && Bw <= 8 && (!dr || Bw * E <= 64), "weight_prep_bwd: Bw=%d E=%d out of range (Bw 1..8, Bw*E <= 64)", Bw, E);
  hipStream_t s = (hipStream_t)stream;
  if (dr && !(zeroed & COMA_ZEROED_OUT) && hipMemsetAsync(dr, 0, sizeof(float) * Bw * E, s) != hipSuccess) { coma_set_error("weight_prep_bwd: memset failed"); return 2; }
  dim3 grid((unsigned)(((int64_t)N * C + 255) / 256));
  static const bool old_path = getenv("COMA_WPREP_BWD_OLD") != nullptr;      // (A/B measurements)
  const bool plain = sc == 27 && sn == (int64_t)C * 27, transposed = sn == 27 && sc == (int64_t)N * 27;
  if (taps == 27 && Bw <= 2 && (plain || transposed) && !old_path) {
    hipLaunchKernelGGL((weight_prep_bwd27_k<2>), grid, dim3(256), 0, s, dwk, master, r, E, Bw, N, C, se, transposed ? 1 : 0, dmaster, dr);
    COMA_LAUNCH_CHECK();
    return 0;
  }
#define L(TP, BBV) hipLaunchKernelGGL((weight_prep_bwd_k<TP, BBV>), grid, dim3(256), 0, s, dwk, master, r, E, Bw, N, C, se, sn, sc, dmaster, dr)
  if (taps == 27) { if (Bw <= 2) L(27, 2); else if (Bw <= 4) L(27, 4); else L(27, 8); }
  else { if (Bw <= 2) L(1, 2); else if (Bw <= 4) L(1, 4); else L(1, 8); }
#undef L
  COMA_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------
// CondConv routing: r = sigmoid(cov . Wr^T + br), bias_mix = r . bias_e   
// A few hundred numbers per layer: one launch forward, one backward, instead of the ~15 tiny
// GEMM / elementwise launches the same algebra costs through a tensor library.
// ---------------------------------------------------------------------------------------
#define ROUTE_MAX_BE 64

__global__ void __launch_bounds__(256) routing_fwd_k(const float* __restrict__ cov, int B, int NC, const float* __restrict__ Wr,
                                                     const float* __restrict__ br, int E, const float* __restrict__ bias_e, int N,
                                                     float* __restrict__ r_out, float* __restrict__ bias_mix) {
  __shared__ float r[ROUTE_MAX_BE];
  if (threadIdx.x < B * E) {
    const int b = threadIdx.x / E, e = threadIdx.x % E;
    float z = br[e];
    for (int c = 0; c < NC; ++c) z = fmaf(cov[b * NC + c], Wr[e * NC + c], z);
    const float v = 1.f / (1.f + expf(-z));
    r[threadIdx.x] = v;
    if (blockIdx.x == 0) r_out[threadIdx.x] = v;
  }
  __syncthreads();
  if (!bias_mix) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < B * N) {
    const int b = i / N, n = i % N;
    float a = 0.f;
    for (int e = 0; e < E; ++e) a = fmaf(r[b * E + e], bias_e[e * N + n], a);
    bias_mix[i] = a;
  }
}

__global__ void __launch_bounds__(256) routing_bwd_k(const float* __restrict__ cov, int B, int NC, const float* __restrict__ r, int E,
                                                     const float* __restrict__ bias_e, int N, const float* __restrict__ dr_w,
                                                     const float* __restrict__ dbias_mix, float* __restrict__ dWr,
                                                     float* __restrict__ dbr, float* __restrict__ dbias_e) {
  __shared__ float dz[ROUTE_MAX_BE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int p = wave; p < B * E; p += 4) {          // dr = dr_w + dbias_mix . bias_e^T, one wave per (b, e)
    const int b = p / E, e = p % E;
    float a = 0.f;
    if (dbias_mix)
      for (int n = lane; n < N; n += 64) a = fmaf(dbias_mix[b * N + n], bias_e[e * N + n], a);
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (lane == 0) {
      const float rv = r[p];
      dz[p] = (a + (dr_w ? dr_w[p] : 0.f)) * rv * (1.f - rv);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < E * NC; i += 256) {
    const int e = i / NC, c = i % NC;
    float a = 0.f;
    for (int b = 0; b < B; ++b) a = fmaf(dz[b * E + e], cov[b * NC + c], a);
    dWr[i] = a;
  }
  for (int e = threadIdx.x; e < E; e += 256) {
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += dz[b * E + e];
    dbr[e] = a;
  }
  if (dbias_e)
    for (int i = threadIdx.x; i < E * N; i += 256) {
      const int e = i / N, n = i % N;
      float a = 0.f;
      if (dbias_mix)
        for (int b = 0; b < B; ++b) a = fmaf(r[b * E + e], dbias_mix[b * N + n], a);
      dbias_e[i] = a;
    }
}

extern "C" int coma_routing_fwd(const float* cov, int32_t B, int32_t NC, const float* Wr, const float* br, int32_t E,
                                const float* bias_e, int32_t N, float* r, float* bias_mix, void* stream) {
  COMA_CHECK(cov && Wr && br && r, "routing_fwd: null argument");
  COMA_CHECK(B >= 1 && E >= 1 && B * E <= ROUTE_MAX_BE && NC >= 1, "routing_fwd: B*E=%d out of range", B * E);
  COMA_CHECK(!bias_mix || (bias_e && N >= 1), "routing_fwd: bias_mix without expert biases");
  const int blocks = bias_mix ? (B * N + 255) / 256 : 1;
  hipLaunchKernelGGL(routing_fwd_k, dim3(blocks), dim3(256), 0, (hipStream_t)stream, cov, B, NC, Wr, br, E, bias_e, N, r, bias_mix);
  COMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int coma_routing_bwd(const float* cov, int32_t B, int32_t NC, const float* r, int32_t E, const float* bias_e, int32_t N,
                                const float* dr_w, const float* dbias_mix, float* dWr, float* dbr, float* dbias_e, void* stream) {
  COMA_CHECK(cov && r && dWr && dbr, "routing_bwd: null argument");
  COMA_CHECK(B >= 1 && E >= 1 && B * E <= ROUTE_MAX_BE && NC >= 1, "routing_bwd: B*E=%d out of range", B * E);
  COMA_CHECK(!(dbias_mix || dbias_e) || (bias_e && N >= 1), "routing_bwd: bias gradient without expert biases");
  hipLaunchKernelGGL(routing_bwd_k, dim3(1), dim3(256), 0, (hipStream_t)stream, cov, B, NC, r, E, bias_e, N, dr_w, dbias_mix, dWr,
                     dbr, dbias_e);
  COMA_LAUNCH_CHECK();
  return 0;
}
