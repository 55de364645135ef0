// Direct (VALU, fp32-accumulate) gather convolution for gfx950.
//
// Permanent path for the tiny-channel full-resolution layers (Cin=1 head conv,
// 32->1 reduce, the 1/2/3/8/16-channel prompt/UQ tail, psi) which are HBM-bound
// and would waste MFMA tiles (SURVEY.md section 7 "hard parts"), and the exact-fp32 path
// of every other conv.  Replaces the cuDNN dispatches behind nn.Conv3d /
// nn.ConvTranspose3d of attn_unet_data_parallel.py (MONAI Convolution call sites
// :229,:495-497,:547-558; CondConv call sites :126,:289-306,:318-325).
#include "common.h"

struct ConvP {
  const void* x; int64_t ldx, sbx; int Di, Hi, Wi, C;
  void* y; int64_t ldy, sby; int Do, Ho, Wo, N;
  const float* w; int64_t wsb;
  const float* bias; int64_t bsb;
  int k, stride, pad, form;
};

// input coordinate of output coordinate o for kernel offset t; returns false if padding
__device__ __forceinline__ bool in_coord(int form, int o, int t, int stride, int pad, int Ni, int& i) {
  if (form == 0) {
    i = o * stride - pad + t;
    return i >= 0 && i < Ni;
  }
  int u = o + pad - t;
  if (u < 0) return false;
  if (stride == 2) { if (u & 1) return false; i = u >> 1; } else { i = u; }
  return i < Ni;
}

template <typename T, int NT, int VEC>
__global__ __launch_bounds__(256) void conv_direct_fwd_k(ConvP p) {
  const int64_t M = (int64_t)p.Do * p.Ho * p.Wo;
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const int b = blockIdx.z;
  const int n0 = blockIdx.y * NT;
  const int ox = (int)(m % p.Wo), oy = (int)((m / p.Wo) % p.Ho), oz = (int)(m / ((int64_t)p.Wo * p.Ho));
  const T* xb = reinterpret_cast<const T*>(p.x) + (int64_t)b * p.sbx;
  const float* wb = p.w + (int64_t)b * p.wsb;
  float acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) acc[n] = p.bias ? p.bias[(int64_t)b * p.bsb + n0 + n] : 0.f;
  const int k = p.k;
  for (int kz = 0; kz < k; ++kz) {
    int iz; const bool vz = in_coord(p.form, oz, kz, p.stride, p.pad, p.Di, iz);
    for (int ky = 0; ky < k; ++ky) {
      int iy; const bool vy = in_coord(p.form, oy, ky, p.stride, p.pad, p.Hi, iy);
      for (int kx = 0; kx < k; ++kx) {
        int ix; const bool vx = in_coord(p.form, ox, kx, p.stride, p.pad, p.Wi, ix);
        if (!(vz && vy && vx)) continue;
        const int tap = (kz * k + ky) * k + kx;
        const T* xp = xb + (((int64_t)iz * p.Hi + iy) * p.Wi + ix) * p.ldx;
        const float* wp = wb + ((int64_t)tap * p.N + n0) * p.C;
        for (int c = 0; c < p.C; c += VEC) {
          float xv[VEC];
          vec_io<T, VEC>::load(xp + c, xv);
#pragma unroll
          for (int n = 0; n < NT; ++n) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[n] = fmaf(xv[j], wp[(int64_t)n * p.C + c + j], acc[n]);
          }
        }
      }
    }
  }
  T* yp = reinterpret_cast<T*>(p.y) + (int64_t)b * p.sby + m * p.ldy + n0;
#pragma unroll
  for (int n = 0; n < NT; ++n) st_f(yp + n, acc[n]);
}

template <typename T, int NT>
static int launch_fwd_vec(const ConvP& p, int B, int vec, hipStream_t s) {
  const int64_t M = (int64_t)p.Do * p.Ho * p.Wo;
  dim3 grid((unsigned)((M + 255) / 256), (unsigned)(p.N / NT), (unsigned)B);
  coma_set_kernel_tag("conv_direct_fwd_k<%s, %d, %d>", sizeof(T) == 4 ? "float" : "__bf16", NT, vec >= 4 ? 4 : 1);
  if (vec >= 4) hipLaunchKernelGGL((conv_direct_fwd_k<T, NT, 4>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((conv_direct_fwd_k<T, NT, 1>), grid, dim3(256), 0, s, p);
  COMA_LAUNCH_CHECK();
  return 0;
}

template <typename T>
static int launch_fwd(const ConvP& p, int B, int vec, hipStream_t s) {
  if (p.N % 16 == 0) return launch_fwd_vec<T, 16>(p, B, vec, s);
  if (p.N % 8 == 0) return launch_fwd_vec<T, 8>(p, B, vec, s);
  if (p.N % 4 == 0) return launch_fwd_vec<T, 4>(p, B, vec, s);
  if (p.N % 3 == 0) return launch_fwd_vec<T, 3>(p, B, vec, s);
  if (p.N % 2 == 0) return launch_fwd_vec<T, 2>(p, B, vec, s);
  return launch_fwd_vec<T, 1>(p, B, vec, s);
}

int conv_check(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* y) {
  COMA_CHECK(d && x && y && x->data && y->data, "conv: null argument");
  COMA_CHECK(d->ksize == 1 || d->ksize == 3, "conv: ksize %d unsupported", d->ksize);
  COMA_CHECK(d->stride == 1 || d->stride == 2, "conv: stride %d unsupported", d->stride);
  COMA_CHECK(d->pad == (d->ksize - 1) / 2, "conv: pad %d unsupported for ksize %d", d->pad, d->ksize);
  COMA_CHECK(x->B == y->B, "conv: batch mismatch %d vs %d", x->B, y->B);
  COMA_CHECK(x->dtype == y->dtype, "conv: dtype mismatch");
  COMA_CHECK(x->ld >= x->C && y->ld >= y->C, "conv: pitch smaller than channels");
  const int s = d->stride;
  if (d->form == 0) {  // y grid = floor((in + 2p - k)/s) + 1
    const int k = d->ksize, pd = d->pad;
    COMA_CHECK(y->D == (x->D + 2 * pd - k) / s + 1 && y->H == (x->H + 2 * pd - k) / s + 1 &&
               y->W == (x->W + 2 * pd - k) / s + 1, "conv: output grid %dx%dx%d does not match input %dx%dx%d",
               y->D, y->H, y->W, x->D, x->H, x->W);
  } else {  // transposed gather: the producing (strided) conv must map y's grid onto x's
    const int k = d->ksize, pd = d->pad;
    COMA_CHECK(x->D == (y->D + 2 * pd - k) / s + 1 && x->H == (y->H + 2 * pd - k) / s + 1 &&
               x->W == (y->W + 2 * pd - k) / s + 1, "tconv: output grid %dx%dx%d does not match input %dx%dx%d",
               y->D, y->H, y->W, x->D, x->H, x->W);
  }
  return 0;
}

int conv_direct_fwd(const coma_conv_desc* d, const coma_tensor* x, const float* wk, const float* bias,
                    const coma_tensor* y, hipStream_t s) {
  ConvP p;
  p.x = x->data; p.ldx = x->ld; p.sbx = x->sb; p.Di = x->D; p.Hi = x->H; p.Wi = x->W; p.C = x->C;
  p.y = y->data; p.ldy = y->ld; p.sby = y->sb; p.Do = y->D; p.Ho = y->H; p.Wo = y->W; p.N = y->C;
  const int taps = d->ksize * d->ksize * d->ksize;
  p.w = wk; p.wsb = d->per_sample_w ? (int64_t)taps * y->C * x->C : 0;
  p.bias = bias; p.bsb = d->per_sample_w ? y->C : 0;
  p.k = d->ksize; p.stride = d->stride; p.pad = d->pad; p.form = d->form;
  const int vec = t_vec(x, 4);
  if (x->dtype == COMA_F32) return launch_fwd<float>(p, x->B, vec, s);
  return launch_fwd<bf16_t>(p, x->B, vec, s);
}

// ---------------------------------------------------------------------------------
// weight gradient: dwk[b][tap][n][c] += sum_m dy[m][n] * x[pos(m,tap)][c]
// LDS-tiled fp32 FMA GEMM over voxel chunks; chunk partials merged with fp32 atomics.
// ---------------------------------------------------------------------------------
struct WgradP {
  const void* x; int64_t ldx, sbx; int Di, Hi, Wi, C;
  const void* dy; int64_t ldy, sby; int Do, Ho, Wo, N;
  float* dwk; int64_t wsb;
  int k, stride, pad, form;
  int mch;        // voxels per block
  int ntn, ntc;   // tiles along n and c
};

template <typename T, int TS>  // thread tile TS x TS, block tile (16*TS) x (16*TS)
__global__ __launch_bounds__(256) void conv_direct_wgrad_k(WgradP p) {
  constexpr int TILE = 16 * TS;
  constexpr int MS = 32;  // voxels per LDS stage
  __shared__ float dyS[MS][TILE + 1];
  __shared__ float xS[MS][TILE + 1];
  const int tid = threadIdx.x;
  const int tn = tid / 16, tc = tid % 16;
  const int b = blockIdx.z;
  int yi = blockIdx.y;
  const int ct = yi % p.ntc; yi /= p.ntc;
  const int nt = yi % p.ntn; yi /= p.ntn;
  const int tap = yi;
  const int k = p.k;
  const int kx = tap % k, ky = (tap / k) % k, kz = tap / (k * k);
  const int64_t M = (int64_t)p.Do * p.Ho * p.Wo;
  const int64_t m_begin = (int64_t)blockIdx.x * p.mch;
  const int64_t m_end = (m_begin + p.mch < M) ? m_begin + p.mch : M;
  const T* xb = reinterpret_cast<const T*>(p.x) + (int64_t)b * p.sbx;
  const T* dyb = reinterpret_cast<const T*>(p.dy) + (int64_t)b * p.sby;
  const int n_base = nt * TILE, c_base = ct * TILE;
  float acc[TS][TS];
#pragma unroll
  for (int i = 0; i < TS; ++i)
#pragma unroll
    for (int j = 0; j < TS; ++j) acc[i][j] = 0.f;

  for (int64_t ms = m_begin; ms < m_end; ms += MS) {
    // cooperative load: MS x TILE elements of each operand
    for (int e = tid; e < MS * TILE; e += 256) {
      const int mm = e / TILE, col = e % TILE;
      const int64_t m = ms + mm;
      float dv = 0.f, xv = 0.f;
      if (m < m_end) {
        const int n = n_base + col;
        if (n < p.N) dv = ld_f(dyb + m * p.ldy + n);
        const int c = c_base + col;
        if (c < p.C) {
          const int ox = (int)(m % p.Wo), oy = (int)((m / p.Wo) % p.Ho), oz = (int)(m / ((int64_t)p.Wo * p.Ho));
          int iz, iy, ix;
          if (in_coord(p.form, oz, kz, p.stride, p.pad, p.Di, iz) &&
              in_coord(p.form, oy, ky, p.stride, p.pad, p.Hi, iy) &&
              in_coord(p.form, ox, kx, p.stride, p.pad, p.Wi, ix))
            xv = ld_f(xb + (((int64_t)iz * p.Hi + iy) * p.Wi + ix) * p.ldx + c);
        }
      }
      dyS[mm][col] = dv;
      xS[mm][col] = xv;
    }
    __syncthreads();
#pragma unroll 4
    for (int mm = 0; mm < MS; ++mm) {
      float a[TS], bb[TS];
#pragma unroll
      for (int i = 0; i < TS; ++i) a[i] = dyS[mm][tn * TS + i];
#pragma unroll
      for (int j = 0; j < TS; ++j) bb[j] = xS[mm][tc * TS + j];
#pragma unroll
      for (int i = 0; i < TS; ++i)
#pragma unroll
        for (int j = 0; j < TS; ++j) acc[i][j] = fmaf(a[i], bb[j], acc[i][j]);
    }
    __syncthreads();
  }
  float* wout = p.dwk + (int64_t)b * p.wsb + (int64_t)tap * p.N * p.C;
#pragma unroll
  for (int i = 0; i < TS; ++i) {
    const int n = n_base + tn * TS + i;
    if (n >= p.N) continue;
#pragma unroll
    for (int j = 0; j < TS; ++j) {
      const int c = c_base + tc * TS + j;
      if (c < p.C && acc[i][j] != 0.f) atomicAdd(wout + (int64_t)n * p.C + c, acc[i][j]);
    }
  }
}

int conv_direct_wgrad(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* dy, float* dwk,
                      hipStream_t s, int zeroed) {
  WgradP p;
  p.x = x->data; p.ldx = x->ld; p.sbx = x->sb; p.Di = x->D; p.Hi = x->H; p.Wi = x->W; p.C = x->C;
  p.dy = dy->data; p.ldy = dy->ld; p.sby = dy->sb; p.Do = dy->D; p.Ho = dy->H; p.Wo = dy->W; p.N = dy->C;
  const int taps = d->ksize * d->ksize * d->ksize;
  const int64_t wsz = (int64_t)taps * dy->C * x->C;
  p.dwk = dwk; p.wsb = d->per_sample_w ? wsz : 0;
  p.k = d->ksize; p.stride = d->stride; p.pad = d->pad; p.form = d->form;
  const int Bw = d->per_sample_w ? x->B : 1;
  if (!(zeroed & COMA_ZEROED_OUT) && hipMemsetAsync(dwk, 0, sizeof(float) * wsz * Bw, s) != hipSuccess) { coma_set_error("wgrad memset failed"); return 2; }
  const int64_t M = (int64_t)dy->D * dy->H * dy->W;
  const int big = dy->C > x->C ? dy->C : x->C;
  const int TS = big > 32 ? 4 : (big > 16 ? 2 : 1);
  const int TILE = 16 * TS;
  int64_t mch = (M + 127) / 128;
  if (mch < 256) mch = 256;
  mch = (mch + 31) / 32 * 32;
  p.mch = (int)mch;
  p.ntn = (dy->C + TILE - 1) / TILE; p.ntc = (x->C + TILE - 1) / TILE;
  dim3 grid((unsigned)((M + mch - 1) / mch), (unsigned)(taps * p.ntn * p.ntc), (unsigned)x->B);
  coma_set_kernel_tag("conv_direct_wgrad_k<%s, %d>", x->dtype == COMA_F32 ? "float" : "__bf16", TS);
#define WG_LAUNCH(T, TSV) hipLaunchKernelGGL((conv_direct_wgrad_k<T, TSV>), grid, dim3(256), 0, s, p)
  if (x->dtype == COMA_F32) {
    if (TS == 4) WG_LAUNCH(float, 4); else if (TS == 2) WG_LAUNCH(float, 2); else WG_LAUNCH(float, 1);
  } else {
    if (TS == 4) WG_LAUNCH(bf16_t, 4); else if (TS == 2) WG_LAUNCH(bf16_t, 2); else WG_LAUNCH(bf16_t, 1);
  }
#undef WG_LAUNCH
  COMA_LAUNCH_CHECK();
  return 0;
}
