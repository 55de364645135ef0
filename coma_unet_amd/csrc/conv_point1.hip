// 1x1x1 convolutions with ONE channel on one side: the attention gate's psi (C -> 1), the U-Net's reduce conv (32 -> 1),
// the final prediction head (2 -> 1), the projection heads' C -> 1 (attn_unet_data_parallel.py:104-137,318-325,556 and
// the MONAI ConvBlock of ProjectionHead), their data-gradients (1 -> C) and weight-gradients.  Per voxel this is a dot
// product / a scaled copy / a weighted column sum: pure HBM streaming.  The general VALU kernel ran them at 0.9 TB/s
// (element-wise bf16 loads, one thread per voxel); here every lane moves 16 bytes and consecutive lanes are contiguous.
// fp32 accumulation and fp32 weights in both activation dtypes (this is also the exact-fp32 mode's path).
#include "common.h"

template <typename T> struct Pk;   // 16 bytes of channels
template <> struct Pk<bf16_t> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void load(const bf16_t* p, float* o) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[2 * j] = __uint_as_float(u[j] << 16); o[2 * j + 1] = __uint_as_float(u[j] & 0xffff0000u); }
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float* o) {
    bf16_t t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = static_cast<bf16_t>(o[j]);
    *reinterpret_cast<uint4*>(p) = *reinterpret_cast<const uint4*>(t);
  }
};
template <> struct Pk<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void load(const float* p, float* o) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
  static __device__ __forceinline__ void store(float* p, const float* o) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
  }
};

struct P1 {
  const void* a; int64_t lda, sba; int K;     // the K-channel tensor
  const void* s; int64_t lds_, sbs;           // the 1-channel tensor (read by to-K / wgrad, written by to-1)
  void* o; int64_t ldo, sbo;                  // output tensor
  const float* w; int64_t wsb;                // K weights per sample group
  const float* bias; int64_t bsb;
  float* dw; int64_t dwsb;
  int64_t V;
  int cp;                                     // 16-byte pieces per voxel (vector path), power of two <= 64; 0 = scalar path
};

// y[v] = bias + sum_k a[v][k] * w[k]
template <typename T>
__global__ __launch_bounds__(256) void p1_dot_k(P1 p) {
  constexpr int PN = Pk<T>::N;
  __shared__ float ws[1024];
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < p.K; i += 256) ws[i] = p.w[b * p.wsb + i];
  __syncthreads();
  const T* ab = reinterpret_cast<const T*>(p.a) + b * p.sba;
  T* ob = reinterpret_cast<T*>(p.o) + b * p.sbo;
  const float bv = p.bias ? p.bias[b * p.bsb] : 0.f;
  if (p.cp) {
    const int cp = p.cp, ch = threadIdx.x & (cp - 1);
    const int64_t vstep = (int64_t)gridDim.x * 256 / cp;
    float wv[PN];
#pragma unroll
    for (int j = 0; j < PN; ++j) wv[j] = ws[ch * PN + j];
    for (int64_t v = ((int64_t)blockIdx.x * 256 + threadIdx.x) / cp; v < (p.V + vstep - 1) / vstep * vstep; v += vstep) {
      float acc = 0.f;
      if (v < p.V) {
        float xv[PN];
        Pk<T>::load(ab + v * p.lda + ch * PN, xv);
#pragma unroll
        for (int j = 0; j < PN; ++j) acc = fmaf(xv[j], wv[j], acc);
      }
      for (int o = 1; o < cp; o <<= 1) acc += __shfl_xor(acc, o, 64);     // (whole wave executes: v is uniform per lane group)
      if (ch == 0 && v < p.V) st_f(ob + v * p.ldo, acc + bv);
    }
  } else {
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < p.V; v += (int64_t)gridDim.x * 256) {
      float acc = bv;
      for (int k = 0; k < p.K; ++k) acc = fmaf(ld_f(ab + v * p.lda + k), ws[k], acc);
      st_f(ob + v * p.ldo, acc);
    }
  }
}

// y[v][k] = bias[k] + s[v] * w[k]
template <typename T>
__global__ __launch_bounds__(256) void p1_scale_k(P1 p) {
  constexpr int PN = Pk<T>::N;
  __shared__ float ws[1024], bs[1024];
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < p.K; i += 256) { ws[i] = p.w[b * p.wsb + i]; bs[i] = p.bias ? p.bias[b * p.bsb + i] : 0.f; }
  __syncthreads();
  const T* sb = reinterpret_cast<const T*>(p.s) + b * p.sbs;
  T* ob = reinterpret_cast<T*>(p.o) + b * p.sbo;
  if (p.cp) {
    const int cp = p.cp, ch = threadIdx.x & (cp - 1);
    const int64_t vstep = (int64_t)gridDim.x * 256 / cp;
    float wv[PN], bb[PN];
#pragma unroll
    for (int j = 0; j < PN; ++j) { wv[j] = ws[ch * PN + j]; bb[j] = bs[ch * PN + j]; }
    for (int64_t v = ((int64_t)blockIdx.x * 256 + threadIdx.x) / cp; v < p.V; v += vstep) {
      const float sv = ld_f(sb + v * p.lds_);
      float ov[PN];
#pragma unroll
      for (int j = 0; j < PN; ++j) ov[j] = fmaf(sv, wv[j], bb[j]);
      Pk<T>::store(ob + v * p.ldo + ch * PN, ov);
    }
  } else {
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < p.V; v += (int64_t)gridDim.x * 256) {
      const float sv = ld_f(sb + v * p.lds_);
      for (int k = 0; k < p.K; ++k) st_f(ob + v * p.ldo + k, fmaf(sv, ws[k], bs[k]));
    }
  }
}

// dw[k] += sum_v a[v][k] * s[v]     (dw pre-zeroed; one atomic per channel per block)
template <typename T>
__global__ __launch_bounds__(256) void p1_wsum_k(P1 p) {
  constexpr int PN = Pk<T>::N;
  __shared__ float acc_s[1024];
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < p.K; i += 256) acc_s[i] = 0.f;
  __syncthreads();
  const T* ab = reinterpret_cast<const T*>(p.a) + b * p.sba;
  const T* sb = reinterpret_cast<const T*>(p.s) + b * p.sbs;
  if (p.cp) {
    const int cp = p.cp, ch = threadIdx.x & (cp - 1);
    const int64_t vstep = (int64_t)gridDim.x * 256 / cp;
    float acc[PN];
#pragma unroll
    for (int j = 0; j < PN; ++j) acc[j] = 0.f;
    for (int64_t v = ((int64_t)blockIdx.x * 256 + threadIdx.x) / cp; v < p.V; v += vstep) {
      const float sv = ld_f(sb + v * p.lds_);
      float xv[PN];
      Pk<T>::load(ab + v * p.lda + ch * PN, xv);
#pragma unroll
      for (int j = 0; j < PN; ++j) acc[j] = fmaf(xv[j], sv, acc[j]);
    }
    // lanes with the same chunk inside a wave: butterfly over the lane bits above log2(cp)
#pragma unroll
    for (int j = 0; j < PN; ++j) {
      float a = acc[j];
      for (int o = cp; o < 64; o <<= 1) a += __shfl_xor(a, o, 64);
      if ((threadIdx.x & 63) < cp) atomicAdd(&acc_s[ch * PN + j], a);
    }
  } else {
    for (int k = 0; k < p.K; ++k) {
      float a = 0.f;
      for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < p.V; v += (int64_t)gridDim.x * 256)
        a = fmaf(ld_f(ab + v * p.lda + k), ld_f(sb + v * p.lds_), a);
      for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
      if ((threadIdx.x & 63) == 0) atomicAdd(&acc_s[k], a);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < p.K; i += 256) atomicAdd(p.dw + b * p.dwsb + i, acc_s[i]);
}

static int pieces(const coma_tensor* t) {       // 16-byte pieces per voxel if the vector path is legal, else 0
  const int pn = t->dtype == COMA_BF16 ? 8 : 4;
  if (t->C % pn || t->ld % pn || t->sb % pn || ((uintptr_t)t->data & 15)) return 0;
  const int cp = t->C / pn;
  return (cp <= 64 && (cp & (cp - 1)) == 0) ? cp : 0;
}
static unsigned p1_grid(int64_t V, int cp) {
  int64_t work = V * (cp ? cp : 1);
  int64_t nb = (work + 256 * 8 - 1) / (256 * 8);
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  return (unsigned)nb;
}

bool conv_point1_ok(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* y) {
  return d->ksize == 1 && d->stride == 1 && (y->C == 1 || x->C == 1) && x->C <= 1024 && y->C <= 1024 && x->dtype == y->dtype;
}

// forward / data-gradient (1x1x1, stride 1: the two gather forms coincide); wk fp32 [Bw][1][N][C]
int conv_point1_fwd(const coma_conv_desc* d, const coma_tensor* x, const float* wk, const float* bias, const coma_tensor* y,
                    hipStream_t s) {
  P1 p{};
  p.V = t_vox(x); p.w = wk; p.bias = bias;
  if (y->C == 1) {
    p.a = x->data; p.lda = x->ld; p.sba = x->sb; p.K = x->C;
    p.o = y->data; p.ldo = y->ld; p.sbo = y->sb;
    p.wsb = d->per_sample_w ? x->C : 0; p.bsb = d->per_sample_w ? 1 : 0;
    p.cp = pieces(x);
    dim3 grid(p1_grid(p.V, p.cp), (unsigned)x->B);
    coma_set_kernel_tag("p1_dot_k<%s>", x->dtype == COMA_F32 ? "float" : "__bf16");
    if (x->dtype == COMA_F32) hipLaunchKernelGGL(p1_dot_k<float>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(p1_dot_k<bf16_t>, grid, dim3(256), 0, s, p);
  } else {
    p.s = x->data; p.lds_ = x->ld; p.sbs = x->sb; p.K = y->C;
    p.o = y->data; p.ldo = y->ld; p.sbo = y->sb;
    p.wsb = d->per_sample_w ? y->C : 0; p.bsb = d->per_sample_w ? y->C : 0;
    p.cp = pieces(y);
    dim3 grid(p1_grid(p.V, p.cp), (unsigned)x->B);
    coma_set_kernel_tag("p1_scale_k<%s>", x->dtype == COMA_F32 ? "float" : "__bf16");
    if (x->dtype == COMA_F32) hipLaunchKernelGGL(p1_scale_k<float>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(p1_scale_k<bf16_t>, grid, dim3(256), 0, s, p);
  }
  COMA_LAUNCH_CHECK();
  return 0;
}

// dwk [Bw][1][N][C] with N == 1 or C == 1: a K-vector per sample group
int conv_point1_wgrad(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* dy, float* dwk, hipStream_t s, int zeroed) {
  P1 p{};
  p.V = t_vox(x);
  const coma_tensor* a = dy->C == 1 ? x : dy;      // the K-channel side
  const coma_tensor* o = dy->C == 1 ? dy : x;      // the 1-channel side
  p.a = a->data; p.lda = a->ld; p.sba = a->sb; p.K = a->C;
  p.s = o->data; p.lds_ = o->ld; p.sbs = o->sb;
  p.dw = dwk; p.dwsb = d->per_sample_w ? a->C : 0;
  p.cp = pieces(a);
  const int Bw = d->per_sample_w ? x->B : 1;
  if (!(zeroed & COMA_ZEROED_OUT) && hipMemsetAsync(dwk, 0, sizeof(float) * a->C * Bw, s) != hipSuccess) { coma_set_error("wgrad memset failed"); return 2; }
  unsigned nb = p1_grid(p.V, p.cp);
  if (nb > 512) nb = 512;
  dim3 grid(nb, (unsigned)x->B);
  coma_set_kernel_tag("p1_wsum_k<%s>", x->dtype == COMA_F32 ? "float" : "__bf16");
  if (x->dtype == COMA_F32) hipLaunchKernelGGL(p1_wsum_k<float>, grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL(p1_wsum_k<bf16_t>, grid, dim3(256), 0, s, p);
  COMA_LAUNCH_CHECK();
  return 0;
}
