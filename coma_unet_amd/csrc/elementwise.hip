// HBM-bound element-wise kernels: attention-gate pieces, strided add/copy (the concat-free
// channel-slice writes), ROI prior painting + prompt select, ROI-weighted MSE / L1 losses and
// fused AdamW.  Reference sites: attn_unet_data_parallel.py:139-150 (gate), :630-656 (UQ
// modulator tail), :736 (AdamW); criterions.py:181-211 (RoiMSE).
#include "common.h"

static unsigned ew_grid(int64_t total) {
  int64_t nb = (total + 255) / 256;
  if (nb > 8192) nb = 8192;
  return (unsigned)(nb < 1 ? 1 : nb);
}

struct T3 { const void* a; int64_t lda, sba; const void* b; int64_t ldb, sbb; void* o; int64_t ldo, sbo; int64_t V; int cv; };

static T3 mk3(const coma_tensor* a, const coma_tensor* b, const coma_tensor* o, int vec) {
  T3 p;
  p.a = a ? a->data : nullptr; p.lda = a ? a->ld : 0; p.sba = a ? (a->B == 1 && o->B > 1 ? 0 : a->sb) : 0;
  p.b = b ? b->data : nullptr; p.ldb = b ? b->ld : 0; p.sbb = b ? (b->B == 1 && o->B > 1 ? 0 : b->sb) : 0;
  p.o = o->data; p.ldo = o->ld; p.sbo = o->sb; p.V = t_vox(o); p.cv = o->C / vec;
  return p;
}
static int vec3(const coma_tensor* a, const coma_tensor* b, const coma_tensor* o) {
  int v = t_vec(o, 4);
  if (a && t_vec(a, 4) < v) v = t_vec(a, 4);
  if (b && t_vec(b, 4) < v) v = t_vec(b, 4);
  return v >= 4 ? 4 : 1;
}

enum { OP_ADD = 0, OP_ADD_RELU = 1, OP_RELU_BWD = 2, OP_COPY = 3 };

template <typename T, int VEC, int OP>
__global__ __launch_bounds__(256) void ew3_k(T3 p) {
  const int b = blockIdx.y;
  const T* ap = reinterpret_cast<const T*>(p.a) + (int64_t)b * p.sba;
  const T* bp = reinterpret_cast<const T*>(p.b) + (int64_t)b * p.sbb;
  T* op = reinterpret_cast<T*>(p.o) + (int64_t)b * p.sbo;
  const int64_t total = p.V * p.cv;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t v = e / p.cv; const int c0 = (int)(e - v * p.cv) * VEC;
    float av[VEC], bv[VEC], ov[VEC];
    vec_io<T, VEC>::load(ap + v * p.lda + c0, av);
    if (OP != OP_COPY) vec_io<T, VEC>::load(bp + v * p.ldb + c0, bv);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      if (OP == OP_ADD) ov[j] = av[j] + bv[j];
      else if (OP == OP_ADD_RELU) { const float t = av[j] + bv[j]; ov[j] = t > 0.f ? t : 0.f; }
      else if (OP == OP_RELU_BWD) ov[j] = av[j] > 0.f ? bv[j] : 0.f;   // a = forward output, b = dout
      else ov[j] = av[j];
    }
    vec_io<T, VEC>::store(op + v * p.ldo + c0, ov);
  }
}

template <int OP>
static int run_ew3(const coma_tensor* a, const coma_tensor* b, const coma_tensor* o, hipStream_t s) {
  const int vec = vec3(a, b, o);
  T3 p = mk3(a, b, o, vec);
  dim3 grid(ew_grid(p.V * p.cv), o->B);
#define L(T, V) hipLaunchKernelGGL((ew3_k<T, V, OP>), grid, dim3(256), 0, s, p)
  if (o->dtype == COMA_F32) { if (vec == 4) L(float, 4); else L(float, 1); }
  else { if (vec == 4) L(bf16_t, 4); else L(bf16_t, 1); }
#undef L
  COMA_LAUNCH_CHECK();
  return 0;
}

static int chk3(const char* name, const coma_tensor* a, const coma_tensor* b, const coma_tensor* o, bool bcast) {
  COMA_CHECK(a && o && a->data && o->data && (!b || b->data), "%s: null argument", name);
  const coma_tensor* ins[2] = {a, b};
  for (int i = 0; i < 2; ++i) {
    const coma_tensor* t = ins[i];
    if (!t) continue;
    COMA_CHECK(t->D == o->D && t->H == o->H && t->W == o->W && t->C == o->C && t->dtype == o->dtype,
               "%s: shape/dtype mismatch", name);
    COMA_CHECK(t->B == o->B || (bcast && t->B == 1), "%s: batch mismatch", name);
  }
  return 0;
}

extern "C" int coma_add(const coma_tensor* a, const coma_tensor* b, const coma_tensor* dst, void* stream) {
  if (int rc = chk3("coma_add", a, b, dst, true)) return rc;
  return b ? run_ew3<OP_ADD>(a, b, dst, (hipStream_t)stream) : run_ew3<OP_COPY>(a, a, dst, (hipStream_t)stream);
}
extern "C" int coma_add_relu_fwd(const coma_tensor* a, const coma_tensor* b, const coma_tensor* out, void* stream) {
  if (int rc = chk3("coma_add_relu_fwd", a, b, out, false)) return rc;
  return run_ew3<OP_ADD_RELU>(a, b, out, (hipStream_t)stream);
}
extern "C" int coma_add_relu_bwd(const coma_tensor* out, const coma_tensor* dout, const coma_tensor* da, void* stream) {
  if (int rc = chk3("coma_add_relu_bwd", out, dout, da, false)) return rc;
  return run_ew3<OP_RELU_BWD>(out, dout, da, (hipStream_t)stream);
}

// dst = T_dst(src): the model's input staging -- the external fp32 volume (B, C, D, H, W) with C == 1 IS a channels-last
// (B, D, H, W, 1) tensor; this writes it into the (pitch-8 padded) bf16 / fp32 internal buffer in one pass instead of an
// ATen strided-copy kernel (35 us at 2 x 128^3: attn_unet_data_parallel.py:661, the model's first touch of x)
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void cast_copy_k(const TS* src, int64_t lds_, int64_t sbs, TD* dst, int64_t ldd, int64_t sbd,
                                                   int64_t V, int C) {
  const int b = blockIdx.y;
  const int64_t total = V * C;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t v = C == 1 ? e : e / C; const int c = C == 1 ? 0 : (int)(e - v * C);
    st_f(dst + b * sbd + v * ldd + c, ld_f(src + b * sbs + v * lds_ + c));
  }
}
extern "C" int coma_cast_copy(const coma_tensor* src, const coma_tensor* dst, void* stream) {
  COMA_CHECK(src && dst && src->data && dst->data, "cast_copy: null argument");
  COMA_CHECK(t_same_grid(src, dst) && src->C == dst->C, "cast_copy: shape mismatch");
  const int64_t V = t_vox(src);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(ew_grid(V * src->C), src->B);
#define L(TS, TD) hipLaunchKernelGGL((cast_copy_k<TS, TD>), grid, dim3(256), 0, s, (const TS*)src->data, src->ld, src->sb, \
                                     (TD*)dst->data, dst->ld, dst->sb, V, src->C)
  if (src->dtype == COMA_F32) { if (dst->dtype == COMA_F32) L(float, float); else L(float, bf16_t); }
  else { if (dst->dtype == COMA_F32) L(bf16_t, float); else L(bf16_t, bf16_t); }
#undef L
  COMA_LAUNCH_CHECK();
  return 0;
}

// base[off .. off + len) = 0 for every (off, len) row of a device table: ONE launch clears the slots of the flat gradient
// buffer that no backward kernel overwrites (the optimizer's zero_grad; the rest of the buffer is written with "=")
__global__ __launch_bounds__(256) void zero_ranges_k(float* __restrict__ base, const int64_t* __restrict__ ranges, int n) {
  for (int r = blockIdx.y; r < n; r += gridDim.y) {
    float* p = base + ranges[2 * r];
    const int64_t len = ranges[2 * r + 1];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < len; i += (int64_t)gridDim.x * 256) p[i] = 0.f;
  }
}
extern "C" int coma_zero_ranges(float* base, const int64_t* ranges, int32_t n, int64_t max_len, void* stream) {
  COMA_CHECK(base && ranges && n >= 0, "zero_ranges: bad argument");
  if (n == 0) return 0;
  int64_t gx = (max_len + 255) / 256;
  if (gx > 64) gx = 64;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(zero_ranges_k, dim3((unsigned)gx, (unsigned)(n > 1024 ? 1024 : n)), dim3(256), 0, (hipStream_t)stream, base, ranges, n);
  COMA_LAUNCH_CHECK();
  return 0;
}

// dst[0] = sum_b src[b]
template <typename T>
__global__ __launch_bounds__(256) void batch_sum_k(const T* src, int64_t lds_, int64_t sbs, int B, T* dst, int64_t ldd,
                                                   int64_t V, int C) {
  const int64_t total = V * C;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t v = e / C; const int c = (int)(e - v * C);
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += ld_f(src + b * sbs + v * lds_ + c);
    st_f(dst + v * ldd + c, s);
  }
}
extern "C" int coma_batch_sum(const coma_tensor* src, const coma_tensor* dst, void* stream) {
  COMA_CHECK(src && dst && src->data && dst->data, "batch_sum: null argument");
  COMA_CHECK(src->D == dst->D && src->H == dst->H && src->W == dst->W && src->C == dst->C && src->dtype == dst->dtype,
             "batch_sum: shape mismatch");
  const int64_t V = t_vox(src);
  hipStream_t s = (hipStream_t)stream;
  if (src->dtype == COMA_F32)
    hipLaunchKernelGGL(batch_sum_k<float>, dim3(ew_grid(V * src->C)), dim3(256), 0, s, (const float*)src->data, src->ld,
                       src->sb, src->B, (float*)dst->data, dst->ld, V, src->C);
  else
    hipLaunchKernelGGL(batch_sum_k<bf16_t>, dim3(ew_grid(V * src->C)), dim3(256), 0, s, (const bf16_t*)src->data,
                       src->ld, src->sb, src->B, (bf16_t*)dst->data, dst->ld, V, src->C);
  COMA_LAUNCH_CHECK();
  return 0;
}

// ---- gate multiply: out[v][c] = x[v][c] * psi[v] ----
struct GateP { const void* x; int64_t ldx, sbx; const void* psi; int64_t ldp, sbp; const void* dout; int64_t ldo, sbo;
               void* out; int64_t ldy, sby; void* dpsi; int64_t lddp, sbdp; int64_t V; int C; int acc; };

template <typename T, int VEC>
__global__ __launch_bounds__(256) void gate_mul_fwd_k(GateP p) {
  const int b = blockIdx.y; const int cv = p.C / VEC;
  const T* xb = reinterpret_cast<const T*>(p.x) + (int64_t)b * p.sbx;
  const T* pb = reinterpret_cast<const T*>(p.psi) + (int64_t)b * p.sbp;
  T* ob = reinterpret_cast<T*>(p.out) + (int64_t)b * p.sby;
  const int64_t total = p.V * cv;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t v = e / cv; const int c0 = (int)(e - v * cv) * VEC;
    float xv[VEC], ov[VEC];
    vec_io<T, VEC>::load(xb + v * p.ldx + c0, xv);
    const float ps = ld_f(pb + v * p.ldp);
#pragma unroll
    for (int j = 0; j < VEC; ++j) ov[j] = xv[j] * ps;
    vec_io<T, VEC>::store(ob + v * p.ldy + c0, ov);
  }
}

// one thread group of (C/VEC) lanes per voxel; dpsi[v] = sum_c dout*x ; dx (=|+=) dout*psi
template <typename T, int VEC>
__global__ __launch_bounds__(256) void gate_mul_bwd_k(GateP p) {
  const int b = blockIdx.y; const int cv = p.C / VEC;   // cv is a power of two <= 64
  const T* xb = reinterpret_cast<const T*>(p.x) + (int64_t)b * p.sbx;
  const T* pb = reinterpret_cast<const T*>(p.psi) + (int64_t)b * p.sbp;
  const T* gb = reinterpret_cast<const T*>(p.dout) + (int64_t)b * p.sbo;
  T* dxb = reinterpret_cast<T*>(p.out) + (int64_t)b * p.sby;
  T* dpb = reinterpret_cast<T*>(p.dpsi) + (int64_t)b * p.sbdp;
  const int64_t total = p.V * cv;   // multiple of cv, so whole voxel groups stay within a wave
  for (int64_t e0 = (int64_t)blockIdx.x * 256; e0 < total; e0 += (int64_t)gridDim.x * 256) {
    const int64_t e = e0 + threadIdx.x;
    const bool live = e < total;
    const int64_t v = live ? e / cv : 0; const int c0 = live ? (int)(e - v * cv) * VEC : 0;
    float xv[VEC], gv[VEC], ov[VEC];
    float dot = 0.f;
    if (live) {
      vec_io<T, VEC>::load(xb + v * p.ldx + c0, xv);
      vec_io<T, VEC>::load(gb + v * p.ldo + c0, gv);
      const float ps = ld_f(pb + v * p.ldp);
      if (p.acc) vec_io<T, VEC>::load(dxb + v * p.ldy + c0, ov);
#pragma unroll
      for (int j = 0; j < VEC; ++j) { dot += xv[j] * gv[j]; ov[j] = (p.acc ? ov[j] : 0.f) + gv[j] * ps; }
      vec_io<T, VEC>::store(dxb + v * p.ldy + c0, ov);
    }
    for (int off = cv >> 1; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
    if (live && c0 == 0) st_f(dpb + v * p.lddp, dot);
  }
}

extern "C" int coma_gate_mul_fwd(const coma_tensor* x, const coma_tensor* psi, const coma_tensor* out, void* stream) {
  COMA_CHECK(x && psi && out && x->data && psi->data && out->data, "gate_mul_fwd: null argument");
  COMA_CHECK(t_same_grid(x, psi) && t_same_grid(x, out) && psi->C == 1 && x->C == out->C && x->dtype == psi->dtype &&
             x->dtype == out->dtype, "gate_mul_fwd: shape/dtype mismatch");
  const int vec = (t_vec(x, 4) >= 4 && t_vec(out, 4) >= 4) ? 4 : 1;
  GateP p{}; p.x = x->data; p.ldx = x->ld; p.sbx = x->sb; p.psi = psi->data; p.ldp = psi->ld; p.sbp = psi->sb;
  p.out = out->data; p.ldy = out->ld; p.sby = out->sb; p.V = t_vox(x); p.C = x->C;
  dim3 grid(ew_grid(p.V * (x->C / vec)), x->B);
  hipStream_t s = (hipStream_t)stream;
#define L(T, V) hipLaunchKernelGGL((gate_mul_fwd_k<T, V>), grid, dim3(256), 0, s, p)
  if (x->dtype == COMA_F32) { if (vec == 4) L(float, 4); else L(float, 1); }
  else { if (vec == 4) L(bf16_t, 4); else L(bf16_t, 1); }
#undef L
  COMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int coma_gate_mul_bwd(const coma_tensor* x, const coma_tensor* psi, const coma_tensor* dout,
                                 const coma_tensor* dx, int32_t accumulate_dx, const coma_tensor* dpsi, void* stream) {
  COMA_CHECK(x && psi && dout && dx && dpsi && x->data && psi->data && dout->data && dx->data && dpsi->data,
             "gate_mul_bwd: null argument");
  COMA_CHECK(t_same_grid(x, psi) && t_same_grid(x, dout) && t_same_grid(x, dx) && t_same_grid(x, dpsi) && psi->C == 1 &&
             dpsi->C == 1 && x->C == dout->C && x->C == dx->C, "gate_mul_bwd: shape mismatch");
  int vec = (t_vec(x, 4) >= 4 && t_vec(dout, 4) >= 4 && t_vec(dx, 4) >= 4) ? 4 : 1;
  int cv = x->C / vec;
  if (cv > 64) { COMA_CHECK(x->C % 64 == 0 && vec == 4 && x->C / 4 <= 64, "gate_mul_bwd: C=%d unsupported", x->C); }
  COMA_CHECK((cv & (cv - 1)) == 0 && cv <= 64, "gate_mul_bwd: C/vec=%d must be a power of two <= 64", cv);
  GateP p{}; p.x = x->data; p.ldx = x->ld; p.sbx = x->sb; p.psi = psi->data; p.ldp = psi->ld; p.sbp = psi->sb;
  p.dout = dout->data; p.ldo = dout->ld; p.sbo = dout->sb; p.out = dx->data; p.ldy = dx->ld; p.sby = dx->sb;
  p.dpsi = dpsi->data; p.lddp = dpsi->ld; p.sbdp = dpsi->sb; p.V = t_vox(x); p.C = x->C; p.acc = accumulate_dx;
  dim3 grid(ew_grid(p.V * cv), x->B);
  hipStream_t s = (hipStream_t)stream;
#define L(T, V) hipLaunchKernelGGL((gate_mul_bwd_k<T, V>), grid, dim3(256), 0, s, p)
  if (x->dtype == COMA_F32) { if (vec == 4) L(float, 4); else L(float, 1); }
  else { if (vec == 4) L(bf16_t, 4); else L(bf16_t, 1); }
#undef L
  COMA_LAUNCH_CHECK();
  return 0;
}

// ---- ROI prior painting + prompt select ----
struct RoiP { const float* roi; int64_t sbr; const void* x; int64_t sbx, ldx; const float* prior; const int32_t* ids; int n_roi;
              const float* abeta; const float* pos; const float* neg; void* out; int64_t ldo, sbo; int64_t V; };

__device__ __forceinline__ int roi_slot(const int32_t* ids, int n, float label) {
  const int li = (int)label;
  if ((float)li != label) return -1;
  for (int i = 0; i < n; ++i) if (ids[i] == li) return i;
  return -1;
}

template <typename T>
__global__ __launch_bounds__(256) void roi_paint_fwd_k(RoiP p) {
  __shared__ int32_t ids[64];
  __shared__ float pri[128];
  __shared__ signed char lut[COMA_ROI_LUT];
  const int b = blockIdx.y;
  if (threadIdx.x < p.n_roi) {
    ids[threadIdx.x] = p.ids[threadIdx.x];
    pri[2 * threadIdx.x] = p.prior[((int64_t)b * p.n_roi + threadIdx.x) * 2];
    pri[2 * threadIdx.x + 1] = p.prior[((int64_t)b * p.n_roi + threadIdx.x) * 2 + 1];
  }
  __syncthreads();
  roi_lut_build(lut, ids, p.n_roi);
  const float* prompt = p.abeta[b] == 1.f ? p.pos : p.neg;   // attn_unet_data_parallel.py:638-639
  const float* rb = p.roi + (int64_t)b * p.sbr;
  const T* xb = reinterpret_cast<const T*>(p.x) + (int64_t)b * p.sbx;
  T* ob = reinterpret_cast<T*>(p.out) + (int64_t)b * p.sbo;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < p.V; v += (int64_t)gridDim.x * 256) {
    float suvr = 0.f, sal = 0.f;
    const int slot = roi_lut_slot(lut, ids, p.n_roi, rb[v]);
    if (slot >= 0 && !(ld_f(xb + v * p.ldx) < 1e-4f)) { suvr = pri[2 * slot]; sal = pri[2 * slot + 1]; }
    T* o = ob + v * p.ldo;
    st_f(o, prompt[v]); st_f(o + 1, sal); st_f(o + 2, suvr);   // cat((prompt, saliency, suvr)) :651
  }
}

extern "C" int coma_roi_paint_fwd(const coma_tensor* roi, const coma_tensor* x, const float* prior,
                                  const int32_t* roi_ids, int32_t n_roi, const float* abeta, const float* pos_prompt,
                                  const float* neg_prompt, const coma_tensor* out3, void* stream) {
  COMA_CHECK(roi && x && out3 && roi->data && x->data && out3->data && prior && roi_ids && abeta && pos_prompt && neg_prompt,
             "roi_paint_fwd: null argument");
  COMA_CHECK(roi->dtype == COMA_F32 && roi->C == 1 && roi->ld == 1 && x->C == 1 && out3->C == 3 &&
             t_same_grid(roi, x) && t_same_grid(roi, out3) && x->dtype == out3->dtype, "roi_paint_fwd: shape/dtype mismatch");
  COMA_CHECK(n_roi > 0 && n_roi <= 64, "roi_paint_fwd: n_roi=%d out of range", n_roi);
  RoiP p; p.roi = (const float*)roi->data; p.sbr = roi->sb; p.x = x->data; p.sbx = x->sb; p.ldx = x->ld; p.prior = prior; p.ids = roi_ids;
  p.n_roi = n_roi; p.abeta = abeta; p.pos = pos_prompt; p.neg = neg_prompt; p.out = out3->data; p.ldo = out3->ld;
  p.sbo = out3->sb; p.V = t_vox(roi);
  dim3 grid(ew_grid(p.V), roi->B);
  hipStream_t s = (hipStream_t)stream;
  if (x->dtype == COMA_F32) hipLaunchKernelGGL(roi_paint_fwd_k<float>, grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL(roi_paint_fwd_k<bf16_t>, grid, dim3(256), 0, s, p);
  COMA_LAUNCH_CHECK();
  return 0;
}

template <typename T>
__global__ __launch_bounds__(256) void roi_paint_bwd_k(const T* d3, int64_t ld, int64_t sb, int B, const float* abeta,
                                                       float* dpos, float* dneg, int64_t V) {
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (int64_t)gridDim.x * 256) {
    float sp = 0.f, sn = 0.f;
    for (int b = 0; b < B; ++b) {
      const float g = ld_f(d3 + b * sb + v * ld);
      if (abeta[b] == 1.f) sp += g; else sn += g;
    }
    dpos[v] += sp; dneg[v] += sn;
  }
}
extern "C" int coma_roi_paint_bwd(const coma_tensor* dout3, const float* abeta, float* dpos, float* dneg, void* stream) {
  COMA_CHECK(dout3 && dout3->data && abeta && dpos && dneg && dout3->C == 3, "roi_paint_bwd: bad argument");
  const int64_t V = t_vox(dout3);
  hipStream_t s = (hipStream_t)stream;
  if (dout3->dtype == COMA_F32)
    hipLaunchKernelGGL(roi_paint_bwd_k<float>, dim3(ew_grid(V)), dim3(256), 0, s, (const float*)dout3->data, dout3->ld,
                       dout3->sb, dout3->B, abeta, dpos, dneg, V);
  else
    hipLaunchKernelGGL(roi_paint_bwd_k<bf16_t>, dim3(ew_grid(V)), dim3(256), 0, s, (const bf16_t*)dout3->data, dout3->ld,
                       dout3->sb, dout3->B, abeta, dpos, dneg, V);
  COMA_LAUNCH_CHECK();
  return 0;
}

// ---- losses ----
// partial[(b*nblk + blk)] = {sum mask, sum (p-g)^2 or |p-g|}
template <typename T, int L1>
__global__ __launch_bounds__(256) void loss_partial_k(const T* pred, int64_t sbp, int64_t ldp, const T* gt, int64_t sbg, const float* roi,
                                                      int64_t sbr, const int32_t* ids_g, const float* w_g, int n_roi,
                                                      int64_t V, double2* partial) {
  __shared__ int32_t ids[64];
  __shared__ float w[64];
  __shared__ double sh[256][2];
  __shared__ signed char lut[COMA_ROI_LUT];
  const int b = blockIdx.y;
  if (roi && threadIdx.x < n_roi) { ids[threadIdx.x] = ids_g[threadIdx.x]; w[threadIdx.x] = w_g[threadIdx.x]; }
  __syncthreads();
  if (roi) roi_lut_build(lut, ids, n_roi);
  double sm = 0.0, se = 0.0;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (int64_t)gridDim.x * 256) {
    const float d = ld_f(pred + b * sbp + v * ldp) - ld_f(gt + b * sbg + v);
    se += L1 ? (double)fabsf(d) : (double)d * (double)d;
    if (roi) { const int slot = roi_lut_slot(lut, ids, n_roi, roi[b * sbr + v]); if (slot >= 0) sm += (double)w[slot]; }
  }
  sh[threadIdx.x][0] = sm; sh[threadIdx.x][1] = se;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { sh[threadIdx.x][0] += sh[threadIdx.x + o][0]; sh[threadIdx.x][1] += sh[threadIdx.x + o][1]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[(int64_t)b * gridDim.x + blockIdx.x] = make_double2(sh[0][0], sh[0][1]);
}
__global__ void loss_finalize_k(const double2* partial, int nblk, int B, int64_t V, float* loss, float* mask_mean, int l1) {
  const int b = blockIdx.x, lane = threadIdx.x;     // one 64-lane wave per sample, fixed-order butterfly
  double sm = 0.0, se = 0.0;
  for (int k = lane; k < nblk; k += 64) { sm += partial[(int64_t)b * nblk + k].x; se += partial[(int64_t)b * nblk + k].y; }
  for (int o = 32; o > 0; o >>= 1) { sm += __shfl_xor(sm, o, 64); se += __shfl_xor(se, o, 64); }
  if (lane != 0) return;
  const double mm = sm / (double)V, ms = se / (double)V;
  if (l1) { loss[b] = (float)ms; }
  else { loss[b] = (float)(mm * ms); mask_mean[b] = (float)mm; }
}

#define LOSS_BLOCKS 512
extern "C" size_t coma_loss_ws_bytes(const coma_tensor* pred) { return (size_t)pred->B * LOSS_BLOCKS * sizeof(double2); }

static int loss_fwd(const coma_tensor* pred, const coma_tensor* gt, const coma_tensor* roi, const int32_t* ids,
                    const float* w, int n_roi, float* loss, float* mask_mean, void* ws, size_t ws_bytes, hipStream_t s, int l1) {
  COMA_CHECK(pred && gt && pred->data && gt->data && loss && ws, "loss: null argument");
  COMA_CHECK(t_same_grid(pred, gt) && pred->C == 1 && gt->C == 1 && gt->ld == 1 && pred->dtype == gt->dtype,
             "loss: pred/gt must be single-channel volumes of equal shape (gt contiguous)");
  COMA_CHECK(ws_bytes >= coma_loss_ws_bytes(pred), "loss: workspace too small");
  if (!l1) COMA_CHECK(roi && roi->data && roi->dtype == COMA_F32 && roi->C == 1 && roi->ld == 1 && t_same_grid(pred, roi) &&
                      ids && w && mask_mean && n_roi > 0 && n_roi <= 64, "roi_mse: bad roi arguments");
  const int64_t V = t_vox(pred);
  int nblk = (int)((V + 256 * 8 - 1) / (256 * 8));
  if (nblk > LOSS_BLOCKS) nblk = LOSS_BLOCKS;
  if (nblk < 1) nblk = 1;
  dim3 grid(nblk, pred->B);
  const float* rp = l1 ? nullptr : (const float*)roi->data;
  const int64_t sbr = l1 ? 0 : roi->sb;
#define L(T, K) hipLaunchKernelGGL((loss_partial_k<T, K>), grid, dim3(256), 0, s, (const T*)pred->data, pred->sb, pred->ld, \
                                   (const T*)gt->data, gt->sb, rp, sbr, ids, w, n_roi, V, (double2*)ws)
  if (pred->dtype == COMA_F32) { if (l1) L(float, 1); else L(float, 0); }
  else { if (l1) L(bf16_t, 1); else L(bf16_t, 0); }
#undef L
  COMA_LAUNCH_CHECK();
  hipLaunchKernelGGL(loss_finalize_k, dim3(pred->B), dim3(64), 0, s, (const double2*)ws, nblk, pred->B, V,
                     loss, mask_mean, l1);
  COMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int coma_roi_mse_fwd(const coma_tensor* pred, const coma_tensor* gt, const coma_tensor* roi, const int32_t* roi_ids,
                                const float* roi_w, int32_t n_roi, float* loss, float* mask_mean, void* ws, size_t ws_bytes,
                                void* stream) {
  return loss_fwd(pred, gt, roi, roi_ids, roi_w, n_roi, loss, mask_mean, ws, ws_bytes, (hipStream_t)stream, 0);
}
extern "C" int coma_l1_fwd(const coma_tensor* pred, const coma_tensor* gt, float* loss, void* ws, size_t ws_bytes, void* stream) {
  return loss_fwd(pred, gt, nullptr, nullptr, nullptr, 0, loss, nullptr, ws, ws_bytes, (hipStream_t)stream, 1);
}

template <typename T, int L1>
__global__ __launch_bounds__(256) void loss_bwd_k(const T* pred, int64_t sbp, int64_t ldp, const T* gt, int64_t sbg, const float* gout,
                                                  const float* mask_mean, T* dp, int64_t sbd, int64_t ldd, int64_t V) {
  const int b = blockIdx.y;
  const float scale = gout[b] * (L1 ? 1.f : mask_mean[b] * 2.f) / (float)V;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (int64_t)gridDim.x * 256) {
    const float d = ld_f(pred + b * sbp + v * ldp) - ld_f(gt + b * sbg + v);
    const float g = L1 ? (d > 0.f ? scale : (d < 0.f ? -scale : 0.f)) : scale * d;
    st_f(dp + b * sbd + v * ldd, g);
  }
}
static int loss_bwd(const coma_tensor* pred, const coma_tensor* gt, const float* gout, const float* mask_mean,
                    const coma_tensor* dpred, hipStream_t s, int l1) {
  COMA_CHECK(pred && gt && dpred && pred->data && gt->data && dpred->data && gout && (l1 || mask_mean), "loss_bwd: null argument");
  COMA_CHECK(t_same_grid(pred, gt) && t_same_grid(pred, dpred) && pred->C == 1 && gt->ld == 1 &&
             pred->dtype == gt->dtype && pred->dtype == dpred->dtype, "loss_bwd: shape/dtype mismatch");
  const int64_t V = t_vox(pred);
  dim3 grid(ew_grid(V), pred->B);
#define L(T, K) hipLaunchKernelGGL((loss_bwd_k<T, K>), grid, dim3(256), 0, s, (const T*)pred->data, pred->sb, pred->ld, \
                                   (const T*)gt->data, gt->sb, gout, mask_mean, (T*)dpred->data, dpred->sb, dpred->ld, V)
  if (pred->dtype == COMA_F32) { if (l1) L(float, 1); else L(float, 0); }
  else { if (l1) L(bf16_t, 1); else L(bf16_t, 0); }
#undef L
  COMA_LAUNCH_CHECK();
  return 0;
}
extern "C" int coma_roi_mse_bwd(const coma_tensor* pred, const coma_tensor* gt, const float* gout, const float* mask_mean,
                                const coma_tensor* dpred, void* stream) {
  return loss_bwd(pred, gt, gout, mask_mean, dpred, (hipStream_t)stream, 0);
}
extern "C" int coma_l1_bwd(const coma_tensor* pred, const coma_tensor* gt, const float* gout, const coma_tensor* dpred, void* stream) {
  return loss_bwd(pred, gt, gout, nullptr, dpred, (hipStream_t)stream, 1);
}

// ---- AdamW (torch.optim.AdamW: decoupled decay, bias-corrected, eps outside sqrt(v_hat)) ----
__global__ __launch_bounds__(256) void adamw_k(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1,
                                               float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                               const int32_t* step_dev, int vec4) {
  if (step_dev) {   // step count lives on the device (hipGraph replays must not freeze it)
    const float st = (float)*step_dev;
    bc1 = 1.f - powf(b1, st);
    bc2_sqrt = sqrtf(1.f - powf(b2, st));
  }
  auto upd = [&](float& pi, float gi, float& mi, float& vi) {
    pi *= (1.f - lr * wd);
    mi = b1 * mi + (1.f - b1) * gi;
    vi = b2 * vi + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= (lr / bc1) * (mi / denom);
  };
  const int64_t n4 = vec4 ? n >> 2 : 0;        // 16 bytes per lane and stream where the four buffers are 16-byte aligned
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 pv = reinterpret_cast<float4*>(p)[i], mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    upd(pv.x, gv.x, mv.x, vv.x); upd(pv.y, gv.y, mv.y, vv.y); upd(pv.z, gv.z, mv.z, vv.z); upd(pv.w, gv.w, mv.w, vv.w);
    reinterpret_cast<float4*>(p)[i] = pv; reinterpret_cast<float4*>(m)[i] = mv; reinterpret_cast<float4*>(v)[i] = vv;
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float pi = p[i], mi = m[i], vi = v[i];
    upd(pi, g[i], mi, vi);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}
extern "C" int coma_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                          float eps, float weight_decay, int32_t step, const int32_t* step_dev, void* stream) {
  COMA_CHECK(p && g && m && v && n >= 0 && (step >= 1 || step_dev), "adamw: bad argument");
  if (n == 0) return 0;
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2 = 1.f - powf(beta2, (float)step);
  const int vec4 = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
  hipLaunchKernelGGL(adamw_k, dim3(ew_grid(vec4 ? (n + 3) / 4 : n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1,
                     beta2, eps, weight_decay, bc1, sqrtf(bc2), step_dev, vec4);
  COMA_LAUNCH_CHECK();
  return 0;
}
