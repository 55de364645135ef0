// Shared device/host helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/coma_unet.h"

typedef __bf16 bf16_t;

void coma_set_error(const char* fmt, ...);
// name of the convolution kernel variant a dispatch is about to launch (as rocprofv3 prints it): the host reads it back
// through coma_last_kernel() to label its per-launch HIP-event timings (bench.py's roofline leg)
void coma_set_kernel_tag(const char* fmt, ...);

#define COMA_CHECK(cond, ...)                         \
  do {                                                \
    if (!(cond)) {                                    \
      coma_set_error(__VA_ARGS__);                    \
      return 1;                                       \
    }                                                 \
  } while (0)

#define COMA_LAUNCH_CHECK()                                           \
  do {                                                                \
    hipError_t e_ = hipGetLastError();                                \
    if (e_ != hipSuccess) {                                           \
      coma_set_error("%s:%d launch: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return 2;                                                       \
    }                                                                 \
  } while (0)

__device__ __forceinline__ float ld_f(const float* p) { return *p; }
__device__ __forceinline__ float ld_f(const bf16_t* p) { return static_cast<float>(*p); }
__device__ __forceinline__ void st_f(float* p, float v) { *p = v; }
__device__ __forceinline__ void st_f(bf16_t* p, float v) { *p = static_cast<bf16_t>(v); }

// Vector of N elements of T <-> float[N]
template <typename T, int N> struct vec_io;
template <> struct vec_io<float, 4> {
  static __device__ __forceinline__ void load(const float* p, float* o) {
    float4 v = *reinterpret_cast<const float4*>(p); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
  static __device__ __forceinline__ void store(float* p, const float* o) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]); }
};
template <> struct vec_io<bf16_t, 4> {
  static __device__ __forceinline__ void load(const bf16_t* p, float* o) {
    uint2 v = *reinterpret_cast<const uint2*>(p);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u); }
  static __device__ __forceinline__ void store(bf16_t* p, const float* o) {
    bf16_t t[4] = {static_cast<bf16_t>(o[0]), static_cast<bf16_t>(o[1]), static_cast<bf16_t>(o[2]), static_cast<bf16_t>(o[3])};
    *reinterpret_cast<uint2*>(p) = *reinterpret_cast<uint2*>(t); }
};
template <> struct vec_io<bf16_t, 8> {
  static __device__ __forceinline__ void load(const bf16_t* p, float* o) {
#ifdef COMA_NT_LOADS
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    const u32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    const uint4 v = make_uint4(t[0], t[1], t[2], t[3]);
#else
    const uint4 v = *reinterpret_cast<const uint4*>(p);
#endif
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
template <typename T> struct vec_io<T, 1> {
  static __device__ __forceinline__ void load(const T* p, float* o) { o[0] = ld_f(p); }
  static __device__ __forceinline__ void store(T* p, const float* o) { st_f(p, o[0]); }
};
template <typename T> struct vec_io<T, 2> {
  static __device__ __forceinline__ void load(const T* p, float* o) { o[0] = ld_f(p); o[1] = ld_f(p + 1); }
  static __device__ __forceinline__ void store(T* p, const float* o) { st_f(p, o[0]); st_f(p + 1, o[1]); }
};

static inline int64_t t_vox(const coma_tensor* t) { return (int64_t)t->D * t->H * t->W; }
static inline bool t_same_grid(const coma_tensor* a, const coma_tensor* b) {
  return a->B == b->B && a->D == b->D && a->H == b->H && a->W == b->W;
}
static inline int esize(int dtype) { return dtype == COMA_BF16 ? 2 : 4; }

// vector width (elements) usable for channel-contiguous access of tensor t
static inline int t_vec(const coma_tensor* t, int want) {
  int es = esize(t->dtype);
  for (int v = want; v > 1; v >>= 1) {
    if (t->C % v == 0 && t->ld % v == 0 && t->sb % v == 0 && ((uintptr_t)t->data % (v * es)) == 0) return v;
  }
  return 1;
}

__device__ __forceinline__ float act_fwd(int act, float z, float a) {
  switch (act) {
    case COMA_ACT_RELU: return z > 0.f ? z : 0.f;
    case COMA_ACT_PRELU: return z > 0.f ? z : a * z;
    case COMA_ACT_LEAKY: return z > 0.f ? z : 0.01f * z;
    case COMA_ACT_SIGMOID: return 1.f / (1.f + expf(-z));
    case COMA_ACT_PRELU_RELU: { float p = z > 0.f ? z : a * z; return p > 0.f ? p : 0.f; }
    default: return z;
  }
}
// d act / d z  and  d act / d slope
__device__ __forceinline__ float act_bwd(int act, float z, float a, float* dslope) {
  *dslope = 0.f;
  switch (act) {
    case COMA_ACT_RELU: return z > 0.f ? 1.f : 0.f;
    case COMA_ACT_PRELU: if (z > 0.f) return 1.f; *dslope = z; return a;
    case COMA_ACT_LEAKY: return z > 0.f ? 1.f : 0.01f;
    case COMA_ACT_SIGMOID: { float s = 1.f / (1.f + expf(-z)); return s * (1.f - s); }
    case COMA_ACT_PRELU_RELU:
      if (z > 0.f) return 1.f;
      if (a * z > 0.f) { *dslope = z; return a; }
      return 0.f;
    default: return 1.f;
  }
}

// ROI label -> slot lookup shared by the loss / painting / evaluation kernels.  Labels are floats holding integer
// atlas ids (17..2035 for the 36 ROIs): a 4096-entry table in LDS replaces a 36-way linear search per voxel.
#define COMA_ROI_LUT 4096
__device__ __forceinline__ void roi_lut_build(signed char* lut, const int32_t* ids, int n) {   // whole block calls this
  for (int i = threadIdx.x; i < COMA_ROI_LUT; i += blockDim.x) lut[i] = -1;
  __syncthreads();
  if (threadIdx.x == 0)
    for (int i = n - 1; i >= 0; --i)             // reverse: the first occurrence of a duplicated id wins, as a search would
      if ((unsigned)ids[i] < COMA_ROI_LUT) lut[ids[i]] = (signed char)i;
  __syncthreads();
}
__device__ __forceinline__ int roi_lut_slot(const signed char* lut, const int32_t* ids, int n, float label) {
  const int li = (int)label;
  if ((float)li != label) return -1;
  if ((unsigned)li < COMA_ROI_LUT) return lut[li];
  for (int i = 0; i < n; ++i) if (ids[i] == li) return i;
  return -1;
}
