// C-ABI entry points that dispatch between kernel families, plus error reporting.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void coma_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static thread_local char g_ktag[160] = "";

void coma_set_kernel_tag(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_ktag, sizeof(g_ktag), fmt, ap);
  va_end(ap);
}

extern "C" const char* coma_last_kernel(void) { return g_ktag; }
extern "C" int coma_abi_version(void) { return COMA_ABI_VERSION; }
extern "C" const char* coma_last_error(void) { return g_err; }

// conv_direct.hip
int conv_check(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* y);
int conv_direct_fwd(const coma_conv_desc* d, const coma_tensor* x, const float* wk, const float* bias,
                    const coma_tensor* y, hipStream_t s);
int conv_direct_wgrad(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* dy, float* dwk, hipStream_t s, int zeroed);
// conv_point1.hip
bool conv_point1_ok(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* y);
int conv_point1_fwd(const coma_conv_desc* d, const coma_tensor* x, const float* wk, const float* bias, const coma_tensor* y,
                    hipStream_t s);
int conv_point1_wgrad(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* dy, float* dwk, hipStream_t s, int zeroed);
// norm.hip
int colsum(const coma_tensor* x, int per_sample, float* out, void* ws, size_t ws_bytes, hipStream_t s);
// conv_mfma.hip
bool conv_mfma_supported(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* y);
bool conv_f32mfma_supported(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* y);
int conv_mfma_fwd(const coma_conv_desc* d, const coma_tensor* x, const void* wk, const float* bias,
                  const coma_tensor* y, hipStream_t s, double2* stats = nullptr, int stats_inst = 0,
                  int* stats_chunks = nullptr, void* ws = nullptr, size_t ws_bytes = 0, int ws_zeroed = 0, int accum = 0);
bool conv_mfma_accumulate_ok(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* y);
size_t conv_mfma_fwd_ws_bytes(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* y);
bool conv_mfma_wgrad_supported(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* dy);
size_t conv_mfma_wgrad_ws_bytes(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* dy);
int conv_mfma_wgrad(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* dy, float* dwk, void* ws,
                    size_t ws_bytes, hipStream_t s, int zeroed);

extern "C" int coma_conv_pick_algo(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* y) {
  if (d->algo == 1) return 1;
  if (conv_point1_ok(d, x, y)) return 1;            // one channel on a side: streaming dot / scale kernels (fp32 weights)
  if (conv_mfma_supported(d, x, y)) return 2;       // bf16 tensors: MFMA wherever the shape allows
  if (conv_f32mfma_supported(d, x, y)) return 3;    // fp32 tensors: fp32 MFMA wherever the shape allows
  return 1;
}

extern "C" int coma_conv_accumulate_ok(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* y) {
  const int algo = coma_conv_pick_algo(d, x, y);
  return algo >= 2 && conv_mfma_accumulate_ok(d, x, y) ? 1 : 0;
}

extern "C" size_t coma_conv_fwd_ws_bytes(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* y) {
  return coma_conv_pick_algo(d, x, y) >= 2 ? conv_mfma_fwd_ws_bytes(d, x, y) : 0;
}

extern "C" int coma_conv_fwd(const coma_conv_desc* d, const coma_tensor* x, const void* wk, int32_t wk_dtype,
                             const float* bias, const coma_tensor* y, void* stream) {
  return coma_conv_fwd_ws(d, x, wk, wk_dtype, bias, y, nullptr, 0, 0, stream);
}

extern "C" int coma_conv_fwd_ws(const coma_conv_desc* d, const coma_tensor* x, const void* wk, int32_t wk_dtype,
                                const float* bias, const coma_tensor* y, void* ws, size_t ws_bytes, int32_t zeroed,
                                void* stream) {
  if (int rc = conv_check(d, x, y)) return rc;
  COMA_CHECK(wk, "conv_fwd: null weights");
  const int wz = (zeroed & COMA_ZEROED_WS) ? 1 : 0, accum = (zeroed & COMA_ACCUMULATE) ? 1 : 0;
  COMA_CHECK(!accum || coma_conv_accumulate_ok(d, x, y), "conv_fwd: COMA_ACCUMULATE is not supported for this problem (ask coma_conv_accumulate_ok)");
  hipStream_t s = (hipStream_t)stream;
  const int algo = coma_conv_pick_algo(d, x, y);
  if (algo == 2) {
    COMA_CHECK(wk_dtype == COMA_BF16, "conv_fwd: MFMA path needs bf16 kernel-layout weights");
    COMA_CHECK(conv_mfma_supported(d, x, y), "conv_fwd: shape not supported by the MFMA path (C=%d N=%d dtype=%d)",
               x->C, y->C, x->dtype);
    return conv_mfma_fwd(d, x, wk, bias, y, s, nullptr, 0, nullptr, ws, ws_bytes, wz, accum);
  }
  COMA_CHECK(wk_dtype == COMA_F32, "conv_fwd: fp32 tensors need fp32 kernel-layout weights");
  if (algo == 3) return conv_mfma_fwd(d, x, wk, bias, y, s, nullptr, 0, nullptr, ws, ws_bytes, wz, accum);
  if (conv_point1_ok(d, x, y)) return conv_point1_fwd(d, x, (const float*)wk, bias, y, s);
  return conv_direct_fwd(d, x, (const float*)wk, bias, y, s);
}

extern "C" int coma_conv_wgrad_algo(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* dy) {
  if (d->algo == 1 || conv_point1_ok(d, x, dy)) return 1;
  if (!conv_mfma_wgrad_supported(d, x, dy)) return 1;
  return x->dtype == COMA_BF16 ? 2 : 3;
}

// conv forward + the statistics of the following BatchNorm(train)/InstanceNorm in one pass where the kernel family
// supports it (the epilogue ADDS its {sum, sumsq} to the caller's zeroed fp64 record `sums[G][C][2]`), otherwise conv
// followed by the stand-alone statistics pass (which adds to the same record).
extern "C" int coma_conv_fwd_norm_stats(const coma_conv_desc* d, const coma_tensor* x, const void* wk, int32_t wk_dtype,
                                        const float* bias, const coma_tensor* y, int32_t mode, double* sums, void* ws,
                                        size_t ws_bytes, int32_t zeroed, void* stream) {
  if (int rc = conv_check(d, x, y)) return rc;
  COMA_CHECK(wk && sums, "conv_fwd_norm_stats: bad argument");
  hipStream_t s = (hipStream_t)stream;
  const int algo_ = coma_conv_pick_algo(d, x, y);
  if ((algo_ == 2 && wk_dtype == COMA_BF16) || (algo_ == 3 && wk_dtype == COMA_F32)) {
    int fused = 0;
    const int inst = mode == COMA_NORM_INSTANCE ? y->B : 0;      // (the kernels' group count; 0 = one BatchNorm group)
    if (int rc = conv_mfma_fwd(d, x, wk, bias, y, s, (double2*)sums, inst, &fused, ws, ws_bytes, (zeroed & COMA_ZEROED_WS) ? 1 : 0)) return rc;
    if (fused) return 0;
  } else {
    if (int rc = coma_conv_fwd(d, x, wk, wk_dtype, bias, y, stream)) return rc;
  }
  return coma_norm_stats(y, mode, sums, stream);
}

extern "C" size_t coma_conv_wgrad_ws_bytes(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* dy) {
  size_t a = coma_norm_ws_bytes(dy);
  size_t b = conv_mfma_wgrad_supported(d, x, dy) ? conv_mfma_wgrad_ws_bytes(d, x, dy) : 0;
  return a > b ? a : b;
}

// the part of coma_conv_wgrad_ws_bytes() the weight-gradient kernels merge into with atomics (their replica scratch): what
// a caller with a pre-zeroed arena hands over as `ws` together with COMA_ZEROED_WS (0: nothing of ws needs to be zero)
extern "C" size_t coma_conv_wgrad_zs_bytes(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* dy) {
  if (conv_point1_ok(d, x, dy) || coma_conv_wgrad_algo(d, x, dy) < 2) return 0;
  return conv_mfma_wgrad_ws_bytes(d, x, dy);
}

extern "C" int coma_conv_wgrad(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* dy, float* dwk,
                               float* dbias, void* ws, size_t ws_bytes, int32_t zeroed, void* stream) {
  if (int rc = conv_check(d, x, dy)) return rc;
  COMA_CHECK(dwk, "conv_wgrad: null dwk");
  hipStream_t s = (hipStream_t)stream;
  if (dbias) {
    // (the column sums write their partial rows into ws: a pre-zeroed ws is no longer zero afterwards)
    COMA_CHECK(ws && ws_bytes >= coma_norm_ws_bytes(dy), "conv_wgrad: workspace too small for the bias gradient");
    if (int rc = colsum(dy, d->per_sample_w, dbias, ws, ws_bytes, s)) return rc;
    zeroed &= ~COMA_ZEROED_WS;
  }
  if (conv_point1_ok(d, x, dy)) return conv_point1_wgrad(d, x, dy, dwk, s, zeroed);
  const int algo = coma_conv_wgrad_algo(d, x, dy);
  if (algo >= 2) return conv_mfma_wgrad(d, x, dy, dwk, ws, ws_bytes, s, zeroed);
  return conv_direct_wgrad(d, x, dy, dwk, s, zeroed);
}
