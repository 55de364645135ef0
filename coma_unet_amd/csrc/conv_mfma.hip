// placeholder until the MFMA implicit-GEMM kernels land
#include "common.h"
bool conv_mfma_supported(const coma_conv_desc*, const coma_tensor*, const coma_tensor*) { return false; }
int conv_mfma_fwd(const coma_conv_desc*, const coma_tensor*, const void*, const float*, const coma_tensor*, hipStream_t) {
  coma_set_error("MFMA conv not built"); return 3; }
bool conv_mfma_wgrad_supported(const coma_conv_desc*, const coma_tensor*, const coma_tensor*) { return false; }
size_t conv_mfma_wgrad_ws_bytes(const coma_conv_desc*, const coma_tensor*, const coma_tensor*) { return 0; }
int conv_mfma_wgrad(const coma_conv_desc*, const coma_tensor*, const coma_tensor*, float*, void*, size_t, hipStream_t) {
  coma_set_error("MFMA wgrad not built"); return 3; }
