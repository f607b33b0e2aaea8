#include "common.h"
#include "kernels.h"
int uvx::gemm_nt_f32(hipStream_t, const GemmDesc&) {
  uvx_set_error("f32 gemm not built yet");
  return UVX_ERR_UNSUPPORTED;
}
