// C-ABI: error reporting + single-op entry points (used by unit tests and by host code that wants
// one kernel at a time).  The coarse-grained model entry points live in model.hip.
#include <stdarg.h>
#include <string.h>
#include "common.h"
#include "kernels.h"
#include "../../include/uvx.h"

static thread_local char g_err[512] = "";

extern "C" void uvx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* uvx_last_error(void) { return g_err; }
extern "C" int32_t uvx_abi_version(void) { return UVX_ABI_VERSION; }

extern "C" int32_t uvx_gemm(void* stream, int32_t dtype, const uvx_gemm_desc_t* g) {
  UVX_CHECK(g != nullptr, UVX_ERR_INVALID, "uvx_gemm: null descriptor");
  uvx::GemmDesc d;
  d.A = g->A; d.B = g->B; d.C = g->C; d.bias = g->bias; d.residual = g->residual;
  d.M = g->M; d.N = g->N; d.K = g->K;
  d.lda = g->lda; d.ldb = g->ldb; d.ldc = g->ldc; d.ldr = g->ldr;
  d.res_mod = g->res_mod; d.batch = g->batch;
  d.sA = g->stride_a; d.sB = g->stride_b; d.sC = g->stride_c; d.sR = g->stride_r;
  d.act = g->act; d.out_f32 = g->out_f32; d.accumulate = g->accumulate; d.alpha = g->alpha;
  return uvx::gemm((hipStream_t)stream, dtype, d);
}
