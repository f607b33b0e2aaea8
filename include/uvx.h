/* libuvx — C ABI of the MI355X-native Ultravox audio->LLM hot path.
 *
 * The reference (fixie-ai/ultravox) has NO native/FFI layer: its hot path is reached through Python
 * classes (ultravox/model/ultravox_model.py:277-352 UltravoxModel.forward, :354-396
 * _prepare_audio_embeds, :768-800 UltravoxProjector.forward, :865-994 ModifiedWhisperEncoder.forward;
 * ultravox/model/ultravox_processing.py:217-370 UltravoxProcessor.__call__).  This header is the
 * boundary a replacement exports instead; each entry point names the reference code it replaces.
 *
 * Conventions
 *  - every data pointer is a DEVICE pointer to a contiguous buffer owned by the caller;
 *  - `stream` is a hipStream_t; all calls are asynchronous on it, no hidden synchronisation;
 *  - return value: 0 ok, <0 uvx error (below), >0 a hipError_t; message via uvx_last_error()
 *    (thread local);
 *  - dtype: UVX_BF16 (production; bf16 storage, f32 accumulate) or UVX_F32 (parity mode).
 */
#ifndef UVX_H_
#define UVX_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define UVX_ABI_VERSION 1
#define UVX_BF16 0
#define UVX_F32 1

#define UVX_OK 0
#define UVX_ERR_INVALID (-1)
#define UVX_ERR_SHAPE (-2)
#define UVX_ERR_WORKSPACE (-3)
#define UVX_ERR_UNSUPPORTED (-4)

const char* uvx_last_error(void);
int32_t uvx_abi_version(void);

/* ---- single-op entry points -------------------------------------------------------------- */

/* C[M,N] = act(alpha * A[M,K] . B[N,K]^T + bias[N]) + residual — torch.nn.Linear semantics
 * (every q/k/v/out/fc/linear_1/linear_2/gate/up/down/lm_head on the path). */
typedef struct {
  const void* A; const void* B; void* C; const void* bias; const void* residual;
  int32_t M, N, K, lda, ldb, ldc, ldr;
  int32_t res_mod, batch;
  int64_t stride_a, stride_b, stride_c, stride_r;
  int32_t act;        /* 0 none, 1 exact-erf GELU */
  int32_t out_f32;    /* bf16 inputs, f32 output (weight gradients) */
  int32_t accumulate; /* C += (f32 output only) */
  float alpha;
} uvx_gemm_desc_t;
int32_t uvx_gemm(void* stream, int32_t dtype, const uvx_gemm_desc_t* desc);

#ifdef __cplusplus
}
#endif
#endif /* UVX_H_ */
