// Global-norm gradient clipping + AdamW for the trainable (projector) parameters.
//
// Reference semantics: HF Trainer's clip_grad_norm_(max_grad_norm = 1.0 default, train.py:256-306 does
// not override it) followed by torch.optim.AdamW (optimizer "adamw_torch", config_base.py:149-154:
// betas (0.9, 0.999), eps 1e-8, weight_decay 0.0).  torch's op sequence is reproduced literally:
//   clip_coef = min(1, max_norm / (total_norm + 1e-6));  g *= clip_coef
//   p *= 1 - lr*wd;  m = lerp(m, g, 1-b1);  v = b2*v + (1-b2) g*g
//   denom = sqrt(v)/sqrt(1-b2^t) + eps;  p -= (lr/(1-b1^t)) * m/denom
// `TS` is the storage type of the parameter and of both moments.  With TS = bf16 every op result is
// rounded to bf16 (what the reference does on GPU, where the projector and hence its optimizer state
// are bf16, ultravox_model.py:433-436); with an f32 `master` copy the update runs in f32 and the bf16
// parameter is a rounded mirror (documented deviation: better numerics, same API).
#include "common.h"
#include "kernels.h"

namespace {

__global__ void sumsq_partial_k(const float* __restrict__ g, long long n, float* __restrict__ partial) {
  __shared__ float red[16];
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) s += g[i] * g[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ void sumsq_final_k(const float* __restrict__ partial, int n, float* __restrict__ out) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += partial[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[0] = s;
}

template <typename TS>
__global__ void adamw_k(TS* __restrict__ param, float* __restrict__ master, const float* __restrict__ grad,
                        TS* __restrict__ m, TS* __restrict__ v, long long n, const float* __restrict__ sumsq,
                        float max_norm, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float coef = 1.0f;
  if (max_norm > 0.f) {
    const float total = sqrtf(sumsq[0]);
    coef = fminf(1.0f, max_norm / (total + 1e-6f));
  }
  if (master) {  // f32 master weights: `m` and `v` then point to f32 moment arrays whatever TS is
    float* mf = reinterpret_cast<float*>(m);
    float* vf = reinterpret_cast<float*>(v);
    const float g = grad[i] * coef;
    float p = master[i] * (1.0f - lr * wd);
    const float mm = mf[i] + (g - mf[i]) * (1.0f - b1);
    const float vv = vf[i] * b2 + (1.0f - b2) * g * g;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    p -= (lr / bc1) * (mm / denom);
    mf[i] = mm; vf[i] = vv; master[i] = p;
    stf<TS>(param + i, p);
    return;
  }
  const float g = rnd<TS>(rnd<TS>(grad[i]) * coef);
  float p = ldf<TS>(param + i);
  if (wd != 0.f) p = rnd<TS>(p * (1.0f - lr * wd));
  const float m0 = ldf<TS>(m + i), v0 = ldf<TS>(v + i);
  const float mm = rnd<TS>(m0 + (g - m0) * (1.0f - b1));                 // lerp_
  const float vv = rnd<TS>(rnd<TS>(v0 * b2) + (1.0f - b2) * g * g);       // mul_ then addcmul_
  const float denom = rnd<TS>(rnd<TS>(rnd<TS>(sqrtf(vv)) / bc2_sqrt) + eps);
  p = rnd<TS>(p - (lr / bc1) * (mm / denom));                            // addcdiv_
  stf<TS>(m + i, mm);
  stf<TS>(v + i, vv);
  stf<TS>(param + i, p);
}

}  // namespace

namespace uvx {

// `partial` scratch: >= 1024 floats.
int grad_sq_norm(hipStream_t st, const float* g, long long n, float* partial, float* out_sumsq) {
  const int blocks = (int)(n / 4096 > 1024 ? 1024 : (n / 4096 < 1 ? 1 : n / 4096));
  hipLaunchKernelGGL(sumsq_partial_k, dim3(blocks), dim3(256), 0, st, g, n, partial);
  hipLaunchKernelGGL(sumsq_final_k, dim3(1), dim3(256), 0, st, partial, blocks, out_sumsq);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

int adamw_clip_step(hipStream_t st, int state_dtype, void* param, float* master, const float* grad, void* m, void* v,
                    long long n, const float* sumsq, float max_norm, float lr, float beta1, float beta2, float eps,
                    float wd, int step) {
  UVX_CHECK(step >= 1, UVX_ERR_INVALID, "adamw: step must start at 1");
  if (n == 0) return UVX_OK;
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
  const int grid = (int)((n + 255) / 256);
  if (state_dtype == DT_BF16)
    hipLaunchKernelGGL(adamw_k<bf16_t>, dim3(grid), dim3(256), 0, st, (bf16_t*)param, master, grad, (bf16_t*)m, (bf16_t*)v, n, sumsq, max_norm, lr, beta1, beta2, eps, wd, bc1, bc2s);
  else
    hipLaunchKernelGGL(adamw_k<float>, dim3(grid), dim3(256), 0, st, (float*)param, master, grad, (float*)m, (float*)v, n, sumsq, max_norm, lr, beta1, beta2, eps, wd, bc1, bc2s);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

}  // namespace uvx
