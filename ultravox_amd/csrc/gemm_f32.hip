// f32 "parity mode" GEMM: same NT contract and epilogues as gemm.hip, every tensor f32, on the exact-f32
// matrix cores (v_mfma_f32_16x16x4_f32: bitwise a k-ordered fmaf chain, 157 TFLOP/s peak = the f32 vector
// rate).  Used when the model is built with UVX_F32 so that logits can be compared with the reference's f32
// CPU path at 1e-3 (north_star); throughput is secondary (simple 64x64x16 LDS tiling, no pipelining).
#include "common.h"
#include "kernels.h"

namespace {

struct GemmF32Args {
  const float* A; const float* B; float* C; const float* bias; const float* residual;
  int M, N, K, lda, ldb, ldc, ldr, res_mod;
  long long sA, sB, sC, sR;
  int act, accumulate;
  float alpha;
};

constexpr int TM = 64, TN = 64, TK = 16, LDT = TK + 1;  // +1 pad: conflict-free column reads

__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(GemmF32Args p) {
  __shared__ float sX[TM * LDT];
  __shared__ float sW[TN * LDT];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wr = w >> 1, wc = w & 1;
  const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN;
  const long long z = blockIdx.z;
  const float* A = p.A + z * p.sA;
  const float* B = p.B + z * p.sB;
  const int lr = tid >> 2, lc = (tid & 3) * 4;  // 64 rows x 4 float4 per tile
  const int am = min(m0 + lr, p.M - 1), bn = min(n0 + lr, p.N - 1);
  const int fr = lane & 15, fk = lane >> 4;

  f32x4_t acc[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < p.K; k0 += TK) {
    const float4 xa = *reinterpret_cast<const float4*>(A + (long long)am * p.lda + k0 + lc);
    const float4 wb = *reinterpret_cast<const float4*>(B + (long long)bn * p.ldb + k0 + lc);
    __syncthreads();
    sX[lr * LDT + lc + 0] = xa.x; sX[lr * LDT + lc + 1] = xa.y; sX[lr * LDT + lc + 2] = xa.z; sX[lr * LDT + lc + 3] = xa.w;
    sW[lr * LDT + lc + 0] = wb.x; sW[lr * LDT + lc + 1] = wb.y; sW[lr * LDT + lc + 2] = wb.z; sW[lr * LDT + lc + 3] = wb.w;
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TK; kk += 4) {
      float xv[2], wv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) xv[i] = sX[(wr * 32 + i * 16 + fr) * LDT + kk + fk];
#pragma unroll
      for (int j = 0; j < 2; ++j) wv[j] = sW[(wc * 32 + j * 16 + fr) * LDT + kk + fk];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j], xv[i], acc[j][i], 0, 0, 0);
    }
  }
  // accumulator: rows = n (4 consecutive per lane), col = m
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wc * 32 + j * 16 + fk * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + wr * 32 + i * 16 + fr;
      if (m >= p.M) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (n + e >= p.N) continue;
        float v = acc[j][i][e] * p.alpha + (p.bias ? p.bias[n + e] : 0.f);
        if (p.act == 1) v = gelu_erf(v);
        if (p.residual) {
          const int rm = p.res_mod > 0 ? (m % p.res_mod) : m;
          v += p.residual[z * p.sR + (long long)rm * p.ldr + n + e];
        }
        float* dst = p.C + z * p.sC + (long long)m * p.ldc + n + e;
        if (p.accumulate) v += *dst;
        *dst = v;
      }
    }
  }
}

}  // namespace

int uvx::gemm_nt_f32(hipStream_t st, const GemmDesc& d) {
  UVX_CHECK(d.M > 0 && d.N > 0 && d.K > 0, UVX_ERR_SHAPE, "gemm_f32: empty problem %dx%dx%d", d.M, d.N, d.K);
  UVX_CHECK(d.K % TK == 0, UVX_ERR_SHAPE, "gemm_f32: K=%d must be a multiple of %d", d.K, TK);
  UVX_CHECK(d.lda % 4 == 0 && d.ldb % 4 == 0, UVX_ERR_SHAPE, "gemm_f32: lda/ldb must be multiples of 4");
  UVX_CHECK(d.act < 2 && !d.b_kn, UVX_ERR_UNSUPPORTED, "gemm_f32: the act 2 / 3 epilogues and the NN form exist on the bf16 path only");
  GemmF32Args a;
  a.A = (const float*)d.A; a.B = (const float*)d.B; a.C = (float*)d.C;
  a.bias = (const float*)d.bias; a.residual = (const float*)d.residual;
  a.M = d.M; a.N = d.N; a.K = d.K; a.lda = d.lda; a.ldb = d.ldb; a.ldc = d.ldc; a.ldr = d.ldr; a.res_mod = d.res_mod;
  a.sA = d.sA; a.sB = d.sB; a.sC = d.sC; a.sR = d.sR; a.act = d.act; a.accumulate = d.accumulate; a.alpha = d.alpha;
  dim3 grid(cdiv(d.M, TM), cdiv(d.N, TN), d.batch > 0 ? d.batch : 1);
  hipLaunchKernelGGL(gemm_nt_f32_kernel, grid, dim3(256), 0, st, a);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}
