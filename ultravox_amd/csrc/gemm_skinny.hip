// bf16 GEMM for M <= 16 rows: the decode step of generate() (M = batch) and other few-row problems.
//
// C[M, N] = epilogue(A[M, K] . B[N, K]^T).  With so few rows the product is weight streaming: every element of B is read
// once and multiplied by M values, so the kernel is HBM-bound and the tiled kernels (128-row tiles, a handful of active
// CUs at N = 4096) are the wrong tool: 25.9 ms per decoded token at C2 against a 2 ms weight-streaming floor.
// Here a block of 8 waves owns 32 output columns (two 16 x 16 MFMA tiles: for the fused SwiGLU epilogue the gate block and
// its up block); the waves split K (k32 chunk i goes to wave i mod 8, so the 8 waves together read 512 contiguous bytes of
// each weight row per step), B fragments are loaded straight from global memory in MFMA operand layout (16 rows x 64 B per
// instruction), A (M x K, at most 448 KB, cache resident) likewise with rows >= M zeroed.  Partial accumulators are summed
// through LDS in a fixed order and wave 0 runs the shared epilogue arithmetic (bias, GELU, residual, SwiGLU; the same
// bf16 rounding points as store_tile in gemm.hip).
#include "common.h"
#include "kernels.h"

namespace {

struct SkinnyArgs {
  const bf16_t* A; const bf16_t* B; bf16_t* C; const bf16_t* bias; const bf16_t* residual; bf16_t* C2;
  int M, N, K, lda, ldb, ldc, ldr, ldc2, res_mod, act, swiglu;
  float alpha;
};

// TILES = 16-column MFMA tiles per block: 2 (needed by the fused SwiGLU: gate block + up block) or 1 (twice the blocks:
// N = 4096 gives 256 blocks instead of 128, one per CU)
template <int TILES>
__global__ __launch_bounds__(512) void gemm_skinny_bf16_k(SkinnyArgs p) {
  __shared__ float red[8][2][64][4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int frow = lane & 15, fg = lane >> 4;
  const int n0 = blockIdx.x * (16 * TILES);
  const bf16_t* b0 = p.B + (long long)min(n0 + frow, p.N - 1) * p.ldb + fg * 8;
  const bf16_t* b1 = p.B + (long long)min(n0 + 16 + frow, p.N - 1) * p.ldb + fg * 8;
  const bool m_ok = frow < p.M;
  const bf16_t* a0 = p.A + (long long)(m_ok ? frow : 0) * p.lda + fg * 8;
  f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const bf16x8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
  const int nchunk = p.K / 32;
  int i = w;
  // UN chunks per iteration: 2-3 x UN independent 16-byte loads in flight per lane (the kernel lives on memory-level parallelism)
  constexpr int UN = TILES == 2 ? 4 : 8;
  for (; i + 8 * (UN - 1) < nchunk; i += 8 * UN) {
    bf16x8_t x[UN], u[UN], v[UN];
#pragma unroll
    for (int t = 0; t < UN; ++t) {
      const int kk = (i + 8 * t) * 32;
      u[t] = *reinterpret_cast<const bf16x8_t*>(b0 + kk);
      if (TILES == 2) v[t] = *reinterpret_cast<const bf16x8_t*>(b1 + kk);
      x[t] = *reinterpret_cast<const bf16x8_t*>(a0 + kk);
    }
#pragma unroll
    for (int t = 0; t < UN; ++t) {
      const bf16x8_t xx = m_ok ? x[t] : zero;
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u[t], xx, acc0, 0, 0, 0);
      if (TILES == 2) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v[t], xx, acc1, 0, 0, 0);
    }
  }
  for (; i < nchunk; i += 8) {
    const int kk = i * 32;
    const bf16x8_t u = *reinterpret_cast<const bf16x8_t*>(b0 + kk);
    bf16x8_t v = zero;
    if (TILES == 2) v = *reinterpret_cast<const bf16x8_t*>(b1 + kk);
    const bf16x8_t x = *reinterpret_cast<const bf16x8_t*>(a0 + kk);
    const bf16x8_t xx = m_ok ? x : zero;
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u, xx, acc0, 0, 0, 0);
    if (TILES == 2) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v, xx, acc1, 0, 0, 0);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[w][0][lane][e] = acc0[e]; red[w][1][lane][e] = acc1[e]; }
  __syncthreads();
  if (w != 0) return;
  float s[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = 0.f;
#pragma unroll
      for (int ww = 0; ww < 8; ++ww) t += red[ww][j][lane][e];     // fixed order: deterministic
      s[j][e] = t * p.alpha;
    }
  const int m = frow;
  if (m >= p.M) return;
  if (p.swiglu) {    // tile 0 = 16 gate columns, tile 1 = the matching up columns (interleaved packing, weights.py)
    const int n = n0 + fg * 4;
    if (n >= p.N) return;
    u16x4_t og, ou, oa;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      og[e] = f2bf(s[0][e]);
      ou[e] = f2bf(s[1][e]);
      const float g = bf2f(og[e]);
      oa[e] = f2bf(bf2f(f2bf(g / (1.0f + __expf(-g)))) * bf2f(ou[e]));
    }
    bf16_t* crow = p.C + (long long)m * p.ldc + n;
    *reinterpret_cast<u16x4_t*>(crow) = og;
    *reinterpret_cast<u16x4_t*>(crow + 16) = ou;
    *reinterpret_cast<u16x4_t*>(p.C2 + (long long)m * p.ldc2 + n0 / 2 + fg * 4) = oa;
    return;
  }
#pragma unroll
  for (int j = 0; j < TILES; ++j) {
    const int n = n0 + 16 * j + fg * 4;
    if (n >= p.N) continue;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = s[j][e] + (p.bias ? bf2f(p.bias[n + e]) : 0.f);
      t = bf2f(f2bf(t));
      if (p.act == 1) t = bf2f(f2bf(gelu_fast(t)));
      v[e] = t;
    }
    if (p.residual) {
      const int rm = p.res_mod > 0 ? (m % p.res_mod) : m;
      const u16x4_t r4 = *reinterpret_cast<const u16x4_t*>(p.residual + (long long)rm * p.ldr + n);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += bf2f(r4[e]);
    }
    u16x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
    *reinterpret_cast<u16x4_t*>(p.C + (long long)m * p.ldc + n) = o;
  }
}

}  // namespace

namespace uvx {

bool gemm_skinny_applicable(const GemmDesc& d) {
  return d.M > 0 && d.M <= 16 && d.batch <= 1 && !d.out_f32 && !d.accumulate && !d.m_dev && d.swiglu != 2 &&
         d.K % 32 == 0 && d.lda % 8 == 0 && d.ldb % 8 == 0 && d.N % 4 == 0 && d.ldc % 4 == 0 &&
         (!d.swiglu || (d.N % 32 == 0 && d.ldc2 % 4 == 0)) && (!d.residual || d.ldr % 4 == 0) && uvx::g_options[4];
}

int gemm_skinny_bf16(hipStream_t st, const GemmDesc& d) {
  SkinnyArgs a;
  a.A = (const bf16_t*)d.A; a.B = (const bf16_t*)d.B; a.C = (bf16_t*)d.C; a.bias = (const bf16_t*)d.bias;
  a.residual = (const bf16_t*)d.residual; a.C2 = (bf16_t*)d.C2;
  a.M = d.M; a.N = d.N; a.K = d.K; a.lda = d.lda; a.ldb = d.ldb; a.ldc = d.ldc; a.ldr = d.ldr; a.ldc2 = d.ldc2;
  a.res_mod = d.res_mod; a.act = d.act; a.swiglu = d.swiglu; a.alpha = d.alpha;
  uvx::ProfScope prof(st, uvx::PROF_GEMM, 2.0 * d.M * d.N * (double)d.K,
                      ((double)d.M * d.K + (double)d.N * d.K) * 2.0 + (double)d.M * d.N * 2.0);
  if (uvx::g_prof_on) uvx::prof_tag(d.M, d.N, d.K, 1, 200);
  if (d.swiglu || d.N >= 16384) hipLaunchKernelGGL(gemm_skinny_bf16_k<2>, dim3((d.N + 31) / 32), dim3(512), 0, st, a);
  else hipLaunchKernelGGL(gemm_skinny_bf16_k<1>, dim3((d.N + 15) / 16), dim3(512), 0, st, a);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

}  // namespace uvx
