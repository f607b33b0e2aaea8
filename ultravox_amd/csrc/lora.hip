// Rank-r LoRA products for the encoder's q_proj / k_proj (audio_model_lora_config, peft layouts: A [r, C], B [C, r]).
//
// With r = 8 every LoRA product is a "skinny" matrix product whose cost is reading the wide operand once: HBM-bound,
// 8 FMAs per loaded element.  Padding the rank to an MFMA tile (r -> 64) and running the GEMM family on it - the
// first version - spent 17 ms of a C2 LoRA-training step in 128x128 tiles that are 87 % zeros or 1 K-tile deep
// (profiles/r01_kernel_stats_lora.txt).  These kernels do the same arithmetic on the VALU, f32 accumulation, with the
// bf16 rounding points of the GEMM path (which are peft's: lora_A output, lora_B output, the sum).
#include "common.h"
#include "kernels.h"
#include "../../include/uvx.h"

namespace {

constexpr int RMAX = 64;

// Y[m, j] = round(alpha * sum_c X[m, c] * W[j, c]),  j < r.   One wave per row; W (r x C) stays in L1 / L2.
// W_CR: W is stored [C][r] (lora_B used as a down-projection in the backward pass) instead of [r][C]
template <typename T, bool W_CR>
__global__ __launch_bounds__(256) void lora_down_k(const T* __restrict__ X, long long ldx, const T* __restrict__ W,
                                                   T* __restrict__ Y, long long ldy, long long M, int C, int r, float alpha) {
  const long long m = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (m >= M) return;
  const T* xr = X + m * ldx;
  for (int j0 = 0; j0 < r; j0 += 8) {            // 8 output columns per pass (r = 8: one pass)
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int c = lane * 8; c < C; c += 64 * 8) {
      float xv[8];
      ld8<T>(xr + c, xv);
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        if (j0 + jj < r) {
          float wv[8];
          if (W_CR) {
#pragma unroll
            for (int i = 0; i < 8; ++i) wv[i] = ldf<T>(W + (long long)(c + i) * r + j0 + jj);
          } else {
            ld8<T>(W + (long long)(j0 + jj) * C + c, wv);
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[jj] += xv[i] * wv[i];
        }
      }
    }
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) acc[jj] = wave_sum(acc[jj]);
    if (lane == 0) {   // whole group of 8 (zeros beyond r): consumers vector-load Y rows
      float o[8];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) o[jj] = acc[jj] * alpha;
      st8<T>(Y + m * ldy + j0, o);
    }
  }
}

// The same for r <= 8, W [r][C] and C = 512 * NCH (the encoder's 1024 columns), bf16: a wave keeps ITS slice of W in registers (8 x NCH
// packed 16-byte vectors per lane) and walks RPW rows, two at a time - the kernel above re-reads W through L1 for every row (9 vector
// loads per 64 FMAs: 19 us for a 24.6 MB activation matrix whose HBM time is 5 us).  Exact bf16 products, f32 accumulation
// (v_dot2c_f32_bf16 over the lane's column pairs in ascending order, butterfly over the lanes), the same rounding of the result.
typedef unsigned int lu32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 lbf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float ldot2(unsigned a, unsigned b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(lbf16x2_t, a), __builtin_bit_cast(lbf16x2_t, b), c, false);
}
__device__ __forceinline__ float ldot8(const lu32x4_t a, const lu32x4_t b, float c) {     // (components through scalars: see gemm_skinny.hip)
  const unsigned a0 = a.x, a1 = a.y, a2 = a.z, a3 = a.w, b0 = b.x, b1 = b.y, b2 = b.z, b3 = b.w;
  return ldot2(a3, b3, ldot2(a2, b2, ldot2(a1, b1, ldot2(a0, b0, c))));
}
template <int NCH>
__device__ __forceinline__ void lora_down_reg_body(const bf16_t* __restrict__ X, long long ldx, const bf16_t* __restrict__ W,
                                                   bf16_t* __restrict__ Y, long long ldy, long long M, int C, int r, float alpha, int rpw) {
  const int lane = threadIdx.x & 63;
  const long long m0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * rpw;
  if (m0 >= M) return;
  const lu32x4_t zero = {0u, 0u, 0u, 0u};
  lu32x4_t wv[8][NCH];
#pragma unroll
  for (int jj = 0; jj < 8; ++jj)
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
      wv[jj][ch] = jj < r ? *reinterpret_cast<const lu32x4_t*>(W + (long long)jj * C + ch * 512 + lane * 8) : zero;
  const long long m1 = min(M, m0 + rpw);
  for (long long m = m0; m < m1; m += 2) {
    const bool two = m + 1 < m1;
    lu32x4_t xa[NCH], xb[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      xa[ch] = *reinterpret_cast<const lu32x4_t*>(X + m * ldx + ch * 512 + lane * 8);
      xb[ch] = *reinterpret_cast<const lu32x4_t*>(X + (two ? m + 1 : m) * ldx + ch * 512 + lane * 8);
    }
    float oa[8], ob[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      float sa = 0.f, sb = 0.f;
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) { sa = ldot8(xa[ch], wv[jj][ch], sa); sb = ldot8(xb[ch], wv[jj][ch], sb); }
      oa[jj] = wave_sum(sa) * alpha;
      ob[jj] = wave_sum(sb) * alpha;
    }
    if (lane == 0) {
      st8<bf16_t>(Y + m * ldy, oa);
      if (two) st8<bf16_t>(Y + (m + 1) * ldy, ob);
    }
  }
}
template <int NCH>
__global__ __launch_bounds__(256) void lora_down_reg_k(const bf16_t* __restrict__ X, long long ldx, const bf16_t* __restrict__ W,
                                                       bf16_t* __restrict__ Y, long long ldy, long long M, int C, int r, float alpha, int rpw) {
  lora_down_reg_body<NCH>(X, ldx, W, Y, ldy, M, C, r, alpha, rpw);
}
// two down-projections over the same M rows in ONE launch (round 6: q_proj and k_proj of a layer - the adapters' products were 12 launches per
// layer and pass): blockIdx.y picks the problem; the arithmetic per output is lora_down_reg_k's
struct LoraDown2 { const bf16_t* X[2]; const bf16_t* W[2]; bf16_t* Y[2]; float alpha[2]; };
template <int NCH>
__global__ __launch_bounds__(256) void lora_down_reg2_k(LoraDown2 q, long long ldx, long long ldy, long long M, int C, int r, int rpw) {
  const int z = blockIdx.y;
  lora_down_reg_body<NCH>(q.X[z], ldx, q.W[z], q.Y[z], ldy, M, C, r, q.alpha[z], rpw);
}

// Z[m, c] = round(Z[m, c] + round(alpha * sum_j Y[m, j] * W[c, j]))   (accumulate = false: Z = round(alpha * ...))
// W_RC: W is stored [r][C] (lora_A used as an up-projection in the backward pass) instead of [C][r]
template <typename T, bool ACC, bool W_RC>
__device__ __forceinline__ void lora_up_body(const T* __restrict__ Y, long long ldy, const T* __restrict__ W, T* __restrict__ Z, long long ldz,
                                             long long M, int C, int r, float alpha) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int cv = C / 8;
  if (i >= M * cv) return;
  const long long m = i / cv;
  const int c = (int)(i % cv) * 8;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (r <= 8 && (W_RC || r == 8)) {
    // ranks up to 8 (the reference's r = 8): Y rows are written in whole groups of 8 (zeros beyond r) and the narrow operand comes in
    // 16-byte vectors - r (W [r][C]) or 8 (W [C][8]) loads per thread instead of 8 r two-byte loads (round 4: 18 us -> the HBM time of Z)
    float y[8];
    ld8<T>(Y + m * ldy, y);
    if (W_RC) {
      for (int j = 0; j < r; ++j) {
        float wv[8];
        ld8<T>(W + (long long)j * C + c, wv);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += y[j] * wv[k];
      }
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float wv[8];
        ld8<T>(W + (long long)(c + k) * 8, wv);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[k] += y[j] * wv[j];
      }
    }
  } else {
    for (int j = 0; j < r; ++j) {
      const float y = ldf<T>(Y + m * ldy + j);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += y * ldf<T>(W_RC ? W + (long long)j * C + c + k : W + (long long)(c + k) * r + j);
    }
  }
  float z[8];
  if (ACC) ld8<T>(Z + m * ldz + c, z);
#pragma unroll
  for (int k = 0; k < 8; ++k) z[k] = ACC ? z[k] + rnd<T>(acc[k] * alpha) : acc[k] * alpha;
  st8<T>(Z + m * ldz + c, z);
}
template <typename T, bool ACC, bool W_RC>
__global__ void lora_up_k(const T* __restrict__ Y, long long ldy, const T* __restrict__ W, T* __restrict__ Z, long long ldz,
                          long long M, int C, int r, float alpha) {
  lora_up_body<T, ACC, W_RC>(Y, ldy, W, Z, ldz, M, C, r, alpha);
}
// two accumulating up-projections over the same M rows in ONE launch (blockIdx.y picks; C = the wider of the two, the narrower one's surplus
// blocks exit).  When both write the SAME rows of Z (the backward's d n += u_q . A_q + u_k . A_k) the launch would race: that case keeps two launches.
// Z = round(round(Z + round(a0 Y0 . W0)) + round(a1 Y1 . W1)): two accumulating up-projections into the SAME rows in one pass over Z (the encoder backward's
// d n += u_q . A_q + u_k . A_k: lora_up twice = two read-modify-write passes).  W [r][C], r <= 8; the intermediate rounding is kept: bit-identical.
template <typename T>
__global__ void lora_up_same2_k(const T* __restrict__ Y0, const T* __restrict__ Y1, long long ldy, const T* __restrict__ W0, const T* __restrict__ W1,
                                T* __restrict__ Z, long long ldz, long long M, int C, int r, float alpha0, float alpha1) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int cv = C / 8;
  if (i >= M * cv) return;
  const long long m = i / cv;
  const int c = (int)(i % cv) * 8;
  float z[8];
  ld8<T>(Z + m * ldz + c, z);
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const T* Y = q ? Y1 : Y0;
    const T* W = q ? W1 : W0;
    const float alpha = q ? alpha1 : alpha0;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, y[8];
    ld8<T>(Y + m * ldy, y);
    for (int j = 0; j < r; ++j) {
      float wv[8];
      ld8<T>(W + (long long)j * C + c, wv);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += y[j] * wv[k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) z[k] = rnd<T>(z[k] + rnd<T>(acc[k] * alpha));
  }
  st8<T>(Z + m * ldz + c, z);
}
template <typename T>
struct LoraUp2 { const T* Y[2]; const T* W[2]; T* Z[2]; int C[2]; float alpha[2]; };
template <typename T, bool W_RC>
__global__ void lora_up2_k(LoraUp2<T> q, long long ldy, long long ldz, long long M, int r) {
  const int z = blockIdx.y;
  lora_up_body<T, true, W_RC>(q.Y[z], ldy, q.W[z], q.Z[z], ldz, M, q.C[z], r, q.alpha[z]);
}

// Weight-gradient contraction over the tokens: P[j][c] = sum_m Y[m, j] * X[m, c]  (j < 8 per pass, c < C).
// Grid = (C / 64 column tiles) x (row chunks of RCH rows).  A block is 4 waves; in a wave, lane = (row lane 0..7) x
// (column vector 0..7): 8 lanes read one 128-byte piece of a row, the 8 row lanes take consecutive rows, the 4 waves
// interleave further.  Each lane keeps acc[8 j][8 c]; row lanes are folded with shuffles, waves through LDS.
// partial[chunk][j][c]; lora_wgrad_reduce_k sums the few chunks in a fixed order.
constexpr int RCH = 256;      // (round 4: 1024 gave 16 x 12 = 192 blocks at the encoder's 12000 x 1024 - fewer than CUs; 256 -> 752 blocks)
template <typename T>
__device__ __forceinline__ void lora_wgrad_body(const T* __restrict__ X, long long ldx, const T* __restrict__ Y, long long ldy,
                                                float* __restrict__ partial, long long M, int C, int r, int j0, float (*red)[8][64]) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int cv = lane & 7, rl = lane >> 3;
  const int c = blockIdx.x * 64 + cv * 8;
  const long long m0 = (long long)blockIdx.y * RCH, m1 = min(M, m0 + RCH);
  float acc[8][8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = 0.f;
  if (c < C) {
    for (long long m = m0 + w * 8 + rl; m < m1; m += 32) {
      float xv[8], yv[8];
      ld8<T>(X + m * ldx + c, xv);
      ld8<T>(Y + m * ldy + j0, yv);
#pragma unroll
      for (int jj = 0; jj < 8; ++jj)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[jj][k] += yv[jj] * xv[k];
    }
  }
#pragma unroll
  for (int jj = 0; jj < 8; ++jj)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float v = acc[jj][k];
      v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
      acc[jj][k] = v;
    }
  if (rl == 0) {
#pragma unroll
    for (int jj = 0; jj < 8; ++jj)
#pragma unroll
      for (int k = 0; k < 8; ++k) red[w][jj][cv * 8 + k] = acc[jj][k];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 8 * 64; i += 256) {
    const int jj = i >> 6, cc = i & 63;
    if (j0 + jj < r && blockIdx.x * 64 + cc < C)
      partial[((long long)blockIdx.y * r + j0 + jj) * C + blockIdx.x * 64 + cc] =
          red[0][jj][cc] + red[1][jj][cc] + red[2][jj][cc] + red[3][jj][cc];
  }
}
template <typename T>
__global__ __launch_bounds__(256) void lora_wgrad_k(const T* __restrict__ X, long long ldx, const T* __restrict__ Y, long long ldy,
                                                    float* __restrict__ partial, long long M, int C, int r, int j0) {
  __shared__ float red[4][8][64];
  lora_wgrad_body<T>(X, ldx, Y, ldy, partial, M, C, r, j0, red);
}
// the partial sums of up to four weight-gradient products in ONE launch (blockIdx.z picks the product; a narrower product's surplus column
// blocks exit): the same per-block arithmetic and the same scratch regions as four lora_wgrad_k launches
template <typename T>
struct LoraWgrad4 { const T* X[4]; const T* Y[4]; float* partial[4]; long long ldx[4], ldy[4]; int C[4]; };
template <typename T>
__global__ __launch_bounds__(256) void lora_wgrad4_k(LoraWgrad4<T> q, long long M, int r, int j0) {
  __shared__ float red[4][8][64];
  const int z = blockIdx.z;
  if ((int)blockIdx.x * 64 >= q.C[z]) return;
  lora_wgrad_body<T>(q.X[z], q.ldx[z], q.Y[z], q.ldy[z], q.partial[z], M, q.C[z], r, j0, red);
}

// out = alpha * sum_chunk partial[chunk]  as [r][C] (transpose_out = 0: lora_A) or [C][r] (1: lora_B)
__global__ void lora_wgrad_reduce_k(const float* __restrict__ partial, int nchunks, int r, int C, float* __restrict__ out,
                                    int transpose_out, float alpha) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= r * C) return;
  const int j = i / C, c = i % C;
  float s = 0.f;
  for (int k = 0; k < nchunks; ++k) s += partial[((long long)k * r + j) * C + c];   // fixed order: deterministic
  out[transpose_out ? (long long)c * r + j : (long long)j * C + c] = s * alpha;
}

// The reduces of several weight-gradient products in ONE launch (round 5): a layer's backward has four of them (d lora_A / d lora_B of q_proj and
// k_proj), each summing ~47 chunk partials for 8 K outputs - 15 us of latency per launch, 96 launches = 1.45 ms of the recipe's step
// (profiles/r05_kernel_stats_kl_lora8.txt).  blockIdx.y = which product; same fixed-order sum per output as lora_wgrad_reduce_k: bit-identical.
struct WgradReduceBatch {
  const float* partial[4]; float* out[4]; int C[4]; int transpose_out[4]; float alpha[4];
  int nchunks, r;
};
__global__ void lora_wgrad_reduce_batch_k(WgradReduceBatch b) {
  const int w = blockIdx.y, C = b.C[w];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b.r * C) return;
  const int j = i / C, c = i % C;
  const float* partial = b.partial[w];
  float s = 0.f;
  for (int k = 0; k < b.nchunks; ++k) s += partial[((long long)k * b.r + j) * C + c];   // fixed order: deterministic
  b.out[w][b.transpose_out[w] ? (long long)c * b.r + j : (long long)j * C + c] = s * b.alpha[w];
}
// two [C, r] -> [r, C] transposes (lora_B of q_proj and k_proj) in one launch: blockIdx.y = which
template <typename T>
__global__ void lora_transpose2_k(const T* __restrict__ in0, T* __restrict__ out0, int C0, const T* __restrict__ in1, T* __restrict__ out1, int C1, int r) {
  const T* in = blockIdx.y ? in1 : in0;
  T* out = blockIdx.y ? out1 : out0;
  const int C = blockIdx.y ? C1 : C0;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= r * C) return;
  const int j = i / C, c = i % C;
  out[i] = in[(long long)c * r + j];
}

// out[j][c] = in[c][j]: lora_B [C, r] -> [r, C] once per use, so that every product reads its narrow operand row-wise
template <typename T>
__global__ void lora_transpose_k(const T* __restrict__ in, T* __restrict__ out, int C, int r) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= r * C) return;
  const int j = i / C, c = i % C;
  out[i] = in[(long long)c * r + j];
}

}  // namespace

namespace uvx {

int lora_down(hipStream_t st, int dtype, const void* X, long long ldx, const void* W, int w_is_cr, void* Y, long long ldy,
              long long M, int C, int r, float alpha) {
  UVX_CHECK(C % 8 == 0 && ldx % 8 == 0 && r > 0 && r <= RMAX, UVX_ERR_SHAPE, "lora_down: C=%d r=%d unsupported", C, r);
  if (M == 0) return UVX_OK;
  if (dtype == DT_BF16 && !w_is_cr && r <= 8 && (C == 512 || C == 1024) && M >= 1024) {      // W in registers, 8 rows per wave
    const int rpw = 8;
    const dim3 g2((unsigned)((M + 4 * rpw - 1) / (4 * rpw)));
    if (C == 512) hipLaunchKernelGGL((lora_down_reg_k<1>), g2, dim3(256), 0, st, (const bf16_t*)X, ldx, (const bf16_t*)W, (bf16_t*)Y, ldy, M, C, r, alpha, rpw);
    else hipLaunchKernelGGL((lora_down_reg_k<2>), g2, dim3(256), 0, st, (const bf16_t*)X, ldx, (const bf16_t*)W, (bf16_t*)Y, ldy, M, C, r, alpha, rpw);
    UVX_LAUNCH_CHECK();
    return UVX_OK;
  }
  const dim3 grid((unsigned)((M + 3) / 4));
#define L(T, F) hipLaunchKernelGGL((lora_down_k<T, F>), grid, dim3(256), 0, st, (const T*)X, ldx, (const T*)W, (T*)Y, ldy, M, C, r, alpha)
  if (dtype == DT_BF16) { if (w_is_cr) L(bf16_t, true); else L(bf16_t, false); }
  else { if (w_is_cr) L(float, true); else L(float, false); }
#undef L
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

// q_proj and k_proj together (same M, ldx, ldy, C, r; bf16 register kernel) - else two lora_down calls
int lora_down2(hipStream_t st, int dtype, const void* X0, const void* X1, long long ldx, const void* W0, const void* W1, void* Y0, void* Y1,
               long long ldy, long long M, int C, int r, float alpha0, float alpha1) {
  if (M > 0 && dtype == DT_BF16 && r <= 8 && (C == 512 || C == 1024) && M >= 1024 && C % 8 == 0 && ldx % 8 == 0 && g_options[22] != 1) {
    const int rpw = 8;
    const dim3 g2((unsigned)((M + 4 * rpw - 1) / (4 * rpw)), 2);
    LoraDown2 q = {{(const bf16_t*)X0, (const bf16_t*)X1}, {(const bf16_t*)W0, (const bf16_t*)W1}, {(bf16_t*)Y0, (bf16_t*)Y1}, {alpha0, alpha1}};
    if (C == 512) hipLaunchKernelGGL((lora_down_reg2_k<1>), g2, dim3(256), 0, st, q, ldx, ldy, M, C, r, rpw);
    else hipLaunchKernelGGL((lora_down_reg2_k<2>), g2, dim3(256), 0, st, q, ldx, ldy, M, C, r, rpw);
    UVX_LAUNCH_CHECK();
    return UVX_OK;
  }
  const int rc = lora_down(st, dtype, X0, ldx, W0, 0, Y0, ldy, M, C, r, alpha0);
  return rc ? rc : lora_down(st, dtype, X1, ldx, W1, 0, Y1, ldy, M, C, r, alpha1);
}

// two ACCUMULATING up-projections into DIFFERENT outputs (q and k columns) with W stored [r][C]; else two lora_up calls
int lora_up2(hipStream_t st, int dtype, const void* Y0, const void* Y1, long long ldy, const void* W0, const void* W1, void* Z0, void* Z1,
             long long ldz, long long M, int C0, int C1, int r, float alpha0, float alpha1) {
  const size_t esz_ = dtype == DT_BF16 ? 2 : 4;
  const bool ok = M > 0 && r <= 8 && C0 % 8 == 0 && C1 % 8 == 0 && ldz % 8 == 0 && ldy % 8 == 0 && g_options[22] != 1 &&
                  ((uintptr_t)Y0 % (8 * esz_)) == 0 && ((uintptr_t)Y1 % (8 * esz_)) == 0 && ((uintptr_t)W0 % 16) == 0 && ((uintptr_t)W1 % 16) == 0 &&
                  ((uintptr_t)Z0 % 16) == 0 && ((uintptr_t)Z1 % 16) == 0;
  if (ok && Z0 == Z1 && C0 == C1) {      // the same rows twice: one pass, the two terms in order
    const long long n = M * (C0 / 8);
    const dim3 grid((unsigned)((n + 255) / 256));
    if (dtype == DT_BF16) hipLaunchKernelGGL(lora_up_same2_k<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)Y0, (const bf16_t*)Y1, ldy, (const bf16_t*)W0, (const bf16_t*)W1,
                                             (bf16_t*)Z0, ldz, M, C0, r, alpha0, alpha1);
    else hipLaunchKernelGGL(lora_up_same2_k<float>, grid, dim3(256), 0, st, (const float*)Y0, (const float*)Y1, ldy, (const float*)W0, (const float*)W1, (float*)Z0, ldz,
                            M, C0, r, alpha0, alpha1);
    UVX_LAUNCH_CHECK();
    return UVX_OK;
  }
  if (ok && Z0 != Z1) {
    const int cmax = C0 > C1 ? C0 : C1;
    const long long n = M * (cmax / 8);
    const dim3 grid((unsigned)((n + 255) / 256), 2);
    if (dtype == DT_BF16) {
      LoraUp2<bf16_t> q = {{(const bf16_t*)Y0, (const bf16_t*)Y1}, {(const bf16_t*)W0, (const bf16_t*)W1}, {(bf16_t*)Z0, (bf16_t*)Z1}, {C0, C1}, {alpha0, alpha1}};
      hipLaunchKernelGGL((lora_up2_k<bf16_t, true>), grid, dim3(256), 0, st, q, ldy, ldz, M, r);
    } else {
      LoraUp2<float> q = {{(const float*)Y0, (const float*)Y1}, {(const float*)W0, (const float*)W1}, {(float*)Z0, (float*)Z1}, {C0, C1}, {alpha0, alpha1}};
      hipLaunchKernelGGL((lora_up2_k<float, true>), grid, dim3(256), 0, st, q, ldy, ldz, M, r);
    }
    UVX_LAUNCH_CHECK();
    return UVX_OK;
  }
  const int rc = lora_up(st, dtype, Y0, ldy, W0, 1, Z0, ldz, M, C0, r, alpha0, 1);
  return rc ? rc : lora_up(st, dtype, Y1, ldy, W1, 1, Z1, ldz, M, C1, r, alpha1, 1);
}

int lora_up(hipStream_t st, int dtype, const void* Y, long long ldy, const void* W, int w_is_rc, void* Z, long long ldz,
            long long M, int C, int r, float alpha, int accumulate) {
  UVX_CHECK(C % 8 == 0 && ldz % 8 == 0 && r > 0 && r <= RMAX, UVX_ERR_SHAPE, "lora_up: C=%d r=%d unsupported", C, r);
  // the r <= 8 path reads Y rows as one 16-byte vector: rows padded to 8 columns (zeros beyond r - lora_down writes them so), ldy % 8 == 0 and
  // 16-byte-aligned Y / W / Z (enforced here, not assumed)
  const size_t esz_ = dtype == DT_BF16 ? 2 : 4;
  UVX_CHECK(r > 8 || (ldy % 8 == 0 && ((uintptr_t)Y % (8 * esz_)) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)Z % 16) == 0), UVX_ERR_SHAPE,
            "lora_up: rank %d <= 8 needs Y rows padded to 8 columns with ldy %% 8 == 0 (ldy = %lld) and 16-byte-aligned operands", r, ldy);
  if (M == 0) return UVX_OK;
  const long long n = M * (C / 8);
  const dim3 grid((unsigned)((n + 255) / 256));
#define L(T, A, F) hipLaunchKernelGGL((lora_up_k<T, A, F>), grid, dim3(256), 0, st, (const T*)Y, ldy, (const T*)W, (T*)Z, ldz, M, C, r, alpha)
#define L2(T, A) do { if (w_is_rc) L(T, A, true); else L(T, A, false); } while (0)
  if (dtype == DT_BF16) { if (accumulate) L2(bf16_t, true); else L2(bf16_t, false); }
  else { if (accumulate) L2(float, true); else L2(float, false); }
#undef L2
#undef L
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

// scratch: lora_wgrad_scratch_floats(M, C, r) floats.  Y rows must be readable in whole groups of 8 (lora_down writes them so).
long long lora_wgrad_scratch_floats(long long M, int C, int r) { return ((M + RCH - 1) / RCH) * (long long)r * C; }

int lora_wgrad(hipStream_t st, int dtype, const void* X, long long ldx, const void* Y, long long ldy, float* out, long long M,
               int C, int r, int transpose_out, float alpha, float* scratch) {
  UVX_CHECK(C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && r > 0 && r <= RMAX, UVX_ERR_SHAPE, "lora_wgrad: C=%d r=%d unsupported", C, r);
  if (M == 0) { UVX_HIP(hipMemsetAsync(out, 0, sizeof(float) * (size_t)r * C, st)); return UVX_OK; }
  const int nchunks = (int)((M + RCH - 1) / RCH);
  const dim3 grid((C + 63) / 64, nchunks);
  for (int j0 = 0; j0 < r; j0 += 8) {
    if (dtype == DT_BF16) hipLaunchKernelGGL(lora_wgrad_k<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)X, ldx, (const bf16_t*)Y, ldy, scratch, M, C, r, j0);
    else hipLaunchKernelGGL(lora_wgrad_k<float>, grid, dim3(256), 0, st, (const float*)X, ldx, (const float*)Y, ldy, scratch, M, C, r, j0);
  }
  hipLaunchKernelGGL(lora_wgrad_reduce_k, dim3((r * C + 255) / 256), dim3(256), 0, st, scratch, nchunks, r, C, out, transpose_out, alpha);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

// n <= 4 weight-gradient products over the same M rows and rank: the partial kernels as in lora_wgrad, into consecutive regions of the scratch,
// then one reduce launch for all of them.  Falls back to n separate lora_wgrad calls when the regions do not fit the scratch.
int lora_wgrad_batch(hipStream_t st, int dtype, const LoraWgradItem* items, int n, long long M, int r, float* scratch, long long scratch_floats) {
  UVX_CHECK(n >= 1 && n <= 4 && r > 0 && r <= RMAX, UVX_ERR_INVALID, "lora_wgrad_batch: n=%d r=%d", n, r);
  const int nchunks = (int)((M + RCH - 1) / RCH);
  long long need = 0;
  int cmax = 0;
  for (int i = 0; i < n; ++i) {
    UVX_CHECK(items[i].C % 8 == 0 && items[i].ldx % 8 == 0 && items[i].ldy % 8 == 0, UVX_ERR_SHAPE, "lora_wgrad_batch: C=%d unsupported", items[i].C);
    need += (long long)nchunks * r * items[i].C;
    cmax = items[i].C > cmax ? items[i].C : cmax;
  }
  if (M == 0 || need > scratch_floats || n == 1) {
    for (int i = 0; i < n; ++i) {
      const int rc = lora_wgrad(st, dtype, items[i].X, items[i].ldx, items[i].Y, items[i].ldy, items[i].out, M, items[i].C, r, items[i].transpose_out,
                                items[i].alpha, scratch);
      if (rc != UVX_OK) return rc;
    }
    return UVX_OK;
  }
  WgradReduceBatch b = {};
  b.nchunks = nchunks; b.r = r;
  float* region = scratch;
  for (int i = 0; i < n; ++i) {
    const LoraWgradItem& it = items[i];
    b.partial[i] = region; b.out[i] = it.out; b.C[i] = it.C; b.transpose_out[i] = it.transpose_out; b.alpha[i] = it.alpha;
    region += (long long)nchunks * r * it.C;
  }
  if (g_options[22] != 1) {      // (round 6) the n products' partial sums in ONE launch per 8 ranks: blockIdx.z = product
    const dim3 grid((cmax + 63) / 64, nchunks, n);
    for (int j0 = 0; j0 < r; j0 += 8) {
      if (dtype == DT_BF16) {
        LoraWgrad4<bf16_t> q = {};
        for (int i = 0; i < n; ++i) { q.X[i] = (const bf16_t*)items[i].X; q.Y[i] = (const bf16_t*)items[i].Y; q.partial[i] = const_cast<float*>(b.partial[i]); q.ldx[i] = items[i].ldx; q.ldy[i] = items[i].ldy; q.C[i] = items[i].C; }
        hipLaunchKernelGGL(lora_wgrad4_k<bf16_t>, grid, dim3(256), 0, st, q, M, r, j0);
      } else {
        LoraWgrad4<float> q = {};
        for (int i = 0; i < n; ++i) { q.X[i] = (const float*)items[i].X; q.Y[i] = (const float*)items[i].Y; q.partial[i] = const_cast<float*>(b.partial[i]); q.ldx[i] = items[i].ldx; q.ldy[i] = items[i].ldy; q.C[i] = items[i].C; }
        hipLaunchKernelGGL(lora_wgrad4_k<float>, grid, dim3(256), 0, st, q, M, r, j0);
      }
    }
  } else {
    for (int i = 0; i < n; ++i) {
      const LoraWgradItem& it = items[i];
      const dim3 grid((it.C + 63) / 64, nchunks);
      for (int j0 = 0; j0 < r; j0 += 8) {
        if (dtype == DT_BF16) hipLaunchKernelGGL(lora_wgrad_k<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)it.X, it.ldx, (const bf16_t*)it.Y, it.ldy, const_cast<float*>(b.partial[i]), M, it.C, r, j0);
        else hipLaunchKernelGGL(lora_wgrad_k<float>, grid, dim3(256), 0, st, (const float*)it.X, it.ldx, (const float*)it.Y, it.ldy, const_cast<float*>(b.partial[i]), M, it.C, r, j0);
      }
    }
  }
  hipLaunchKernelGGL(lora_wgrad_reduce_batch_k, dim3((r * cmax + 255) / 256, n), dim3(256), 0, st, b);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

int lora_transpose2(hipStream_t st, int dtype, const void* in0, void* out0, int C0, const void* in1, void* out1, int C1, int r) {
  const int cmax = C0 > C1 ? C0 : C1;
  const dim3 grid((r * cmax + 255) / 256, 2);
  if (dtype == DT_BF16) hipLaunchKernelGGL(lora_transpose2_k<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)in0, (bf16_t*)out0, C0, (const bf16_t*)in1, (bf16_t*)out1, C1, r);
  else hipLaunchKernelGGL(lora_transpose2_k<float>, grid, dim3(256), 0, st, (const float*)in0, (float*)out0, C0, (const float*)in1, (float*)out1, C1, r);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

int lora_transpose(hipStream_t st, int dtype, const void* in, void* out, int C, int r) {
  if (dtype == DT_BF16) hipLaunchKernelGGL(lora_transpose_k<bf16_t>, dim3((r * C + 255) / 256), dim3(256), 0, st, (const bf16_t*)in, (bf16_t*)out, C, r);
  else hipLaunchKernelGGL(lora_transpose_k<float>, dim3((r * C + 255) / 256), dim3(256), 0, st, (const float*)in, (float*)out, C, r);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

// ---- whole adapted linears: what the towers' training paths (model.hip, wav2vec2.hip) call per projection ----
#define UVX_LORA_RC(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)
// One adapted linear (peft Linear.forward, dropout 0): y[M, cout] += round(scale * (x A^T) B^T).  t [M, 64 of a 128-column row] keeps
// x A^T and bT [r, cout] the transposed lora_B for the backward.
int lora_apply(hipStream_t st, int dt, const void* x, long long ldx, const uvx_lora_proj_t& P, void* bT, void* t, void* y, long long ldy,
                      long long M, int cin, int cout, int r, float scale) {
  UVX_LORA_RC(lora_transpose(st, dt, P.b, bT, cout, r));
  UVX_LORA_RC(lora_down(st, dt, x, ldx, P.a, 0, t, 128, M, cin, r, 1.0f));
  return lora_up(st, dt, t, 128, bT, 1, y, ldy, M, cout, r, scale, 1);
}
// ... and its backward: u [M, 64 of 128] = round(scale * d y . B);  d A [r, cin] = u^T . x;  d B [cout, r] = scale * d y^T . t.  The caller adds
// u . A to d x once the base dgrad has written it (lora_up, accumulate).
int lora_apply_bwd(hipStream_t st, int dt, const void* x, long long ldx, const void* dy, long long lddy, const void* bT, const void* t, void* u,
                          const uvx_lora_proj_grad_t& G, long long M, int cin, int cout, int r, float scale, float* scratch, long long scratch_floats) {
  UVX_LORA_RC(lora_down(st, dt, dy, lddy, bT, 0, u, 128, M, cout, r, scale));
  const LoraWgradItem items[2] = {{x, ldx, u, 128, G.a, cin, 0, 1.0f}, {dy, lddy, t, 128, G.b, cout, 1, scale}};
  return lora_wgrad_batch(st, dt, items, 2, M, r, scratch, scratch_floats);
}

// descriptor sanity: an adapted projection (a != NULL) has its lora_B and - with `grads` - both gradient buffers
int lora_check(const uvx_encoder_lora_t* lora, int n_layers, const uvx_encoder_lora_grads_t* grads, const char* who) {
  UVX_CHECK(lora && lora->layers && lora->r > 0 && lora->r <= 64, UVX_ERR_INVALID, "%s: bad LoRA descriptor (rank must be in 1..64)", who);
  for (int l = 0; l < n_layers; ++l) {
    const uvx_enc_lora_layer_t& R = lora->layers[l];
    const uvx_lora_proj_t* P[7] = {&R.q, &R.k, &R.v, &R.o, &R.g, &R.u, &R.d};
    for (int j = 0; j < 7; ++j) {
      UVX_CHECK(!P[j]->a || P[j]->b, UVX_ERR_INVALID, "%s: layer %d, projection %d has lora_A but no lora_B", who, l, j);
      if (grads && P[j]->a) {
        const uvx_enc_lora_layer_grads_t& GG = grads->layers[l];
        const uvx_lora_proj_grad_t* G[7] = {&GG.q, &GG.k, &GG.v, &GG.o, &GG.g, &GG.u, &GG.d};
        UVX_CHECK(G[j]->a && G[j]->b, UVX_ERR_INVALID, "%s: layer %d, projection %d is adapted but has no gradient buffers", who, l, j);
      }
    }
  }
  return UVX_OK;
}

#undef UVX_LORA_RC

}  // namespace uvx
