// LayerNorm / RMSNorm kernels (HBM-bound; 16-byte vector access, f32 statistics).
//
//  layernorm_fwd      — [3P] nn.LayerNorm inside WhisperEncoderLayer + final layer_norm
//                       (reference call sites ultravox_model.py:966-973, :980), eps 1e-5.
//  rmsnorm_fwd/bwd    — LlamaRMSNorm semantics (ultravox_model.py:733-736): statistics in f32,
//                       x_hat cast to the activation dtype BEFORE the weight multiply.
//  stack_rmsnorm_fwd  — StackAudioFrames (ultravox_model.py:722-730) fused with ln_pre (:791).
#include "common.h"
#include "kernels.h"

namespace {

// 8 elements in their storage type, kept in registers between two uses
template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> {
  u16x8_t v;
  __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const u16x8_t*>(p); }
  __device__ __forceinline__ void get(float* o) const {
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = bf2f(v[i]);
  }
};
template <> struct Raw8<float> {
  float v[8];
  __device__ __forceinline__ void load(const float* p) { ld8<float>(p, v); }
  __device__ __forceinline__ void get(float* o) const {
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = v[i];
  }
};
constexpr int MAXV = 6;  // 8-element vectors per thread kept in registers (cols <= 256*8*6 = 12288)

template <typename T>
__global__ void layernorm_fwd_k(const T* __restrict__ x, const T* __restrict__ w, const T* __restrict__ b,
                                T* __restrict__ y, int cols, float eps) {
  __shared__ float red[16];
  const long long row = blockIdx.x;
  const T* xr = x + row * cols;
  T* yr = y + row * cols;
  float s = 0.f;
  for (int c = threadIdx.x * 8; c < cols; c += blockDim.x * 8) {
    float v[8];
    ld8<T>(xr + c, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
  }
  const float mean = block_sum(s, red) / cols;
  float q = 0.f;
  for (int c = threadIdx.x * 8; c < cols; c += blockDim.x * 8) {
    float v[8];
    ld8<T>(xr + c, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float d = v[i] - mean; q += d * d; }
  }
  const float rstd = rsqrtf(block_sum(q, red) / cols + eps);
  for (int c = threadIdx.x * 8; c < cols; c += blockDim.x * 8) {
    float v[8], wv[8], bv[8], o[8];
    ld8<T>(xr + c, v);
    ld8<T>(w + c, wv);
    ld8<T>(b + c, bv);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (v[i] - mean) * rstd * wv[i] + bv[i];
    st8<T>(yr + c, o);
  }
}

// bf16 rows of NV * 512 elements, ONE WAVE PER ROW (4 rows per block): the row is read once into registers, the two
// reductions are wave shuffles - no LDS, no block barrier.  Same arithmetic as the block kernels (a different, fixed
// summation order).
template <int NV>
__global__ __launch_bounds__(256) void layernorm_fwd_wave_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                            const bf16_t* __restrict__ b, bf16_t* __restrict__ y,
                                                            long long rows, float eps) {
  constexpr int cols = NV * 512;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const bf16_t* xr = x + row * cols;
  Raw8<bf16_t> xv[NV];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    xv[k].load(xr + (k * 64 + lane) * 8);
    float v[8];
    xv[k].get(v);
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
  }
  const float mean = wave_sum(s) / cols;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    float v[8];
    xv[k].get(v);
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float d = v[i] - mean; q += d * d; }
  }
  const float rstd = rsqrtf(wave_sum(q) / cols + eps);
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = (k * 64 + lane) * 8;
    float v[8], wv[8], bv[8], o[8];
    xv[k].get(v);
    ld8<bf16_t>(w + c, wv);
    ld8<bf16_t>(b + c, bv);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (v[i] - mean) * rstd * wv[i] + bv[i];
    st8<bf16_t>(y + row * cols + c, o);
  }
}

template <int NV>
__global__ __launch_bounds__(256) void rmsnorm_fwd_wave_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                          bf16_t* __restrict__ y, float* __restrict__ rstd_out,
                                                          long long rows, float eps, int flavor) {
  constexpr int cols = NV * 512;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const bf16_t* xr = x + row * cols;
  Raw8<bf16_t> xv[NV];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    xv[k].load(xr + (k * 64 + lane) * 8);
    float v[8];
    xv[k].get(v);
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i] * v[i];
  }
  const float rstd = rsqrtf(wave_sum(s) / cols + eps);
  if (rstd_out && lane == 0) rstd_out[row] = rstd;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = (k * 64 + lane) * 8;
    float v[8], wv[8], o[8];
    xv[k].get(v);
    ld8<bf16_t>(w + c, wv);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = flavor ? (v[i] * rstd) * (1.0f + wv[i]) : wv[i] * rnd<bf16_t>(v[i] * rstd);
    st8<bf16_t>(y + row * cols + c, o);
  }
}

// LayerNorm backward for a FROZEN affine (Whisper encoder under LoRA: only the input gradient is needed):
//   g = dy * w,  x_hat = (x - mean) * rstd,  dx = rstd * (g - mean(g) - x_hat * mean(g * x_hat)) [+ dx_add]
// mean / rstd are recomputed exactly as the forward kernel does (two passes), one block per row.
template <typename T>
__global__ void layernorm_bwd_k(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ w,
                                const T* __restrict__ dx_add, T* __restrict__ dx, int cols, float eps) {
  __shared__ float red[16];
  __shared__ float red2[16];
  const long long row = blockIdx.x;
  const T* xr = x + row * cols;
  const T* gr = dy + row * cols;
  float s = 0.f;
  for (int c = threadIdx.x * 8; c < cols; c += blockDim.x * 8) {
    float v[8];
    ld8<T>(xr + c, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
  }
  const float mean = block_sum(s, red) / cols;
  float q = 0.f;
  for (int c = threadIdx.x * 8; c < cols; c += blockDim.x * 8) {
    float v[8];
    ld8<T>(xr + c, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float d = v[i] - mean; q += d * d; }
  }
  const float rstd = rsqrtf(block_sum(q, red) / cols + eps);
  float a = 0.f, b = 0.f;
  for (int c = threadIdx.x * 8; c < cols; c += blockDim.x * 8) {
    float v[8], g[8], wv[8];
    ld8<T>(xr + c, v);
    ld8<T>(gr + c, g);
    ld8<T>(w + c, wv);
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float gw = g[i] * wv[i]; a += gw; b += gw * (v[i] - mean) * rstd; }
  }
  a = block_sum(a, red) / cols;
  b = block_sum(b, red2) / cols;
  for (int c = threadIdx.x * 8; c < cols; c += blockDim.x * 8) {
    float v[8], g[8], wv[8], o[8];
    ld8<T>(xr + c, v);
    ld8<T>(gr + c, g);
    ld8<T>(w + c, wv);
    if (dx_add) ld8<T>(dx_add + row * cols + c, o);
    else {
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] += rstd * (g[i] * wv[i] - a - (v[i] - mean) * rstd * b);
    st8<T>(dx + row * cols + c, o);
  }
}

// Row addressing shared by the plain and the frame-stacking RMSNorm: a logical row of `cols`
// elements starts at `base` and only the first `valid` elements exist (the rest read as zero).
struct RowMap {
  int J, T, S, C;  // S == 0: plain [rows, cols]
};
__device__ __forceinline__ void row_span(const RowMap& m, long long row, int cols, long long& base, int& valid) {
  if (m.S == 0) { base = row * cols; valid = cols; return; }
  const long long b = row / m.J;
  const int j = (int)(row % m.J);
  base = (b * m.T + (long long)j * m.S) * m.C;
  const int frames = min(m.S, m.T - j * m.S);
  valid = frames * m.C;
}

template <typename T>
__global__ void rmsnorm_fwd_k(const T* __restrict__ x, const T* __restrict__ w, T* __restrict__ y,
                              T* __restrict__ stacked, float* __restrict__ rstd_out, int cols, float eps,
                              RowMap map, int flavor, const int32_t* __restrict__ rows_dev, const T* __restrict__ resid) {
  __shared__ float red[16];
  const long long row = blockIdx.x;
  if (rows_dev && row >= *rows_dev) return;     // block-uniform: the tail of a row-compacted buffer holds no data
  long long base; int valid;
  row_span(map, row, cols, base, valid);
  const T* xr = x + base;
  T* yr = y + row * cols;
  float s = 0.f;
  for (int c = threadIdx.x * 8; c < valid; c += blockDim.x * 8) {
    float v[8];
    ld8<T>(xr + c, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i] * v[i];
  }
  const float rstd = rsqrtf(block_sum(s, red) / cols + eps);
  if (rstd_out && threadIdx.x == 0) rstd_out[row] = rstd;
  for (int c = threadIdx.x * 8; c < cols; c += blockDim.x * 8) {
    float v[8], wv[8], o[8];
    if (c < valid) ld8<T>(xr + c, v);
    else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = 0.f;
    }
    ld8<T>(w + c, wv);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = flavor ? (v[i] * rstd) * (1.0f + wv[i]) : wv[i] * rnd<T>(v[i] * rstd);
    if (resid) {      // the norm's output is a tensor of its own (rounded) before the residual add
      float rv[8];
      ld8<T>(resid + row * cols + c, rv);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = rnd<T>(o[i]) + rv[i];
    }
    st8<T>(yr + c, o);
    if (stacked) st8<T>(stacked + row * cols + c, v);
  }
}

// The same for rows of at most 8 x blockDim x NV columns that are plain (no frame stacking): the row stays in registers between the two passes and
// the weight vectors are requested together with it, BEFORE the reduction - one round trip to memory instead of two.  At the decode step's few rows
// the kernel is pure latency (8.7 us per launch, 161 launches per 70B token at 3..16 sequences: profiles/r05_decode70_b8_kernel_stats.txt); same
// per-thread and block summation order, same arithmetic and rounding points as rmsnorm_fwd_k: bit-identical.
template <typename T, int NV>
__global__ void rmsnorm_fwd_reg_k(const T* __restrict__ x, const T* __restrict__ w, T* __restrict__ y, float* __restrict__ rstd_out, int cols,
                                  float eps, int flavor, const int32_t* __restrict__ rows_dev, const T* __restrict__ resid) {
  __shared__ float red[16];
  const long long row = blockIdx.x;
  if (rows_dev && row >= *rows_dev) return;     // block-uniform
  const T* xr = x + row * cols;
  float v[NV][8], wv[NV][8];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = (threadIdx.x + k * blockDim.x) * 8;
    if (c < cols) {
      ld8<T>(xr + c, v[k]);
      ld8<T>(w + c, wv[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = (threadIdx.x + k * blockDim.x) * 8;
    if (c < cols) {
#pragma unroll
      for (int i = 0; i < 8; ++i) s += v[k][i] * v[k][i];
    }
  }
  const float rstd = rsqrtf(block_sum(s, red) / cols + eps);
  if (rstd_out && threadIdx.x == 0) rstd_out[row] = rstd;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = (threadIdx.x + k * blockDim.x) * 8;
    if (c >= cols) continue;
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = flavor ? (v[k][i] * rstd) * (1.0f + wv[k][i]) : wv[k][i] * rnd<T>(v[k][i] * rstd);
    if (resid) {
      float rv[8];
      ld8<T>(resid + row * cols + c, rv);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = rnd<T>(o[i]) + rv[i];
    }
    st8<T>(y + row * cols + c, o);
  }
}

// One block handles `rpb` consecutive rows; each thread owns fixed columns so the weight gradient
// is accumulated in registers and flushed once per column per block: into dw_part [block][cols] when the caller lends that scratch
// (round 6: rmsnorm_dw_reduce_k then sums the blocks in block order - the projector's norm-weight gradients were the one output of a
// training step that moved by an ulp from run to run), else with an atomicAdd into dw (uvx_rmsnorm_bwd: the ABI has no scratch argument).
// MV: 8-element vectors per thread (static trip count; the launcher picks the smallest that covers the row - at the LLM's 4096
// columns MV = 2 needs half the registers of MV = 6 and twice as many rows are in flight per CU)
template <typename T, bool WANT_DX, bool WANT_DW, int MV>
__global__ void rmsnorm_bwd_k(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ w,
                              const T* __restrict__ dx_add, T* __restrict__ dx, float* __restrict__ dw,
                              int rows, int cols, float eps, int rpb, int flavor, const int32_t* __restrict__ rows_dev,
                              float* __restrict__ dw_part, uvx::RowSkip x_map, bool map_dx) {
  if (rows_dev) rows = min(rows, *rows_dev);     // device-side row count (row-compacted buffers)
  // flavor 1 (Gemma): y = x_hat * (1 + w) with no intermediate rounding -> the effective weight is 1 + w and
  // d w gets the UNROUNDED x_hat
  __shared__ float red[16];
  __shared__ float red2[16];
  float dwacc[MV][8];
  if (WANT_DW) {
#pragma unroll
    for (int j = 0; j < MV; ++j)
#pragma unroll
      for (int i = 0; i < 8; ++i) dwacc[j][i] = 0.f;
  }
  const int r0 = blockIdx.x * rpb;
  const int r1 = min(rows, r0 + rpb);
  for (int row = r0; row < r1; ++row) {
    const long long xrow = x_map.skip ? row + (row / x_map.tc + 1) * x_map.skip : row;      // (kernels.h RowSkip: the stash keeps every row)
    const T* xr = x + xrow * cols;
    const T* dyr = dy + (long long)row * cols;
    float s1 = 0.f, s2 = 0.f;
    // the row is read from global memory ONCE: x, dy and w stay in registers between the reduction and the update
    // (static trip count MV; the launcher guarantees cols <= MV * blockDim.x * 8)
    // (only where it fits the register budget: bf16 without the weight-gradient accumulators - the hot, frozen-LLM case)
    constexpr bool CACHE = sizeof(T) == 2 && !WANT_DW;
    Raw8<T> xc[CACHE ? MV : 1], gc[CACHE ? MV : 1];   // raw storage type: 4 registers per 8 bf16
#pragma unroll
    for (int j = 0; j < MV; ++j) {
      const int c = (j * blockDim.x + threadIdx.x) * 8;
      if (c >= cols) continue;
      Raw8<T> xr8, gr8;
      xr8.load(xr + c);
      gr8.load(dyr + c);
      if (CACHE) { xc[j] = xr8; gc[j] = gr8; }
      float xv[8], gv[8], wv[8];
      xr8.get(xv); gr8.get(gv);
      ld8<T>(w + c, wv);
#pragma unroll
      for (int i = 0; i < 8; ++i) { s1 += xv[i] * xv[i]; s2 += gv[i] * (flavor ? 1.0f + wv[i] : wv[i]) * xv[i]; }
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red2);
    const float r = rsqrtf(s1 / cols + eps);
    const float coef = r * r * r * s2 / cols;
#pragma unroll
    for (int j = 0; j < MV; ++j) {  // static trip count: dwacc[j] must stay in registers
      const int c = (j * blockDim.x + threadIdx.x) * 8;
      if (c >= cols) continue;
      float xv[8], gv[8], wv[8];
      if (CACHE) { xc[j].get(xv); gc[j].get(gv); }
      else { ld8<T>(xr + c, xv); ld8<T>(dyr + c, gv); }
      if (WANT_DX) {
        ld8<T>(w + c, wv);   // the weight row is shared by every block: L2 hit
        float o[8];
        if (dx_add) ld8<T>(dx_add + (long long)row * cols + c, o);
        else {
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] += r * gv[i] * (flavor ? 1.0f + wv[i] : wv[i]) - xv[i] * coef;
        st8<T>(dx + (map_dx ? xrow : (long long)row) * cols + c, o);
      }
      if (WANT_DW) {
#pragma unroll
        for (int i = 0; i < 8; ++i) dwacc[j][i] += gv[i] * (flavor ? xv[i] * r : rnd<T>(xv[i] * r));
      }
    }
  }
  if (WANT_DW) {
#pragma unroll
    for (int j = 0; j < MV; ++j) {
      const int c = (j * blockDim.x + threadIdx.x) * 8;
      if (c >= cols) continue;
      if (dw_part) {
        *reinterpret_cast<float4*>(dw_part + (long long)blockIdx.x * cols + c) = make_float4(dwacc[j][0], dwacc[j][1], dwacc[j][2], dwacc[j][3]);
        *reinterpret_cast<float4*>(dw_part + (long long)blockIdx.x * cols + c + 4) = make_float4(dwacc[j][4], dwacc[j][5], dwacc[j][6], dwacc[j][7]);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) atomicAdd(dw + c + i, dwacc[j][i]);
      }
    }
  }
}

// dw[c] += sum over blocks (ascending) of dw_part[block][c]: fixed order, bit-reproducible
__global__ void rmsnorm_dw_reduce_k(const float* __restrict__ part, int nblocks, int cols, float* __restrict__ dw) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  float s = 0.f;
  for (int b = 0; b < nblocks; ++b) s += part[(long long)b * cols + c];
  dw[c] += s;
}

int norm_threads(int cols) {
  int t = ((cols / 8) + 63) / 64 * 64;
  return t < 64 ? 64 : (t > 256 ? 256 : t);
}

}  // namespace

namespace uvx {

int layernorm_fwd(hipStream_t st, int dtype, const void* x, const void* w, const void* b, void* y, int rows,
                  int cols, float eps) {
  UVX_CHECK(cols % 8 == 0, UVX_ERR_SHAPE, "layernorm: cols=%d must be a multiple of 8", cols);
  if (rows == 0) return UVX_OK;
  const int th = norm_threads(cols);
  if (dtype == DT_BF16 && (cols == 512 || cols == 1024 || cols == 2048 || cols == 4096)) {
    const dim3 grid((rows + 3) / 4), blk(256);
#define LW(NV) hipLaunchKernelGGL(layernorm_fwd_wave_k<NV>, grid, blk, 0, st, (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, (long long)rows, eps)
    if (cols == 512) LW(1); else if (cols == 1024) LW(2); else if (cols == 2048) LW(4); else LW(8);
#undef LW
  } else if (dtype == DT_BF16)
    hipLaunchKernelGGL(layernorm_fwd_k<bf16_t>, dim3(rows), dim3(th), 0, st, (const bf16_t*)x, (const bf16_t*)w,
                       (const bf16_t*)b, (bf16_t*)y, cols, eps);
  else
    hipLaunchKernelGGL(layernorm_fwd_k<float>, dim3(rows), dim3(th), 0, st, (const float*)x, (const float*)w,
                       (const float*)b, (float*)y, cols, eps);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

int layernorm_bwd(hipStream_t st, int dtype, const void* dy, const void* x, const void* w, const void* dx_add, void* dx,
                  int rows, int cols, float eps) {
  UVX_CHECK(cols % 8 == 0, UVX_ERR_SHAPE, "layernorm_bwd: cols=%d must be a multiple of 8", cols);
  if (rows == 0) return UVX_OK;
  const int th = norm_threads(cols);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(layernorm_bwd_k<bf16_t>, dim3(rows), dim3(th), 0, st, (const bf16_t*)dy, (const bf16_t*)x,
                       (const bf16_t*)w, (const bf16_t*)dx_add, (bf16_t*)dx, cols, eps);
  else
    hipLaunchKernelGGL(layernorm_bwd_k<float>, dim3(rows), dim3(th), 0, st, (const float*)dy, (const float*)x,
                       (const float*)w, (const float*)dx_add, (float*)dx, cols, eps);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

static int rms_launch(hipStream_t st, int dtype, const void* x, const void* w, void* y, void* stacked,
                      float* rstd, long long rows, int cols, float eps, RowMap map, int flavor = 0, const int32_t* rows_dev = nullptr,
                      const void* resid = nullptr) {
  UVX_CHECK(cols % 8 == 0, UVX_ERR_SHAPE, "rmsnorm: cols=%d must be a multiple of 8", cols);
  if (rows == 0) return UVX_OK;
  const int th = norm_threads(cols);
  // (at 4096 columns the block kernel is as fast - 11.5 vs 12.4 us at 2528 rows - and has 4x the blocks: keep it)
  if (dtype == DT_BF16 && map.S == 0 && !stacked && !rows_dev && !resid && (cols == 512 || cols == 1024 || cols == 2048)) {
    const dim3 grid((unsigned)((rows + 3) / 4)), blk(256);
#define LW(NV) hipLaunchKernelGGL(rmsnorm_fwd_wave_k<NV>, grid, blk, 0, st, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, rstd, rows, eps, flavor)
    if (cols == 512) LW(1); else if (cols == 1024) LW(2); else LW(4);
#undef LW
  } else if (dtype == DT_BF16 && map.S == 0 && !stacked && cols <= th * 8 * 4 && rows <= 512 && uvx::g_options[18] == 0) {
    // few rows - the decode step, a prompt's prefill: latency-bound, one round trip instead of two (70B decode 25.64 -> 25.34 ms per token at
    // B = 8, the 8B model 4.08 -> 3.98; profiles/r05_rmsnorm_reg_ab.txt).  The training step's 2528 rows are throughput-bound and read
    // 0.1-0.2 ms per step SLOWER with it: they keep the two-pass kernel.  (option 18 = 1: the two-pass kernel everywhere, for A/B)
    if (cols <= th * 8) hipLaunchKernelGGL((rmsnorm_fwd_reg_k<bf16_t, 1>), dim3(rows), dim3(th), 0, st, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, rstd, cols, eps, flavor, rows_dev, (const bf16_t*)resid);
    else if (cols <= th * 8 * 2) hipLaunchKernelGGL((rmsnorm_fwd_reg_k<bf16_t, 2>), dim3(rows), dim3(th), 0, st, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, rstd, cols, eps, flavor, rows_dev, (const bf16_t*)resid);
    else hipLaunchKernelGGL((rmsnorm_fwd_reg_k<bf16_t, 4>), dim3(rows), dim3(th), 0, st, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, rstd, cols, eps, flavor, rows_dev, (const bf16_t*)resid);
  } else if (dtype == DT_BF16)
    hipLaunchKernelGGL(rmsnorm_fwd_k<bf16_t>, dim3(rows), dim3(th), 0, st, (const bf16_t*)x, (const bf16_t*)w,
                       (bf16_t*)y, (bf16_t*)stacked, rstd, cols, eps, map, flavor, rows_dev, (const bf16_t*)resid);
  else
    hipLaunchKernelGGL(rmsnorm_fwd_k<float>, dim3(rows), dim3(th), 0, st, (const float*)x, (const float*)w,
                       (float*)y, (float*)stacked, rstd, cols, eps, map, flavor, rows_dev, (const float*)resid);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

int rmsnorm_fwd(hipStream_t st, int dtype, const void* x, const void* w, void* y, float* rstd, int rows,
                int cols, float eps, int flavor, const int32_t* rows_dev, const void* resid) {
  return rms_launch(st, dtype, x, w, y, nullptr, rstd, rows, cols, eps, RowMap{0, 0, 0, 0}, flavor, rows_dev, resid);
}

int stack_rmsnorm_fwd(hipStream_t st, int dtype, const void* x, const void* w, void* y, void* stacked, int B,
                      int T, int C, int S, float eps) {
  UVX_CHECK(S > 0 && C % 8 == 0, UVX_ERR_SHAPE, "stack_rmsnorm: bad S=%d C=%d", S, C);
  const int J = (T + S - 1) / S;
  return rms_launch(st, dtype, x, w, y, stacked, nullptr, (long long)B * J, C * S, eps, RowMap{J, T, S, C});
}

template <typename T>
static int rms_bwd_launch(hipStream_t st, const void* dy, const void* x, const void* w, const void* dx_add,
                          void* dx, float* dw, int rows, int cols, float eps, int flavor, const int32_t* rows_dev, float* dw_part,
                          uvx::RowSkip x_map, bool map_dx) {
  const int th = 256;   // (512 threads with one vector each: 20.4 vs 17.7 us at 2528 x 4096 - profiles/r03_rmsnorm_bwd_variants.txt)
  UVX_CHECK(cols % 8 == 0 && cols <= th * 8 * MAXV, UVX_ERR_SHAPE, "rmsnorm_bwd: cols=%d unsupported", cols);
  if (rows == 0) return UVX_OK;
  const int rpb = dw ? 16 : 1;
  const int grid = (rows + rpb - 1) / rpb;
#define L(DX, DW)                                                                                                 \
  do {                                                                                                            \
    if (cols <= th * 8)                                                                                           \
      hipLaunchKernelGGL((rmsnorm_bwd_k<T, DX, DW, 1>), dim3(grid), dim3(th), 0, st, (const T*)dy, (const T*)x,   \
                         (const T*)w, (const T*)dx_add, (T*)dx, dw, rows, cols, eps, rpb, flavor, rows_dev, dw ? dw_part : nullptr, x_map, map_dx);     \
    else if (cols <= th * 8 * 2)                                                                                  \
      hipLaunchKernelGGL((rmsnorm_bwd_k<T, DX, DW, 2>), dim3(grid), dim3(th), 0, st, (const T*)dy, (const T*)x,   \
                         (const T*)w, (const T*)dx_add, (T*)dx, dw, rows, cols, eps, rpb, flavor, rows_dev, dw ? dw_part : nullptr, x_map, map_dx);     \
    else                                                                                                          \
      hipLaunchKernelGGL((rmsnorm_bwd_k<T, DX, DW, MAXV>), dim3(grid), dim3(th), 0, st, (const T*)dy, (const T*)x, \
                         (const T*)w, (const T*)dx_add, (T*)dx, dw, rows, cols, eps, rpb, flavor, rows_dev, dw ? dw_part : nullptr, x_map, map_dx);     \
  } while (0)
  if (dx && dw) L(true, true);
  else if (dx) L(true, false);
  else if (dw) L(false, true);
#undef L
  if (dw && dw_part) hipLaunchKernelGGL(rmsnorm_dw_reduce_k, dim3(cdiv(cols, 256)), dim3(256), 0, st, dw_part, grid, cols, dw);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

long long rmsnorm_bwd_dw_scratch_floats(int rows, int cols) { return (long long)((rows + 15) / 16) * cols; }

int rmsnorm_bwd(hipStream_t st, int dtype, const void* dy, const void* x, const void* w, const void* dx_add,
                void* dx, float* dw, int rows, int cols, float eps, int flavor, const int32_t* rows_dev, float* dw_part, RowSkip x_map, bool map_dx) {
  return dtype == DT_BF16 ? rms_bwd_launch<bf16_t>(st, dy, x, w, dx_add, dx, dw, rows, cols, eps, flavor, rows_dev, dw_part, x_map, map_dx)
                          : rms_bwd_launch<float>(st, dy, x, w, dx_add, dx, dw, rows, cols, eps, flavor, rows_dev, dw_part, x_map, map_dx);
}

}  // namespace uvx
