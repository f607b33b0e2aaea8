// HBM-bound glue kernels of the Ultravox hot path (16-byte vector access everywhere).
//   swiglu fwd/bwd      — SwiGLU (ultravox_model.py:739-742: first half = value, second half = gate)
//                         and the [3P] LlamaMLP act_fn(gate)*up (gate_first = 1 in our fused layout)
//   rope_inplace        — [3P] apply_rotary_pos_emb (rotate_half form), forward and inverse
//   embed_gather        — embed_tokens(input_ids)            (ultravox_model.py:314-316)
//   merge_audio(+bwd)   — the sequential in-place overwrite  (ultravox_model.py:390-394, :259-275)
//   transposes / im2col — layout changes feeding the one NT GEMM kernel
#include "common.h"
#include "kernels.h"

namespace {

// GLU activations.  ACT 0: silu(g) = g * sigmoid(g); ACT 1: [3P] gelu_pytorch_tanh(g) = 0.5 g (1 + tanh(k (g + 0.044715 g^3))),
// k = sqrt(2 / pi) - Gemma's hidden_act; ACT 2: exact GELU 0.5 g (1 + erf(g / sqrt 2)) - what ACT2FN["gelu"] is, i.e. what a
// Gemma checkpoint whose config.json says hidden_act = "gelu" computes ([3P] GemmaMLP reads config.hidden_act).  dact = d act / d g.
template <int ACT>
__device__ __forceinline__ float glu_act(float g) {
  if (ACT == 0) return g / (1.0f + expf(-g));
  if (ACT == 2) return 0.5f * g * (1.0f + erff(g * 0.70710678118654752440f));
  const float u = 0.7978845608028654f * (g + 0.044715f * g * g * g);
  return 0.5f * g * (1.0f + tanhf(u));
}
template <int ACT>
__device__ __forceinline__ void glu_act_grad(float g, float& a, float& da) {
  if (ACT == 0) {
    const float s = 1.0f / (1.0f + expf(-g));
    a = g * s;
    da = s * (1.0f + g * (1.0f - s));
  } else if (ACT == 2) {
    const float cdf = 0.5f * (1.0f + erff(g * 0.70710678118654752440f));
    a = g * cdf;
    da = cdf + g * 0.3989422804014327f * expf(-0.5f * g * g);      // Phi(g) + g phi(g)
  } else {
    const float u = 0.7978845608028654f * (g + 0.044715f * g * g * g);
    const float t = tanhf(u);
    a = 0.5f * g * (1.0f + t);
    da = 0.5f * (1.0f + t) + 0.5f * g * (1.0f - t * t) * 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * g * g);
  }
}

// Plain (un-gated) activations of the projector when config.projector_act is not "swiglu" (ultravox_model.py:754: transformers.activations
// .get_activation): ACT 0 silu / swish, 1 gelu_new / gelu_pytorch_tanh, 2 gelu (exact), 3 relu.  One rounding of the f32 result, as the torch
// module's output; the backward multiplies the incoming gradient with the exact derivative at the saved pre-activation.
template <typename T, int ACT>
__global__ void act_fwd_k(const T* __restrict__ in, T* __restrict__ out, long long n8) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  float v[8];
  ld8<T>(in + i * 8, v);
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = ACT == 3 ? fmaxf(v[k], 0.f) : glu_act<ACT == 3 ? 0 : ACT>(v[k]);
  st8<T>(out + i * 8, v);
}
template <typename T, int ACT>
__global__ void act_bwd_k(const T* __restrict__ dout, const T* __restrict__ in, T* __restrict__ din, long long n8) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  float x[8], g[8];
  ld8<T>(in + i * 8, x);
  ld8<T>(dout + i * 8, g);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (ACT == 3) g[k] = x[k] > 0.f ? g[k] : 0.f;
    else {
      float a, da;
      glu_act_grad<ACT == 3 ? 0 : ACT>(x[k], a, da);
      g[k] *= da;
    }
  }
  st8<T>(din + i * 8, g);
}

template <typename T>
__global__ void scale_k(T* __restrict__ x, long long n8, float s) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  float v[8];
  ld8<T>(x + i * 8, v);
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] *= s;
  st8<T>(x + i * 8, v);
}

template <typename T, int ACT>
__global__ void swiglu_fwd_k(const T* __restrict__ in, T* __restrict__ out, long long n8, int half, int gate_first,
                             const int32_t* __restrict__ rows_dev) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const int hv = half / 8;
  const long long row = i / hv;
  if (rows_dev && row >= *rows_dev) return;    // device-side row count (compacted supervised rows): the tail holds no data
  const int c = (int)(i % hv) * 8;
  const T* r = in + row * 2 * half;
  float a[8], g[8], o[8];
  const int voff = gate_first == 2 ? (c / 16) * 32 + (c % 16) + 16 : (gate_first ? half + c : c);
  const int goff = gate_first == 2 ? (c / 16) * 32 + (c % 16) : (gate_first ? c : half + c);
  ld8<T>(r + voff, a);   // value / up
  ld8<T>(r + goff, g);   // gate
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = rnd<T>(glu_act<ACT>(g[k])) * a[k];
  st8<T>(out + row * half + c, o);
}

template <typename T, int ACT>
__global__ void swiglu_bwd_k(const T* __restrict__ dout, const T* __restrict__ in, T* __restrict__ din,
                             long long n8, int half, int gate_first, const int32_t* __restrict__ rows_dev, uvx::RowSkip in_map) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const int hv = half / 8;
  const long long row = i / hv;
  if (rows_dev && row >= *rows_dev) return;
  const int c = (int)(i % hv) * 8;
  const T* r = in + (in_map.skip ? row + (row / in_map.tc + 1) * in_map.skip : row) * 2 * half;
  T* dr = din + row * 2 * half;
  const int voff = gate_first == 2 ? (c / 16) * 32 + (c % 16) + 16 : (gate_first ? half + c : c);
  const int goff = gate_first == 2 ? (c / 16) * 32 + (c % 16) : (gate_first ? c : half + c);
  float a[8], g[8], d[8], da[8], dg[8];
  ld8<T>(r + voff, a);
  ld8<T>(r + goff, g);
  ld8<T>(dout + row * half + c, d);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float act, dact;
    glu_act_grad<ACT>(g[k], act, dact);
    da[k] = d[k] * rnd<T>(act);
    dg[k] = d[k] * a[k] * dact;
  }
  st8<T>(dr + voff, da);
  st8<T>(dr + goff, dg);
}

// cos_sin: [T_table, D/2, 2] f32.  x: [rows, ld]; heads 0..H-1 of width D start at column 0.
template <typename T>
__global__ void rope_k(T* __restrict__ x, const float* __restrict__ cs, const int32_t* __restrict__ pos,
                       long long n_items, int Tlen, int H, int D, int ld, float sgn) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_items) return;
  const int per_head = D / 16;  // 8-element vectors in the first half
  const int per_row = H * per_head;
  const long long row = i / per_row;
  const int rem = (int)(i % per_row);
  const int h = rem / per_head, c = (rem % per_head) * 8;
  const int p = pos ? pos[row] : (int)(row % Tlen);
  T* base = x + row * ld + h * D;
  const float* t = cs + ((long long)p * (D / 2) + c) * 2;
  float lo[8], hi[8], olo[8], ohi[8];
  ld8<T>(base + c, lo);
  ld8<T>(base + D / 2 + c, hi);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float co = rnd<T>(t[2 * k]), si = rnd<T>(t[2 * k + 1]) * sgn;
    olo[k] = rnd<T>(lo[k] * co) + rnd<T>(-hi[k] * si);
    ohi[k] = rnd<T>(hi[k] * co) + rnd<T>(lo[k] * si);
  }
  st8<T>(base + c, olo);
  st8<T>(base + D / 2 + c, ohi);
}

// Qwen3: per-head RMSNorm of q and k ([3P] transformers modeling_qwen3.py Qwen3Attention: q_norm / k_norm = Qwen3RMSNorm(head_dim)
// on the projections viewed as [.., heads, head_dim], BEFORE the rotary embedding; Qwen3RMSNorm == LlamaRMSNorm: w * round(x_hat)),
// fused with rope_k's rotary embedding (same arithmetic and rounding points as rmsnorm_fwd_k followed by rope_k: the normalised
// row is rounded to the storage type before it is rotated).  One thread = the 8-column chunk c of the first half of a head and
// its rotary partner c + D/2; the D/16 threads of a head share the sum of squares through wave shuffles.  raw != null: the
// un-normalised q | k rows are kept ([rows, (Hq + Hkv) * D], the backward needs them); v is untouched.
template <typename T>
__global__ void qk_norm_rope_k(T* __restrict__ x, const T* __restrict__ wq, const T* __restrict__ wk, T* __restrict__ raw,
                               const float* __restrict__ cs, const int32_t* __restrict__ pos, long long n_items, int Tlen, int Hq,
                               int Hkv, int D, int ld, float eps, int flavor) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_items) return;      // (whole heads: n_items is a multiple of D/16, a power of two that divides the block)
  const int per_head = D / 16, H = Hq + Hkv;
  const int per_row = H * per_head;
  const long long row = i / per_row;
  const int rem = (int)(i % per_row);
  const int h = rem / per_head, c = (rem % per_head) * 8;
  const int p = pos ? pos[row] : (int)(row % Tlen);
  T* base = x + row * ld + h * D;
  const T* w = h < Hq ? wq : wk;
  float lo[8], hi[8], wl[8], wh[8], olo[8], ohi[8];
  ld8<T>(base + c, lo);
  ld8<T>(base + D / 2 + c, hi);
  if (raw) {
    T* r = raw + (row * H + h) * D;
    st8<T>(r + c, lo);
    st8<T>(r + D / 2 + c, hi);
  }
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) ss += lo[k] * lo[k] + hi[k] * hi[k];
  for (int m = 1; m < per_head; m <<= 1) ss += __shfl_xor(ss, m, 64);
  const float rstd = rsqrtf(ss / D + eps);
  ld8<T>(w + c, wl);
  ld8<T>(w + D / 2 + c, wh);
  const float* t = cs + ((long long)p * (D / 2) + c) * 2;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    // flavor 1 (Gemma-3's Gemma3RMSNorm): x_hat * (1 + w) in f32, one rounding; 0 (Qwen3): w * round(x_hat)
    const float nl = flavor ? rnd<T>((lo[k] * rstd) * (1.0f + wl[k])) : rnd<T>(wl[k] * rnd<T>(lo[k] * rstd));
    const float nh = flavor ? rnd<T>((hi[k] * rstd) * (1.0f + wh[k])) : rnd<T>(wh[k] * rnd<T>(hi[k] * rstd));
    const float co = rnd<T>(t[2 * k]), si = rnd<T>(t[2 * k + 1]);
    olo[k] = rnd<T>(nl * co) + rnd<T>(-nh * si);
    ohi[k] = rnd<T>(nh * co) + rnd<T>(nl * si);
  }
  st8<T>(base + c, olo);
  st8<T>(base + D / 2 + c, ohi);
}

// Backward of the per-head norm (frozen weights: activation gradient only), in place on the q | k columns of d_qkv: dy = the
// gradient of the NORMALISED rows (the attention backward's RoPE-inverted dq / dk), x = the raw rows the forward kept.
// rmsnorm_bwd_k's arithmetic per row of D: r = rsqrt(mean(x^2) + eps), dx = r dy w - x r^3 mean(dy w x).  One thread = 8 columns.
template <typename T>
__global__ void qk_norm_bwd_k(T* __restrict__ dqk, const T* __restrict__ raw, const T* __restrict__ wq, const T* __restrict__ wk,
                              long long n_items, int Hq, int Hkv, int D, int ld, float eps, int flavor, uvx::RowSkip raw_map) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_items) return;
  const int per_head = D / 8, H = Hq + Hkv;
  const int per_row = H * per_head;
  const long long row = i / per_row;
  const int rem = (int)(i % per_row);
  const int h = rem / per_head, c = (rem % per_head) * 8;
  T* g = dqk + row * ld + h * D + c;
  const T* w = (h < Hq ? wq : wk) + c;
  float xv[8], gv[8], wv[8], o[8];
  const long long xrow = raw_map.skip ? row + (row / raw_map.tc + 1) * raw_map.skip : row;      // (kernels.h RowSkip: the stash keeps every row)
  ld8<T>(raw + (xrow * H + h) * D + c, xv);
  ld8<T>(g, gv);
  ld8<T>(w, wv);
  if (flavor) {
#pragma unroll
    for (int k = 0; k < 8; ++k) wv[k] += 1.0f;      // Gemma: the effective weight is 1 + w
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) { s1 += xv[k] * xv[k]; s2 += gv[k] * wv[k] * xv[k]; }
  for (int m = 1; m < per_head; m <<= 1) { s1 += __shfl_xor(s1, m, 64); s2 += __shfl_xor(s2, m, 64); }
  const float r = rsqrtf(s1 / D + eps);
  const float coef = r * r * r * s2 / D;
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = r * gv[k] * wv[k] - xv[k] * coef;
  st8<T>(g, o);
}

template <typename T>
__global__ void embed_gather_k(const T* __restrict__ table, const int64_t* __restrict__ ids, T* __restrict__ out,
                               long long n8, int D, int vocab) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const int dv = D / 8;
  const long long row = i / dv;
  const int c = (int)(i % dv) * 8;
  long long id = ids[row];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  float v[8];
  ld8<T>(table + id * D + c, v);
  st8<T>(out + row * D + c, v);
}

// ---- merge: "last writer wins" reproduces the reference's sequential python loop exactly ----
// owner[b*T + t] = largest audio index i_a whose [start, start+len) range covers row t of batch b.
__global__ void merge_owner_k(int32_t* __restrict__ owner, int32_t* __restrict__ item_batch,
                              const int64_t* __restrict__ audio_batch_size, const int64_t* __restrict__ start,
                              const int32_t* __restrict__ len, int B, int n_items, int T, int Na) {
  __shared__ int ib[1024];
  if (threadIdx.x == 0) {
    int a = 0;
    for (int b = 0; b < B; ++b) {
      const int cnt = (int)audio_batch_size[b];
      for (int k = 0; k < cnt && a < n_items; ++k, ++a) { ib[a & 1023] = b; item_batch[a] = b; }
    }
    for (; a < n_items; ++a) item_batch[a] = -1;  // more audio items than audio_batch_size accounts for
  }
  __syncthreads();
  for (long long i = threadIdx.x; i < (long long)n_items * Na; i += blockDim.x) {
    const int a = (int)(i / Na), j = (int)(i % Na);
    const int b = item_batch[a];
    if (b < 0) continue;
    const int l = min((int)len[a], Na);
    const long long t = start[a] + j;
    if (j < l && t >= 0 && t < T) atomicMax(owner + (long long)b * T + t, a);
  }
}

template <typename T>
__global__ void merge_copy_k(T* __restrict__ embeds, const T* __restrict__ audio, const int32_t* __restrict__ owner,
                             const int32_t* __restrict__ item_batch, const int64_t* __restrict__ start,
                             const int32_t* __restrict__ len, long long n8, int Tlen, int D, int Na, int bwd,
                             T* __restrict__ daudio) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const int dv = D / 8;
  const long long r = i / dv;  // audio row = a*Na + j
  const int c = (int)(i % dv) * 8;
  const int a = (int)(r / Na), j = (int)(r % Na);
  const int b = item_batch[a];
  const int l = min((int)len[a], Na);
  const long long t = start[a] + j;
  const bool live = b >= 0 && j < l && t >= 0 && t < Tlen && owner[(long long)b * Tlen + t] == a;
  float v[8];
  if (!bwd) {
    if (!live) return;
    ld8<T>(audio + r * D + c, v);
    st8<T>(embeds + ((long long)b * Tlen + t) * D + c, v);
  } else {
    if (live) ld8<T>(embeds + ((long long)b * Tlen + t) * D + c, v);
    else {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = 0.f;
    }
    st8<T>(daudio + r * D + c, v);
  }
}

// ---- batched 2-D transpose through LDS: out[z][c, r] = in[z][r, c]; the padded region r in [rows, ld_out)
// is written as zeros.  Two-level batch index z = zo * nzi + zi (e.g. batch x heads).
template <typename T>
__global__ void transpose_k(const T* __restrict__ in, T* __restrict__ out, int rows, int cols, int ld_in, int ld_out,
                            int nzi, long long s_in_o, long long s_in_i, long long s_out) {
  __shared__ T tile[64][65];
  const int zo = blockIdx.z / nzi, zi = blockIdx.z % nzi;
  const T* ib = in + zo * s_in_o + zi * s_in_i;
  T* ob = out + blockIdx.z * s_out;
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 256 threads: 4 rows per pass
  for (int rr = ty; rr < 64; rr += 4) {
    const int r = r0 + rr, c = c0 + tx;
    tile[rr][tx] = (r < rows && c < cols) ? ib[(long long)r * ld_in + c] : (T)0;
  }
  __syncthreads();
  for (int cc = ty; cc < 64; cc += 4) {
    const int c = c0 + cc, r = r0 + tx;
    if (c < cols && r < ld_out) ob[(long long)c * ld_out + r] = tile[tx][cc];
  }
}

// bf16, 16-byte global access on both sides (ld_in, ld_out, cols and all strides multiples of 8, 16-byte aligned bases):
// 2 loads + 2 stores per thread for a 64 x 64 tile instead of 16 + 16 two-byte ones; the transposition itself is done
// with 2-byte LDS accesses on a 65-element row stride (conflict-free for both the row-wise writes and the column gathers).
// NT: non-temporal loads and stores - a stream of weights that is transposed once per step (llm_wt_stream) and far exceeds the
// caches should not evict the GEMM operands that live there
template <bool NT>
__global__ void transpose_bf16_v8_k(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int rows, int cols, int ld_in,
                                    int ld_out, int nzi, long long s_in_o, long long s_in_i, long long s_out) {
  __shared__ bf16_t tile[64][65];
  const int zo = blockIdx.z / nzi, zi = blockIdx.z % nzi;
  const bf16_t* ib = in + zo * s_in_o + zi * s_in_i;
  bf16_t* ob = out + blockIdx.z * s_out;
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int idx = threadIdx.x + k * 256, rr = idx >> 3, ch = idx & 7;
    const int r = r0 + rr, c = c0 + ch * 8;
    u16x8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (r < rows && c < cols) {
      const u16x8_t* src = reinterpret_cast<const u16x8_t*>(ib + (long long)r * ld_in + c);
      v = NT ? __builtin_nontemporal_load(src) : *src;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) tile[rr][ch * 8 + e] = v[e];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int idx = threadIdx.x + k * 256, cc = idx >> 3, r8 = idx & 7;
    const int c = c0 + cc, r = r0 + r8 * 8;
    if (c < cols && r < ld_out) {
      u16x8_t v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tile[r8 * 8 + e][cc];
      u16x8_t* dst = reinterpret_cast<u16x8_t*>(ob + (long long)c * ld_out + r);
      if (NT) __builtin_nontemporal_store(v, dst);
      else *dst = v;
    }
  }
}

// conv1 im2col: out[(b*F + t), k*n_mels + c] = mel[b, c, t + k - 1]  (zero outside [0, F)), cols padded to Kp
template <typename T, typename TIN>
__global__ void im2col_conv1_k(const TIN* __restrict__ mel, T* __restrict__ out, int n_mels, int F, int F_stride,
                               int Kp) {
  extern __shared__ float sm[];  // [n_mels][66]
  const int b = blockIdx.y, t0 = blockIdx.x * 64;
  const TIN* mb = mel + (long long)b * n_mels * F_stride;
  for (int i = threadIdx.x; i < n_mels * 66; i += blockDim.x) {
    const int c = i / 66, tt = t0 - 1 + (i % 66);
    float v = 0.f;
    if (tt >= 0 && tt < F) {
      if constexpr (sizeof(TIN) == 4) v = rnd<T>((float)mb[(long long)c * F_stride + tt]);
      else v = ldf<T>((const T*)mb + (long long)c * F_stride + tt);
    }
    sm[i] = v;
  }
  __syncthreads();
  const int tmax = min(64, F - t0);
  for (long long i = threadIdx.x; i < (long long)tmax * Kp; i += blockDim.x) {
    const int tl = (int)(i / Kp), col = (int)(i % Kp);
    float v = 0.f;
    if (col < 3 * n_mels) {
      const int k = col / n_mels, c = col % n_mels;
      v = sm[c * 66 + tl + k];
    }
    stf<T>(out + ((long long)b * F + t0 + tl) * Kp + col, v);
  }
}

template <typename T>
__global__ void add_k(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o, long long n8) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  float x[8], y[8];
  ld8<T>(a + i * 8, x);
  ld8<T>(b + i * 8, y);
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] += y[k];
  st8<T>(o + i * 8, x);
}

template <typename T>
__global__ void cast_from_f32_k(const float* __restrict__ in, T* __restrict__ out, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) stf<T>(out + i, in[i]);
}

// GELU as a separate pass (encoder LoRA training keeps the pre-activation for the backward pass).  Forward = the
// arithmetic of the fused GEMM epilogue (bf16: gelu_fast + round; f32: erff); backward = the derivative (bf16: gelu_fast_grad,
// the arithmetic of the act == 3 GEMM epilogue; f32: erff / expf).
template <typename T>
__global__ void gelu_fwd_k(const T* __restrict__ pre, T* __restrict__ out, long long n8) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  float v[8];
  ld8<T>(pre + i * 8, v);
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = sizeof(T) == 2 ? gelu_fast(v[k]) : gelu_erf(v[k]);
  st8<T>(out + i * 8, v);
}
template <typename T>
__global__ void gelu_bwd_k(const T* __restrict__ dout, const T* __restrict__ pre, T* __restrict__ din, long long n8) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  float x[8], g[8];
  ld8<T>(pre + i * 8, x);
  ld8<T>(dout + i * 8, g);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (sizeof(T) == 2) { g[k] *= gelu_fast_grad(x[k]); continue; }     // bf16: the arithmetic of the fused epilogue (GemmDesc::act == 3)
    const float cdf = 0.5f * (1.0f + erff(x[k] * 0.70710678118654752440f));
    const float pdf = 0.3989422804014327f * expf(-0.5f * x[k] * x[k]);
    g[k] *= cdf + x[k] * pdf;
  }
  st8<T>(din + i * 8, g);
}
// rows: compaction list of sup_rows (count at rows[n]).  GATHER: dst[c] = src[rows[c]];  SCATTER: dst[rows[c]] = src[c].
template <typename T, bool SCATTER>
__global__ void move_rows_k(const T* __restrict__ src, const int32_t* __restrict__ rows, long long n, T* __restrict__ dst, int D,
                            int first) {
  const long long c = blockIdx.x + first;
  if (c >= rows[n]) return;
  const long long r = rows[c];
  const T* s = src + (SCATTER ? c - first : r) * D;
  T* d = dst + (SCATTER ? r : c) * D;
  for (int k = threadIdx.x * 8; k < D; k += blockDim.x * 8) {
    float v[8];
    ld8<T>(s + k, v);
    st8<T>(d + k, v);
  }
}

template <typename T>
__global__ void splitk_reduce_scatter_k(const float* __restrict__ partial, int nsplit, int cap, const int32_t* __restrict__ rows,
                                        long long n, T* __restrict__ dst, int D) {
  const int c = blockIdx.x;
  if (c >= rows[n] || c >= cap) return;
  T* d = dst + (long long)rows[c] * D;
  for (int k = threadIdx.x * 4; k < D; k += blockDim.x * 4) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < nsplit; ++z) {                 // fixed order: deterministic
      const float4 v = *reinterpret_cast<const float4*>(partial + ((long long)z * cap + c) * D + k);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    stf<T>(d + k, a.x); stf<T>(d + k + 1, a.y); stf<T>(d + k + 2, a.z); stf<T>(d + k + 3, a.w);
  }
}

inline int grid1d(long long n, int th) { return (int)((n + th - 1) / th); }

}  // namespace

#define DISPATCH_T(dtype, KERNEL, grid, block, shmem, st, ...)                                        \
  do {                                                                                                \
    if ((dtype) == uvx::DT_BF16) hipLaunchKernelGGL(KERNEL<bf16_t>, grid, block, shmem, st, __VA_ARGS__); \
    else hipLaunchKernelGGL(KERNEL<float>, grid, block, shmem, st, __VA_ARGS__);                      \
    UVX_LAUNCH_CHECK();                                                                               \
  } while (0)

namespace uvx {

int swiglu_fwd(hipStream_t st, int dtype, const void* in, void* out, int rows, int half, int gate_first, int act, const int32_t* rows_dev) {
  UVX_CHECK(half % 8 == 0 && (gate_first != 2 || half % 16 == 0), UVX_ERR_SHAPE, "swiglu: half=%d must be a multiple of 8 (16 when interleaved)", half);
  UVX_CHECK(act >= 0 && act <= 2, UVX_ERR_INVALID, "swiglu: unknown activation %d", act);
  const long long n8 = (long long)rows * half / 8;
  if (n8 == 0) return UVX_OK;
#define L(T, A) hipLaunchKernelGGL((swiglu_fwd_k<T, A>), dim3(grid1d(n8, 256)), dim3(256), 0, st, (const T*)in, (T*)out, n8, half, gate_first, rows_dev)
  if (dtype == DT_BF16) { if (act == 2) L(bf16_t, 2); else if (act) L(bf16_t, 1); else L(bf16_t, 0); }
  else { if (act == 2) L(float, 2); else if (act) L(float, 1); else L(float, 0); }
#undef L
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

int swiglu_bwd(hipStream_t st, int dtype, const void* dout, const void* in, void* din, int rows, int half,
               int gate_first, int act, const int32_t* rows_dev, RowSkip in_map) {
  UVX_CHECK(half % 8 == 0, UVX_ERR_SHAPE, "swiglu_bwd: half=%d must be a multiple of 8", half);
  UVX_CHECK(act >= 0 && act <= 2, UVX_ERR_INVALID, "swiglu_bwd: unknown activation %d", act);
  const long long n8 = (long long)rows * half / 8;
  if (n8 == 0) return UVX_OK;
#define L(T, A) hipLaunchKernelGGL((swiglu_bwd_k<T, A>), dim3(grid1d(n8, 256)), dim3(256), 0, st, (const T*)dout, (const T*)in, (T*)din, n8, half, gate_first, rows_dev, in_map)
  if (dtype == DT_BF16) { if (act == 2) L(bf16_t, 2); else if (act) L(bf16_t, 1); else L(bf16_t, 0); }
  else { if (act == 2) L(float, 2); else if (act) L(float, 1); else L(float, 0); }
#undef L
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

}  // namespace uvx
namespace {
using namespace uvx;
// One half (gate: which = 0, up: which = 1) of a gate|up tensor [M, 2 I] in the GEMM's interleaved layout (16-column gate / up blocks alternating:
// column c of the half sits at (c / 16) * 32 + 16 which + c % 16) <-> a contiguous [M, I] tensor.  ADD: gu = round(gu + src) (the accumulate rounding
// of lora_up); else dst = gu.  For the MLP adapters of the LLM (ABI 18), whose rank-r kernels work on contiguous columns.
template <typename T, bool ADD>
__global__ void gu_half_k(T* __restrict__ gu, T* __restrict__ flat, long long n8, int I, int which) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const int cv = I / 8;
  const long long m = i / cv;
  const int c = (int)(i % cv) * 8;
  T* g = gu + m * 2 * I + (c / 16) * 32 + 16 * which + (c % 16);
  float a[8], b[8];
  ld8<T>(g, a);
  if (ADD) {
    ld8<T>(flat + m * I + c, b);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = rnd<T>(a[k] + b[k]);
    st8<T>(g, a);
  } else {
    st8<T>(flat + m * I + c, a);
  }
}
}  // namespace
namespace uvx {
int gu_half(hipStream_t st, int dtype, void* gu, void* flat, long long M, int I, int which, int add) {
  UVX_CHECK(I % 16 == 0 && (which == 0 || which == 1), UVX_ERR_SHAPE, "gu_half: I=%d which=%d", I, which);
  const long long n8 = M * (I / 8);
  if (n8 == 0) return UVX_OK;
#define L(T, A) hipLaunchKernelGGL((gu_half_k<T, A>), dim3(grid1d(n8, 256)), dim3(256), 0, st, (T*)gu, (T*)flat, n8, I, which)
  if (dtype == DT_BF16) { if (add) L(bf16_t, true); else L(bf16_t, false); }
  else { if (add) L(float, true); else L(float, false); }
#undef L
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

}  // namespace uvx
namespace {
__global__ void compact_row_list_k(const int32_t* __restrict__ rows, int32_t* __restrict__ out, int n, int T, int skip) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  if (i == n) { out[n] = rows[n]; return; }
  if (i >= rows[n]) return;
  const int r = rows[i], b = r / T, t = r % T;
  out[i] = t >= skip ? r - (b + 1) * skip : n - 1;
}
template <typename T>
__global__ void take_rows_from_k(const T* __restrict__ src, T* __restrict__ dst, long long n8, int cols, uvx::RowSkip map) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const int cv = cols / 8;
  const long long r = i / cv;
  const int c = (int)(i % cv) * 8;
  float v[8];
  ld8<T>(src + (r + (r / map.tc + 1) * map.skip) * cols + c, v);
  st8<T>(dst + r * cols + c, v);
}
}  // namespace
namespace uvx {
int take_rows_from(hipStream_t st, int dtype, const void* src, void* dst, int rows, int cols, RowSkip map) {
  UVX_CHECK(cols % 8 == 0 && map.tc > 0 && src != dst, UVX_ERR_SHAPE, "take_rows_from: cols=%d tc=%d", cols, map.tc);
  const long long n8 = (long long)rows * (cols / 8);
  if (n8 == 0) return UVX_OK;
  if (dtype == DT_BF16) hipLaunchKernelGGL(take_rows_from_k<bf16_t>, dim3(grid1d(n8, 256)), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, n8, cols, map);
  else hipLaunchKernelGGL(take_rows_from_k<float>, dim3(grid1d(n8, 256)), dim3(256), 0, st, (const float*)src, (float*)dst, n8, cols, map);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}
int compact_row_list(hipStream_t st, const int32_t* rows, int32_t* out, int n, int T, int skip) {
  hipLaunchKernelGGL(compact_row_list_k, dim3((n + 256) / 256), dim3(256), 0, st, rows, out, n, T, skip);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

int scale_inplace(hipStream_t st, int dtype, void* x, long long n, float s) {
  UVX_CHECK(n % 8 == 0, UVX_ERR_SHAPE, "scale: n=%lld must be a multiple of 8", n);
  if (n == 0) return UVX_OK;
  if (dtype == DT_BF16) hipLaunchKernelGGL(scale_k<bf16_t>, dim3(grid1d(n / 8, 256)), dim3(256), 0, st, (bf16_t*)x, n / 8, s);
  else hipLaunchKernelGGL(scale_k<float>, dim3(grid1d(n / 8, 256)), dim3(256), 0, st, (float*)x, n / 8, s);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

int rope_inplace(hipStream_t st, int dtype, void* x, const float* cos_sin, const int32_t* pos, int rows, int T,
                 int n_heads_rot, int head_dim, int ld, int inverse) {
  UVX_CHECK(head_dim % 16 == 0 && ld % 8 == 0, UVX_ERR_SHAPE, "rope: head_dim=%d ld=%d unsupported", head_dim, ld);
  const long long n = (long long)rows * n_heads_rot * (head_dim / 16);
  if (n == 0) return UVX_OK;
  const float sgn = inverse ? -1.f : 1.f;
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(rope_k<bf16_t>, dim3(grid1d(n, 256)), dim3(256), 0, st, (bf16_t*)x, cos_sin, pos, n, T, n_heads_rot, head_dim, ld, sgn);
  else
    hipLaunchKernelGGL(rope_k<float>, dim3(grid1d(n, 256)), dim3(256), 0, st, (float*)x, cos_sin, pos, n, T, n_heads_rot, head_dim, ld, sgn);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

int qk_norm_rope(hipStream_t st, int dtype, void* qkv, const void* wq, const void* wk, void* raw, const float* cos_sin,
                 const int32_t* pos, int rows, int T, int Hq, int Hkv, int head_dim, int ld, float eps, int flavor) {
  UVX_CHECK((head_dim == 64 || head_dim == 128 || head_dim == 256) && ld % 8 == 0, UVX_ERR_SHAPE,
            "qk_norm_rope: head_dim=%d ld=%d unsupported (64, 128 or 256; row stride a multiple of 8)", head_dim, ld);
  UVX_CHECK(qkv && wq && wk && cos_sin, UVX_ERR_INVALID, "qk_norm_rope: null argument");
  const long long n = (long long)rows * (Hq + Hkv) * (head_dim / 16);
  if (n == 0) return UVX_OK;
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(qk_norm_rope_k<bf16_t>, dim3(grid1d(n, 256)), dim3(256), 0, st, (bf16_t*)qkv, (const bf16_t*)wq, (const bf16_t*)wk,
                       (bf16_t*)raw, cos_sin, pos, n, T, Hq, Hkv, head_dim, ld, eps, flavor);
  else
    hipLaunchKernelGGL(qk_norm_rope_k<float>, dim3(grid1d(n, 256)), dim3(256), 0, st, (float*)qkv, (const float*)wq, (const float*)wk,
                       (float*)raw, cos_sin, pos, n, T, Hq, Hkv, head_dim, ld, eps, flavor);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

int qk_norm_bwd(hipStream_t st, int dtype, void* d_qkv, const void* raw, const void* wq, const void* wk, int rows, int Hq, int Hkv,
                int head_dim, int ld, float eps, int flavor, RowSkip raw_map) {
  UVX_CHECK((head_dim == 64 || head_dim == 128 || head_dim == 256) && ld % 8 == 0, UVX_ERR_SHAPE,
            "qk_norm_bwd: head_dim=%d ld=%d unsupported", head_dim, ld);
  UVX_CHECK(d_qkv && raw && wq && wk, UVX_ERR_INVALID, "qk_norm_bwd: null argument");
  const long long n = (long long)rows * (Hq + Hkv) * (head_dim / 8);
  if (n == 0) return UVX_OK;
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(qk_norm_bwd_k<bf16_t>, dim3(grid1d(n, 256)), dim3(256), 0, st, (bf16_t*)d_qkv, (const bf16_t*)raw, (const bf16_t*)wq,
                       (const bf16_t*)wk, n, Hq, Hkv, head_dim, ld, eps, flavor, raw_map);
  else
    hipLaunchKernelGGL(qk_norm_bwd_k<float>, dim3(grid1d(n, 256)), dim3(256), 0, st, (float*)d_qkv, (const float*)raw, (const float*)wq,
                       (const float*)wk, n, Hq, Hkv, head_dim, ld, eps, flavor, raw_map);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

int embed_gather(hipStream_t st, int dtype, const void* table, const int64_t* ids, void* out, int rows, int D,
                 int vocab) {
  UVX_CHECK(D % 8 == 0, UVX_ERR_SHAPE, "embed_gather: D=%d must be a multiple of 8", D);
  const long long n8 = (long long)rows * D / 8;
  if (n8 == 0) return UVX_OK;
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(embed_gather_k<bf16_t>, dim3(grid1d(n8, 256)), dim3(256), 0, st, (const bf16_t*)table, ids, (bf16_t*)out, n8, D, vocab);
  else
    hipLaunchKernelGGL(embed_gather_k<float>, dim3(grid1d(n8, 256)), dim3(256), 0, st, (const float*)table, ids, (float*)out, n8, D, vocab);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

int merge_owner(hipStream_t st, int32_t* owner, int32_t* item_batch, const int64_t* audio_batch_size,
                const int64_t* start, const int32_t* len, int B, int n_items, int T, int Na) {
  UVX_CHECK(n_items <= 1024, UVX_ERR_SHAPE, "merge: more than 1024 audio items per step (%d)", n_items);
  UVX_HIP(hipMemsetAsync(owner, 0xff, sizeof(int32_t) * (size_t)B * T, st));
  if (n_items == 0) return UVX_OK;
  hipLaunchKernelGGL(merge_owner_k, dim3(1), dim3(1024), 0, st, owner, item_batch, audio_batch_size, start, len, B,
                     n_items, T, Na);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

int merge_audio(hipStream_t st, int dtype, void* embeds, const void* audio, void* daudio, const int32_t* owner,
                const int32_t* item_batch, const int64_t* start, const int32_t* len, int n_items, int T, int D,
                int Na, int bwd) {
  UVX_CHECK(D % 8 == 0, UVX_ERR_SHAPE, "merge: D=%d must be a multiple of 8", D);
  const long long n8 = (long long)n_items * Na * D / 8;
  if (n8 == 0) return UVX_OK;
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(merge_copy_k<bf16_t>, dim3(grid1d(n8, 256)), dim3(256), 0, st, (bf16_t*)embeds, (const bf16_t*)audio, owner, item_batch, start, len, n8, T, D, Na, bwd, (bf16_t*)daudio);
  else
    hipLaunchKernelGGL(merge_copy_k<float>, dim3(grid1d(n8, 256)), dim3(256), 0, st, (float*)embeds, (const float*)audio, owner, item_batch, start, len, n8, T, D, Na, bwd, (float*)daudio);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

static int transpose_launch(hipStream_t st, int dtype, const void* in, void* out, int rows, int cols, int ld_in,
                            int ld_out, int nzo, int nzi, long long s_in_o, long long s_in_i, long long s_out, bool nt = false) {
  if (rows == 0 || cols == 0 || nzo * nzi == 0) return UVX_OK;
  // the zero padded region out[:, rows..ld_out) is written too (the GEMM K dimension must be 64-aligned)
  dim3 grid(cdiv(ld_out, 64), cdiv(cols, 64), nzo * nzi);
  const bool v8 = dtype == DT_BF16 && ((ld_in | ld_out | cols) & 7) == 0 && ((s_in_o | s_in_i | s_out) & 7) == 0 &&
                  (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
  if (v8 && nt)
    hipLaunchKernelGGL(transpose_bf16_v8_k<true>, grid, dim3(256), 0, st, (const bf16_t*)in, (bf16_t*)out, rows, cols, ld_in, ld_out, nzi, s_in_o, s_in_i, s_out);
  else if (v8)
    hipLaunchKernelGGL(transpose_bf16_v8_k<false>, grid, dim3(256), 0, st, (const bf16_t*)in, (bf16_t*)out, rows, cols, ld_in, ld_out, nzi, s_in_o, s_in_i, s_out);
  else if (dtype == DT_BF16)
    hipLaunchKernelGGL(transpose_k<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)in, (bf16_t*)out, rows, cols, ld_in, ld_out, nzi, s_in_o, s_in_i, s_out);
  else
    hipLaunchKernelGGL(transpose_k<float>, grid, dim3(256), 0, st, (const float*)in, (float*)out, rows, cols, ld_in, ld_out, nzi, s_in_o, s_in_i, s_out);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

int transpose2d(hipStream_t st, int dtype, const void* in, void* out, int rows, int cols, int ld_in, int ld_out,
                int batch, long long s_in, long long s_out) {
  return transpose_launch(st, dtype, in, out, rows, cols, ld_in, ld_out, batch, 1, s_in, 0, s_out);
}

int transpose2d_streaming(hipStream_t st, int dtype, const void* in, void* out, int rows, int cols, int ld_in, int ld_out) {
  return transpose_launch(st, dtype, in, out, rows, cols, ld_in, ld_out, 1, 1, 0, 0, 0, g_options[5] == 0);   // (probe option 5 = 1: plain accesses, for A/B)
}

int heads_transpose(hipStream_t st, int dtype, const void* in, void* out, int B, int T, int Tp, int H, int D, int ld) {
  // [B, T, H, D] (token stride ld) -> [B, H, D, Tp] in ONE launch: z = b * H + h
  return transpose_launch(st, dtype, in, out, T, D, ld, Tp, B, H, (long long)T * ld, D, (long long)D * Tp);
}

int im2col_conv1(hipStream_t st, int dtype, const void* mel, int mel_is_f32, void* out, int B, int n_mels, int F,
                 int F_stride, int Kp) {
  UVX_CHECK(Kp >= 3 * n_mels, UVX_ERR_SHAPE, "im2col: Kp=%d < 3*n_mels", Kp);
  if (B == 0 || F == 0) return UVX_OK;
  dim3 grid(cdiv(F, 64), B);
  const size_t sh = sizeof(float) * n_mels * 66;
  if (dtype == DT_BF16) {
    if (mel_is_f32) hipLaunchKernelGGL((im2col_conv1_k<bf16_t, float>), grid, dim3(256), sh, st, (const float*)mel, (bf16_t*)out, n_mels, F, F_stride, Kp);
    else hipLaunchKernelGGL((im2col_conv1_k<bf16_t, bf16_t>), grid, dim3(256), sh, st, (const bf16_t*)mel, (bf16_t*)out, n_mels, F, F_stride, Kp);
  } else {
    UVX_CHECK(mel_is_f32, UVX_ERR_INVALID, "im2col: f32 mode needs f32 mel");
    hipLaunchKernelGGL((im2col_conv1_k<float, float>), grid, dim3(256), sh, st, (const float*)mel, (float*)out, n_mels, F, F_stride, Kp);
  }
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

int add_rows(hipStream_t st, int dtype, const void* a, const void* b, void* out, long long n) {
  UVX_CHECK(n % 8 == 0, UVX_ERR_SHAPE, "add: n must be a multiple of 8");
  if (n == 0) return UVX_OK;
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(add_k<bf16_t>, dim3(grid1d(n / 8, 256)), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n / 8);
  else
    hipLaunchKernelGGL(add_k<float>, dim3(grid1d(n / 8, 256)), dim3(256), 0, st, (const float*)a, (const float*)b, (float*)out, n / 8);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

int cast_f32_to(hipStream_t st, int dtype, const float* in, void* out, long long n) {
  if (n == 0) return UVX_OK;
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(cast_from_f32_k<bf16_t>, dim3(grid1d(n, 256)), dim3(256), 0, st, in, (bf16_t*)out, n);
  else
    hipLaunchKernelGGL(cast_from_f32_k<float>, dim3(grid1d(n, 256)), dim3(256), 0, st, in, (float*)out, n);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

int fill_zero(hipStream_t st, void* p, long long bytes) {
  if (bytes > 0) UVX_HIP(hipMemsetAsync(p, 0, (size_t)bytes, st));
  return UVX_OK;
}

int gather_rows(hipStream_t st, int dtype, const void* src, const int32_t* rows, long long n, void* dst, int D) {
  UVX_CHECK(D % 8 == 0, UVX_ERR_SHAPE, "gather_rows: D=%d must be a multiple of 8", D);
  if (n == 0) return UVX_OK;
  if (dtype == DT_BF16) hipLaunchKernelGGL((move_rows_k<bf16_t, false>), dim3(n), dim3(256), 0, st, (const bf16_t*)src, rows, n, (bf16_t*)dst, D, 0);
  else hipLaunchKernelGGL((move_rows_k<float, false>), dim3(n), dim3(256), 0, st, (const float*)src, rows, n, (float*)dst, D, 0);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

int scatter_rows(hipStream_t st, int dtype, const void* src, const int32_t* rows, long long n, void* dst, int D, int first) {
  UVX_CHECK(D % 8 == 0, UVX_ERR_SHAPE, "scatter_rows: D=%d must be a multiple of 8", D);
  if (n - first <= 0) return UVX_OK;
  if (dtype == DT_BF16) hipLaunchKernelGGL((move_rows_k<bf16_t, true>), dim3(n - first), dim3(256), 0, st, (const bf16_t*)src, rows, n, (bf16_t*)dst, D, first);
  else hipLaunchKernelGGL((move_rows_k<float, true>), dim3(n - first), dim3(256), 0, st, (const float*)src, rows, n, (float*)dst, D, first);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

int splitk_reduce_scatter(hipStream_t st, int dtype, const float* partial, int nsplit, int cap, const int32_t* rows,
                          long long n, void* dst, int D) {
  UVX_CHECK(D % 4 == 0, UVX_ERR_SHAPE, "splitk_reduce: D=%d must be a multiple of 4", D);
  if (cap <= 0) return UVX_OK;
  if (dtype == DT_BF16) hipLaunchKernelGGL(splitk_reduce_scatter_k<bf16_t>, dim3(cap), dim3(256), 0, st, partial, nsplit, cap, rows, n, (bf16_t*)dst, D);
  else hipLaunchKernelGGL(splitk_reduce_scatter_k<float>, dim3(cap), dim3(256), 0, st, partial, nsplit, cap, rows, n, (float*)dst, D);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

int act_fwd(hipStream_t st, int dtype, const void* in, void* out, long long n, int act) {
  UVX_CHECK(n % 8 == 0, UVX_ERR_SHAPE, "act_fwd: element count must be a multiple of 8");
  UVX_CHECK(act >= 0 && act <= 3, UVX_ERR_INVALID, "act_fwd: unknown activation %d", act);
  if (n == 0) return UVX_OK;
#define L(T, A) hipLaunchKernelGGL((act_fwd_k<T, A>), dim3(grid1d(n / 8, 256)), dim3(256), 0, st, (const T*)in, (T*)out, n / 8)
  if (dtype == DT_BF16) { if (act == 0) L(bf16_t, 0); else if (act == 1) L(bf16_t, 1); else if (act == 2) L(bf16_t, 2); else L(bf16_t, 3); }
  else { if (act == 0) L(float, 0); else if (act == 1) L(float, 1); else if (act == 2) L(float, 2); else L(float, 3); }
#undef L
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}
int act_bwd(hipStream_t st, int dtype, const void* dout, const void* in, void* din, long long n, int act) {
  UVX_CHECK(n % 8 == 0, UVX_ERR_SHAPE, "act_bwd: element count must be a multiple of 8");
  UVX_CHECK(act >= 0 && act <= 3, UVX_ERR_INVALID, "act_bwd: unknown activation %d", act);
  if (n == 0) return UVX_OK;
#define L(T, A) hipLaunchKernelGGL((act_bwd_k<T, A>), dim3(grid1d(n / 8, 256)), dim3(256), 0, st, (const T*)dout, (const T*)in, (T*)din, n / 8)
  if (dtype == DT_BF16) { if (act == 0) L(bf16_t, 0); else if (act == 1) L(bf16_t, 1); else if (act == 2) L(bf16_t, 2); else L(bf16_t, 3); }
  else { if (act == 0) L(float, 0); else if (act == 1) L(float, 1); else if (act == 2) L(float, 2); else L(float, 3); }
#undef L
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

int gelu_fwd(hipStream_t st, int dtype, const void* pre, void* out, long long n) {
  UVX_CHECK(n % 8 == 0, UVX_ERR_SHAPE, "gelu: element count must be a multiple of 8");
  if (n == 0) return UVX_OK;
  if (dtype == DT_BF16) hipLaunchKernelGGL(gelu_fwd_k<bf16_t>, dim3(grid1d(n / 8, 256)), dim3(256), 0, st, (const bf16_t*)pre, (bf16_t*)out, n / 8);
  else hipLaunchKernelGGL(gelu_fwd_k<float>, dim3(grid1d(n / 8, 256)), dim3(256), 0, st, (const float*)pre, (float*)out, n / 8);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

int gelu_bwd(hipStream_t st, int dtype, const void* dout, const void* pre, void* din, long long n) {
  UVX_CHECK(n % 8 == 0, UVX_ERR_SHAPE, "gelu_bwd: element count must be a multiple of 8");
  if (n == 0) return UVX_OK;
  if (dtype == DT_BF16) hipLaunchKernelGGL(gelu_bwd_k<bf16_t>, dim3(grid1d(n / 8, 256)), dim3(256), 0, st, (const bf16_t*)dout, (const bf16_t*)pre, (bf16_t*)din, n / 8);
  else hipLaunchKernelGGL(gelu_bwd_k<float>, dim3(grid1d(n / 8, 256)), dim3(256), 0, st, (const float*)dout, (const float*)pre, (float*)din, n / 8);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

}  // namespace uvx
