// Alt audio tower (BASELINE.json config 5): [3P] transformers Wav2Vec2Model.forward behind the C ABI - the AutoModel branch
// of UltravoxModel._create_audio_tower (ultravox_model.py:460-467, :476-485) for facebook/wav2vec2-large-960h-style towers
// (GroupNorm after the first conv layer, bias-free convs, post-LN encoder) and the layer-norm (-lv60) family.  Frozen tower: forward only; under
// apply_lora (audio_model_lora_config.r > 0): uvx_wav2vec2_fwd_train / uvx_wav2vec2_bwd at the end of this file.
//
//   input_values [B, L]  (zero-mean / unit-variance waveform; the `input_values` fallback of ultravox_processing.py:308)
//   -> 7 x Conv1d(+GELU): layer 0 (1 -> C, k 10, s 5) as im2col + GEMM, then GroupNorm(C groups) over time + GELU;
//      layers 1..6 (C -> C, k 3/2, s 2) as GEMMs on a STRIDED VIEW of the time-major activations: output frame t reads the
//      contiguous K = k C run starting at row s t (row stride s C) - no im2col copy for the 146 GFLOP/clip conv stack
//   -> LayerNorm(C) -> Linear(C -> d)
//   -> + GELU(grouped Conv1d(d, d, k 128, pad 64, groups 16))  [weight norm folded at pack time]: per group a GEMM over the
//      contiguous K = 128 x d/G run of a zero-padded per-group copy of the hidden states
//   -> LayerNorm -> layers x { x = LN(x + attn(x));  x = LN(x + fc2(gelu(fc1(x)))) }   -> last_hidden_state [B, T, d]
//
// Rounding points are torch's module-by-module bf16 rounding (conv / norm / activation / linear / residual outputs).
#include "common.h"
#include "kernels.h"
#include "../../include/uvx.h"

namespace {

using namespace uvx;

struct Arena {
  char* base;
  size_t cap;
  size_t off = 0;
  Arena(void* b, size_t c) : base((char*)b), cap(c) {}
  void* take(size_t bytes) {
    const size_t a = (off + 255) & ~(size_t)255;
    off = a + bytes;
    return base ? (void*)(base + a) : nullptr;
  }
  bool fits() const { return !base || off <= cap; }
};
inline size_t esz(int dtype) { return dtype == DT_BF16 ? 2 : 4; }
inline int rup(int x, int m) { return (x + m - 1) / m * m; }
inline char* at(const void* p, size_t elems, int dtype) { return (char*)p + elems * esz(dtype); }
#define RC(expr)            \
  do {                      \
    int _rc = (expr);       \
    if (_rc) return _rc;    \
  } while (0)

GemmDesc lin(const void* A, const void* W, void* C, int M, int N, int K) {
  GemmDesc g;
  g.A = A; g.B = W; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N;
  return g;
}

constexpr int K0P = 64;     // the first conv's kernel (10 taps) zero padded to one GEMM K-tile
constexpr int GN_ROWS = 512;  // time rows per GroupNorm partial block

// out[(b T0 + t), k] = round(in[b, s t + k]) for k < k0, 0 for k0 <= k < 64
template <typename T, typename TIN>
__global__ void w2v_im2col0_k(const TIN* __restrict__ in, T* __restrict__ out, long long rows, int T0, int L, int k0, int s0) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (row, 8-column group)
  if (i >= rows * (K0P / 8)) return;
  const long long row = i / (K0P / 8);
  const int c0 = (int)(i % (K0P / 8)) * 8;
  const int b = (int)(row / T0), t = (int)(row % T0);
  const TIN* src = in + (long long)b * L + (long long)t * s0;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = c0 + j;
    v[j] = k < k0 ? (float)src[k] : 0.f;
    if (sizeof(TIN) == 2) v[j] = k < k0 ? bf2f(((const bf16_t*)src)[k]) : 0.f;
  }
  st8<T>(out + row * K0P + c0, v);
}

// GroupNorm(num_groups = C): per (clip, channel) statistics over the T0 frames of a time-major [B, T0, C] tensor.
// pass 1: each block sums GN_ROWS rows for all channels (thread = 8 channels); pass 2: combines the chunks in double.
template <typename T>
__global__ void w2v_gn_partial_k(const T* __restrict__ x, float* __restrict__ part, int T0, int C, int nchunk) {
  const int b = blockIdx.y, ch = blockIdx.x;
  const int c = threadIdx.x * 8;
  if (c >= C) return;
  const int r0 = ch * GN_ROWS, r1 = min(T0, r0 + GN_ROWS);
  const T* p = x + ((long long)b * T0 + r0) * C + c;
  float s[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
  for (int r = r0; r < r1; ++r, p += C) {
    float v[8];
    ld8<T>(p, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] += v[j]; q[j] += v[j] * v[j]; }
  }
  float* o = part + (((long long)b * nchunk + ch) * C + c) * 2;
#pragma unroll
  for (int j = 0; j < 8; ++j) { o[2 * j] = s[j]; o[2 * j + 1] = q[j]; }
}
__global__ void w2v_gn_final_k(const float* __restrict__ part, float* __restrict__ stat, int T0, int C, int nchunk, float eps) {
  const int b = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0., q = 0.;
  for (int ch = 0; ch < nchunk; ++ch) {
    const float* p = part + (((long long)b * nchunk + ch) * C + c) * 2;
    s += p[0]; q += p[1];
  }
  const double mean = s / T0, var = fmax(q / T0 - mean * mean, 0.);   // biased variance, as torch.nn.GroupNorm
  stat[((long long)b * C + c) * 2] = (float)mean;
  stat[((long long)b * C + c) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
}
// x = round(gelu(round((x - mean) rstd gamma + beta)))  in place
template <typename T>
__global__ void w2v_gn_gelu_k(T* __restrict__ x, const float* __restrict__ stat, const T* __restrict__ gamma,
                              const T* __restrict__ beta, long long n8, int T0, int C) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const int cv = C / 8;
  const int c = (int)(i % cv) * 8;
  const int b = (int)(i / cv / T0);
  float v[8], g[8], be[8];
  ld8<T>(x + i * 8, v);
  ld8<T>(gamma + c, g);
  ld8<T>(beta + c, be);
  const float* st = stat + ((long long)b * C + c) * 2;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float y = rnd<T>((v[j] - st[2 * j]) * st[2 * j + 1] * g[j] + be[j]);
    v[j] = sizeof(T) == 2 ? gelu_fast(y) : gelu_erf(y);
  }
  st8<T>(x + i * 8, v);
}

// xg[b][g][r][c] = h[b, r - K/2, g dg + c] for K/2 <= r < K/2 + T, else 0;  r in [0, T + K)
template <typename T>
__global__ void w2v_pos_pack_k(const T* __restrict__ h, T* __restrict__ xg, long long n8, int Tn, int d, int G, int K) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const int dg = d / G, cv = dg / 8, R = Tn + K;
  const int c = (int)(i % cv) * 8;
  long long rest = i / cv;
  const int r = (int)(rest % R); rest /= R;
  const int g = (int)(rest % G);
  const int b = (int)(rest / G);
  float v[8];
  const int t = r - K / 2;
  if (t >= 0 && t < Tn) ld8<T>(h + ((long long)b * Tn + t) * d + g * dg + c, v);
  else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
  }
  st8<T>(xg + i * 8, v);
}
// h = round(h + round(gelu(round(acc + bias))))   (conv output, activation and residual add each rounded once)
template <typename T>
__global__ void w2v_pos_finish_k(T* __restrict__ h, const float* __restrict__ acc, const T* __restrict__ bias, long long n8, int d) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const int c = (int)(i % (d / 8)) * 8;
  float x[8], a[8], bv[8];
  ld8<T>(h + i * 8, x);
  ld8<float>(acc + i * 8, a);
  ld8<T>(bias + c, bv);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float y = rnd<T>(a[j] + bv[j]);
    x[j] += rnd<T>(sizeof(T) == 2 ? gelu_fast(y) : gelu_erf(y));
  }
  st8<T>(h + i * 8, x);
}

inline unsigned g1(long long n, int b) { return (unsigned)((n + b - 1) / b); }

// per-layer stash of the LoRA-training forward (uvx_wav2vec2_fwd_train -> uvx_wav2vec2_bwd).  x_in = the layer's input (pre-LN: the residual
// stream h; post-LN: the normalised x), mid = x_in + attention branch, y2 = LN1(mid) + feed-forward branch (post-LN only: the input of
// final_layer_norm), pre = fc1's pre-activation; t / t2 = the adapters' down-projections [M, 128] (q | k and v | out_proj), b?T = lora_B^T [r, d]
struct W2vLayerStash {
  void *x_in, *qkv, *o, *mid, *y2, *pre, *t, *t2, *bqT, *bkT, *bvT, *boT;
  float* lse;
};
struct W2vWs {
  int T[9];          // frames after conv layer i
  int Tn, Tp, M, nchunk;
  void *im2col, *bufA, *bufB, *x, *n, *qkv, *vt, *o, *o2, *f, *xg;
  float *part, *stat, *acc;
  // training only
  char* slots; size_t slot_bytes; W2vLayerStash ls0;
  void *dx, *d_n, *d_o, *d_f, *d_qkv, *qT, *kT, *doT, *u, *u2;
  float *delta, *wg;
};
W2vLayerStash w2v_layer(const W2vWs& w, int l) {
  W2vLayerStash s = w.ls0;
  const size_t off = w.slot_bytes * l;
  void** ps[] = {&s.x_in, &s.qkv, &s.o, &s.mid, &s.y2, &s.pre, &s.t, &s.t2, &s.bqT, &s.bkT, &s.bvT, &s.boT, (void**)&s.lse};
  for (void** q : ps) if (*q) *q = (char*)*q + off;
  return s;
}

int frames(const uvx_w2v_config_t& c, int L, int* T) {
  int n = L;
  for (int i = 0; i < c.n_conv; ++i) {
    if (n < c.conv_kernel[i]) return -1;
    n = (n - c.conv_kernel[i]) / c.conv_stride[i] + 1;
    T[i] = n;
  }
  return n;
}

W2vWs carve(Arena& a, const uvx_w2v_config_t& c, int B, int L, bool train = false) {
  W2vWs w = {};
  w.Tn = frames(c, L, w.T);
  if (w.Tn <= 0) return w;
  const size_t es = esz(c.dtype);
  const int C = c.conv_dim, d = c.d, dh = d / c.heads;
  w.Tp = rup(w.Tn, 64);
  w.M = B * w.Tn;
  w.nchunk = (w.T[0] + GN_ROWS - 1) / GN_ROWS;
  w.im2col = a.take((size_t)B * w.T[0] * K0P * es);
  w.bufA = a.take((size_t)B * w.T[0] * C * es);
  w.bufB = a.take((size_t)B * (c.n_conv > 1 ? w.T[1] : 1) * C * es);
  w.part = (float*)a.take(sizeof(float) * (size_t)B * w.nchunk * C * 2);
  w.stat = (float*)a.take(sizeof(float) * (size_t)B * C * 2);
  w.x = a.take((size_t)w.M * d * es);
  w.n = a.take((size_t)w.M * d * es);
  w.qkv = a.take((size_t)w.M * 3 * d * es);
  w.vt = a.take((size_t)B * c.heads * dh * w.Tp * es);
  w.o = a.take((size_t)w.M * d * es);
  w.o2 = c.stable_ln ? a.take((size_t)w.M * d * es) : nullptr;      // second residual-stream buffer of the pre-LN layers
  w.f = a.take((size_t)w.M * c.ffn * es);
  w.xg = a.take((size_t)B * c.pos_groups * (w.Tn + c.pos_k) * (d / c.pos_groups) * es);
  w.acc = (float*)a.take(sizeof(float) * (size_t)w.M * d);
  if (train) {
    const size_t M = (size_t)w.M;
    const size_t start = (a.off + 255) & ~(size_t)255;
    a.off = start;
    W2vLayerStash& s = w.ls0;
    s.x_in = a.take(M * d * es); s.qkv = a.take(M * 3 * d * es); s.o = a.take(M * d * es); s.mid = a.take(M * d * es);
    s.y2 = c.stable_ln ? nullptr : a.take(M * d * es);
    s.pre = a.take(M * c.ffn * es); s.t = a.take(M * 128 * es); s.t2 = a.take(M * 128 * es);
    s.bqT = a.take((size_t)64 * d * es); s.bkT = a.take((size_t)64 * d * es); s.bvT = a.take((size_t)64 * d * es); s.boT = a.take((size_t)64 * d * es);
    s.lse = (float*)a.take(sizeof(float) * (size_t)B * c.heads * w.Tn);
    a.off = (a.off + 255) & ~(size_t)255;
    w.slot_bytes = a.off - start;
    w.slots = a.base ? a.base + start : nullptr;
    a.off = start + w.slot_bytes * c.layers;
    w.dx = a.take(M * d * es); w.d_n = a.take(M * d * es); w.d_o = a.take(M * d * es);
    w.d_f = a.take(M * c.ffn * es); w.d_qkv = a.take(M * 3 * d * es);
    const size_t ht = (size_t)B * c.heads * dh * w.Tp * es;
    w.qT = a.take(ht); w.kT = a.take(ht); w.doT = a.take(ht);
    w.u = a.take(M * 128 * es); w.u2 = a.take(M * 128 * es);
    w.delta = (float*)a.take(sizeof(float) * (size_t)B * c.heads * w.Tn);
    w.wg = (float*)a.take(sizeof(float) * (size_t)lora_wgrad_scratch_floats(w.M, d, 64));
  }
  return w;
}

int check(const uvx_w2v_config_t* c) {
  UVX_CHECK(c != nullptr, UVX_ERR_INVALID, "wav2vec2: null config");
  UVX_CHECK(c->dtype == DT_BF16 || c->dtype == DT_F32, UVX_ERR_INVALID, "wav2vec2: bad dtype %d", c->dtype);
  UVX_CHECK(c->n_conv >= 2 && c->n_conv <= 8, UVX_ERR_SHAPE, "wav2vec2: %d conv layers (2..8)", c->n_conv);
  UVX_CHECK(c->conv_dim % 64 == 0 && c->conv_dim <= 8192, UVX_ERR_SHAPE, "wav2vec2: conv_dim %d must be a multiple of 64", c->conv_dim);
  UVX_CHECK(c->conv_kernel[0] <= K0P, UVX_ERR_SHAPE, "wav2vec2: first conv kernel %d > %d", c->conv_kernel[0], K0P);
  UVX_CHECK(!c->feat_norm_layer || c->conv_dim % 8 == 0, UVX_ERR_SHAPE, "wav2vec2: conv_dim %d", c->conv_dim);
  UVX_CHECK(c->d % c->heads == 0 && c->d % c->pos_groups == 0 && (c->d / c->pos_groups) % 8 == 0 &&
                (c->pos_k * (c->d / c->pos_groups)) % 64 == 0,
            UVX_ERR_SHAPE, "wav2vec2: hidden %d / heads %d / positional-conv groups %d, kernel %d unsupported", c->d, c->heads,
            c->pos_groups, c->pos_k);
  return UVX_OK;
}

}  // namespace

extern "C" int32_t uvx_wav2vec2_frames(const uvx_w2v_config_t* cfg, int32_t L) {
  if (!cfg) return -1;
  int T[9];
  return frames(*cfg, L, T);
}

extern "C" size_t uvx_wav2vec2_ws_bytes(const uvx_w2v_config_t* cfg, int32_t B, int32_t L) {
  if (!cfg) return 0;
  Arena a(nullptr, 0);
  carve(a, *cfg, B, L);
  return a.off + 256;
}

// lora != NULL: the LoRA-training forward (apply_lora on the AutoModel tower, ultravox_model.py:460-467, 690-709) - adapters on the attention
// projections and a per-layer stash for uvx_wav2vec2_bwd; the feature encoder, projection and positional conv have no adapter and stay forward-only
static int w2v_forward(void* stream, const uvx_w2v_config_t* cfg, const uvx_w2v_weights_t* w, const uvx_encoder_lora_t* lora, const void* input_values,
                       int32_t values_is_f32, int32_t B, int32_t L, void* out, void* workspace, size_t ws_bytes) {
  RC(check(cfg));
  UVX_CHECK(w && w->layers && input_values && out && workspace, UVX_ERR_INVALID, "wav2vec2_fwd: null argument");
  const uvx_w2v_config_t& c = *cfg;
  hipStream_t st = (hipStream_t)stream;
  const bool train = lora != nullptr;
  if (train) RC(lora_check(lora, c.layers, nullptr, "wav2vec2 LoRA"));
  for (int l = 0; train && l < c.layers; ++l)
    UVX_CHECK(!lora->layers[l].g.a && !lora->layers[l].u.a && !lora->layers[l].d.a, UVX_ERR_UNSUPPORTED,
              "wav2vec2 LoRA: layer %d has a feed-forward adapter (the attention projections q / k / v / out_proj are built for this tower)", l);
  for (int i = 0; i < c.n_conv; ++i) {
    UVX_CHECK(!c.conv_bias || w->conv_b[i], UVX_ERR_INVALID, "wav2vec2_fwd: conv_bias is set but conv layer %d has no bias", i);
    UVX_CHECK(!c.feat_norm_layer || (w->conv_ln_w[i] && w->conv_ln_b[i]), UVX_ERR_INVALID, "wav2vec2_fwd: conv layer %d has no layer norm", i);
  }
  UVX_CHECK(c.feat_norm_layer || (w->gn_w && w->gn_b), UVX_ERR_INVALID, "wav2vec2_fwd: the first conv layer's GroupNorm weights are missing");
  if (B == 0) return UVX_OK;
  Arena a(workspace, ws_bytes);
  W2vWs s = carve(a, c, B, L, train);
  UVX_CHECK(s.Tn > 0, UVX_ERR_SHAPE, "wav2vec2_fwd: %d samples are shorter than the conv stack's receptive field", L);
  UVX_CHECK(a.fits(), UVX_ERR_WORKSPACE, "wav2vec2_fwd: workspace %zu < %zu bytes", ws_bytes, a.off);
  const int dt = c.dtype, C = c.conv_dim, d = c.d, dh = d / c.heads, M = s.M, Tn = s.Tn;
  UVX_CHECK(dt == DT_BF16 || values_is_f32, UVX_ERR_INVALID, "wav2vec2_fwd: f32 mode needs f32 input_values");

  // ---- feature encoder ----
  {
    const long long rows = (long long)B * s.T[0], n = rows * (K0P / 8);
    if (dt == DT_BF16) {
      if (values_is_f32) hipLaunchKernelGGL((w2v_im2col0_k<bf16_t, float>), dim3(g1(n, 256)), dim3(256), 0, st, (const float*)input_values, (bf16_t*)s.im2col, rows, s.T[0], L, c.conv_kernel[0], c.conv_stride[0]);
      else hipLaunchKernelGGL((w2v_im2col0_k<bf16_t, bf16_t>), dim3(g1(n, 256)), dim3(256), 0, st, (const bf16_t*)input_values, (bf16_t*)s.im2col, rows, s.T[0], L, c.conv_kernel[0], c.conv_stride[0]);
    } else {
      hipLaunchKernelGGL((w2v_im2col0_k<float, float>), dim3(g1(n, 256)), dim3(256), 0, st, (const float*)input_values, (float*)s.im2col, rows, s.T[0], L, c.conv_kernel[0], c.conv_stride[0]);
    }
    UVX_LAUNCH_CHECK();
    {
      GemmDesc g0 = lin(s.im2col, w->conv0_w, s.bufA, (int)rows, C, K0P);
      g0.bias = c.conv_bias ? w->conv_b[0] : nullptr;
      RC(gemm(st, dt, g0));
    }
    const dim3 gp(s.nchunk, B);
    const long long n8 = rows * C / 8;
    if (c.feat_norm_layer) {      // Wav2Vec2LayerNormConvLayer: conv -> LayerNorm over the channels -> GELU, each a rounded tensor
      RC(layernorm_fwd(st, dt, s.bufA, w->conv_ln_w[0], w->conv_ln_b[0], s.bufA, (int)rows, C, 1e-5f));
      RC(gelu_fwd(st, dt, s.bufA, s.bufA, rows * C));
    } else if (dt == DT_BF16) {      // GroupNorm over time (per clip and channel) + GELU
      hipLaunchKernelGGL(w2v_gn_partial_k<bf16_t>, gp, dim3(C / 8), 0, st, (const bf16_t*)s.bufA, s.part, s.T[0], C, s.nchunk);
      hipLaunchKernelGGL(w2v_gn_final_k, dim3(g1(C, 128), B), dim3(128), 0, st, s.part, s.stat, s.T[0], C, s.nchunk, 1e-5f);
      hipLaunchKernelGGL(w2v_gn_gelu_k<bf16_t>, dim3(g1(n8, 256)), dim3(256), 0, st, (bf16_t*)s.bufA, s.stat, (const bf16_t*)w->gn_w, (const bf16_t*)w->gn_b, n8, s.T[0], C);
    } else {
      hipLaunchKernelGGL(w2v_gn_partial_k<float>, gp, dim3(C / 8), 0, st, (const float*)s.bufA, s.part, s.T[0], C, s.nchunk);
      hipLaunchKernelGGL(w2v_gn_final_k, dim3(g1(C, 128), B), dim3(128), 0, st, s.part, s.stat, s.T[0], C, s.nchunk, 1e-5f);
      hipLaunchKernelGGL(w2v_gn_gelu_k<float>, dim3(g1(n8, 256)), dim3(256), 0, st, (float*)s.bufA, s.stat, (const float*)w->gn_w, (const float*)w->gn_b, n8, s.T[0], C);
    }
    UVX_LAUNCH_CHECK();
  }
  void* cur = s.bufA;
  void* nxt = s.bufB;
  for (int i = 1; i < c.n_conv; ++i) {   // Conv1d(C, C, k, s) + GELU on the strided view (no bias: conv_bias = False)
    UVX_CHECK(w->conv_w[i] != nullptr, UVX_ERR_INVALID, "wav2vec2_fwd: conv layer %d has no weights", i);
    const int k = c.conv_kernel[i], sd = c.conv_stride[i];
    GemmDesc g = lin(cur, w->conv_w[i], nxt, s.T[i], C, k * C);
    g.lda = sd * C; g.act = c.feat_norm_layer ? 0 : 1; g.batch = B;
    g.sA = (long long)s.T[i - 1] * C; g.sC = (long long)s.T[i] * C;
    g.bias = c.conv_bias ? w->conv_b[i] : nullptr;
    RC(gemm(st, dt, g));
    if (c.feat_norm_layer) {
      RC(layernorm_fwd(st, dt, nxt, w->conv_ln_w[i], w->conv_ln_b[i], nxt, B * s.T[i], C, 1e-5f));
      RC(gelu_fwd(st, dt, nxt, nxt, (long long)B * s.T[i] * C));
    }
    void* t = cur; cur = nxt; nxt = t;
  }
  // ---- feature projection: LayerNorm(C) -> Linear(C, d) ----
  RC(layernorm_fwd(st, dt, cur, w->fp_ln_w, w->fp_ln_b, nxt, M, C, c.ln_eps));
  {
    GemmDesc g = lin(nxt, w->fp_w, s.x, M, d, C);
    g.bias = w->fp_b;
    RC(gemm(st, dt, g));
  }
  // ---- positional conv embedding (grouped, k = pos_k, "same" padding, last frame dropped for even k) ----
  {
    const int G = c.pos_groups, dg = d / G, K = c.pos_k, R = Tn + K;
    const long long n8 = (long long)B * G * R * dg / 8;
    if (dt == DT_BF16) hipLaunchKernelGGL(w2v_pos_pack_k<bf16_t>, dim3(g1(n8, 256)), dim3(256), 0, st, (const bf16_t*)s.x, (bf16_t*)s.xg, n8, Tn, d, G, K);
    else hipLaunchKernelGGL(w2v_pos_pack_k<float>, dim3(g1(n8, 256)), dim3(256), 0, st, (const float*)s.x, (float*)s.xg, n8, Tn, d, G, K);
    UVX_LAUNCH_CHECK();
    for (int b = 0; b < B; ++b) {   // one batched GEMM per clip: batch = groups (their weights differ, the clip's do not)
      GemmDesc g = lin(at(s.xg, (size_t)b * G * R * dg, dt), w->pos_w, s.acc + (size_t)b * Tn * d, Tn, dg, K * dg);
      g.lda = dg; g.ldc = d; g.out_f32 = 1; g.batch = G;
      g.sA = (long long)R * dg; g.sB = (long long)dg * K * dg; g.sC = dg;
      RC(gemm(st, dt, g));
    }
    const long long m8 = (long long)M * d / 8;
    if (dt == DT_BF16) hipLaunchKernelGGL(w2v_pos_finish_k<bf16_t>, dim3(g1(m8, 256)), dim3(256), 0, st, (bf16_t*)s.x, s.acc, (const bf16_t*)w->pos_b, m8, d);
    else hipLaunchKernelGGL(w2v_pos_finish_k<float>, dim3(g1(m8, 256)), dim3(256), 0, st, (float*)s.x, s.acc, (const float*)w->pos_b, m8, d);
    UVX_LAUNCH_CHECK();
  }
  const float qscale = 1.0f / sqrtf((float)dh);
  // one attention branch: q|k|v projection of `in` (+ adapters), attention, out_proj + residual `res` (+ adapter) -> `dst`
  auto attn_branch = [&](int l, const void* in, const void* res, void* dst) -> int {
    const uvx_enc_layer_t& Lw = w->layers[l];
    W2vLayerStash S = train ? w2v_layer(s, l) : W2vLayerStash{};
    void* qkv = train ? S.qkv : s.qkv;
    void* o = train ? S.o : s.o;
    {
      GemmDesc g = lin(in, Lw.wqkv, qkv, M, 3 * d, d);
      g.bias = Lw.bqkv;
      RC(gemm(st, dt, g));
    }
    if (train) {      // peft: result += lora_B(lora_A(x)) * scaling; q carries head_dim^-0.5 (folded into the packed q rows)
      const uvx_enc_lora_layer_t& R = lora->layers[l];
      if (R.q.a) RC(lora_apply(st, dt, in, d, R.q, S.bqT, S.t, qkv, 3 * d, M, d, d, lora->r, lora->scaling * qscale));
      if (R.k.a) RC(lora_apply(st, dt, in, d, R.k, S.bkT, at(S.t, 64, dt), at(qkv, d, dt), 3 * d, M, d, d, lora->r, lora->scaling));
      if (R.v.a) RC(lora_apply(st, dt, in, d, R.v, S.bvT, S.t2, at(qkv, 2 * d, dt), 3 * d, M, d, d, lora->r, lora->scaling));
    }
    if (attention_needs_transposed_copies(dt)) RC(heads_transpose(st, dt, at(qkv, 2 * d, dt), s.vt, B, Tn, s.Tp, c.heads, dh, 3 * d));
    AttnDesc ad;
    ad.q = qkv; ad.k = at(qkv, d, dt); ad.v = at(qkv, 2 * d, dt); ad.vt = s.vt; ad.o = o; ad.lse = train ? S.lse : nullptr;
    ad.B = B; ad.T = Tn; ad.Tp = s.Tp; ad.Hq = c.heads; ad.Hkv = c.heads; ad.D = dh;
    ad.ldq = ad.ldk = ad.ldv = 3 * d; ad.ldo = d; ad.causal = 0; ad.block = 0;
    ad.scale = 1.0f;   // q_proj (weight and bias) pre-scaled by head_dim^-0.5 at pack time: exact for a power of two
    RC(attention_fwd(st, dt, ad));
    {
      GemmDesc g = lin(o, Lw.wo, dst, M, d, d);
      g.bias = Lw.bo; g.residual = res; g.ldr = d;
      RC(gemm(st, dt, g));
    }
    if (train && lora->layers[l].o.a) RC(lora_apply(st, dt, o, d, lora->layers[l].o, S.boT, at(S.t2, 64, dt), dst, d, M, d, d, lora->r, lora->scaling));
    return UVX_OK;
  };
  // one feed-forward branch: fc1 + GELU (training: the pre-activation is kept), fc2 + residual `res` -> `dst`
  auto ffn_branch = [&](int l, const void* in, const void* res, void* dst) -> int {
    const uvx_enc_layer_t& Lw = w->layers[l];
    if (train) {
      W2vLayerStash S = w2v_layer(s, l);
      GemmDesc g = lin(in, Lw.fc1_w, S.pre, M, c.ffn, d);
      g.bias = Lw.fc1_b;
      RC(gemm(st, dt, g));
      RC(gelu_fwd(st, dt, S.pre, s.f, (long long)M * c.ffn));
    } else {
      GemmDesc g = lin(in, Lw.fc1_w, s.f, M, c.ffn, d);
      g.bias = Lw.fc1_b; g.act = 1;
      RC(gemm(st, dt, g));
    }
    GemmDesc g = lin(s.f, Lw.fc2_w, dst, M, d, c.ffn);
    g.bias = Lw.fc2_b; g.residual = res; g.ldr = d;
    return gemm(st, dt, g);
  };
  if (c.stable_ln) {
    // ---- encoder, do_stable_layer_norm = True ([3P] Wav2Vec2EncoderStableLayerNorm / ...EncoderLayerStableLayerNorm): per layer
    //      h = h + attention(layer_norm(h));  h = h + feed_forward(final_layer_norm(h));  encoder.layer_norm after the last layer ----
    // inference: the residual stream alternates between s.x and s.o2; training: it lives in the layer stashes (x_in -> mid -> next x_in)
    void* h = s.x;
    if (train && c.layers > 0) {
      UVX_HIP(hipMemcpyAsync(w2v_layer(s, 0).x_in, s.x, (size_t)M * d * esz(dt), hipMemcpyDeviceToDevice, st));
      h = w2v_layer(s, 0).x_in;
    }
    for (int l = 0; l < c.layers; ++l) {
      const uvx_enc_layer_t& Lw = w->layers[l];
      void* h2 = train ? w2v_layer(s, l).mid : s.o2;
      void* h_next = !train ? h : (l + 1 < c.layers ? w2v_layer(s, l + 1).x_in : s.x);
      RC(layernorm_fwd(st, dt, h, Lw.ln1_w, Lw.ln1_b, s.n, M, d, c.ln_eps));
      RC(attn_branch(l, s.n, h, h2));
      RC(layernorm_fwd(st, dt, h2, Lw.ln2_w, Lw.ln2_b, s.n, M, d, c.ln_eps));
      RC(ffn_branch(l, s.n, h2, h_next));
      h = h_next;
    }
    return layernorm_fwd(st, dt, h, w->ln_w, w->ln_b, out, M, d, c.ln_eps);      // (training: h == s.x, kept for the backward)
  }
  // ---- encoder (post-LN, do_stable_layer_norm = False): x = LN(x + attention(x)); x = LN(x + feed_forward(x)) ----
  // inference: x alternates between s.n and s.x; training: x_in / mid / y2 of the layer stashes, x1 = LN1(mid) in s.n
  void* x = train && c.layers > 0 ? w2v_layer(s, 0).x_in : s.n;
  void* y = s.x;
  RC(layernorm_fwd(st, dt, s.x, w->ln_w, w->ln_b, x, M, d, c.ln_eps));
  for (int l = 0; l < c.layers; ++l) {
    const uvx_enc_layer_t& Lw = w->layers[l];
    void* mid = train ? w2v_layer(s, l).mid : y;
    void* x1 = train ? s.n : x;
    void* y2 = train ? w2v_layer(s, l).y2 : y;
    RC(attn_branch(l, x, x, mid));
    RC(layernorm_fwd(st, dt, mid, Lw.ln1_w, Lw.ln1_b, x1, M, d, c.ln_eps));       // layers.N.layer_norm
    RC(ffn_branch(l, x1, x1, y2));
    void* dst = l + 1 == c.layers ? out : (train ? w2v_layer(s, l + 1).x_in : x);
    RC(layernorm_fwd(st, dt, y2, Lw.ln2_w, Lw.ln2_b, dst, M, d, c.ln_eps));     // layers.N.final_layer_norm
    x = dst;
  }
  if (c.layers == 0) UVX_HIP(hipMemcpyAsync(out, x, (size_t)M * d * esz(dt), hipMemcpyDeviceToDevice, st));
  return UVX_OK;
}

extern "C" int32_t uvx_wav2vec2_fwd(void* stream, const uvx_w2v_config_t* cfg, const uvx_w2v_weights_t* w, const void* input_values,
                                    int32_t values_is_f32, int32_t B, int32_t L, void* out, void* workspace, size_t ws_bytes) {
  return w2v_forward(stream, cfg, w, nullptr, input_values, values_is_f32, B, L, out, workspace, ws_bytes);
}

extern "C" size_t uvx_wav2vec2_train_ws_bytes(const uvx_w2v_config_t* cfg, int32_t B, int32_t L) {
  if (!cfg) return 0;
  Arena a(nullptr, 0);
  carve(a, *cfg, B, L, true);
  return a.off + 256;
}

extern "C" int32_t uvx_wav2vec2_fwd_train(void* stream, const uvx_w2v_config_t* cfg, const uvx_w2v_weights_t* w, const uvx_encoder_lora_t* lora,
                                          const void* input_values, int32_t values_is_f32, int32_t B, int32_t L, void* out, void* workspace,
                                          size_t ws_bytes) {
  UVX_CHECK(lora != nullptr, UVX_ERR_INVALID, "wav2vec2_fwd_train: null LoRA descriptor");
  return w2v_forward(stream, cfg, w, lora, input_values, values_is_f32, B, L, out, workspace, ws_bytes);
}

// Backward of the LoRA-adapted wav2vec2 encoder: d last_hidden_state [B, frames, d] -> the adapters' gradients in every layer (everything else is
// frozen: apply_lora, ultravox_model.py:690-709).  Walks the stash of uvx_wav2vec2_fwd_train; nothing below layer 0's q|k|v input is trainable, so the
// walk stops there (no gradient for the positional conv / feature projection / feature encoder).
extern "C" int32_t uvx_wav2vec2_bwd(void* stream, const uvx_w2v_config_t* cfg, const uvx_w2v_weights_t* w, const uvx_encoder_lora_t* lora,
                                    const void* d_out, int32_t B, int32_t L, const uvx_encoder_lora_grads_t* grads, void* workspace, size_t ws_bytes) {
  RC(check(cfg));
  UVX_CHECK(w && w->layers && lora && d_out && grads && grads->layers && workspace, UVX_ERR_INVALID, "wav2vec2_bwd: null argument");
  const uvx_w2v_config_t& c = *cfg;
  RC(lora_check(lora, c.layers, grads, "wav2vec2_bwd"));
  hipStream_t st = (hipStream_t)stream;
  if (B == 0 || c.layers == 0) return UVX_OK;
  Arena a(workspace, ws_bytes);
  W2vWs s = carve(a, c, B, L, true);
  UVX_CHECK(s.Tn > 0, UVX_ERR_SHAPE, "wav2vec2_bwd: %d samples are shorter than the conv stack's receptive field", L);
  UVX_CHECK(a.fits(), UVX_ERR_WORKSPACE, "wav2vec2_bwd: workspace %zu < %zu bytes", ws_bytes, a.off);
  const int dt = c.dtype, d = c.d, dh = d / c.heads, M = s.M, Tn = s.Tn, r = lora->r;
  const float qscale = 1.0f / sqrtf((float)dh);
  const long long wg_floats = lora_wgrad_scratch_floats(s.M, d, 64);
  // gradient of the attention branch's output `dy` [M, d] (d_o scratch: s.d_o) -> adapter gradients; unless `last`, d (branch input) in s.d_n =
  // d qkv . Wqkv + u . A (+ `add`: the residual path's gradient, folded into the dgrad's epilogue)
  auto attn_branch_bwd = [&](int l, const void* dy, const void* add, bool last) -> int {
    const uvx_enc_layer_t& Lw = w->layers[l];
    UVX_CHECK(Lw.wqkv_t && Lw.wo_t && Lw.fc1_t && Lw.fc2_t, UVX_ERR_INVALID, "wav2vec2_bwd: layer %d lacks transposed weights", l);
    W2vLayerStash S = w2v_layer(s, l);
    const uvx_enc_lora_layer_t& R = lora->layers[l];
    const uvx_enc_lora_layer_grads_t& G = grads->layers[l];
    // the branch input: pre-LN = layer_norm(x_in), recomputed; post-LN = x_in itself
    const void* in = S.x_in;
    if (c.stable_ln) {
      RC(layernorm_fwd(st, dt, S.x_in, Lw.ln1_w, Lw.ln1_b, s.n, M, d, c.ln_eps));
      in = s.n;
    }
    RC(gemm(st, dt, lin(dy, Lw.wo_t, s.d_o, M, d, d)));
    if (R.o.a) {
      RC(lora_apply_bwd(st, dt, S.o, d, dy, d, S.boT, at(S.t2, 64, dt), at(s.u2, 64, dt), G.o, M, d, d, r, lora->scaling, s.wg, wg_floats));
      RC(lora_up(st, dt, at(s.u2, 64, dt), 128, R.o.a, 1, s.d_o, d, M, d, r, 1.0f, 1));
    }
    if (attention_needs_transposed_copies(dt)) RC(heads_transpose(st, dt, S.qkv, s.qT, B, Tn, s.Tp, c.heads, dh, 3 * d));
    if (attention_needs_transposed_copies(dt)) RC(heads_transpose(st, dt, at(S.qkv, d, dt), s.kT, B, Tn, s.Tp, c.heads, dh, 3 * d));
    if (attention_needs_transposed_copies(dt)) RC(heads_transpose(st, dt, s.d_o, s.doT, B, Tn, s.Tp, c.heads, dh, d));
    AttnBwdDesc bd;
    AttnDesc& ad = bd.f;
    ad.q = S.qkv; ad.k = at(S.qkv, d, dt); ad.v = at(S.qkv, 2 * d, dt); ad.o = S.o; ad.lse = S.lse;
    ad.B = B; ad.T = Tn; ad.Tp = s.Tp; ad.Hq = c.heads; ad.Hkv = c.heads; ad.D = dh;
    ad.ldq = ad.ldk = ad.ldv = 3 * d; ad.ldo = d; ad.causal = 0; ad.block = 0; ad.scale = 1.0f;
    bd.dout = s.d_o; bd.qt = s.qT; bd.kt = s.kT; bd.dot = s.doT; bd.delta = s.delta; bd.dkv_part = nullptr;
    bd.dq = s.d_qkv; bd.dk = at(s.d_qkv, d, dt); bd.dv = at(s.d_qkv, 2 * d, dt);
    bd.lddq = bd.lddk = bd.lddv = 3 * d;
    RC(attention_bwd(st, dt, bd));
    if (R.q.a) RC(lora_apply_bwd(st, dt, in, d, s.d_qkv, 3 * d, S.bqT, S.t, s.u, G.q, M, d, d, r, lora->scaling * qscale, s.wg, wg_floats));
    if (R.k.a) RC(lora_apply_bwd(st, dt, in, d, at(s.d_qkv, d, dt), 3 * d, S.bkT, at(S.t, 64, dt), at(s.u, 64, dt), G.k, M, d, d, r, lora->scaling, s.wg, wg_floats));
    if (R.v.a) RC(lora_apply_bwd(st, dt, in, d, at(s.d_qkv, 2 * d, dt), 3 * d, S.bvT, S.t2, s.u2, G.v, M, d, d, r, lora->scaling, s.wg, wg_floats));
    if (last) return UVX_OK;
    {
      GemmDesc g = lin(s.d_qkv, Lw.wqkv_t, s.d_n, M, d, 3 * d);
      g.residual = add; g.ldr = d;
      RC(gemm(st, dt, g));
    }
    if (R.q.a) RC(lora_up(st, dt, s.u, 128, R.q.a, 1, s.d_n, d, M, d, r, 1.0f, 1));
    if (R.k.a) RC(lora_up(st, dt, at(s.u, 64, dt), 128, R.k.a, 1, s.d_n, d, M, d, r, 1.0f, 1));
    if (R.v.a) RC(lora_up(st, dt, s.u2, 128, R.v.a, 1, s.d_n, d, M, d, r, 1.0f, 1));
    return UVX_OK;
  };
  // gradient of the feed-forward branch's output `dy` -> s.d_n = (dy . W_fc2 * gelu'(pre)) . W_fc1 (+ `add`)
  auto ffn_branch_bwd = [&](int l, const void* dy, const void* add) -> int {
    const uvx_enc_layer_t& Lw = w->layers[l];
    W2vLayerStash S = w2v_layer(s, l);
    RC(gemm(st, dt, lin(dy, Lw.fc2_t, s.d_f, M, c.ffn, d)));
    RC(gelu_bwd(st, dt, s.d_f, S.pre, s.d_f, (long long)M * c.ffn));
    GemmDesc g = lin(s.d_f, Lw.fc1_t, s.d_n, M, d, c.ffn);
    g.residual = add; g.ldr = d;
    return gemm(st, dt, g);
  };
  if (c.stable_ln) {
    // out = encoder.layer_norm(h_final); h_final = s.x (left there by the forward)
    RC(layernorm_bwd(st, dt, d_out, s.x, w->ln_w, nullptr, s.dx, M, d, c.ln_eps));
    for (int l = c.layers - 1; l >= 0; --l) {
      const uvx_enc_layer_t& Lw = w->layers[l];
      W2vLayerStash S = w2v_layer(s, l);
      RC(ffn_branch_bwd(l, s.dx, nullptr));                                                     // d n2 in s.d_n
      RC(layernorm_bwd(st, dt, s.d_n, S.mid, Lw.ln2_w, s.dx, s.dx, M, d, c.ln_eps));           // d h2 = d h + LN2'(d n2)
      RC(attn_branch_bwd(l, s.dx, nullptr, l == 0));                                            // d n1 in s.d_n
      if (l == 0) break;
      RC(layernorm_bwd(st, dt, s.d_n, S.x_in, Lw.ln1_w, s.dx, s.dx, M, d, c.ln_eps));          // d h = d h2 + LN1'(d n1)
    }
    return UVX_OK;
  }
  // post-LN: the layer's output = final_layer_norm(y2), y2 = x1 + ffn(x1), x1 = layer_norm(mid), mid = x_in + attention(x_in)
  const void* dx = d_out;
  for (int l = c.layers - 1; l >= 0; --l) {
    const uvx_enc_layer_t& Lw = w->layers[l];
    W2vLayerStash S = w2v_layer(s, l);
    RC(layernorm_bwd(st, dt, dx, S.y2, Lw.ln2_w, nullptr, s.dx, M, d, c.ln_eps));               // d y2 in s.dx
    RC(ffn_branch_bwd(l, s.dx, s.dx));                                                          // d x1 = d y2 + ffn'(d y2) in s.d_n
    RC(layernorm_bwd(st, dt, s.d_n, S.mid, Lw.ln1_w, nullptr, s.dx, M, d, c.ln_eps));           // d mid in s.dx
    RC(attn_branch_bwd(l, s.dx, s.dx, l == 0));                                                 // d x_in = d mid + attention'(d mid) in s.d_n
    dx = s.d_n;
  }
  return UVX_OK;
}
