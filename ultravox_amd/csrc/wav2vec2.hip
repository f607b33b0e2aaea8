// Alt audio tower (BASELINE.json config 5): [3P] transformers Wav2Vec2Model.forward behind the C ABI - the AutoModel branch
// of UltravoxModel._create_audio_tower (ultravox_model.py:460-467, :476-485) for facebook/wav2vec2-large-960h-style towers
// (GroupNorm after the first conv layer, bias-free convs, post-LN encoder).  Frozen tower: forward only.
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

struct W2vWs {
  int T[9];          // frames after conv layer i
  int Tn, Tp, M, nchunk;
  void *im2col, *bufA, *bufB, *x, *n, *qkv, *vt, *o, *o2, *f, *xg;
  float *part, *stat, *acc;
};

int frames(const uvx_w2v_config_t& c, int L, int* T) {
  int n = L;
  for (int i = 0; i < c.n_conv; ++i) {
    if (n < c.conv_kernel[i]) return -1;
    n = (n - c.conv_kernel[i]) / c.conv_stride[i] + 1;
    T[i] = n;
  }
  return n;
}

W2vWs carve(Arena& a, const uvx_w2v_config_t& c, int B, int L) {
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

extern "C" int32_t uvx_wav2vec2_fwd(void* stream, const uvx_w2v_config_t* cfg, const uvx_w2v_weights_t* w, const void* input_values,
                                    int32_t values_is_f32, int32_t B, int32_t L, void* out, void* workspace, size_t ws_bytes) {
  RC(check(cfg));
  UVX_CHECK(w && w->layers && input_values && out && workspace, UVX_ERR_INVALID, "wav2vec2_fwd: null argument");
  const uvx_w2v_config_t& c = *cfg;
  hipStream_t st = (hipStream_t)stream;
  for (int i = 0; i < c.n_conv; ++i) {
    UVX_CHECK(!c.conv_bias || w->conv_b[i], UVX_ERR_INVALID, "wav2vec2_fwd: conv_bias is set but conv layer %d has no bias", i);
    UVX_CHECK(!c.feat_norm_layer || (w->conv_ln_w[i] && w->conv_ln_b[i]), UVX_ERR_INVALID, "wav2vec2_fwd: conv layer %d has no layer norm", i);
  }
  UVX_CHECK(c.feat_norm_layer || (w->gn_w && w->gn_b), UVX_ERR_INVALID, "wav2vec2_fwd: the first conv layer's GroupNorm weights are missing");
  if (B == 0) return UVX_OK;
  Arena a(workspace, ws_bytes);
  W2vWs s = carve(a, c, B, L);
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
  if (c.stable_ln) {
    // ---- encoder, do_stable_layer_norm = True ([3P] Wav2Vec2EncoderStableLayerNorm / ...EncoderLayerStableLayerNorm): per layer
    //      h = h + attention(layer_norm(h));  h = h + feed_forward(final_layer_norm(h));  encoder.layer_norm after the last layer ----
    void* h = s.x;       // the residual stream alternates between s.x and s.o2
    void* h2 = s.o2;
    for (int l = 0; l < c.layers; ++l) {
      const uvx_enc_layer_t& Lw = w->layers[l];
      RC(layernorm_fwd(st, dt, h, Lw.ln1_w, Lw.ln1_b, s.n, M, d, c.ln_eps));
      {
        GemmDesc g = lin(s.n, Lw.wqkv, s.qkv, M, 3 * d, d);
        g.bias = Lw.bqkv;
        RC(gemm(st, dt, g));
      }
      if (attention_needs_transposed_copies(dt)) RC(heads_transpose(st, dt, at(s.qkv, 2 * d, dt), s.vt, B, Tn, s.Tp, c.heads, dh, 3 * d));
      AttnDesc ad;
      ad.q = s.qkv; ad.k = at(s.qkv, d, dt); ad.v = at(s.qkv, 2 * d, dt); ad.vt = s.vt; ad.o = s.o;
      ad.B = B; ad.T = Tn; ad.Tp = s.Tp; ad.Hq = c.heads; ad.Hkv = c.heads; ad.D = dh;
      ad.ldq = ad.ldk = ad.ldv = 3 * d; ad.ldo = d; ad.causal = 0; ad.block = 0;
      ad.scale = 1.0f;
      RC(attention_fwd(st, dt, ad));
      {
        GemmDesc g = lin(s.o, Lw.wo, h2, M, d, d);
        g.bias = Lw.bo; g.residual = h; g.ldr = d;
        RC(gemm(st, dt, g));
      }
      RC(layernorm_fwd(st, dt, h2, Lw.ln2_w, Lw.ln2_b, s.n, M, d, c.ln_eps));
      {
        GemmDesc g = lin(s.n, Lw.fc1_w, s.f, M, c.ffn, d);
        g.bias = Lw.fc1_b; g.act = 1;
        RC(gemm(st, dt, g));
      }
      {
        GemmDesc g = lin(s.f, Lw.fc2_w, h, M, d, c.ffn);
        g.bias = Lw.fc2_b; g.residual = h2; g.ldr = d;
        RC(gemm(st, dt, g));
      }
    }
    return layernorm_fwd(st, dt, h, w->ln_w, w->ln_b, out, M, d, c.ln_eps);
  }
  // ---- encoder (post-LN, do_stable_layer_norm = False) ----
  void* x = s.n;     // x alternates between s.n and s.x
  void* y = s.x;
  RC(layernorm_fwd(st, dt, s.x, w->ln_w, w->ln_b, x, M, d, c.ln_eps));
  for (int l = 0; l < c.layers; ++l) {
    const uvx_enc_layer_t& Lw = w->layers[l];
    {
      GemmDesc g = lin(x, Lw.wqkv, s.qkv, M, 3 * d, d);
      g.bias = Lw.bqkv;
      RC(gemm(st, dt, g));
    }
    if (attention_needs_transposed_copies(dt)) RC(heads_transpose(st, dt, at(s.qkv, 2 * d, dt), s.vt, B, Tn, s.Tp, c.heads, dh, 3 * d));
    AttnDesc ad;
    ad.q = s.qkv; ad.k = at(s.qkv, d, dt); ad.v = at(s.qkv, 2 * d, dt); ad.vt = s.vt; ad.o = s.o;
    ad.B = B; ad.T = Tn; ad.Tp = s.Tp; ad.Hq = c.heads; ad.Hkv = c.heads; ad.D = dh;
    ad.ldq = ad.ldk = ad.ldv = 3 * d; ad.ldo = d; ad.causal = 0; ad.block = 0;
    ad.scale = 1.0f;   // q_proj (weight and bias) pre-scaled by head_dim^-0.5 at pack time: exact for a power of two
    RC(attention_fwd(st, dt, ad));
    {
      GemmDesc g = lin(s.o, Lw.wo, y, M, d, d);
      g.bias = Lw.bo; g.residual = x; g.ldr = d;
      RC(gemm(st, dt, g));
    }
    RC(layernorm_fwd(st, dt, y, Lw.ln1_w, Lw.ln1_b, x, M, d, c.ln_eps));       // layers.N.layer_norm
    {
      GemmDesc g = lin(x, Lw.fc1_w, s.f, M, c.ffn, d);
      g.bias = Lw.fc1_b; g.act = 1;
      RC(gemm(st, dt, g));
    }
    {
      GemmDesc g = lin(s.f, Lw.fc2_w, y, M, d, c.ffn);
      g.bias = Lw.fc2_b; g.residual = x; g.ldr = d;
      RC(gemm(st, dt, g));
    }
    void* dst = l + 1 == c.layers ? out : x;
    RC(layernorm_fwd(st, dt, y, Lw.ln2_w, Lw.ln2_b, dst, M, d, c.ln_eps));     // layers.N.final_layer_norm
  }
  if (c.layers == 0) UVX_HIP(hipMemcpyAsync(out, x, (size_t)M * d * esz(dt), hipMemcpyDeviceToDevice, st));
  return UVX_OK;
}
