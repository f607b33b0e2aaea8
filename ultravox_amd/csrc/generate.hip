// Inference path of the LLM: prefill + KV-cache decode.
//
// Reference: UltravoxModel.generate (ultravox_model.py:398-426) builds the merged inputs_embeds once and
// delegates to the [3P] HF language_model.generate: one prefill pass over the prompt, then a 1-token
// decode loop over a KV cache.  With an attention_mask HF derives position ids as cumsum(mask) - 1
// (1 where masked) — left-padded batches (infer.py:155-194) start counting at their first real token.
//
// The prefill reuses the training-path kernels (no stashes).  The decode step is weight streaming and launch-bound and has
// its own kernels: gemm_skinny.hip (M <= 16) and attn_decode_grp_k below (one block per sequence and KV head).  A further
// chunk of tokens on top of a filled cache (conversation turns) goes through uvx_llm_prefill_chunk.
// KV cache layout: [layer][k | v][B][Tmax][kv_heads * head_dim], caller-owned.
#include <algorithm>
#include "common.h"
#include "kernels.h"
#include "../../include/uvx.h"

namespace {
using namespace uvx;

struct Arena {
  char* base; size_t cap; size_t off = 0;
  Arena(void* b, size_t c) : base((char*)b), cap(c) {}
  void* take(size_t bytes) { const size_t a = (off + 255) & ~(size_t)255; off = a + bytes; return base ? (void*)(base + a) : nullptr; }
  bool fits() const { return !base || off <= cap; }
};
inline size_t esz(int dtype) { return dtype == DT_BF16 ? 2 : 4; }
inline char* at(const void* p, size_t elems, int dtype) { return (char*)p + elems * esz(dtype); }
#define RC(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

struct InferWs {
  void *x, *x2, *n, *qkv, *vt, *o, *gu, *act, *last, *hn;
  int32_t *kvs, *kvl, *pos;
  int M, Tp, QKV, OD;
  void* sk; size_t sk_bytes;     // split-K scratch of the GEMMs (bf16 prefill of a few hundred rows; gemm.hip "Split-K"), or null
};
InferWs carve(Arena& a, const uvx_config_t& c, int B, int T) {
  InferWs w;
  const size_t es = esz(c.dtype), M = (size_t)B * T;
  w.M = B * T; w.Tp = (T + 63) / 64 * 64;
  w.QKV = (c.llm_heads + 2 * c.llm_kv_heads) * c.llm_head_dim; w.OD = c.llm_heads * c.llm_head_dim;
  w.x = a.take(M * c.llm_d * es); w.x2 = a.take(M * c.llm_d * es); w.n = a.take(M * c.llm_d * es);
  w.qkv = a.take(M * w.QKV * es);
  w.vt = a.take((size_t)B * c.llm_kv_heads * c.llm_head_dim * w.Tp * es);
  w.o = a.take(M * w.OD * es);
  w.gu = a.take(M * 2 * c.llm_inter * es); w.act = a.take(M * c.llm_inter * es);
  w.last = a.take((size_t)B * c.llm_d * es); w.hn = a.take((size_t)B * c.llm_d * es);
  w.kvs = (int32_t*)a.take(sizeof(int32_t) * B); w.kvl = (int32_t*)a.take(sizeof(int32_t) * B);
  w.pos = (int32_t*)a.take(sizeof(int32_t) * M);
  // a decode batch beyond 16 sequences, or one or two prompts' worth of rows: too many rows for the weight-streaming kernels (the staged
  // MFMA kernel serves 16-row tiles: B = 32 runs at 0.44, B = 64 at 0.23 of the HBM roofline), too few tiles for the 256 CUs
  w.sk_bytes = c.dtype == DT_BF16 && M > 16 && M <= 1536 ? gemm_splitk_ws_bytes((int)M, std::max(std::max(w.QKV, 2 * c.llm_inter), c.llm_d)) : 0;
  w.sk = w.sk_bytes ? a.take(w.sk_bytes) : nullptr;
  return w;
}
// lends the split-K scratch to a GEMM of the layer loop
GemmDesc sk(GemmDesc g, const InferWs& s) { g.splitk_ws = s.sk; g.splitk_ws_bytes = s.sk_bytes; return g; }

// position ids + valid key range from the attention mask (HF prepare_inputs_for_generation semantics)
__global__ void mask_positions_k(const int64_t* __restrict__ mask, int32_t* __restrict__ pos, int32_t* __restrict__ kv_start,
                                 int32_t* __restrict__ kv_len, int32_t* __restrict__ next_pos, int T) {
  const int b = blockIdx.x;
  if (threadIdx.x != 0) return;
  int run = 0, lo = T, hi = 0;
  for (int t = 0; t < T; ++t) {
    const bool keep = mask ? mask[(long long)b * T + t] != 0 : true;
    if (keep) { pos[b * T + t] = run++; lo = min(lo, t); hi = t + 1; }
    else pos[b * T + t] = 1;
  }
  kv_start[b] = lo < hi ? lo : 0;
  kv_len[b] = hi;
  next_pos[b] = run;
}

// cache[layer][0|1][b][t0 + t][:] = k|v part of qkv row (b, t)
template <typename T>
__global__ void kv_append_k(const T* __restrict__ qkv, T* __restrict__ cache_k, T* __restrict__ cache_v, int B, int Tn,
                            int Tmax, int t0, int QKV, int koff, int KVD) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int per_row = KVD / 8;
  if (i >= (long long)B * Tn * per_row) return;
  const int c = (int)(i % per_row) * 8;
  const long long row = i / per_row;
  const int b = (int)(row / Tn), t = (int)(row % Tn);
  float k[8], v[8];
  ld8<T>(qkv + row * QKV + koff + c, k);
  ld8<T>(qkv + row * QKV + koff + KVD + c, v);
  const long long dst = ((long long)b * Tmax + t0 + t) * KVD + c;
  st8<T>(cache_k + dst, k);
  st8<T>(cache_v + dst, v);
}

// Rotary embedding and cache append in ONE launch (round 4, decode: rope_k + kv_append_k were two 4.6 us launches per layer, 0.7 ms of a
// 26 ms Llama-3.3-70B token; round 5: the prefill's Tn new positions per sequence go the same way - the pair re-read the k rows it had
// just written).  Row r = b Tn + t of qkv sits at position pos[r] and goes to cache row t0 + t of sequence b: q heads are rotated in
// place, k heads are rotated and written to the cache row (and back to the qkv row, as rope_k does - the prefill's attention reads them
// there), v heads are copied to the cache.  Same arithmetic and rounding points as rope_k.
// item = (row, head of q | k, 8-column chunk of the first half of the head) or (row, 8-column chunk of v).
template <typename T>
__global__ void rope_kv_append_k(T* __restrict__ qkv, const float* __restrict__ cs, const int32_t* __restrict__ pos, T* __restrict__ cache_k,
                                 T* __restrict__ cache_v, int B, int Tn, int Tmax, int t0, int Hq, int Hkv, int D, int QKV) {
  const int per_head = D / 16, KVD = Hkv * D;
  const int rope_items = (Hq + Hkv) * per_head, v_items = KVD / 8, per_row = rope_items + v_items;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * Tn * per_row) return;
  const long long r = i / per_row;
  const int rem = (int)(i % per_row);
  const int b = (int)(r / Tn), tq = (int)(r % Tn);
  T* row = qkv + r * QKV;
  const long long crow = ((long long)b * Tmax + t0 + tq) * KVD;
  if (rem >= rope_items) {                      // v: copy
    const int c = (rem - rope_items) * 8;
    float v[8];
    ld8<T>(row + (Hq + Hkv) * D + c, v);
    st8<T>(cache_v + crow + c, v);
    return;
  }
  const int h = rem / per_head, c = (rem % per_head) * 8;
  T* base = row + h * D;
  const float* t = cs + ((long long)pos[r] * (D / 2) + c) * 2;
  float lo[8], hi[8], olo[8], ohi[8];
  ld8<T>(base + c, lo);
  ld8<T>(base + D / 2 + c, hi);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float co = rnd<T>(t[2 * k]), si = rnd<T>(t[2 * k + 1]);
    olo[k] = rnd<T>(lo[k] * co) + rnd<T>(-hi[k] * si);
    ohi[k] = rnd<T>(hi[k] * co) + rnd<T>(lo[k] * si);
  }
  st8<T>(base + c, olo);
  st8<T>(base + D / 2 + c, ohi);
  if (h >= Hq) {                                // k: the rotated row is the cache row
    T* kc = cache_k + crow + (h - Hq) * D;
    st8<T>(kc + c, olo);
    st8<T>(kc + D / 2 + c, ohi);
  }
}
// q | k | v projection, rotary embedding and cache append of Tn new positions per sequence (prefill, chunked prefill).  bf16 without
// per-head q / k norms: the projection, then rope_kv_append_k; otherwise qkv_rope + kv_append_k (option 15 = 0 forces that pair: A/B)
template <typename T>
void launch_rope_kv_append(hipStream_t st, void* qkv, const float* cs, const int32_t* pos, void* ck, void* cv, int B, int Tn, int Tmax, int t0,
                           int Hq, int Hkv, int dh, int QKV) {
  const long long items = (long long)B * Tn * ((Hq + Hkv) * (dh / 16) + Hkv * dh / 8);
  hipLaunchKernelGGL(rope_kv_append_k<T>, dim3(cdiv(items, 256)), dim3(256), 0, st, (T*)qkv, cs, pos, (T*)ck, (T*)cv, B, Tn, Tmax, t0, Hq, Hkv, dh, QKV);
}

// full[b][t][koff + c] = cache_k[b][t][c], full[b][t][koff + KVD + c] = cache_v[b][t][c] for t < Tf: the cached keys / values
// laid back into the [B, Tf, QKV] row format the prefill attention reads (chunked prefill over a cached prefix)
template <typename T>
__global__ void kv_gather_k(const T* __restrict__ cache_k, const T* __restrict__ cache_v, T* __restrict__ full, int B, int Tf,
                            int Tmax, int QKV, int koff, int KVD) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int per_row = KVD / 8;
  if (i >= (long long)B * Tf * per_row) return;
  const int c = (int)(i % per_row) * 8;
  const long long row = i / per_row;
  const int b = (int)(row / Tf), t = (int)(row % Tf);
  float k[8], v[8];
  const long long src = ((long long)b * Tmax + t) * KVD + c;
  ld8<T>(cache_k + src, k);
  ld8<T>(cache_v + src, v);
  st8<T>(full + row * QKV + koff + c, k);
  st8<T>(full + row * QKV + koff + KVD + c, v);
}

// RoPE positions of a chunk that continues each sequence: pos[b * Tn + i] = pos0[b] + i
__global__ void chunk_positions_k(const int32_t* __restrict__ pos0, int32_t* __restrict__ pos, int B, int Tn) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * Tn) pos[i] = pos0[i / Tn] + i % Tn;
}

// One wave per (sequence, query head): q . K^T over the cached keys, online softmax, P . V.
template <typename T, int D>
__global__ void attn_decode_k(const T* __restrict__ qkv, const T* __restrict__ cache_k, const T* __restrict__ cache_v,
                              T* __restrict__ out, const int32_t* __restrict__ kv_start, int B, int Hq, int Hkv, int Tmax,
                              int len, int QKV, float scale, int lo_clamp) {
  const int wid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wid >= B * Hq) return;
  const int b = wid / Hq, h = wid % Hq, hk = h / (Hq / Hkv);
  constexpr int E = D / 64;
  float q[E], acc[E];
#pragma unroll
  for (int e = 0; e < E; ++e) { q[e] = ldf<T>(qkv + (long long)b * QKV + h * D + lane + 64 * e); acc[e] = 0.f; }
  const int KVD = Hkv * D;
  float m = -__builtin_huge_valf(), l = 0.f;
  // (lo_clamp: Gemma-3's sliding-window layers see the last `window` cache slots only)
  for (int j = max(kv_start ? kv_start[b] : 0, lo_clamp); j < len; ++j) {
    const T* kr = cache_k + ((long long)b * Tmax + j) * KVD + hk * D;
    const T* vr = cache_v + ((long long)b * Tmax + j) * KVD + hk * D;
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) s += q[e] * ldf<T>(kr + lane + 64 * e);
    s = wave_sum(s) * scale;
    const float mn = fmaxf(m, s);
    const float alpha = expf(m - mn), p = rnd<T>(expf(s - mn));
    l = l * alpha + p;
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = acc[e] * alpha + p * ldf<T>(vr + lane + 64 * e);
    m = mn;
  }
  const float inv = l > 0.f ? 1.f / l : 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e) stf<T>(out + (long long)b * Hq * D + h * D + lane + 64 * e, acc[e] * inv);
}

// One block per (sequence, KV head): the G query heads that share the KV head are served together (K and V rows are read
// once from HBM / L2).  Round 4: 1024 threads (was 256: 55 us per layer at 332 keys on Llama-3.3-70B, 12 % of a decoded token -
// a latency chain on 8 of 256 CUs, profiles/r04_decode70_kernel_stats.txt).
//   phase 1  scores: thread = (key lane, 16-byte chunk of the head dimension), NTH / (D / 8) keys in flight per pass, the chunk
//            partial sums of a key folded by shuffles -> LDS sc[g][key]
//   phase 2  softmax statistics: one wave per query head; probabilities rounded to the storage type (as the tiled kernels do)
//   phase 3  P . V: thread = (key residue class, query head, 8 output dimensions): 16-byte V loads, four keys in flight, the residue
//            classes folded through LDS in a fixed order.
// sc: dynamic LDS, G * len floats.
// rope_cs != NULL (round 6, tuning option 23): the kernel ALSO does what rope_kv_append_k did for the new token in a launch of its own (4.9 us per
// layer: 5 % of a Llama-3-8B token) - the block's G query heads are rotated while they are loaded, and the new token's key (rotated) and value rows of
// its KV head are written to cache row len - 1 before the scores are taken.  With hsplit > 1 several blocks write the same values to the same row
// (benign); a block reads the row only after its own stores are fenced and barriered.  rope_kv_append_k's arithmetic and rounding points: bit-identical.
template <typename T, int D, int G, int NTH>
__global__ __launch_bounds__(NTH) void attn_decode_grp_k(const T* __restrict__ qkv, T* __restrict__ cache_k,
                                                         T* __restrict__ cache_v, T* __restrict__ out,
                                                         const int32_t* __restrict__ kv_start, int Hq, int Hkv, int Tmax,
                                                         int len, int QKV, float scale, int lo_clamp, int hsplit,
                                                         const float* __restrict__ rope_cs, const int32_t* __restrict__ rope_pos) {
  // hsplit: the query heads of a KV head are dealt to `hsplit` blocks of G heads each (G = group size / hsplit): at batch 1 a 70B
  // decode step had 8 blocks of 8 heads - 20 us of latency on 8 of 256 CUs; 64 blocks of one head re-read K / V from L2 and finish sooner
  extern __shared__ float sc[];                 // [G][len]
  __shared__ float qs[G][D];
  __shared__ float red[NTH * 8];                // phase 3 partial sums: [residue class][query head][dimension]
  __shared__ float stat[G];
  constexpr int CH = D / 8;                     // 16-byte chunks per row
  constexpr int KL = NTH / CH;                  // keys in flight per pass
  constexpr int NW = NTH / 64;
  constexpr int ITEMS = G * D / 8;              // (query head, 8-dimension chunk) pairs of the block
  constexpr int KS = NTH / ITEMS;               // key residue classes in phase 3
  static_assert(NTH % ITEMS == 0 && ITEMS <= NTH, "block size vs outputs");
  const int b = blockIdx.x / (Hkv * hsplit), hq0 = (blockIdx.x % (Hkv * hsplit)) * G, hk = hq0 / (G * hsplit);   // first query head, KV head
  const int tid = threadIdx.x, ch = tid % CH, kl = tid / CH, lane = tid & 63, w = tid >> 6;
  const int j0 = max(kv_start ? kv_start[b] : 0, lo_clamp);
  const int KVD = Hkv * D;
  const T* kbase = cache_k + (long long)b * Tmax * KVD + hk * D + ch * 8;
  const T* vbase = cache_v + (long long)b * Tmax * KVD + hk * D;
  if (rope_cs) {
    const float* t = rope_cs + (long long)rope_pos[b] * (D / 2) * 2;
    const T* row = qkv + (long long)b * QKV;
    // items: (head among the G query heads + the KV head's key, pair index d < D / 2); then the value row's 8-element chunks
    for (int i = tid; i < (G + 1) * (D / 2); i += NTH) {
      const int hh = i / (D / 2), d = i % (D / 2);
      const T* base = hh < G ? row + (hq0 + hh) * D : row + (Hq + hk) * D;
      const float lo = ldf<T>(base + d), hi = ldf<T>(base + D / 2 + d);
      const float co = rnd<T>(t[2 * d]), si = rnd<T>(t[2 * d + 1]);
      const float olo = rnd<T>(rnd<T>(lo * co) + rnd<T>(-hi * si)), ohi = rnd<T>(rnd<T>(hi * co) + rnd<T>(lo * si));
      if (hh < G) { qs[hh][d] = olo; qs[hh][D / 2 + d] = ohi; }
      else {
        T* kc = cache_k + ((long long)b * Tmax + len - 1) * KVD + hk * D;
        stf<T>(kc + d, olo); stf<T>(kc + D / 2 + d, ohi);
      }
    }
    if (tid < D / 8) {
      float v[8];
      ld8<T>(row + (Hq + Hkv) * D + hk * D + tid * 8, v);
      st8<T>(cache_v + ((long long)b * Tmax + len - 1) * KVD + hk * D + tid * 8, v);
    }
    // the row is read back below by other threads of THIS block only: a workgroup-scope release (the stores have left the wave) + the barrier;
    // a device-scope fence here costs a cache write-back per block (first form of this change: 3.31 -> 3.44 ms per token, B = 8 3.9 -> 5.7)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  } else {
    for (int i = tid; i < G * D; i += NTH) qs[i / D][i % D] = ldf<T>(qkv + (long long)b * QKV + (hq0 + i / D) * D + i % D);
  }
  __syncthreads();
  if (rope_cs) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  // ---- phase 1: scores (two passes of keys in flight per trip) ----
  {
    float q[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int e = 0; e < 8; ++e) q[g][e] = qs[g][ch * 8 + e];
    for (int j = j0 + kl; j < len; j += 2 * KL) {
      float kv[2][8];
      const bool two = j + KL < len;
      ld8<T>(kbase + (long long)j * KVD, kv[0]);
      ld8<T>(kbase + (long long)(two ? j + KL : j) * KVD, kv[1]);
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int g = 0; g < G; ++g) {
          float s = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) s += q[g][e] * kv[u][e];
#pragma unroll
          for (int o = 1; o < CH; o <<= 1) s += __shfl_xor(s, o, 64);      // the CH threads of a key are consecutive lanes
          if (ch == 0 && (u == 0 || two)) sc[g * len + j + u * KL] = s * scale;
        }
    }
  }
  __syncthreads();
  // ---- phase 2: softmax statistics (one wave per head, heads round-robin) ----
  for (int g = w; g < G; g += NW) {
    float m = -__builtin_huge_valf();
    for (int j = j0 + lane; j < len; j += 64) m = fmaxf(m, sc[g * len + j]);
    m = wave_max(m);
    float l = 0.f;
    for (int j = j0 + lane; j < len; j += 64) {
      const float p = rnd<T>(expf(sc[g * len + j] - m));
      sc[g * len + j] = p;
      l += p;
    }
    l = wave_sum(l);
    if (lane == 0) stat[g] = l;
  }
  __syncthreads();
  // ---- phase 3: P . V ----
  // thread = (key residue class `part`, query head g, 8 output dimensions): 16-byte V loads, four keys in flight per trip, KS classes
  // walk the keys in parallel (the first version - one thread per (g, dimension) over ALL keys - was a chain of ~40 dependent
  // load batches: 30 us per layer)
  const int i = tid % ITEMS, part = tid / ITEMS;
  const int g = (i * 8) / D, dd = (i * 8) % D;
  const T* vp = vbase + dd;
  const float* pg = sc + g * len;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  int j = j0 + part;
  for (; j + 3 * KS < len; j += 4 * KS) {
    float v[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) ld8<T>(vp + (long long)(j + u * KS) * KVD, v[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float pj = pg[j + u * KS];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += pj * v[u][e];
    }
  }
  for (; j < len; j += KS) {
    float v[8];
    ld8<T>(vp + (long long)j * KVD, v);
    const float pj = pg[j];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += pj * v[e];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[(part * ITEMS + i) * 8 + e] = acc[e];
  __syncthreads();
  for (int o = tid; o < G * D; o += NTH) {       // fold the residue classes in a fixed order
    float t = 0.f;
    for (int q2 = 0; q2 < KS; ++q2) t += red[q2 * ITEMS * 8 + o];
    const float l = stat[o / D];
    stf<T>(out + (long long)b * Hq * D + hq0 * D + o, l > 0.f ? t / l : 0.f);
  }
}

// launch helper: static LDS (~36 KB) + the dynamic score array (<= 48 KB) exceed the 64 KB a launch gets by default
template <int DD, int GG>
int launch_decode_grp(hipStream_t st, size_t sh, int blocks, const bf16_t* qkv, bf16_t* ck, bf16_t* cv, bf16_t* o,
                      const int32_t* kv_start, int Hq, int Hkv, int Tmax, int len, int QKV, float scale, int lo, int hsplit,
                      const float* rope_cs, const int32_t* rope_pos) {
  static PerDeviceOnce attr_set;
  UVX_SET_ATTR_ONCE(attr_set, (attn_decode_grp_k<bf16_t, DD, GG, 1024>), 64 * 1024);
  hipLaunchKernelGGL((attn_decode_grp_k<bf16_t, DD, GG, 1024>), dim3(blocks), dim3(1024), sh, st, qkv, ck, cv, o, kv_start, Hq, Hkv, Tmax, len,
                     QKV, scale, lo, hsplit, rope_cs, rope_pos);
  return UVX_OK;
}

// argmax of one logits row by the whole block (1024 threads; bv / bi: LDS scratch of blockDim.x entries); the result is valid in thread 0
template <typename T>
__device__ __forceinline__ int row_argmax(const T* __restrict__ r, int V, float* bv, int* bi) {
  float best = -__builtin_huge_valf();
  int idx = 0x7fffffff;
  // 16-byte loads, 1024 threads per row (a 128256-wide row took 190 us with 256 threads and 2-byte loads)
  const int V8 = (V % 8 == 0 && (reinterpret_cast<uintptr_t>(r) & 15) == 0) ? V : 0;
  for (int c = threadIdx.x * 8; c < V8; c += blockDim.x * 8) {
    float v[8];
    ld8<T>(r + c, v);
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (v[e] > best) { best = v[e]; idx = c + e; }    // ascending c within a thread: strict > keeps the lowest index
  }
  for (int c = V8 + threadIdx.x; c < V; c += blockDim.x) {
    const float v = ldf<T>(r + c);
    if (v > best) { best = v; idx = c; }
  }
  bv[threadIdx.x] = best; bi[threadIdx.x] = idx;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      const float o = bv[threadIdx.x + s];
      const int oi = bi[threadIdx.x + s];
      if (o > bv[threadIdx.x] || (o == bv[threadIdx.x] && oi < bi[threadIdx.x])) { bv[threadIdx.x] = o; bi[threadIdx.x] = oi; }
    }
    __syncthreads();
  }
  return bi[0] == 0x7fffffff ? 0 : bi[0];   // nothing above -inf: index 0, like torch.argmax
}

template <typename T>
__global__ void argmax_k(const T* __restrict__ logits, int64_t* __restrict__ out, int V) {
  __shared__ float bv[1024];
  __shared__ int bi[1024];
  const int idx = row_argmax<T>(logits + (long long)blockIdx.x * V, V, bv, bi);
  if (threadIdx.x == 0) out[blockIdx.x] = idx;
}

// The per-token bookkeeping of a greedy generate() loop ([3P] GenerationMixin._sample with do_sample = False, as the reference's
// inference path reaches it: ultravox_model.py:422-426 -> language_model.generate) in ONE launch instead of argmax + eight small torch
// kernels: next token = argmax of the row while the sequence is unfinished, pad_token_id afterwards; it is appended to the sequences
// buffer; a sequence is finished once it has produced an EOS id; the next decode step's RoPE position is written; the number of still
// unfinished rows goes to counter[step & 1] (block 0 zeroes the other slot for the next launch: launches of one loop are stream-ordered).
template <typename T>
__global__ void greedy_select_k(const T* __restrict__ logits, int V, const int64_t* __restrict__ eos_ids, int n_eos, int64_t pad,
                                int32_t* __restrict__ unfinished, int64_t* __restrict__ next_tokens, int64_t* __restrict__ sequences,
                                long long stride, long long col, const int32_t* __restrict__ pos0, int32_t* __restrict__ pos, int step,
                                int32_t* __restrict__ counter) {
  __shared__ float bv[1024];
  __shared__ int bi[1024];
  const int b = blockIdx.x;
  const int idx = row_argmax<T>(logits + (long long)b * V, V, bv, bi);
  if (threadIdx.x != 0) return;
  if (b == 0) counter[(step & 1) ^ 1] = 0;
  const bool live = unfinished[b] != 0;
  const int64_t tok = live ? (int64_t)idx : pad;
  next_tokens[b] = tok;
  sequences[(long long)b * stride + col] = tok;
  bool still = live;
  for (int e = 0; e < n_eos; ++e) still = still && tok != eos_ids[e];
  unfinished[b] = still ? 1 : 0;
  if (pos) pos[b] = pos0[b] + step;
  if (still) atomicAdd(&counter[step & 1], 1);
}

// [3P] transformers 4.51.3 GemmaModel.forward: hidden_states * tensor(hidden_size ** 0.5, dtype) - the normaliser is rounded
// to the model dtype first (same helper as model.hip)
float gemma_normalizer(const uvx_config_t& c) {
  const float n = sqrtf((float)c.llm_d);
  return c.dtype == DT_BF16 ? bf2f(f2bf(n)) : n;
}

GemmDesc lin(const void* A, const void* W, void* C, int M, int N, int K) {
  GemmDesc g;
  g.A = A; g.B = W; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N;
  return g;
}

// one decoder layer on M = B*T rows; attention supplied by the caller
struct LayerIO { void *x_in, *x_mid; };

// n_ready: s.n already holds post_attention_layernorm(x_mid) (attn_out's split-K reduce wrote it).  next_ln1 / next_ready: the NEXT
// layer's input_layernorm is asked of the down projection's reduce (GemmDesc::norm_*); *next_ready says whether s.n holds it on return.
int mlp_block(hipStream_t st, const uvx_config_t& c, const uvx_llm_layer_t& L, InferWs& s, int M, void* x_mid, void* x_out,
              bool n_ready = false, const void* next_ln1 = nullptr, bool* next_ready = nullptr) {
  const int dt = c.dtype, D = c.llm_d;
  if (next_ready) *next_ready = false;
  if (!n_ready && dt == DT_BF16 && M <= 16 && c.llm_flavor != UVX_LLM_GEMMA3) {     // decode: post_attention_layernorm inside the gate|up GEMV (M <= 2) / the staged skinny kernel (M <= 16)
    GemmDesc g = lin(x_mid, L.wgu, s.gu, M, 2 * c.llm_inter, D);
    const bool fused = c.llm_flavor == UVX_LLM_LLAMA;
    if (fused) { g.C2 = s.act; g.ldc2 = c.llm_inter; g.swiglu = 1; }
    const int rc = gemm_skinny_rmsnorm_bf16(st, g, L.ln2, c.rms_eps, c.llm_flavor);
    if (rc != UVX_ERR_UNSUPPORTED) {
      RC(rc);
      if (!fused) RC(swiglu_fwd(st, dt, s.gu, s.act, M, c.llm_inter, 2, c.llm_act));
      GemmDesc d = lin(s.act, L.wd, x_out, M, D, c.llm_inter);
      d.residual = x_mid; d.ldr = D;
      return gemm(st, dt, d);
    }
  }
  if (!n_ready) RC(rmsnorm_fwd(st, dt, x_mid, L.ln2, s.n, nullptr, M, D, c.rms_eps, c.llm_flavor));
  if (c.llm_flavor == UVX_LLM_GEMMA3) {      // x_out = x_mid + post_feedforward_norm(mlp(pre_feedforward_norm(x_mid)))
    RC(gemm(st, dt, sk(lin(s.n, L.wgu, s.gu, M, 2 * c.llm_inter, D), s)));
    RC(swiglu_fwd(st, dt, s.gu, s.act, M, c.llm_inter, 2, c.llm_act));
    RC(gemm(st, dt, sk(lin(s.act, L.wd, s.n, M, D, c.llm_inter), s)));        // (s.n is free again: the gate|up GEMM has consumed it)
    return rmsnorm_fwd(st, dt, s.n, L.ln2_post, x_out, nullptr, M, D, c.rms_eps, c.llm_flavor, nullptr, x_mid);
  }
  GemmDesc g = lin(s.n, L.wgu, s.gu, M, 2 * c.llm_inter, D);
  const bool fused = dt == DT_BF16 && c.llm_flavor == UVX_LLM_LLAMA;   // SwiGLU in the epilogue; Gemma's GeGLU: separate kernel
  if (fused) { g.C2 = s.act; g.ldc2 = c.llm_inter; g.swiglu = 1; }
  RC(gemm(st, dt, sk(g, s)));
  if (!fused) RC(swiglu_fwd(st, dt, s.gu, s.act, M, c.llm_inter, 2, c.llm_act));
  GemmDesc d = lin(s.act, L.wd, x_out, M, D, c.llm_inter);
  d.residual = x_mid; d.ldr = D;
  if (dt == DT_BF16 && next_ln1 && next_ready) {      // (s.n is free: the gate|up GEMM has consumed it)
    d.norm_w = next_ln1; d.norm_out = s.n; d.norm_ld = D; d.norm_eps = c.rms_eps; d.norm_flavor = c.llm_flavor; d.norm_done = next_ready;
  }
  return gemm(st, dt, sk(d, s));
}

// x_out = x + o_proj(o)  -  Gemma-3: x + post_attention_norm(o_proj(o))
// n_ready (may be null): set when s.n holds post_attention_layernorm(x_out) on return (written by the o projection's split-K reduce)
int attn_out(hipStream_t st, const uvx_config_t& c, const uvx_llm_layer_t& L, InferWs& s, int M, const void* o, int OD, const void* x, void* x_out,
             bool* n_ready = nullptr) {
  const int dt = c.dtype, D = c.llm_d;
  if (n_ready) *n_ready = false;
  if (c.llm_flavor == UVX_LLM_GEMMA3) {
    RC(gemm(st, dt, sk(lin(o, L.wo, s.n, M, D, OD), s)));             // (s.n is free: the q|k|v GEMM has consumed it)
    return rmsnorm_fwd(st, dt, s.n, L.ln1_post, x_out, nullptr, M, D, c.rms_eps, c.llm_flavor, nullptr, x);
  }
  GemmDesc g = lin(o, L.wo, x_out, M, D, OD);
  g.residual = x; g.ldr = D;
  if (dt == DT_BF16 && n_ready) {      // (s.n is free: the q|k|v GEMM has consumed it)
    g.norm_w = L.ln2; g.norm_out = s.n; g.norm_ld = D; g.norm_eps = c.rms_eps; g.norm_flavor = c.llm_flavor; g.norm_done = n_ready;
  }
  return gemm(st, dt, sk(g, s));
}
float attn_scale_of(const uvx_config_t& c) { return c.llm_attn_scale > 0.f ? c.llm_attn_scale : 1.0f / sqrtf((float)c.llm_head_dim); }
// Gemma-3: post norms present and a local rotary table where layers are flagged
int g3_check(const uvx_config_t& c, const uvx_llm_weights_t* w, int Tmax) {
  if (c.llm_flavor != UVX_LLM_GEMMA3) return UVX_OK;
  bool any_local = false;
  for (int l = 0; l < c.llm_layers; ++l) {
    UVX_CHECK(w->layers[l].ln1_post && w->layers[l].ln2_post, UVX_ERR_INVALID, "llm: Gemma-3 layer %d has no post norms", l);
    any_local = any_local || (w->layer_local && w->layer_local[l]);
  }
  UVX_CHECK(!any_local || w->rope_cos_sin_local, UVX_ERR_INVALID, "llm: Gemma-3 sliding-window layers need rope_cos_sin_local");
  (void)Tmax;
  return UVX_OK;
}

// q | k | v projection of a layer (+ Qwen2's biases), then the rotary embedding - for Qwen3 / Gemma-3 behind the per-head q_norm / k_norm
// (l: the layer index - Gemma-3's sliding-window layers rotate with their own table)
int qkv_rope(hipStream_t st, const uvx_config_t& c, const uvx_llm_weights_t* w, const uvx_llm_layer_t& L, const void* n, void* qkv,
             const int32_t* pos, int rows, int T, int QKV, int l, const InferWs* ws = nullptr) {
  const int dt = c.dtype, dh = c.llm_head_dim, Hq = c.llm_heads, Hkv = c.llm_kv_heads;
  const bool g3 = c.llm_flavor == UVX_LLM_GEMMA3;
  const float* rope = g3 && w->layer_local && w->layer_local[l] ? w->rope_cos_sin_local : w->rope_cos_sin;
  GemmDesc g = lin(n, L.wqkv, qkv, rows, QKV, c.llm_d);
  g.bias = L.bqkv;
  RC(gemm(st, dt, ws ? sk(g, *ws) : g));
  if (c.llm_qk_norm) {
    UVX_CHECK(L.q_norm && L.k_norm, UVX_ERR_INVALID, "llm: llm_qk_norm is set but a layer has no q_norm / k_norm");
    return qk_norm_rope(st, dt, qkv, L.q_norm, L.k_norm, nullptr, rope, pos, rows, T, Hq, Hkv, dh, QKV, c.rms_eps, g3 ? 1 : 0);
  }
  return rope_inplace(st, dt, qkv, rope, pos, rows, T, Hq + Hkv, dh, QKV, 0);
}

// q | k | v projection, rotary embedding and cache append of Tn new positions per sequence (rows b Tn + t at positions pos[row] -> cache
// rows t0 + t): bf16 without per-head q / k norms runs the projection and then ONE launch that rotates q / k and writes the cache rows
// (rope_kv_append_k); everything else - and tuning option 16 = 0, for A/B - runs qkv_rope and kv_append_k.  Same values either way.
int qkv_rope_append(hipStream_t st, const uvx_config_t& c, const uvx_llm_weights_t* w, const uvx_llm_layer_t& L, const InferWs& s, const void* n,
                    const int32_t* pos, int B, int Tn, int Tmax, int t0, void* ck, void* cv, int l) {
  const int dt = c.dtype, dh = c.llm_head_dim, Hq = c.llm_heads, Hkv = c.llm_kv_heads, KVD = Hkv * dh, rows = B * Tn;
  if (dt == DT_BF16 && !c.llm_qk_norm && uvx::g_options[16]) {
    GemmDesc g = lin(n, L.wqkv, s.qkv, rows, s.QKV, c.llm_d);
    g.bias = L.bqkv;
    RC(gemm(st, dt, sk(g, s)));
    const bool g3 = c.llm_flavor == UVX_LLM_GEMMA3;
    launch_rope_kv_append<bf16_t>(st, s.qkv, g3 && w->layer_local && w->layer_local[l] ? w->rope_cos_sin_local : w->rope_cos_sin, pos, ck, cv,
                                  B, Tn, Tmax, t0, Hq, Hkv, dh, s.QKV);
    UVX_LAUNCH_CHECK();
    return UVX_OK;
  }
  RC(qkv_rope(st, c, w, L, n, s.qkv, pos, rows, Tn, s.QKV, l, &s));
  const long long na = (long long)rows * (KVD / 8);
  if (dt == DT_BF16)
    hipLaunchKernelGGL(kv_append_k<bf16_t>, dim3(cdiv(na, 256)), dim3(256), 0, st, (const bf16_t*)s.qkv, (bf16_t*)ck, (bf16_t*)cv, B, Tn, Tmax, t0, s.QKV, Hq * dh, KVD);
  else
    hipLaunchKernelGGL(kv_append_k<float>, dim3(cdiv(na, 256)), dim3(256), 0, st, (const float*)s.qkv, (float*)ck, (float*)cv, B, Tn, Tmax, t0, s.QKV, Hq * dh, KVD);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

}  // namespace

extern "C" size_t uvx_kv_cache_bytes(const uvx_config_t* cfg, int32_t B, int32_t Tmax) {
  if (!cfg) return 0;
  return (size_t)cfg->llm_layers * 2 * B * Tmax * cfg->llm_kv_heads * cfg->llm_head_dim * esz(cfg->dtype);
}

extern "C" size_t uvx_llm_infer_ws_bytes(const uvx_config_t* cfg, int32_t B, int32_t T) {
  if (!cfg) return 0;
  Arena a(nullptr, 0);
  carve(a, *cfg, B, T);
  return a.off + 256;
}

extern "C" int32_t uvx_llm_prefill(void* stream, const uvx_config_t* cfg, const uvx_llm_weights_t* w, const void* inputs_embeds,
                                   const int64_t* attention_mask, int32_t B, int32_t T, void* kv_cache, int32_t Tmax,
                                   int32_t* next_pos, int32_t* kv_start, void* logits_last, void* workspace, size_t ws_bytes) {
  UVX_CHECK(cfg && w && inputs_embeds && kv_cache && next_pos && kv_start && logits_last && workspace, UVX_ERR_INVALID,
            "llm_prefill: null argument");
  const uvx_config_t& c = *cfg;
  UVX_CHECK(T >= 1 && T <= Tmax, UVX_ERR_SHAPE, "llm_prefill: prompt length %d exceeds the cache length %d", T, Tmax);
  UVX_CHECK(w->rope_len >= Tmax, UVX_ERR_SHAPE, "llm_prefill: rope table (%d) shorter than the cache (%d)", w->rope_len, Tmax);
  RC(g3_check(c, w, T));          // (Gemma-3: post norms + local rotary table present; prompts beyond the window run the WINDOWED kernels)
  hipStream_t st = (hipStream_t)stream;
  Arena a(workspace, ws_bytes);
  InferWs s = carve(a, c, B, T);
  UVX_CHECK(a.fits(), UVX_ERR_WORKSPACE, "llm_prefill: workspace %zu < %zu bytes", ws_bytes, a.off);
  const int dt = c.dtype, D = c.llm_d, M = s.M, dh = c.llm_head_dim, Hq = c.llm_heads, Hkv = c.llm_kv_heads, KVD = Hkv * dh;
  const size_t es = esz(dt);
  hipLaunchKernelGGL(mask_positions_k, dim3(B), dim3(64), 0, st, attention_mask, s.pos, kv_start, s.kvl, next_pos, T);
  UVX_LAUNCH_CHECK();
  UVX_HIP(hipMemcpyAsync(s.x, inputs_embeds, (size_t)M * D * es, hipMemcpyDeviceToDevice, st));
  if (c.llm_flavor == UVX_LLM_GEMMA) RC(scale_inplace(st, dt, s.x, (long long)M * D, gemma_normalizer(c)));   // 4.51.3: inside the model
  const size_t layer_stride = (size_t)2 * B * Tmax * KVD;  // elements
  bool n1_ready = false, n2_ready = false;      // s.n already holds this layer's input_layernorm / post_attention_layernorm (fused reduces)
  for (int l = 0; l < c.llm_layers; ++l) {
    const uvx_llm_layer_t& L = w->layers[l];
    if (!n1_ready) RC(rmsnorm_fwd(st, dt, s.x, L.ln1, s.n, nullptr, M, D, c.rms_eps, c.llm_flavor));
    {
      char* ck = at(kv_cache, l * layer_stride, dt);
      char* cv = at(kv_cache, l * layer_stride + (size_t)B * Tmax * KVD, dt);
      RC(qkv_rope_append(st, c, w, L, s, s.n, s.pos, B, T, Tmax, 0, ck, cv, l));
    }
    if (attention_needs_transposed_copies(dt)) RC(heads_transpose(st, dt, at(s.qkv, (size_t)(Hq + Hkv) * dh, dt), s.vt, B, T, s.Tp, Hkv, dh, s.QKV));
    AttnDesc ad;
    ad.q = s.qkv; ad.k = at(s.qkv, (size_t)Hq * dh, dt); ad.v = at(s.qkv, (size_t)(Hq + Hkv) * dh, dt);
    ad.vt = s.vt; ad.o = s.o; ad.lse = nullptr; ad.kv_start = kv_start; ad.kv_len = s.kvl;
    ad.B = B; ad.T = T; ad.Tp = s.Tp; ad.Hq = Hq; ad.Hkv = Hkv; ad.D = dh;
    ad.ldq = ad.ldk = ad.ldv = s.QKV; ad.ldo = s.OD; ad.causal = 1; ad.scale = attn_scale_of(c);
    ad.window = c.llm_window > 0 && T > c.llm_window && w->layer_local && w->layer_local[l] ? c.llm_window : 0;
    RC(attention_fwd(st, dt, ad));
    RC(attn_out(st, c, L, s, M, s.o, s.OD, s.x, s.x2, &n2_ready));
    RC(mlp_block(st, c, L, s, M, s.x2, s.x, n2_ready, l + 1 < c.llm_layers ? w->layers[l + 1].ln1 : nullptr, &n1_ready));
  }
  // logits of the LAST position of every sequence only (what generate() consumes)
  UVX_HIP(hipMemcpy2DAsync(s.last, (size_t)D * es, at(s.x, (size_t)(T - 1) * D, dt), (size_t)T * D * es, (size_t)D * es, B,
                           hipMemcpyDeviceToDevice, st));
  RC(rmsnorm_fwd(st, dt, s.last, w->norm, s.hn, nullptr, B, D, c.rms_eps, c.llm_flavor));
  return gemm(st, dt, lin(s.hn, w->lm_head, logits_last, B, c.vocab, D));
}

namespace {
struct ChunkWs { void *fq, *fvt, *fo; int Tfp; };
InferWs carve_chunk(Arena& a, const uvx_config_t& c, int B, int Tn, int Tf, ChunkWs& k) {
  InferWs s = carve(a, c, B, Tn);
  const size_t es = esz(c.dtype);
  k.Tfp = (Tf + 63) / 64 * 64;
  k.fq = a.take((size_t)B * Tf * s.QKV * es);
  k.fvt = a.take((size_t)B * c.llm_kv_heads * c.llm_head_dim * k.Tfp * es);
  k.fo = a.take((size_t)B * Tf * s.OD * es);
  return s;
}
}  // namespace

extern "C" size_t uvx_llm_prefill_chunk_ws_bytes(const uvx_config_t* cfg, int32_t B, int32_t Tn, int32_t cur_len) {
  if (!cfg) return 0;
  Arena a(nullptr, 0);
  ChunkWs k;
  carve_chunk(a, *cfg, B, Tn, cur_len + Tn, k);
  return a.off + 256;
}

// Prefill of Tn further tokens per sequence on top of `cur_len` cached positions (HF generate(past_key_values=...): only
// input_ids[:, cur_len:] are run).  GEMMs, norms and the MLP run on the B*Tn new rows only; attention runs the prefill
// kernel over the whole [B, cur_len + Tn] key range with the query blocks of the prefix skipped (AttnDesc::q_begin): K / V
// of the prefix are gathered from the cache into the kernel's row layout (a few MB per layer against the GBs of weights
// the layer streams), the new rows are appended to the cache first so one gather serves both.
// `all_rows`: logits of every new position [B, Tn, vocab] (what HF's forward returns for a cached call without logits_to_keep,
// ultravox_model.py:328-334) instead of the last one [B, vocab].
static int32_t prefill_chunk_impl(void* stream, const uvx_config_t* cfg, const uvx_llm_weights_t* w, const void* inputs_embeds,
                                  int32_t B, int32_t Tn, void* kv_cache, int32_t Tmax, int32_t cur_len,
                                  const int32_t* positions0, const int32_t* kv_start, void* logits_last, void* workspace,
                                  size_t ws_bytes, bool all_rows) {
  UVX_CHECK(cfg && w && inputs_embeds && kv_cache && positions0 && logits_last && workspace, UVX_ERR_INVALID,
            "llm_prefill_chunk: null argument");
  const uvx_config_t& c = *cfg;
  const int Tf = cur_len + Tn;
  UVX_CHECK(Tn >= 1 && cur_len >= 0 && Tf <= Tmax, UVX_ERR_SHAPE, "llm_prefill_chunk: %d cached + %d new positions exceed the cache length %d",
            cur_len, Tn, Tmax);
  UVX_CHECK(w->rope_len >= Tmax, UVX_ERR_SHAPE, "llm_prefill_chunk: rope table (%d) shorter than the cache (%d)", w->rope_len, Tmax);
  RC(g3_check(c, w, cur_len + Tn));   // (Gemma-3 weight check; cache + chunk beyond the window run the windowed kernels)
  hipStream_t st = (hipStream_t)stream;
  Arena a(workspace, ws_bytes);
  ChunkWs k;
  InferWs s = carve_chunk(a, c, B, Tn, Tf, k);
  UVX_CHECK(a.fits(), UVX_ERR_WORKSPACE, "llm_prefill_chunk: workspace %zu < %zu bytes", ws_bytes, a.off);
  const int dt = c.dtype, D = c.llm_d, M = s.M, dh = c.llm_head_dim, Hq = c.llm_heads, Hkv = c.llm_kv_heads, KVD = Hkv * dh;
  const size_t es = esz(dt);
  hipLaunchKernelGGL(chunk_positions_k, dim3(cdiv(M, 256)), dim3(256), 0, st, positions0, s.pos, B, Tn);
  UVX_LAUNCH_CHECK();
  UVX_HIP(hipMemcpyAsync(s.x, inputs_embeds, (size_t)M * D * es, hipMemcpyDeviceToDevice, st));
  if (c.llm_flavor == UVX_LLM_GEMMA) RC(scale_inplace(st, dt, s.x, (long long)M * D, gemma_normalizer(c)));
  UVX_HIP(hipMemsetAsync(k.fq, 0, (size_t)B * Tf * s.QKV * es, st));   // the prefix rows' (skipped) query part stays defined
  const size_t layer_stride = (size_t)2 * B * Tmax * KVD;  // elements
  bool n1_ready = false, n2_ready = false;
  for (int l = 0; l < c.llm_layers; ++l) {
    const uvx_llm_layer_t& L = w->layers[l];
    if (!n1_ready) RC(rmsnorm_fwd(st, dt, s.x, L.ln1, s.n, nullptr, M, D, c.rms_eps, c.llm_flavor));
    char* ck = at(kv_cache, l * layer_stride, dt);
    char* cv = at(kv_cache, l * layer_stride + (size_t)B * Tmax * KVD, dt);
    RC(qkv_rope_append(st, c, w, L, s, s.n, s.pos, B, Tn, Tmax, cur_len, ck, cv, l));
    const long long ng = (long long)B * Tf * (KVD / 8);
    if (dt == DT_BF16)
      hipLaunchKernelGGL(kv_gather_k<bf16_t>, dim3(cdiv(ng, 256)), dim3(256), 0, st, (const bf16_t*)ck, (const bf16_t*)cv, (bf16_t*)k.fq, B, Tf, Tmax, s.QKV, Hq * dh, KVD);
    else
      hipLaunchKernelGGL(kv_gather_k<float>, dim3(cdiv(ng, 256)), dim3(256), 0, st, (const float*)ck, (const float*)cv, (float*)k.fq, B, Tf, Tmax, s.QKV, Hq * dh, KVD);
    UVX_LAUNCH_CHECK();
    for (int b = 0; b < B; ++b)   // the new rows' queries into their place in the full-length layout
      UVX_HIP(hipMemcpy2DAsync(at(k.fq, ((size_t)b * Tf + cur_len) * s.QKV, dt), (size_t)s.QKV * es, at(s.qkv, (size_t)b * Tn * s.QKV, dt),
                               (size_t)s.QKV * es, (size_t)Hq * dh * es, Tn, hipMemcpyDeviceToDevice, st));
    if (attention_needs_transposed_copies(dt)) RC(heads_transpose(st, dt, at(k.fq, (size_t)(Hq + Hkv) * dh, dt), k.fvt, B, Tf, k.Tfp, Hkv, dh, s.QKV));
    AttnDesc ad;
    ad.q = k.fq; ad.k = at(k.fq, (size_t)Hq * dh, dt); ad.v = at(k.fq, (size_t)(Hq + Hkv) * dh, dt);
    ad.vt = k.fvt; ad.o = k.fo; ad.lse = nullptr; ad.kv_start = kv_start; ad.kv_len = nullptr;
    ad.B = B; ad.T = Tf; ad.Tp = k.Tfp; ad.Hq = Hq; ad.Hkv = Hkv; ad.D = dh;
    ad.ldq = ad.ldk = ad.ldv = s.QKV; ad.ldo = s.OD; ad.causal = 1; ad.q_begin = cur_len; ad.scale = attn_scale_of(c);
    ad.window = c.llm_window > 0 && Tf > c.llm_window && w->layer_local && w->layer_local[l] ? c.llm_window : 0;
    RC(attention_fwd(st, dt, ad));
    for (int b = 0; b < B; ++b)
      UVX_HIP(hipMemcpyAsync(at(s.o, (size_t)b * Tn * s.OD, dt), at(k.fo, ((size_t)b * Tf + cur_len) * s.OD, dt), (size_t)Tn * s.OD * es,
                             hipMemcpyDeviceToDevice, st));
    RC(attn_out(st, c, L, s, M, s.o, s.OD, s.x, s.x2, &n2_ready));
    RC(mlp_block(st, c, L, s, M, s.x2, s.x, n2_ready, l + 1 < c.llm_layers ? w->layers[l + 1].ln1 : nullptr, &n1_ready));
  }
  if (all_rows) {      // final norm and LM head on the B * Tn new rows (x2 is free after the last layer)
    RC(rmsnorm_fwd(st, dt, s.x, w->norm, s.x2, nullptr, M, D, c.rms_eps, c.llm_flavor));
    return gemm(st, dt, lin(s.x2, w->lm_head, logits_last, M, c.vocab, D));
  }
  UVX_HIP(hipMemcpy2DAsync(s.last, (size_t)D * es, at(s.x, (size_t)(Tn - 1) * D, dt), (size_t)Tn * D * es, (size_t)D * es, B,
                           hipMemcpyDeviceToDevice, st));
  RC(rmsnorm_fwd(st, dt, s.last, w->norm, s.hn, nullptr, B, D, c.rms_eps, c.llm_flavor));
  return gemm(st, dt, lin(s.hn, w->lm_head, logits_last, B, c.vocab, D));
}

extern "C" int32_t uvx_llm_prefill_chunk(void* stream, const uvx_config_t* cfg, const uvx_llm_weights_t* w, const void* inputs_embeds,
                                         int32_t B, int32_t Tn, void* kv_cache, int32_t Tmax, int32_t cur_len,
                                         const int32_t* positions0, const int32_t* kv_start, void* logits_last, void* workspace,
                                         size_t ws_bytes) {
  return prefill_chunk_impl(stream, cfg, w, inputs_embeds, B, Tn, kv_cache, Tmax, cur_len, positions0, kv_start, logits_last, workspace,
                            ws_bytes, false);
}
extern "C" int32_t uvx_llm_prefill_chunk_logits(void* stream, const uvx_config_t* cfg, const uvx_llm_weights_t* w, const void* inputs_embeds,
                                                int32_t B, int32_t Tn, void* kv_cache, int32_t Tmax, int32_t cur_len,
                                                const int32_t* positions0, const int32_t* kv_start, void* logits_all, void* workspace,
                                                size_t ws_bytes) {
  return prefill_chunk_impl(stream, cfg, w, inputs_embeds, B, Tn, kv_cache, Tmax, cur_len, positions0, kv_start, logits_all, workspace,
                            ws_bytes, true);
}

extern "C" int32_t uvx_llm_decode(void* stream, const uvx_config_t* cfg, const uvx_llm_weights_t* w, const void* token_embeds,
                                  const int32_t* positions, const int32_t* kv_start, void* kv_cache, int32_t Tmax,
                                  int32_t cur_len, int32_t B, void* logits, void* workspace, size_t ws_bytes) {
  UVX_CHECK(cfg && w && token_embeds && positions && kv_cache && logits && workspace, UVX_ERR_INVALID, "llm_decode: null argument");
  const uvx_config_t& c = *cfg;
  UVX_CHECK(cur_len >= 0 && cur_len < Tmax, UVX_ERR_SHAPE, "llm_decode: cache full (%d of %d)", cur_len, Tmax);
  hipStream_t st = (hipStream_t)stream;
  Arena a(workspace, ws_bytes);
  InferWs s = carve(a, c, B, 1);
  UVX_CHECK(a.fits(), UVX_ERR_WORKSPACE, "llm_decode: workspace %zu < %zu bytes", ws_bytes, a.off);
  const int dt = c.dtype, D = c.llm_d, dh = c.llm_head_dim, Hq = c.llm_heads, Hkv = c.llm_kv_heads, KVD = Hkv * dh;
  UVX_CHECK(dh == 64 || dh == 128 || dh == 256, UVX_ERR_UNSUPPORTED, "llm_decode: head_dim %d not supported", dh);
  RC(g3_check(c, w, 0));          // (weights / tables only: the sliding-window layers clamp their key range below)
  const size_t es = esz(dt);
  UVX_HIP(hipMemcpyAsync(s.x, token_embeds, (size_t)B * D * es, hipMemcpyDeviceToDevice, st));
  if (c.llm_flavor == UVX_LLM_GEMMA) RC(scale_inplace(st, dt, s.x, (long long)B * D, gemma_normalizer(c)));
  const size_t layer_stride = (size_t)2 * B * Tmax * KVD;
  const float scale = attn_scale_of(c);
  bool n1_ready = false, n2_ready = false;      // batches beyond 16 rows (tiled split-K linears): the norms ride in the reduce kernels
  for (int l = 0; l < c.llm_layers; ++l) {
    const uvx_llm_layer_t& L = w->layers[l];
    const bool fuse_rope_append = dt == DT_BF16 && !c.llm_qk_norm;     // (Qwen3 / Gemma-3: q_norm / k_norm + RoPE is its own kernel)
    if (fuse_rope_append) {
      // input_layernorm inside the q|k|v GEMV (B <= 2); otherwise the two launches
      GemmDesc g = lin(s.x, L.wqkv, s.qkv, B, s.QKV, D);
      g.bias = L.bqkv;
      const int rc = n1_ready ? UVX_ERR_UNSUPPORTED : gemm_skinny_rmsnorm_bf16(st, g, L.ln1, c.rms_eps, c.llm_flavor);
      if (rc == UVX_ERR_UNSUPPORTED) {
        if (!n1_ready) RC(rmsnorm_fwd(st, dt, s.x, L.ln1, s.n, nullptr, B, D, c.rms_eps, c.llm_flavor));
        g.A = s.n;
        RC(gemm(st, dt, sk(g, s)));
      } else {
        RC(rc);
      }
    } else {
      if (!n1_ready) RC(rmsnorm_fwd(st, dt, s.x, L.ln1, s.n, nullptr, B, D, c.rms_eps, c.llm_flavor));
      RC(qkv_rope(st, c, w, L, s.n, s.qkv, positions, B, 1, s.QKV, l, &s));
    }
    // Gemma-3 sliding-window layer: the new token attends to the last `window` positions = cache slots (the slots of a sequence are
    // contiguous, so the window is a clamp of the first visible slot)
    const int lo = (c.llm_window > 0 && w->layer_local && w->layer_local[l]) ? max(0, cur_len + 1 - c.llm_window) : 0;
    char* ck = at(kv_cache, l * layer_stride, dt);
    char* cv = at(kv_cache, l * layer_stride + (size_t)B * Tmax * KVD, dt);
    const long long n = (long long)B * (KVD / 8);
    const int nw = B * Hq;
    if (dt == DT_BF16) {
      const int Gall = Hq / Hkv, len = cur_len + 1;
      // query heads of a KV head per block: split the group until ~256 blocks are in flight (powers of two that divide the group)
      int hsplit = 1;
      while (hsplit * 2 <= Gall && Gall % (hsplit * 2) == 0 && B * Hkv * hsplit < 256) hsplit *= 2;
      const int G = Gall / hsplit;
      const size_t sh = sizeof(float) * (size_t)G * len;
      // the grouped attention kernel rotates q / k and appends the new k / v rows itself (option 23; Gemma-3's local layers have their own table:
      // they keep the separate launch) - otherwise rope_kv_append_k (or, with q / k norms, kv_append_k after qkv_rope) runs first
      const bool grp = sh <= 48 * 1024 && ((dh == 128 && (G == 1 || G == 2 || G == 4 || G == 8)) || (dh == 64 && (G == 1 || G == 2 || G == 4)));
      const bool in_attn = fuse_rope_append && grp && g_options[23] != 1 && c.llm_flavor != UVX_LLM_GEMMA3;
      const float* rcs = in_attn ? w->rope_cos_sin : nullptr;
      if (in_attn) {
      } else if (fuse_rope_append) {
        launch_rope_kv_append<bf16_t>(st, s.qkv, w->rope_cos_sin, positions, ck, cv, B, 1, Tmax, cur_len, Hq, Hkv, dh, s.QKV);
      } else {
        hipLaunchKernelGGL(kv_append_k<bf16_t>, dim3(cdiv(n, 256)), dim3(256), 0, st, (const bf16_t*)s.qkv, (bf16_t*)ck, (bf16_t*)cv, B, 1, Tmax, cur_len, s.QKV, Hq * dh, KVD);
      }
#define UVX_DEC(DD, GG) RC((launch_decode_grp<DD, GG>(st, sh, B * Hkv * hsplit, (const bf16_t*)s.qkv, (bf16_t*)ck, (bf16_t*)cv, (bf16_t*)s.o, kv_start, Hq, Hkv, Tmax, len, s.QKV, scale, lo, hsplit, rcs, positions)))
      if (sh <= 48 * 1024 && dh == 128 && G == 4) UVX_DEC(128, 4);
      else if (sh <= 48 * 1024 && dh == 128 && G == 8) UVX_DEC(128, 8);
      else if (sh <= 48 * 1024 && dh == 128 && G == 2) UVX_DEC(128, 2);
      else if (sh <= 48 * 1024 && dh == 128 && G == 1) UVX_DEC(128, 1);
      else if (sh <= 48 * 1024 && dh == 64 && G == 4) UVX_DEC(64, 4);
      else if (sh <= 48 * 1024 && dh == 64 && G == 2) UVX_DEC(64, 2);
      else if (sh <= 48 * 1024 && dh == 64 && G == 1) UVX_DEC(64, 1);
#undef UVX_DEC
      else if (dh == 256) hipLaunchKernelGGL((attn_decode_k<bf16_t, 256>), dim3(cdiv(nw, 4)), dim3(256), 0, st, (const bf16_t*)s.qkv, (const bf16_t*)ck, (const bf16_t*)cv, (bf16_t*)s.o, kv_start, B, Hq, Hkv, Tmax, cur_len + 1, s.QKV, scale, lo);
      else if (dh == 64) hipLaunchKernelGGL((attn_decode_k<bf16_t, 64>), dim3(cdiv(nw, 4)), dim3(256), 0, st, (const bf16_t*)s.qkv, (const bf16_t*)ck, (const bf16_t*)cv, (bf16_t*)s.o, kv_start, B, Hq, Hkv, Tmax, cur_len + 1, s.QKV, scale, lo);
      else hipLaunchKernelGGL((attn_decode_k<bf16_t, 128>), dim3(cdiv(nw, 4)), dim3(256), 0, st, (const bf16_t*)s.qkv, (const bf16_t*)ck, (const bf16_t*)cv, (bf16_t*)s.o, kv_start, B, Hq, Hkv, Tmax, cur_len + 1, s.QKV, scale, lo);
    } else {
      hipLaunchKernelGGL(kv_append_k<float>, dim3(cdiv(n, 256)), dim3(256), 0, st, (const float*)s.qkv, (float*)ck, (float*)cv, B, 1, Tmax, cur_len, s.QKV, Hq * dh, KVD);
      if (dh == 256) hipLaunchKernelGGL((attn_decode_k<float, 256>), dim3(cdiv(nw, 4)), dim3(256), 0, st, (const float*)s.qkv, (const float*)ck, (const float*)cv, (float*)s.o, kv_start, B, Hq, Hkv, Tmax, cur_len + 1, s.QKV, scale, lo);
      else if (dh == 64) hipLaunchKernelGGL((attn_decode_k<float, 64>), dim3(cdiv(nw, 4)), dim3(256), 0, st, (const float*)s.qkv, (const float*)ck, (const float*)cv, (float*)s.o, kv_start, B, Hq, Hkv, Tmax, cur_len + 1, s.QKV, scale, lo);
      else hipLaunchKernelGGL((attn_decode_k<float, 128>), dim3(cdiv(nw, 4)), dim3(256), 0, st, (const float*)s.qkv, (const float*)ck, (const float*)cv, (float*)s.o, kv_start, B, Hq, Hkv, Tmax, cur_len + 1, s.QKV, scale, lo);
    }
    UVX_LAUNCH_CHECK();
    RC(attn_out(st, c, L, s, B, s.o, s.OD, s.x, s.x2, &n2_ready));
    RC(mlp_block(st, c, L, s, B, s.x2, s.x, n2_ready, l + 1 < c.llm_layers ? w->layers[l + 1].ln1 : nullptr, &n1_ready));
  }
  RC(rmsnorm_fwd(st, dt, s.x, w->norm, s.hn, nullptr, B, D, c.rms_eps, c.llm_flavor));
  return gemm(st, dt, sk(lin(s.hn, w->lm_head, logits, B, c.vocab, D), s));
}

extern "C" int32_t uvx_greedy_select(void* stream, int32_t dtype, const void* logits, int32_t B, int32_t V, const int64_t* eos_ids,
                                     int32_t n_eos, int64_t pad, int32_t* unfinished, int64_t* next_tokens, int64_t* sequences,
                                     int64_t stride, int64_t col, const int32_t* positions0, int32_t* positions, int32_t step,
                                     int32_t* counter) {
  UVX_CHECK(logits && unfinished && next_tokens && sequences && counter && (n_eos == 0 || eos_ids) && (!positions || positions0), UVX_ERR_INVALID,
            "greedy_select: null argument");
  UVX_CHECK(B >= 0 && V > 0 && n_eos >= 0 && step >= 0 && col >= 0 && col < stride, UVX_ERR_SHAPE, "greedy_select: B=%d V=%d col=%lld stride=%lld", B, V,
            (long long)col, (long long)stride);
  if (B == 0) return UVX_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_BF16) hipLaunchKernelGGL(greedy_select_k<bf16_t>, dim3(B), dim3(1024), 0, st, (const bf16_t*)logits, V, eos_ids, n_eos, pad, unfinished,
                                           next_tokens, sequences, (long long)stride, (long long)col, positions0, positions, step, counter);
  else hipLaunchKernelGGL(greedy_select_k<float>, dim3(B), dim3(1024), 0, st, (const float*)logits, V, eos_ids, n_eos, pad, unfinished, next_tokens,
                          sequences, (long long)stride, (long long)col, positions0, positions, step, counter);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

extern "C" int32_t uvx_argmax(void* stream, int32_t dtype, const void* logits, int32_t rows, int32_t V, int64_t* out) {
  UVX_CHECK(logits && out, UVX_ERR_INVALID, "argmax: null argument");
  if (rows == 0) return UVX_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_BF16) hipLaunchKernelGGL(argmax_k<bf16_t>, dim3(rows), dim3(1024), 0, st, (const bf16_t*)logits, out, V);
  else hipLaunchKernelGGL(argmax_k<float>, dim3(rows), dim3(1024), 0, st, (const float*)logits, out, V);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}
