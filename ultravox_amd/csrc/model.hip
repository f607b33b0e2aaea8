// Host-side orchestration of the hot path behind the C ABI (include/uvx.h): which kernels run, in
// what order, on which slices of the caller's workspace.  No device allocation, no synchronisation.
#include <mutex>
#include <type_traits>
#include <unordered_map>
#include <algorithm>
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

__global__ void enc_kvlen_k(const int64_t* __restrict__ audio_lens, int32_t* __restrict__ kv_len, int B, int Te) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  // _get_feat_extract_output_lengths: (len - 1) // 2 + 1   (python floor division)
  const long long l = audio_lens[b];
  long long o = (l - 1 >= 0 ? (l - 1) / 2 : -((2 - l) / 2)) + 1;
  kv_len[b] = (int32_t)(o < 0 ? 0 : (o > Te ? Te : o));
}

__global__ void mask_range_k(const int64_t* __restrict__ mask, int32_t* __restrict__ kv_start, int32_t* __restrict__ kv_len,
                             int T) {
  __shared__ int lo, hi;
  if (threadIdx.x == 0) { lo = T; hi = 0; }
  __syncthreads();
  const int64_t* m = mask + (long long)blockIdx.x * T;
  int l = T, h = 0;
  for (int t = threadIdx.x; t < T; t += blockDim.x)
    if (m[t] != 0) { l = min(l, t); h = max(h, t + 1); }
  atomicMin(&lo, l);
  atomicMax(&hi, h);
  __syncthreads();
  if (threadIdx.x == 0) { kv_start[blockIdx.x] = lo < hi ? lo : 0; kv_len[blockIdx.x] = hi; }
}

__global__ void full_range_k(int32_t* __restrict__ kv_start, int32_t* __restrict__ kv_len, int B, int T) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) { kv_start[b] = 0; kv_len[b] = T; }
}

// ------------------------------------------------------------------ encoder
// per-layer stash of the LoRA-training forward (uvx_encoder_fwd_train -> uvx_encoder_bwd)
struct EncLayerStash {
  void *x_in, *qkv, *o, *x_mid, *pre, *t;   // t = [lora_A_q(n) | lora_A_k(n)], [M, 128] (columns 0..r-1 and 64..64+r-1)
  float* lse;
  void *bqT, *bkT;                           // lora_B^T [r, d] of this layer: every rank-r product reads rows
  void *t2, *bvT, *boT;                      // v_proj / out_proj adapters (ABI 17): t2 = [lora_A_v(n) | lora_A_o(attention output)] [M, 128]
  void* t3;                                  // fc1 / fc2 adapters (ABI 18): [lora_A_fc1(n2) | lora_A_fc2(gelu(fc1))] [M, 128]
};
struct EncWs {
  void *im2col, *c1, *x, *n, *qkv, *vt, *o, *f;
  int32_t* kvlen;
  int Te, Tp, M, Kp1;
  void* sk; size_t sk_bytes;    // inference at one or two clips: split-K scratch of the layer GEMMs (gemm.hip "Split-K"), else null
  // training only
  char* slots; size_t slot_bytes; EncLayerStash ls0;
  void *dx, *d_n, *d_o, *d_f, *d_qkv, *qT, *kT, *doT, *u, *u2;   // u2: [d v . B_v | d x_mid . B_o] [M, 128]
  void *u3, *lbT;   // u3: [d pre . B_fc1 | d x_out . B_fc2] [M, 128]; lbT: one MLP adapter's lora_B^T [r, max(d, ffn)] (transposed again in the backward)
  float *delta, *wg;   // wg: lora_wgrad scratch
  int Mp;
};
void enc_slot(Arena& a, const uvx_config_t& c, int B, int Te, EncLayerStash& s) {
  const size_t es = esz(c.dtype), M = (size_t)B * Te, d = c.enc_d;
  s.x_in = a.take(M * d * es); s.qkv = a.take(M * 3 * d * es); s.o = a.take(M * d * es); s.x_mid = a.take(M * d * es);
  s.pre = a.take(M * c.enc_ffn * es); s.t = a.take(M * 128 * es);
  s.lse = (float*)a.take(sizeof(float) * (size_t)B * c.enc_heads * Te);
  s.bqT = a.take(64 * d * es); s.bkT = a.take(64 * d * es);
  s.t2 = a.take(M * 128 * es); s.bvT = a.take(64 * d * es); s.boT = a.take(64 * d * es);
  s.t3 = a.take(M * 128 * es);
}
EncLayerStash enc_layer(const EncWs& w, int l) {
  EncLayerStash s = w.ls0;
  const size_t off = w.slot_bytes * l;
  void** ps[] = {&s.x_in, &s.qkv, &s.o, &s.x_mid, &s.pre, &s.t, (void**)&s.lse, &s.bqT, &s.bkT, &s.t2, &s.bvT, &s.boT, &s.t3};
  for (void** q : ps) if (*q) *q = (char*)*q + off;
  return s;
}
EncWs enc_carve(Arena& a, const uvx_config_t& c, int B, int F, bool train = false) {
  EncWs w = {};
  const size_t es = esz(c.dtype);
  w.Te = (F - 1) / 2 + 1;
  w.Tp = rup(w.Te, 64);
  w.M = B * w.Te;
  w.Kp1 = rup(3 * c.n_mels, 64);
  const int dh = c.enc_d / c.enc_heads;
  w.im2col = a.take((size_t)B * F * w.Kp1 * es);
  w.c1 = a.take((size_t)B * (F + 2) * c.enc_d * es);
  w.x = a.take((size_t)w.M * c.enc_d * es);
  w.n = a.take((size_t)w.M * c.enc_d * es);
  w.qkv = a.take((size_t)w.M * 3 * c.enc_d * es);
  w.vt = a.take((size_t)B * c.enc_heads * dh * w.Tp * es);
  w.o = a.take((size_t)w.M * c.enc_d * es);
  w.f = a.take((size_t)w.M * c.enc_ffn * es);
  w.kvlen = (int32_t*)a.take(sizeof(int32_t) * B);
  // one or two clips (generate() at B = 1, 2): 1500 / 3000 rows are 48 ... 144 of the 128- / 256-wide tiles on 256 CUs
  w.sk_bytes = !train && c.dtype == DT_BF16 && w.M > 64 && w.M <= 3072 ? gemm_splitk_ws_bytes(w.M, c.enc_ffn) : 0;
  w.sk = w.sk_bytes ? a.take(w.sk_bytes) : nullptr;
  if (train) {
    const size_t M = (size_t)w.M, d = c.enc_d;
    w.Mp = rup(w.M, 64);
    const size_t start = (a.off + 255) & ~(size_t)255;
    a.off = start;
    enc_slot(a, c, B, w.Te, w.ls0);
    a.off = (a.off + 255) & ~(size_t)255;
    w.slot_bytes = a.off - start;
    w.slots = a.base ? a.base + start : nullptr;
    a.off = start + w.slot_bytes * c.enc_layers;
    w.dx = a.take(M * d * es); w.d_n = a.take(M * d * es); w.d_o = a.take(M * d * es);
    w.d_f = a.take(M * c.enc_ffn * es); w.d_qkv = a.take(M * 3 * d * es);
    const size_t ht = (size_t)B * c.enc_heads * dh * w.Tp * es;
    w.qT = a.take(ht); w.kT = a.take(ht); w.doT = a.take(ht);
    w.u = a.take(M * 128 * es); w.u2 = a.take(M * 128 * es); w.u3 = a.take(M * 128 * es);
    w.lbT = a.take((size_t)64 * std::max(c.enc_d, c.enc_ffn) * es);
    w.delta = (float*)a.take(sizeof(float) * (size_t)B * c.enc_heads * w.Te);
    w.wg = (float*)a.take(sizeof(float) * (size_t)lora_wgrad_scratch_floats(w.M, std::max(c.enc_d, c.enc_ffn), 64));
  }
  return w;
}

// ------------------------------------------------------------------ projector
struct ProjWs {
  void *stacked, *xn, *h1, *a, *an, *ypre;                                        // forward stash
  void *dy2, *dyT, *anT, *w2T, *d_an, *d_a, *d_h1, *dh1T, *xnT, *w1T, *d_xn;      // backward temps
  float* dwp;                                                                     // per-block partials of the RMSNorm weight gradients (fixed-order sum)
  int J, R, Rp, C8, H, Hh, D;
};
ProjWs proj_carve(Arena& a, const uvx_config_t& c, int B, int Te) {
  ProjWs w;
  const size_t es = esz(c.dtype);
  w.J = (Te + c.stack_factor - 1) / c.stack_factor;
  w.R = B * w.J;
  w.Rp = rup(w.R, 64);
  w.C8 = c.enc_d * c.stack_factor;
  w.H = c.proj_hidden;
  w.Hh = c.proj_act == UVX_PROJ_SWIGLU ? c.proj_hidden / 2 : c.proj_hidden;     // SwiGLU halves the width, a plain activation keeps it
  w.D = c.llm_d;
  w.stacked = a.take((size_t)w.R * w.C8 * es);
  w.xn = a.take((size_t)w.R * w.C8 * es);
  w.h1 = a.take((size_t)w.R * w.H * es);
  w.a = a.take((size_t)w.R * w.Hh * es);
  w.an = c.proj_ln_mid ? a.take((size_t)w.R * w.Hh * es) : w.a;
  w.ypre = c.proj_ln_mid ? nullptr : a.take((size_t)w.R * w.D * es);
  w.dy2 = c.proj_ln_mid ? nullptr : a.take((size_t)w.R * w.D * es);
  w.dyT = a.take((size_t)w.D * w.Rp * es);
  w.anT = a.take((size_t)w.Hh * w.Rp * es);
  w.w2T = a.take((size_t)w.Hh * w.D * es);
  w.d_an = a.take((size_t)w.R * w.Hh * es);
  w.d_a = c.proj_ln_mid ? a.take((size_t)w.R * w.Hh * es) : w.d_an;
  w.d_h1 = a.take((size_t)w.R * w.H * es);
  w.dh1T = a.take((size_t)w.H * w.Rp * es);
  w.xnT = a.take((size_t)w.C8 * w.Rp * es);
  w.w1T = a.take((size_t)w.C8 * w.H * es);
  w.d_xn = a.take((size_t)w.R * w.C8 * es);
  w.dwp = (float*)a.take(sizeof(float) * (size_t)rmsnorm_bwd_dw_scratch_floats(w.R, std::max(std::max(w.C8, w.D), w.Hh)));
  return w;
}

// ------------------------------------------------------------------ LLM
struct LlmLayerStash {
  void *x_in, *qkv, *o, *x_mid, *gu;
  float* lse;
  void *t, *bqT, *bkT;   // LLM LoRA (text_model_lora_config): [lora_A_q(n) | lora_A_k(n)] [M, 128]; lora_B^T of q / k
  void *t2, *bvT, *boT;  // v_proj / o_proj adapters (ABI 17): [lora_A_v(n) | lora_A_o(attention output)] [M, 128]; lora_B^T of v / o
  void *t3, *t4;         // MLP adapters (ABI 18): t3 = [lora_A_gate(n2) | lora_A_up(n2)], t4 = [lora_A_down(act) | -]  [M, 128] each
  void* qk_raw;          // Qwen3 / Gemma-3 (llm_qk_norm): the q | k projections before q_norm / k_norm [M, (Hq + Hkv) * dh]
  void *o_pre, *m_pre;   // Gemma-3: o_proj / down_proj outputs BEFORE their post norms [M, D] (the post norms' backward needs them)
};
struct LlmWs {
  LlmLayerStash ls[1];   // layer-0 slot; slot i starts slot_bytes * i later
  size_t slot_bytes;
  char* slots;
  void *x_final, *hn, *n, *act, *vt, *logits;
  float* ce_scratch;
  int32_t* sup;          // supervised-row compaction list (sup_rows), count at [M]
  int32_t* sup_c;        // the same rows' indices among the row-compacted gradients (llm_backward, first_pos > 0)
  int32_t *kvs, *kvl;
  // backward
  void *dx, *d_hn, *d_act, *d_gu, *d_n, *d_o, *d_qkv, *qT, *kT, *doT;
  float* delta;
  float* dkv_part;
  void *lu, *lu2; // LoRA backward: u = [dq . B_q | dk . B_k] [M, 128]; lu2 = [dv . B_v | d (o_proj output) . B_o]
  void *lu3, *lu4, *lbT;   // MLP adapters: lu3 = [d gate . B_g | d up . B_u], lu4 = [d (down output) . B_d | -]; lbT = one adapter's lora_B^T [r, max(D, I)]
  float* lwg;    // lora_wgrad scratch
  void *wt[2], *head_t;   // llm_wt_stream: two alternating sets of one layer's transposed weights, and lm_head^T
  int M, Tp, QKV, OD;
};
// elements of one layer's four transposed matrices (wqkv_t | wo_t | wgu_t | wd_t, in this order)
inline size_t layer_wt_elems(const uvx_config_t& c) {
  const size_t QKV = (size_t)(c.llm_heads + 2 * c.llm_kv_heads) * c.llm_head_dim, OD = (size_t)c.llm_heads * c.llm_head_dim;
  return (QKV + OD + 3 * (size_t)c.llm_inter) * c.llm_d;
}
// lora_wgrad scratch of the LLM's adapters: the widest adapted linear (hidden, heads * head_dim or the MLP width)
inline long long llm_wg_floats(const uvx_config_t& c, int M) {
  return lora_wgrad_scratch_floats(M, std::max(std::max(c.llm_d, c.llm_heads * c.llm_head_dim), c.llm_inter), 64);
}
void llm_slot(Arena& a, const uvx_config_t& c, int B, int T, LlmLayerStash& s) {
  const size_t es = esz(c.dtype);
  const size_t M = (size_t)B * T;
  const int QKV = (c.llm_heads + 2 * c.llm_kv_heads) * c.llm_head_dim;
  s.x_in = a.take(M * c.llm_d * es);
  s.qkv = a.take(M * QKV * es);
  s.o = a.take(M * c.llm_heads * c.llm_head_dim * es);
  s.x_mid = a.take(M * c.llm_d * es);
  s.gu = a.take(M * 2 * c.llm_inter * es);
  s.lse = (float*)a.take(sizeof(float) * (size_t)B * c.llm_heads * T);
  s.t = a.take(M * 128 * es);
  s.bqT = a.take((size_t)64 * c.llm_heads * c.llm_head_dim * es);
  s.bkT = a.take((size_t)64 * c.llm_kv_heads * c.llm_head_dim * es);
  s.t2 = a.take(M * 128 * es);
  s.bvT = a.take((size_t)64 * c.llm_kv_heads * c.llm_head_dim * es);
  s.boT = a.take((size_t)64 * c.llm_d * es);
  s.t3 = a.take(M * 128 * es);
  s.t4 = a.take(M * 128 * es);
  s.qk_raw = a.take(c.llm_qk_norm ? M * (c.llm_heads + c.llm_kv_heads) * c.llm_head_dim * es : 0);
  s.o_pre = a.take(c.llm_flavor == UVX_LLM_GEMMA3 ? M * c.llm_d * es : 0);
  s.m_pre = a.take(c.llm_flavor == UVX_LLM_GEMMA3 ? M * c.llm_d * es : 0);
}
LlmWs llm_carve(Arena& a, const uvx_config_t& c, int B, int T, int save) {
  LlmWs w;
  const size_t es = esz(c.dtype);
  w.M = B * T;
  w.Tp = rup(T, 64);
  w.QKV = (c.llm_heads + 2 * c.llm_kv_heads) * c.llm_head_dim;
  w.OD = c.llm_heads * c.llm_head_dim;
  const size_t M = (size_t)w.M;
  // layer slots: `n_slots` identical records laid out back to back
  const int n_slots = save ? c.llm_layers : 2;
  const size_t start = (a.off + 255) & ~(size_t)255;
  a.off = start;
  llm_slot(a, c, B, T, w.ls[0]);
  a.off = (a.off + 255) & ~(size_t)255;
  w.slot_bytes = a.off - start;
  w.slots = a.base ? a.base + start : nullptr;
  a.off = start + w.slot_bytes * n_slots;
  w.x_final = a.take(M * c.llm_d * es);
  w.hn = a.take(M * c.llm_d * es);
  w.n = a.take(M * c.llm_d * es);
  w.act = a.take(M * c.llm_inter * es);
  w.vt = a.take((size_t)B * c.llm_kv_heads * c.llm_head_dim * w.Tp * es);
  w.logits = a.take(M * c.vocab * es);
  w.lbT = a.take((size_t)64 * std::max(c.llm_d, c.llm_inter) * es);
  w.ce_scratch = (float*)a.take(sizeof(float) * (2 + M));
  w.sup = (int32_t*)a.take(sizeof(int32_t) * (M + 1));
  w.sup_c = (int32_t*)a.take(sizeof(int32_t) * (M + 1));
  w.kvs = (int32_t*)a.take(sizeof(int32_t) * B);
  w.kvl = (int32_t*)a.take(sizeof(int32_t) * B);
  if (save) {
    w.dx = a.take(M * c.llm_d * es);
    w.d_hn = a.take(M * c.llm_d * es);
    w.d_act = a.take(M * c.llm_inter * es);
    w.d_gu = a.take(M * 2 * c.llm_inter * es);
    w.d_n = a.take(M * c.llm_d * es);
    w.d_o = a.take(M * w.OD * es);
    w.d_qkv = a.take(M * w.QKV * es);
    w.qT = a.take((size_t)B * c.llm_heads * c.llm_head_dim * w.Tp * es);
    w.kT = a.take((size_t)B * c.llm_kv_heads * c.llm_head_dim * w.Tp * es);
    w.doT = a.take((size_t)B * c.llm_heads * c.llm_head_dim * w.Tp * es);
    w.delta = (float*)a.take(sizeof(float) * (size_t)B * c.llm_heads * T);
    w.dkv_part = (float*)a.take(sizeof(float) * 2 * M * w.OD);
    w.lu = a.take(M * 128 * es); w.lu2 = a.take(M * 128 * es); w.lu3 = a.take(M * 128 * es); w.lu4 = a.take(M * 128 * es);
    w.lwg = (float*)a.take(sizeof(float) * (size_t)llm_wg_floats(c, w.M));
    w.wt[0] = a.take(c.llm_wt_stream ? layer_wt_elems(c) * es : 0);
    w.wt[1] = a.take(c.llm_wt_stream ? layer_wt_elems(c) * es : 0);
    w.head_t = a.take(c.llm_wt_stream ? (size_t)c.vocab * c.llm_d * es : 0);
  }
  return w;
}
// stash record of layer l (slot l when saving, slot l&1 otherwise)
LlmLayerStash llm_layer(const LlmWs& w, int slot) {
  LlmLayerStash s = w.ls[0];
  const size_t d = w.slot_bytes * slot;
  s.x_in = (char*)s.x_in + d; s.qkv = (char*)s.qkv + d; s.o = (char*)s.o + d;
  s.x_mid = (char*)s.x_mid + d; s.gu = (char*)s.gu + d; s.lse = (float*)((char*)s.lse + d);
  s.t = (char*)s.t + d; s.bqT = (char*)s.bqT + d; s.bkT = (char*)s.bkT + d; s.qk_raw = (char*)s.qk_raw + d;
  s.t2 = (char*)s.t2 + d; s.bvT = (char*)s.bvT + d; s.boT = (char*)s.boT + d;
  s.t3 = (char*)s.t3 + d; s.t4 = (char*)s.t4 + d;
  s.o_pre = (char*)s.o_pre + d; s.m_pre = (char*)s.m_pre + d;
  return s;
}

// What uvx_llm_fwd_train left in a workspace (host-side note keyed by the workspace address): whether the last layer's stash is
// row-compacted.  uvx_llm_bwd_train re-derives that from tuning option 3; if the option changed in between it would misread the
// stash silently - now it is an error.
std::mutex g_pair_mu;
std::unordered_map<const void*, bool> g_pair_compact;
void note_pair(const void* ws, bool compact) { std::lock_guard<std::mutex> l(g_pair_mu); g_pair_compact[ws] = compact; }
int check_pair(const void* ws, bool compact) {
  std::lock_guard<std::mutex> l(g_pair_mu);
  auto it = g_pair_compact.find(ws);
  UVX_CHECK(it == g_pair_compact.end() || it->second == compact, UVX_ERR_INVALID,
            "llm_bwd_train: the forward pass left a %s last-layer stash in this workspace, the backward expects %s (uvx_set_option(3, ..) "
            "changed between uvx_llm_fwd_train and uvx_llm_bwd_train)", it->second ? "row-compacted" : "full-row", compact ? "row-compacted" : "full-row");
  return UVX_OK;
}

// Timing probe (tuning option 15, default 0): bit mask of kernel classes NOT launched - the step's results are then garbage, its
// time says what that class costs inside the overlapped schedule (bench.py --opt 15=<mask>; never set by the product).
// 1 LLM attention backward, 2 LLM attention forward, 4 SwiGLU backward, 8 RMSNorm backward, 16 RMSNorm forward, 32 RoPE forward,
// 64 encoder attention, 128 encoder LayerNorm
inline bool probe_skip(int bit) { return (g_options[15] & bit) != 0; }

// Batch slice [b0, b0 + nb) of the carved LLM workspace: every buffer is batch-major at the top level ([B*T, ld] rows or
// [B, ...]), so a slice is the same record with its pointers advanced.  Used by the two-stream schedule below.
LlmWs llm_view(const LlmWs& w, const uvx_config_t& c, int b0, int nb, int T) {
  LlmWs v = w;
  if (b0 == 0 && nb * T == w.M) return v;
  const size_t es = esz(c.dtype), r0 = (size_t)b0 * T, D = c.llm_d, I = c.llm_inter;
  auto adv = [](auto& p, size_t bytes) { if (p) p = (typename std::remove_reference<decltype(p)>::type)((char*)p + bytes); };
  v.M = nb * T;
  LlmLayerStash& s = v.ls[0];
  adv(s.x_in, r0 * D * es); adv(s.qkv, r0 * w.QKV * es); adv(s.o, r0 * w.OD * es); adv(s.x_mid, r0 * D * es);
  adv(s.gu, r0 * 2 * I * es); adv(s.lse, sizeof(float) * (size_t)b0 * c.llm_heads * T); adv(s.t, r0 * 128 * es);
  if (c.llm_qk_norm) adv(s.qk_raw, r0 * (c.llm_heads + c.llm_kv_heads) * c.llm_head_dim * es);
  if (c.llm_flavor == UVX_LLM_GEMMA3) { adv(s.o_pre, r0 * D * es); adv(s.m_pre, r0 * D * es); }
  adv(v.x_final, r0 * D * es); adv(v.hn, r0 * D * es); adv(v.n, r0 * D * es); adv(v.act, r0 * I * es);
  adv(v.vt, (size_t)b0 * c.llm_kv_heads * c.llm_head_dim * w.Tp * es);
  adv(v.logits, r0 * c.vocab * es);
  adv(v.kvs, sizeof(int32_t) * (size_t)b0); adv(v.kvl, sizeof(int32_t) * (size_t)b0);
  adv(v.dx, r0 * D * es); adv(v.d_hn, r0 * D * es); adv(v.d_act, r0 * I * es); adv(v.d_gu, r0 * 2 * I * es);
  adv(v.d_n, r0 * D * es); adv(v.d_o, r0 * w.OD * es); adv(v.d_qkv, r0 * w.QKV * es);
  adv(v.qT, (size_t)b0 * c.llm_heads * c.llm_head_dim * w.Tp * es);
  adv(v.kT, (size_t)b0 * c.llm_kv_heads * c.llm_head_dim * w.Tp * es);
  adv(v.doT, (size_t)b0 * c.llm_heads * c.llm_head_dim * w.Tp * es);
  adv(v.delta, sizeof(float) * (size_t)b0 * c.llm_heads * T);
  adv(v.dkv_part, sizeof(float) * 2 * r0 * w.OD);
  return v;
}

// Multi-stream schedule (tuning option 11 = number of chains, default 1 = one chain on the caller's stream): the batch is cut into slices whose layer chains are
// independent (frozen LLM: no weight gradient couples them); they run on the caller's stream and on side streams.  Every kernel
// of a chain depends on its predecessor, so on ONE stream the tail of each GEMM (a partly filled last round of tiles: 1120
// tiles = 4.4 rounds of 256 CUs at N = 28672, 560 = 2.2 at N = 14336) and every HBM-bound elementwise kernel leave CUs
// idle; with two chains in flight the other half's kernel takes those CUs.  Same kernels on the same rows: results are
// bit-identical to the one-stream schedule.  Fork / join by events (legal under stream capture as well).
struct Fork {
  hipStream_t side[3] = {nullptr, nullptr, nullptr};
  hipEvent_t e_fork = nullptr, e_join[3] = {nullptr, nullptr, nullptr};
  bool ok = false;
};
Fork* fork_for_device() {
  static Fork forks[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  Fork& f = forks[dev];
  // One set of side streams / events per DEVICE, created once (under a lock: two host threads may make their first call together).
  // The set is shared by every call on the device: calls that use it must be issued from one stream at a time (the trainer's
  // usage); a partly failed creation is torn down so that a retry starts clean.
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  if (!f.ok) {
    bool good = hipEventCreateWithFlags(&f.e_fork, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < 3 && good; ++i)
      good = hipStreamCreateWithFlags(&f.side[i], hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&f.e_join[i], hipEventDisableTiming) == hipSuccess;
    if (!good) {
      if (f.e_fork) (void)hipEventDestroy(f.e_fork);
      for (int i = 0; i < 3; ++i) {
        if (f.side[i]) (void)hipStreamDestroy(f.side[i]);
        if (f.e_join[i]) (void)hipEventDestroy(f.e_join[i]);
      }
      f = Fork{};
      return nullptr;
    }
    f.ok = true;
  }
  return &f;
}
// llm_wt_stream: the side stream that transposes layer l - 1's weights while layer l is differentiated, and its events
struct WtStream {
  hipStream_t side = nullptr;
  hipEvent_t e_start = nullptr, e_head = nullptr, e_ready[2] = {nullptr, nullptr}, e_free[2] = {nullptr, nullptr};
  bool ok = false;
};
WtStream* wt_stream_for_device() {
  static WtStream all[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  WtStream& f = all[dev];
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  if (!f.ok) {
    hipEvent_t* ev[6] = {&f.e_start, &f.e_head, &f.e_ready[0], &f.e_ready[1], &f.e_free[0], &f.e_free[1]};
    bool good = true;
    for (hipEvent_t* e : ev) good = good && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
    good = good && hipStreamCreateWithFlags(&f.side, hipStreamNonBlocking) == hipSuccess;
    if (!good) {      // tear down what exists: a retry starts clean
      for (hipEvent_t* e : ev)
        if (*e) (void)hipEventDestroy(*e);
      if (f.side) (void)hipStreamDestroy(f.side);
      f = WtStream{};
      return nullptr;
    }
    f.ok = true;
  }
  return &f;
}
// The chains of one call: batch slices [b0[i], b0[i + 1]) with their workspace views and streams (chain 0 = the caller's).
// Option 11 = number of chains (2 by default, up to 4; 0 / 1 = one chain); a chain needs at least one sequence.
struct Chains {
  int n = 1;
  int b0[5] = {0, 0, 0, 0, 0};
  LlmWs v[4];
  hipStream_t st[4];
  Fork* fk = nullptr;
};
Chains make_chains(hipStream_t st, const LlmWs& s, const uvx_config_t& c, int B, int T, bool allowed) {
  Chains ch;
  int want = g_options[11] < 2 ? 1 : (g_options[11] > 4 ? 4 : g_options[11]);
  if (want > B) want = B;
  ch.fk = (allowed && want >= 2) ? fork_for_device() : nullptr;
  ch.n = ch.fk ? want : 1;
  for (int i = 0; i <= ch.n; ++i) ch.b0[i] = (int)(((long long)B * i + ch.n - 1) / ch.n);   // sizes differ by at most one, larger first
  for (int i = 0; i < ch.n; ++i) {
    ch.v[i] = ch.n == 1 ? s : llm_view(s, c, ch.b0[i], ch.b0[i + 1] - ch.b0[i], T);
    ch.st[i] = i == 0 ? st : ch.fk->side[i - 1];
  }
  return ch;
}
int chains_fork(const Chains& ch) {
  if (ch.n < 2) return UVX_OK;
  UVX_HIP(hipEventRecord(ch.fk->e_fork, ch.st[0]));
  for (int i = 1; i < ch.n; ++i) UVX_HIP(hipStreamWaitEvent(ch.st[i], ch.fk->e_fork, 0));
  return UVX_OK;
}
int chains_join(const Chains& ch) {
  for (int i = 1; i < ch.n; ++i) {
    UVX_HIP(hipEventRecord(ch.fk->e_join[i - 1], ch.st[i]));
    UVX_HIP(hipStreamWaitEvent(ch.st[0], ch.fk->e_join[i - 1], 0));
  }
  return UVX_OK;
}

int check_cfg(const uvx_config_t* c) {
  UVX_CHECK(c != nullptr, UVX_ERR_INVALID, "null config");
  UVX_CHECK(c->dtype == DT_BF16 || c->dtype == DT_F32, UVX_ERR_INVALID, "bad dtype %d", c->dtype);
  UVX_CHECK(c->llm_flavor >= UVX_LLM_LLAMA && c->llm_flavor <= UVX_LLM_GEMMA3, UVX_ERR_INVALID, "bad llm_flavor %d", c->llm_flavor);
  UVX_CHECK(c->llm_act >= UVX_ACT_SILU && c->llm_act <= UVX_ACT_GELU_ERF && (c->llm_flavor != UVX_LLM_LLAMA) == (c->llm_act != UVX_ACT_SILU),
            UVX_ERR_INVALID, "llm_act %d does not fit llm_flavor %d (Llama: SiLU; Gemma: tanh- or erf-GELU)", c->llm_act, c->llm_flavor);
  UVX_CHECK(c->llm_wt_stream == 0 || c->llm_wt_stream == 1, UVX_ERR_INVALID, "llm_wt_stream %d: 0 or 1", c->llm_wt_stream);
  UVX_CHECK(c->llm_qk_norm == 0 || (c->llm_qk_norm == 1 && c->llm_flavor != UVX_LLM_GEMMA), UVX_ERR_INVALID,
            "llm_qk_norm %d: 0 or 1 (Qwen3: Llama-flavoured; Gemma-3: Gemma-flavoured)", c->llm_qk_norm);
  UVX_CHECK((c->llm_flavor == UVX_LLM_GEMMA3) == (c->llm_qk_norm == 1 && c->llm_flavor == UVX_LLM_GEMMA3) && c->llm_attn_scale >= 0.f &&
            c->llm_window >= 0, UVX_ERR_INVALID, "Gemma-3 needs llm_qk_norm = 1; llm_attn_scale / llm_window must not be negative");
  return UVX_OK;
}

// [3P] transformers 4.51.3 GemmaModel.forward: normalizer = torch.tensor(hidden_size ** 0.5, dtype=hidden_states.dtype) -
// the square root is ROUNDED to the model dtype before it multiplies (55.5 for hidden_size 3072 in bf16)
float gemma_normalizer(const uvx_config_t& c) {
  const float n = sqrtf((float)c.llm_d);
  return c.dtype == DT_BF16 ? bf2f(f2bf(n)) : n;
}

GemmDesc lin(const void* A, const void* W, void* C, int M, int N, int K) {
  GemmDesc g;
  g.A = A; g.B = W; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N;
  return g;
}
// The dgrad of a frozen linear y = x . W^T (W [N_out, N_in]):  d x [M, N_in] = d y [M, N_out] . W.  With the transposed copy Wt [N_in, N_out]
// it is the NT problem lin(d y, Wt, ..); without one (Wt == NULL; bf16, round 6) the NN form reads W as it lies (GemmDesc::b_kn: B [K, N],
// row stride N) - bit-identical, and the copy (16 GB of them for Llama-3-8B) need not exist.
GemmDesc lin_dgrad(const void* dY, const void* Wt, const void* W, void* dX, int M, int N_in, int N_out) {
  GemmDesc g = lin(dY, Wt ? Wt : W, dX, M, N_in, N_out);
  if (!Wt) { g.b_kn = 1; g.ldb = N_in; }
  return g;
}

}  // namespace

// =====================================================================================
extern "C" int32_t uvx_logmel(void* stream, const float* pcm, const float* window, const float* tw_cos,
                              const float* tw_sin, const float* mel_fb, float* out, float* scratch, int32_t B, int32_t L,
                              int32_t n_mels, int32_t F_stride) {
  return uvx::logmel((hipStream_t)stream, pcm, window, tw_cos, tw_sin, mel_fb, out, scratch, B, L, n_mels, F_stride);
}

extern "C" size_t uvx_encoder_ws_bytes(const uvx_config_t* cfg, int32_t B, int32_t F) {
  if (!cfg) return 0;
  Arena a(nullptr, 0);
  enc_carve(a, *cfg, B, F);
  return a.off + 256;
}

static GemmDesc enc_sk(GemmDesc g, const EncWs& s) { g.splitk_ws = s.sk; g.splitk_ws_bytes = s.sk_bytes; return g; }
static int enc_forward(hipStream_t st, const uvx_config_t& c, const uvx_encoder_weights_t* w, const uvx_encoder_lora_t* lora,
                       const void* mel, int mel_is_f32, const int64_t* audio_lens, int B, int F, void* out, void* workspace,
                       size_t ws_bytes) {
  const bool train = lora != nullptr;
  // ultravox_model.py:874-878: the mel length may not exceed max_source_positions * conv strides
  UVX_CHECK(F <= c.enc_max_pos * 2, UVX_ERR_SHAPE,
            "Whisper expects the mel input features to be of length %d or less, but found %d", c.enc_max_pos * 2, F);
  UVX_CHECK(c.enc_d % c.enc_heads == 0, UVX_ERR_SHAPE, "encoder: d=%d not divisible by heads=%d", c.enc_d, c.enc_heads);
  UVX_CHECK(c.enc_block == 0 || (c.enc_max_pos * 2) % c.enc_block == 0, UVX_ERR_SHAPE,
            "audio_latency_block_size %d must divide %d evenly.", c.enc_block, c.enc_max_pos * 2);
  if (train) RC(lora_check(lora, c.enc_layers, nullptr, "encoder LoRA"));
  for (int l = 0; train && l < c.enc_layers; ++l)
    UVX_CHECK(!lora->layers[l].u.a, UVX_ERR_INVALID, "encoder LoRA: layer %d has an up_proj adapter (the Whisper MLP is fc1 / fc2: g / d)", l);
  if (B == 0 || F == 0) return UVX_OK;
  Arena a(workspace, ws_bytes);
  EncWs s = enc_carve(a, c, B, F, train);
  UVX_CHECK(a.fits(), UVX_ERR_WORKSPACE, "encoder_fwd: workspace %zu < %zu bytes", ws_bytes, a.off);
  const int dt = c.dtype, d = c.enc_d, Te = s.Te, M = s.M, dh = d / c.enc_heads;
  const size_t es = esz(dt);

  // conv1 + GELU (ultravox_model.py:893) as im2col + GEMM, written time-major into a buffer with one
  // zero frame before and after each clip so that conv2 (k3, s2, p1) reads 3 consecutive frames as ONE
  // contiguous K = 3d row: A row t = frames 2t-1..2t+1, row stride 2d.
  RC(im2col_conv1(st, dt, mel, mel_is_f32, s.im2col, B, c.n_mels, F, F, s.Kp1));
  RC(fill_zero(st, s.c1, (long long)B * (F + 2) * d * es));
  {
    GemmDesc g = lin(s.im2col, w->conv1_w, at(s.c1, d, dt), F, d, s.Kp1);
    g.bias = w->conv1_b; g.act = 1; g.batch = B;
    g.sA = (long long)F * s.Kp1; g.sC = (long long)(F + 2) * d;
    RC(gemm(st, dt, g));
  }
  void* x = train ? enc_layer(s, 0).x_in : s.x;   // the running hidden state (training: lives in the layer stashes)
  {  // conv2 + GELU (:894), permute (:896), + embed_positions[:Te] (:897-899)
    GemmDesc g = lin(s.c1, w->conv2_w, x, Te, d, 3 * d);
    g.lda = 2 * d; g.bias = w->conv2_b; g.act = 1; g.batch = B;
    g.sA = (long long)(F + 2) * d; g.sC = (long long)Te * d;
    g.residual = w->pos; g.ldr = d; g.sR = 0;
    RC(gemm(st, dt, enc_sk(g, s)));
  }
  // key padding mask from audio_len (:915-926)
  const int32_t* kvlen = nullptr;
  if (audio_lens) {
    hipLaunchKernelGGL(enc_kvlen_k, dim3(cdiv(B, 64)), dim3(64), 0, st, audio_lens, s.kvlen, B, Te);
    UVX_LAUNCH_CHECK();
    kvlen = s.kvlen;
  }
  const float qscale = 1.0f / sqrtf((float)dh);
  for (int l = 0; l < c.enc_layers; ++l) {
    const uvx_enc_layer_t& L = w->layers[l];
    EncLayerStash S = train ? enc_layer(s, l) : EncLayerStash{};
    void* qkv = train ? S.qkv : s.qkv;
    void* o = train ? S.o : s.o;
    void* x_mid = train ? S.x_mid : x;            // inference: the residual stream is updated in place
    void* x_out = !train ? x : (l + 1 < c.enc_layers ? enc_layer(s, l + 1).x_in : s.x);
    if (!probe_skip(128)) RC(layernorm_fwd(st, dt, x, L.ln1_w, L.ln1_b, s.n, M, d, c.ln_eps));
    {
      GemmDesc g = lin(s.n, L.wqkv, qkv, M, 3 * d, d);
      g.bias = L.bqkv;
      RC(gemm(st, dt, enc_sk(g, s)));
    }
    if (train) {
      // peft LoRA on q_proj / k_proj: result += lora_B(lora_A(x)) * scaling (and q carries Whisper's head_dim^-0.5,
      // folded into wqkv at pack time).  Rank-r products on the VALU (lora.hip): HBM-bound, no padding to an MFMA tile.
      // (Round 6, tried and removed: the two up-projections as terms of the q|k|v GEMM's whole-line epilogue - and of its dgrad's - instead
      //  of read-modify-write passes over qkv / d n.  Bit-identical, 96 launches fewer per step and 0.25 ms per step SLOWER: the terms'
      //  loads sit in an epilogue nothing overlaps, and their registers cost the 256-row tile its spill-free budget - profiles/r06_flavours.txt.)
      const uvx_enc_lora_layer_t& R = lora->layers[l];
      const int r = lora->r;
      if (R.q.a && R.k.a) {      // the default target_modules: the pair in one launch each
        RC(lora_transpose2(st, dt, R.q.b, S.bqT, d, R.k.b, S.bkT, d, r));
        RC(lora_down2(st, dt, s.n, s.n, d, R.q.a, R.k.a, S.t, at(S.t, 64, dt), 128, M, d, r, 1.0f, 1.0f));
        RC(lora_up2(st, dt, S.t, at(S.t, 64, dt), 128, S.bqT, S.bkT, qkv, at(qkv, d, dt), 3 * d, M, d, d, r, lora->scaling * qscale, lora->scaling));
      } else {
        if (R.q.a) RC(lora_apply(st, dt, s.n, d, R.q, S.bqT, S.t, qkv, 3 * d, M, d, d, r, lora->scaling * qscale));
        if (R.k.a) RC(lora_apply(st, dt, s.n, d, R.k, S.bkT, at(S.t, 64, dt), at(qkv, d, dt), 3 * d, M, d, d, r, lora->scaling));
      }
      // v_proj (target_modules beyond the default; ABI 17): the same product into the v columns
      if (R.v.a) RC(lora_apply(st, dt, s.n, d, R.v, S.bvT, S.t2, at(qkv, 2 * d, dt), 3 * d, M, d, d, r, lora->scaling));
    }
    if (attention_needs_transposed_copies(dt)) RC(heads_transpose(st, dt, at(qkv, 2 * d, dt), s.vt, B, Te, s.Tp, c.enc_heads, dh, 3 * d));
    AttnDesc ad;
    ad.q = qkv; ad.k = at(qkv, d, dt); ad.v = at(qkv, 2 * d, dt); ad.vt = s.vt; ad.o = o; ad.lse = train ? S.lse : nullptr;
    ad.kv_len = kvlen; ad.B = B; ad.T = Te; ad.Tp = s.Tp; ad.Hq = c.enc_heads; ad.Hkv = c.enc_heads; ad.D = dh;
    ad.ldq = ad.ldk = ad.ldv = 3 * d; ad.ldo = d; ad.causal = 0; ad.block = c.enc_block;
    ad.scale = 1.0f;  // q_proj is pre-scaled by head_dim^-0.5 at pack time (exact in bf16 for dh = 64)
    if (!probe_skip(64)) RC(attention_fwd(st, dt, ad));
    {
      GemmDesc g = lin(o, L.wo, x_mid, M, d, d);
      g.bias = L.bo; g.residual = x; g.ldr = d;
      RC(gemm(st, dt, enc_sk(g, s)));
    }
    // out_proj adapter: x_mid += lora_B(lora_A(attention output)) * scaling
    if (train && lora->layers[l].o.a) RC(lora_apply(st, dt, o, d, lora->layers[l].o, S.boT, at(S.t2, 64, dt), x_mid, d, M, d, d, lora->r, lora->scaling));
    if (!probe_skip(128)) RC(layernorm_fwd(st, dt, x_mid, L.ln2_w, L.ln2_b, s.n, M, d, c.ln_eps));
    if (train) {   // keep the fc1 pre-activation for the GELU backward
      GemmDesc g = lin(s.n, L.fc1_w, S.pre, M, c.enc_ffn, d);
      g.bias = L.fc1_b;
      // bf16: the GELU runs in the GEMM's epilogue, which writes the pre-activation AND the activation (act 2; tuning option 21 = 1: the
      // separate gelu_fwd pass of rounds 3-5 - bit-identical)
      const uvx_lora_proj_t& A1 = lora->layers[l].g;      // fc1 adapter (ABI 18): joins the pre-activation, so the GELU runs after it
      const bool fused = dt == DT_BF16 && g_options[21] != 1 && !A1.a;
      if (fused) { g.act = 2; g.C2 = s.f; g.ldc2 = c.enc_ffn; }
      RC(gemm(st, dt, g));
      if (A1.a) RC(lora_apply(st, dt, s.n, d, A1, s.lbT, S.t3, S.pre, c.enc_ffn, M, d, c.enc_ffn, lora->r, lora->scaling));
      if (!fused) RC(gelu_fwd(st, dt, S.pre, s.f, (long long)M * c.enc_ffn));
    } else {
      GemmDesc g = lin(s.n, L.fc1_w, s.f, M, c.enc_ffn, d);
      g.bias = L.fc1_b; g.act = 1;
      RC(gemm(st, dt, enc_sk(g, s)));
    }
    {
      GemmDesc g = lin(s.f, L.fc2_w, x_out, M, d, c.enc_ffn);
      g.bias = L.fc2_b; g.residual = x_mid; g.ldr = d;
      RC(gemm(st, dt, enc_sk(g, s)));
    }
    if (train && lora->layers[l].d.a)      // fc2 adapter
      RC(lora_apply(st, dt, s.f, c.enc_ffn, lora->layers[l].d, s.lbT, at(S.t3, 64, dt), x_out, d, M, c.enc_ffn, d, lora->r, lora->scaling));
    x = x_out;
  }
  RC(layernorm_fwd(st, dt, x, w->lnf_w, w->lnf_b, out, M, d, c.ln_eps));  // :980
  return UVX_OK;
}

extern "C" int32_t uvx_encoder_fwd(void* stream, const uvx_config_t* cfg, const uvx_encoder_weights_t* w, const void* mel,
                                   int32_t mel_is_f32, const int64_t* audio_lens, int32_t B, int32_t F, void* out,
                                   void* workspace, size_t ws_bytes) {
  RC(check_cfg(cfg));
  UVX_CHECK(w && mel && out && workspace, UVX_ERR_INVALID, "encoder_fwd: null argument");
  return enc_forward((hipStream_t)stream, *cfg, w, nullptr, mel, mel_is_f32, audio_lens, B, F, out, workspace, ws_bytes);
}

extern "C" size_t uvx_encoder_train_ws_bytes(const uvx_config_t* cfg, int32_t B, int32_t F) {
  if (!cfg) return 0;
  Arena a(nullptr, 0);
  enc_carve(a, *cfg, B, F, true);
  return a.off + 256;
}

extern "C" int32_t uvx_encoder_fwd_train(void* stream, const uvx_config_t* cfg, const uvx_encoder_weights_t* w,
                                         const uvx_encoder_lora_t* lora, const void* mel, int32_t mel_is_f32,
                                         const int64_t* audio_lens, int32_t B, int32_t F, void* out, void* workspace,
                                         size_t ws_bytes) {
  RC(check_cfg(cfg));
  UVX_CHECK(w && lora && mel && out && workspace, UVX_ERR_INVALID, "encoder_fwd_train: null argument");
  return enc_forward((hipStream_t)stream, *cfg, w, lora, mel, mel_is_f32, audio_lens, B, F, out, workspace, ws_bytes);
}

// Backward of the LoRA-adapted encoder: d out [B, Te, d] -> gradients of lora_A / lora_B of q_proj and k_proj in every
// layer (everything else is frozen: apply_lora, ultravox_model.py:690-709).  Walks the stash of uvx_encoder_fwd_train.
extern "C" int32_t uvx_encoder_bwd(void* stream, const uvx_config_t* cfg, const uvx_encoder_weights_t* w,
                                   const uvx_encoder_lora_t* lora, const void* d_out, const int64_t* audio_lens, int32_t B,
                                   int32_t F, const uvx_encoder_lora_grads_t* grads, void* workspace, size_t ws_bytes) {
  RC(check_cfg(cfg));
  UVX_CHECK(w && lora && d_out && grads && grads->layers && workspace, UVX_ERR_INVALID, "encoder_bwd: null argument");
  const uvx_config_t& c = *cfg;
  RC(lora_check(lora, c.enc_layers, grads, "encoder_bwd"));
  hipStream_t st = (hipStream_t)stream;
  if (B == 0 || F == 0) return UVX_OK;
  Arena a(workspace, ws_bytes);
  EncWs s = enc_carve(a, c, B, F, true);
  UVX_CHECK(a.fits(), UVX_ERR_WORKSPACE, "encoder_bwd: workspace %zu < %zu bytes", ws_bytes, a.off);
  const int dt = c.dtype, d = c.enc_d, Te = s.Te, M = s.M, dh = d / c.enc_heads, r = lora->r;
  const float qscale = 1.0f / sqrtf((float)dh);
  // ln_post backward: out = LN(x_final); x_final is the running state after the last layer (s.x)
  RC(layernorm_bwd(st, dt, d_out, s.x, w->lnf_w, nullptr, s.dx, M, d, c.ln_eps));
  for (int l = c.enc_layers - 1; l >= 0; --l) {
    const uvx_enc_layer_t& L = w->layers[l];
    UVX_CHECK(L.wqkv_t && L.wo_t && L.fc1_t && L.fc2_t, UVX_ERR_INVALID, "encoder_bwd: layer %d lacks transposed weights", l);
    EncLayerStash S = enc_layer(s, l);
    // ---- MLP: x_out = x_mid + fc2(gelu(fc1(LN2(x_mid)))) ----
    {  // d f = (d x . W_fc2) * gelu'(pre): bf16 - in the dgrad GEMM's epilogue (act 3; option 21 = 1: the separate gelu_bwd pass, bit-identical)
      const uvx_enc_lora_layer_t& Rm = lora->layers[l];
      const uvx_enc_lora_layer_grads_t& Gm = grads->layers[l];
      const long long wgf = lora_wgrad_scratch_floats(s.M, std::max(c.enc_d, c.enc_ffn), 64);
      GemmDesc g = lin(s.dx, L.fc2_t, s.d_f, M, c.enc_ffn, d);
      const bool fused = dt == DT_BF16 && g_options[21] != 1 && !Rm.d.a;
      if (fused) { g.act = 3; g.C2 = S.pre; g.ldc2 = c.enc_ffn; }
      RC(gemm(st, dt, g));
      if (Rm.d.a) {      // fc2 adapter: its input gelu(pre) is recomputed; d f += (d x . B * scaling) . A BEFORE the GELU backward
        RC(gelu_fwd(st, dt, S.pre, s.f, (long long)M * c.enc_ffn));
        RC(lora_transpose(st, dt, Rm.d.b, s.lbT, d, r));
        RC(lora_apply_bwd(st, dt, s.f, c.enc_ffn, s.dx, d, s.lbT, at(S.t3, 64, dt), at(s.u3, 64, dt), Gm.d, M, c.enc_ffn, d, r, lora->scaling, s.wg, wgf));
        RC(lora_up(st, dt, at(s.u3, 64, dt), 128, Rm.d.a, 1, s.d_f, c.enc_ffn, M, c.enc_ffn, r, 1.0f, 1));
      }
      if (!fused) RC(gelu_bwd(st, dt, s.d_f, S.pre, s.d_f, (long long)M * c.enc_ffn));
    }
    RC(gemm(st, dt, lin(s.d_f, L.fc1_t, s.d_n, M, d, c.enc_ffn)));
    if (lora->layers[l].g.a) {      // fc1 adapter: its input LN2(x_mid) is recomputed; d n2 += (d pre . B * scaling) . A
      const long long wgf = lora_wgrad_scratch_floats(s.M, std::max(c.enc_d, c.enc_ffn), 64);
      RC(layernorm_fwd(st, dt, S.x_mid, L.ln2_w, L.ln2_b, s.n, M, d, c.ln_eps));
      RC(lora_transpose(st, dt, lora->layers[l].g.b, s.lbT, c.enc_ffn, r));
      RC(lora_apply_bwd(st, dt, s.n, d, s.d_f, c.enc_ffn, s.lbT, S.t3, s.u3, grads->layers[l].g, M, d, c.enc_ffn, r, lora->scaling, s.wg, wgf));
      RC(lora_up(st, dt, s.u3, 128, lora->layers[l].g.a, 1, s.d_n, d, M, d, r, 1.0f, 1));
    }
    RC(layernorm_bwd(st, dt, s.d_n, S.x_mid, L.ln2_w, s.dx, s.dx, M, d, c.ln_eps));
    // ---- attention: x_mid = x_in + wo(attn(q, k, v)) ----
    RC(gemm(st, dt, lin(s.dx, L.wo_t, s.d_o, M, d, d)));
    const uvx_enc_lora_layer_t& R = lora->layers[l];
    const uvx_enc_lora_layer_grads_t& G = grads->layers[l];
    const long long wg_floats = lora_wgrad_scratch_floats(s.M, std::max(c.enc_d, c.enc_ffn), 64);
    if (R.o.a) {   // out_proj adapter: its gradients, and d o += (d x_mid . B_o * scaling) . A_o
      RC(lora_apply_bwd(st, dt, S.o, d, s.dx, d, S.boT, at(S.t2, 64, dt), at(s.u2, 64, dt), G.o, M, d, d, r, lora->scaling, s.wg, wg_floats));
      RC(lora_up(st, dt, at(s.u2, 64, dt), 128, R.o.a, 1, s.d_o, d, M, d, r, 1.0f, 1));
    }
    if (attention_needs_transposed_copies(dt)) RC(heads_transpose(st, dt, S.qkv, s.qT, B, Te, s.Tp, c.enc_heads, dh, 3 * d));
    if (attention_needs_transposed_copies(dt)) RC(heads_transpose(st, dt, at(S.qkv, d, dt), s.kT, B, Te, s.Tp, c.enc_heads, dh, 3 * d));
    if (attention_needs_transposed_copies(dt)) RC(heads_transpose(st, dt, s.d_o, s.doT, B, Te, s.Tp, c.enc_heads, dh, d));
    AttnBwdDesc bd;
    AttnDesc& ad = bd.f;
    ad.q = S.qkv; ad.k = at(S.qkv, d, dt); ad.v = at(S.qkv, 2 * d, dt); ad.o = S.o; ad.lse = S.lse;
    ad.kv_len = audio_lens ? s.kvlen : nullptr;          // written by the forward pass (same audio_lens)
    ad.B = B; ad.T = Te; ad.Tp = s.Tp; ad.Hq = c.enc_heads; ad.Hkv = c.enc_heads; ad.D = dh;
    ad.ldq = ad.ldk = ad.ldv = 3 * d; ad.ldo = d; ad.causal = 0; ad.block = c.enc_block; ad.scale = 1.0f;
    bd.dout = s.d_o; bd.qt = s.qT; bd.kt = s.kT; bd.dot = s.doT; bd.delta = s.delta; bd.dkv_part = nullptr;
    bd.dq = s.d_qkv; bd.dk = at(s.d_qkv, d, dt); bd.dv = at(s.d_qkv, 2 * d, dt);
    bd.lddq = bd.lddk = bd.lddv = 3 * d;
    RC(attention_bwd(st, dt, bd));
    // ---- LoRA gradients of q_proj / k_proj (/ v_proj) (rank-r products on the VALU, lora.hip) ----
    const bool qk_pair = R.q.a && R.k.a;
    if (R.q.a || R.k.a || R.v.a) RC(layernorm_fwd(st, dt, S.x_in, L.ln1_w, L.ln1_b, s.n, M, d, c.ln_eps));      // n1 recomputed (not stashed)
    if (qk_pair) {
      // u = [dq . B_q * (scaling * qscale) | dk . B_k * scaling]  [M, 128] (columns 0..r-1 and 64..64+r-1)
      RC(lora_down2(st, dt, s.d_qkv, at(s.d_qkv, d, dt), 3 * d, S.bqT, S.bkT, s.u, at(s.u, 64, dt), 128, M, d, r, lora->scaling * qscale, lora->scaling));
      // d lora_A [r, d] = u^T . n;  d lora_B [d, r] = scale * dq^T . t
      const LoraWgradItem items[4] = {{s.n, d, s.u, 128, G.q.a, d, 0, 1.0f}, {s.n, d, at(s.u, 64, dt), 128, G.k.a, d, 0, 1.0f},
                                      {s.d_qkv, 3 * d, S.t, 128, G.q.b, d, 1, lora->scaling * qscale},
                                      {at(s.d_qkv, d, dt), 3 * d, at(S.t, 64, dt), 128, G.k.b, d, 1, lora->scaling}};
      RC(lora_wgrad_batch(st, dt, items, 4, M, r, s.wg, wg_floats));      // (one reduce launch for the four)
    } else {
      if (R.q.a) RC(lora_apply_bwd(st, dt, s.n, d, s.d_qkv, 3 * d, S.bqT, S.t, s.u, G.q, M, d, d, r, lora->scaling * qscale, s.wg, wg_floats));
      if (R.k.a) RC(lora_apply_bwd(st, dt, s.n, d, at(s.d_qkv, d, dt), 3 * d, S.bkT, at(S.t, 64, dt), at(s.u, 64, dt), G.k, M, d, d, r, lora->scaling, s.wg, wg_floats));
    }
    if (R.v.a) RC(lora_apply_bwd(st, dt, s.n, d, at(s.d_qkv, 2 * d, dt), 3 * d, S.bvT, S.t2, s.u2, G.v, M, d, d, r, lora->scaling, s.wg, wg_floats));
    if (l == 0) break;   // nothing trainable below layer 0
    // ---- d n1 = d qkv . Wqkv + u . [A_q ; A_k (; A_v)], then LN1 backward into the residual stream ----
    if (l == 0) break;      // nothing below the first layer is trainable (frozen conv stem and positions): its input gradient has no consumer
    RC(gemm(st, dt, lin(s.d_qkv, L.wqkv_t, s.d_n, M, d, 3 * d)));
    if (qk_pair) {
      RC(lora_up2(st, dt, s.u, at(s.u, 64, dt), 128, R.q.a, R.k.a, s.d_n, s.d_n, d, M, d, d, r, 1.0f, 1.0f));      // (same rows: one pass, q term then k term)
    } else {
      if (R.q.a) RC(lora_up(st, dt, s.u, 128, R.q.a, 1, s.d_n, d, M, d, r, 1.0f, 1));
      if (R.k.a) RC(lora_up(st, dt, at(s.u, 64, dt), 128, R.k.a, 1, s.d_n, d, M, d, r, 1.0f, 1));
    }
    if (R.v.a) RC(lora_up(st, dt, s.u2, 128, R.v.a, 1, s.d_n, d, M, d, r, 1.0f, 1));
    RC(layernorm_bwd(st, dt, s.d_n, S.x_in, L.ln1_w, s.dx, s.dx, M, d, c.ln_eps));
  }
  return UVX_OK;
}

// =====================================================================================
extern "C" size_t uvx_projector_ws_bytes(const uvx_config_t* cfg, int32_t B, int32_t Te) {
  if (!cfg) return 0;
  Arena a(nullptr, 0);
  proj_carve(a, *cfg, B, Te);
  return a.off + 256;
}

extern "C" int32_t uvx_projector_fwd(void* stream, const uvx_config_t* cfg, const uvx_projector_weights_t* w,
                                     const void* enc_out, int32_t B, int32_t Te, void* out, void* workspace,
                                     size_t ws_bytes) {
  RC(check_cfg(cfg));
  UVX_CHECK(w && enc_out && out && workspace, UVX_ERR_INVALID, "projector_fwd: null argument");
  const uvx_config_t& c = *cfg;
  UVX_CHECK(c.proj_hidden % 16 == 0, UVX_ERR_SHAPE, "projector hidden %d must be a multiple of 16", c.proj_hidden);
  UVX_CHECK(c.proj_ln_mid ? (w->ln_mid != nullptr) : (w->ln_post != nullptr), UVX_ERR_INVALID,
            "projector: ln_%s weight missing", c.proj_ln_mid ? "mid" : "post");
  hipStream_t st = (hipStream_t)stream;
  if (B == 0 || Te == 0) return UVX_OK;
  Arena a(workspace, ws_bytes);
  ProjWs s = proj_carve(a, c, B, Te);
  UVX_CHECK(a.fits(), UVX_ERR_WORKSPACE, "projector_fwd: workspace %zu < %zu bytes", ws_bytes, a.off);
  const int dt = c.dtype;
  // _pad_and_stack + ln_pre (:790-791)
  RC(stack_rmsnorm_fwd(st, dt, enc_out, w->ln_pre, s.xn, s.stacked, B, Te, c.enc_d, c.stack_factor, c.proj_eps));
  // a few clips (generate(): 188 rows per 30 s): the two linears are 2 x 16 tiles - split-K (gemm.hip) with the backward's W1^T buffer,
  // idle during any forward, as the scratch for the partial tiles.  Larger batches (training at C2: 1504 rows) keep the plain path.
  auto psk = [&](GemmDesc g) {
    if (dt == DT_BF16 && s.R > 64 && s.R <= 1024) { g.splitk_ws = s.w1T; g.splitk_ws_bytes = (size_t)s.C8 * s.H * esz(dt); }
    return g;
  };
  RC(gemm(st, dt, psk(lin(s.xn, w->w1, s.h1, s.R, s.H, s.C8))));            // linear_1 (:793)
  UVX_CHECK(c.proj_act >= UVX_PROJ_SWIGLU && c.proj_act <= UVX_PROJ_RELU, UVX_ERR_INVALID, "projector: unknown proj_act %d", c.proj_act);
  if (c.proj_act == UVX_PROJ_SWIGLU) RC(swiglu_fwd(st, dt, s.h1, s.a, s.R, s.Hh, /*gate_first=*/0));           // SwiGLU (:739-742, :795)
  else RC(act_fwd(st, dt, s.h1, s.a, (long long)s.R * s.H, c.proj_act - 1));                                   // ACT2FN[projector_act] (:754, :795)
  if (c.proj_ln_mid) {
    RC(rmsnorm_fwd(st, dt, s.a, w->ln_mid, s.an, nullptr, s.R, s.Hh, c.proj_eps));  // ln_mid (:796)
    RC(gemm(st, dt, psk(lin(s.an, w->w2, out, s.R, s.D, s.Hh))));                   // linear_2 (:798)
  } else {
    RC(gemm(st, dt, psk(lin(s.a, w->w2, s.ypre, s.R, s.D, s.Hh))));
    RC(rmsnorm_fwd(st, dt, s.ypre, w->ln_post, out, nullptr, s.R, s.D, c.proj_eps));  // ln_post (:799)
  }
  return UVX_OK;
}

extern "C" int32_t uvx_projector_bwd(void* stream, const uvx_config_t* cfg, const uvx_projector_weights_t* w,
                                     const void* dout, int32_t B, int32_t Te, const uvx_projector_grads_t* gr,
                                     void* d_enc_out, void* workspace, size_t ws_bytes) {
  RC(check_cfg(cfg));
  UVX_CHECK(w && dout && gr && workspace, UVX_ERR_INVALID, "projector_bwd: null argument");
  const uvx_config_t& c = *cfg;
  hipStream_t st = (hipStream_t)stream;
  if (B == 0 || Te == 0) return UVX_OK;
  Arena a(workspace, ws_bytes);
  ProjWs s = proj_carve(a, c, B, Te);
  UVX_CHECK(a.fits(), UVX_ERR_WORKSPACE, "projector_bwd: workspace %zu < %zu bytes", ws_bytes, a.off);
  const int dt = c.dtype;
  const void* dy = dout;
  RC(fill_zero(st, gr->ln_pre, sizeof(float) * s.C8));
  if (c.proj_ln_mid) RC(fill_zero(st, gr->ln_mid, sizeof(float) * s.Hh));
  else {
    RC(fill_zero(st, gr->ln_post, sizeof(float) * s.D));
    RC(rmsnorm_bwd(st, dt, dout, s.ypre, w->ln_post, nullptr, s.dy2, gr->ln_post, s.R, s.D, c.proj_eps, 0, nullptr, s.dwp));
    dy = s.dy2;
  }
  // linear_2: dW2[D, Hh] = dy^T . an ; d_an = dy . W2
  RC(transpose2d(st, dt, dy, s.dyT, s.R, s.D, s.D, s.Rp, 1, 0, 0));
  RC(transpose2d(st, dt, s.an, s.anT, s.R, s.Hh, s.Hh, s.Rp, 1, 0, 0));
  {
    GemmDesc g = lin(s.dyT, s.anT, gr->w2, s.D, s.Hh, s.Rp);
    g.out_f32 = 1;
    RC(gemm(st, dt, g));
  }
  RC(transpose2d(st, dt, w->w2, s.w2T, s.D, s.Hh, s.Hh, s.D, 1, 0, 0));
  RC(gemm(st, dt, lin(dy, s.w2T, s.d_an, s.R, s.Hh, s.D)));
  if (c.proj_ln_mid)
    RC(rmsnorm_bwd(st, dt, s.d_an, s.a, w->ln_mid, nullptr, s.d_a, gr->ln_mid, s.R, s.Hh, c.proj_eps, 0, nullptr, s.dwp));
  if (c.proj_act == UVX_PROJ_SWIGLU) RC(swiglu_bwd(st, dt, s.d_a, s.h1, s.d_h1, s.R, s.Hh, 0));
  else RC(act_bwd(st, dt, s.d_a, s.h1, s.d_h1, (long long)s.R * s.H, c.proj_act - 1));
  // linear_1: dW1[H, C8] = d_h1^T . xn ; d_xn = d_h1 . W1 (only needed for the ln_pre weight gradient)
  RC(transpose2d(st, dt, s.d_h1, s.dh1T, s.R, s.H, s.H, s.Rp, 1, 0, 0));
  RC(transpose2d(st, dt, s.xn, s.xnT, s.R, s.C8, s.C8, s.Rp, 1, 0, 0));
  {
    GemmDesc g = lin(s.dh1T, s.xnT, gr->w1, s.H, s.C8, s.Rp);
    g.out_f32 = 1;
    RC(gemm(st, dt, g));
  }
  RC(transpose2d(st, dt, w->w1, s.w1T, s.H, s.C8, s.C8, s.H, 1, 0, 0));
  RC(gemm(st, dt, lin(s.d_h1, s.w1T, s.d_xn, s.R, s.C8, s.H)));
  if (!d_enc_out) {
    RC(rmsnorm_bwd(st, dt, s.d_xn, s.stacked, w->ln_pre, nullptr, nullptr, gr->ln_pre, s.R, s.C8, c.proj_eps, 0, nullptr, s.dwp));
    return UVX_OK;
  }
  // the encoder trains too (LoRA): d stacked [R, S*C] in place of d_xn, then un-stack: clip b's frames are the first
  // Te * C elements of its J * S * C block (the padded tail frames get no gradient consumer)
  RC(rmsnorm_bwd(st, dt, s.d_xn, s.stacked, w->ln_pre, nullptr, s.d_xn, gr->ln_pre, s.R, s.C8, c.proj_eps, 0, nullptr, s.dwp));
  const size_t es = esz(dt);
  UVX_HIP(hipMemcpy2DAsync(d_enc_out, (size_t)Te * c.enc_d * es, s.d_xn, (size_t)s.J * s.C8 * es, (size_t)Te * c.enc_d * es, B,
                           hipMemcpyDeviceToDevice, st));
  return UVX_OK;
}

// =====================================================================================
extern "C" int32_t uvx_embed_merge(void* stream, const uvx_config_t* cfg, const void* embed_table, const int64_t* input_ids,
                                   const void* audio_embeds, const int64_t* audio_batch_size,
                                   const int64_t* audio_token_start_idx, const int32_t* audio_token_len, int32_t B,
                                   int32_t T, int32_t n_items, int32_t Na, void* inputs_embeds, int32_t* scratch) {
  RC(check_cfg(cfg));
  UVX_CHECK(inputs_embeds && scratch, UVX_ERR_INVALID, "embed_merge: null argument");
  hipStream_t st = (hipStream_t)stream;
  const uvx_config_t& c = *cfg;
  if (input_ids) RC(embed_gather(st, c.dtype, embed_table, input_ids, inputs_embeds, B * T, c.llm_d, c.vocab));
  // Gemma-3: the embedding MODULE scales its rows ([3P] Gemma3TextScaledWordEmbedding: lookup * sqrt(hidden) in the table's dtype);
  // the audio rows merged over them below are not scaled
  if (input_ids && c.llm_flavor == UVX_LLM_GEMMA3) RC(scale_inplace(st, c.dtype, inputs_embeds, (long long)B * T * c.llm_d, gemma_normalizer(c)));
  int32_t* owner = scratch;
  int32_t* item_batch = scratch + (size_t)B * T;
  if (n_items > 0) {
    UVX_CHECK(audio_embeds && audio_batch_size && audio_token_start_idx && audio_token_len, UVX_ERR_INVALID,
              "inputs_embeds/audio_values/audio_token_start_idx/audio_token_len/audio_lens/audio_batch_size must be provided.");
  }
  RC(merge_owner(st, owner, item_batch, audio_batch_size, audio_token_start_idx, audio_token_len, B, n_items, T, Na));
  RC(merge_audio(st, c.dtype, inputs_embeds, audio_embeds, nullptr, owner, item_batch, audio_token_start_idx,
                 audio_token_len, n_items, T, c.llm_d, Na, 0));
  return UVX_OK;
}

extern "C" int32_t uvx_merge_embeds_bwd(void* stream, const uvx_config_t* cfg, const void* d_inputs_embeds,
                                        const int64_t* audio_token_start_idx, const int32_t* audio_token_len, int32_t B,
                                        int32_t T, int32_t n_items, int32_t Na, void* d_audio_embeds,
                                        const int32_t* scratch) {
  RC(check_cfg(cfg));
  hipStream_t st = (hipStream_t)stream;
  const int32_t* owner = scratch;
  const int32_t* item_batch = scratch + (size_t)B * T;
  return merge_audio(st, cfg->dtype, const_cast<void*>(d_inputs_embeds), nullptr, d_audio_embeds, owner, item_batch,
                     audio_token_start_idx, audio_token_len, n_items, T, cfg->llm_d, Na, 1);
}

// =====================================================================================
extern "C" size_t uvx_llm_ws_bytes(const uvx_config_t* cfg, int32_t B, int32_t T, int32_t save_for_bwd) {
  if (!cfg) return 0;
  Arena a(nullptr, 0);
  llm_carve(a, *cfg, B, T, save_for_bwd);
  return a.off + 256;
}

static int llm_check(const uvx_config_t& c, const uvx_llm_weights_t* w, int T) {
  UVX_CHECK(c.llm_heads % c.llm_kv_heads == 0, UVX_ERR_SHAPE, "llm: heads %d not a multiple of kv heads %d", c.llm_heads, c.llm_kv_heads);
  UVX_CHECK(c.llm_inter % 16 == 0, UVX_ERR_SHAPE, "llm: intermediate size %d must be a multiple of 16", c.llm_inter);
  UVX_CHECK(w->rope_len >= T, UVX_ERR_SHAPE, "llm: rope table (%d) shorter than sequence (%d)", w->rope_len, T);
  if (c.llm_qk_norm)
    for (int l = 0; l < c.llm_layers; ++l)
      UVX_CHECK(w->layers[l].q_norm && w->layers[l].k_norm, UVX_ERR_INVALID, "llm: llm_qk_norm is set but layer %d has no q_norm / k_norm", l);
  if (c.llm_flavor == UVX_LLM_GEMMA3) {
    bool any_local = false;
    for (int l = 0; l < c.llm_layers; ++l) {
      UVX_CHECK(w->layers[l].ln1_post && w->layers[l].ln2_post, UVX_ERR_INVALID, "llm: Gemma-3 layer %d has no post norms", l);
      any_local = any_local || (w->layer_local && w->layer_local[l]);
    }
    UVX_CHECK(!any_local || w->rope_cos_sin_local, UVX_ERR_INVALID, "llm: Gemma-3 sliding-window layers need rope_cos_sin_local");
  }
  return UVX_OK;
}

__global__ void set_i32_k(int32_t* p, int32_t v) { *p = v; }

// rows != nullptr (uvx_llm_fwd_rows): the LM head is evaluated only for the listed positions (device list, ascending,
// host-known length); the compact logits stay in the workspace with the list, like the supervised-row CE path.
static int llm_forward(void* stream, const uvx_config_t* cfg, const uvx_llm_weights_t* w, const void* inputs_embeds,
                       const int64_t* attention_mask, const int64_t* labels, int32_t B, int32_t T, void* logits,
                       float* loss, int32_t save_for_bwd, void* workspace, size_t ws_bytes, const int32_t* rows,
                       int32_t n_rows, void* logits_rows, const uvx_encoder_lora_t* lora = nullptr, bool top_rows = false) {
  RC(check_cfg(cfg));
  UVX_CHECK(w && inputs_embeds && workspace, UVX_ERR_INVALID, "llm_fwd: null argument");
  UVX_CHECK(!labels || loss, UVX_ERR_INVALID, "llm_fwd: labels given but no loss output");
  const uvx_config_t& c = *cfg;
  RC(llm_check(c, w, T));
  hipStream_t st = (hipStream_t)stream;
  if (B == 0 || T == 0) return UVX_OK;
  Arena a(workspace, ws_bytes);
  LlmWs s = llm_carve(a, c, B, T, save_for_bwd);
  UVX_CHECK(a.fits(), UVX_ERR_WORKSPACE, "llm_fwd: workspace %zu < %zu bytes", ws_bytes, a.off);
  const int dt = c.dtype, D = c.llm_d, M = s.M, dh = c.llm_head_dim, Hq = c.llm_heads, Hkv = c.llm_kv_heads;
  const size_t es = esz(dt);

  // valid-key range per sequence (kept in the workspace for the backward pass)
  if (attention_mask) hipLaunchKernelGGL(mask_range_k, dim3(B), dim3(256), 0, st, attention_mask, s.kvs, s.kvl, T);
  else hipLaunchKernelGGL(full_range_k, dim3(cdiv(B, 64)), dim3(64), 0, st, s.kvs, s.kvl, B, T);
  UVX_LAUNCH_CHECK();
  // top_rows (uvx_llm_fwd_train): only the supervised positions enter the loss, and in the LAST layer nothing downstream of
  // its attention mixes positions any more - o_proj, the MLP and the final norm are row-wise.  Their results are needed
  // (and have a gradient) on the supervised rows alone, so the last layer's post-attention half runs on the compacted
  // rows (device-side list, no host sync; GEMMs clamp to the device count).  Same loss, same gradients.
  UVX_CHECK(!top_rows || (labels && loss && dt == DT_BF16 && save_for_bwd && !logits && !rows), UVX_ERR_INVALID,
            "llm_fwd_train: labels and a loss output are required, bf16 only");
  UVX_CHECK(!rows || dt == DT_BF16, UVX_ERR_UNSUPPORTED, "llm_fwd_rows: bf16 only");
  UVX_CHECK(!rows || (n_rows >= 0 && n_rows <= M), UVX_ERR_SHAPE, "llm_fwd_rows: %d rows of %d", n_rows, M);
  const int fl = c.llm_flavor;   // 0 Llama, 1 Gemma, 2 Gemma-3 (norm flavour - any non-zero value is Gemma's -, GLU activation, embedding scale)
  const bool g3 = fl == UVX_LLM_GEMMA3;
  // uvx_llm_fwd_rows (round 6): the same holds for a caller-supplied row list - only the listed positions' logits leave the call (teacher)
  // or are differentiated (student, uvx_llm_bwd_rows), so the last layer's row-wise half runs on them alone
  const bool top = top_rows || (rows && !lora);
  const bool tc = top && g_options[3] && !g3;   // (tuning option 3 off, or Gemma-3's post norms: the plain full-row path, in both calls of the pair)
  if (top && save_for_bwd) note_pair(workspace, tc);
  const float attn_scale = c.llm_attn_scale > 0.f ? c.llm_attn_scale : 1.0f / sqrtf((float)dh);
  {
    LlmLayerStash l0 = llm_layer(s, 0);
    UVX_HIP(hipMemcpyAsync(l0.x_in, inputs_embeds, (size_t)M * D * es, hipMemcpyDeviceToDevice, st));
    if (fl == UVX_LLM_GEMMA) RC(scale_inplace(st, dt, l0.x_in, (long long)M * D, gemma_normalizer(c)));
  }
  auto slot_of = [&](int l) { return save_for_bwd ? l : (l & 1); };
  // first half of a layer, rows of the view v (a batch slice): norm, q|k|v projection, RoPE, V^T, causal GQA flash attention
  auto layer_attn = [&](hipStream_t sx, const LlmWs& v, int Bv, int l) -> int {
    const uvx_llm_layer_t& L = w->layers[l];
    LlmLayerStash cur = llm_layer(v, slot_of(l));
    const int Mv = v.M;
    // (probe bit 256: the kernel runs but writes elsewhere - the GEMM then reads a buffer nobody has just written)
    if (!probe_skip(16)) RC(rmsnorm_fwd(sx, dt, cur.x_in, L.ln1, probe_skip(256) && save_for_bwd ? v.d_n : v.n, nullptr, Mv, D, c.rms_eps, fl));
    {
      GemmDesc g = lin(v.n, L.wqkv, cur.qkv, Mv, s.QKV, D);
      g.bias = L.bqkv;     // Qwen2: q / k / v projection biases (null otherwise)
      RC(gemm(sx, dt, g));
    }
    if (lora) {   // peft LoRA on q_proj / k_proj (/ v_proj) (text_model_lora_config): added to the projections, before q_norm / RoPE
      const uvx_enc_lora_layer_t& R = lora->layers[l];
      const int r = lora->r, qc = Hq * dh, kc = Hkv * dh;
      if (R.q.a && R.k.a) {
        RC(lora_transpose2(sx, dt, R.q.b, cur.bqT, qc, R.k.b, cur.bkT, kc, r));
        RC(lora_down(sx, dt, v.n, D, R.q.a, 0, cur.t, 128, Mv, D, r, 1.0f));
        RC(lora_down(sx, dt, v.n, D, R.k.a, 0, at(cur.t, 64, dt), 128, Mv, D, r, 1.0f));
        RC(lora_up(sx, dt, cur.t, 128, cur.bqT, 1, cur.qkv, s.QKV, Mv, qc, r, lora->scaling, 1));
        RC(lora_up(sx, dt, at(cur.t, 64, dt), 128, cur.bkT, 1, at(cur.qkv, (size_t)qc, dt), s.QKV, Mv, kc, r, lora->scaling, 1));
      } else {
        if (R.q.a) RC(lora_apply(sx, dt, v.n, D, R.q, cur.bqT, cur.t, cur.qkv, s.QKV, Mv, D, qc, r, lora->scaling));
        if (R.k.a) RC(lora_apply(sx, dt, v.n, D, R.k, cur.bkT, at(cur.t, 64, dt), at(cur.qkv, (size_t)qc, dt), s.QKV, Mv, D, kc, r, lora->scaling));
      }
      if (R.v.a) RC(lora_apply(sx, dt, v.n, D, R.v, cur.bvT, cur.t2, at(cur.qkv, (size_t)(qc + kc), dt), s.QKV, Mv, D, kc, r, lora->scaling));
    }
    // (Gemma-3: the sliding-window layers rotate with their own table)
    const float* rope = g3 && w->layer_local && w->layer_local[l] ? w->rope_cos_sin_local : w->rope_cos_sin;
    if (c.llm_qk_norm)   // Qwen3 / Gemma-3: q_norm / k_norm per head, then RoPE - one pass; the raw rows stay for the backward
      RC(qk_norm_rope(sx, dt, cur.qkv, L.q_norm, L.k_norm, save_for_bwd ? cur.qk_raw : nullptr, rope, nullptr, Mv, T, Hq, Hkv,
                      dh, s.QKV, c.rms_eps, g3 ? 1 : 0));
    else if (!probe_skip(32)) RC(rope_inplace(sx, dt, cur.qkv, rope, nullptr, Mv, T, Hq + Hkv, dh, s.QKV, 0));
    if (attention_needs_transposed_copies(dt)) RC(heads_transpose(sx, dt, at(cur.qkv, (size_t)(Hq + Hkv) * dh, dt), v.vt, Bv, T, s.Tp, Hkv, dh, s.QKV));
    AttnDesc ad;
    ad.q = cur.qkv; ad.k = at(cur.qkv, (size_t)Hq * dh, dt); ad.v = at(cur.qkv, (size_t)(Hq + Hkv) * dh, dt);
    ad.vt = v.vt; ad.o = cur.o; ad.lse = cur.lse; ad.kv_start = v.kvs; ad.kv_len = v.kvl;
    ad.B = Bv; ad.T = T; ad.Tp = s.Tp; ad.Hq = Hq; ad.Hkv = Hkv; ad.D = dh;
    ad.ldq = ad.ldk = ad.ldv = s.QKV; ad.ldo = s.OD; ad.causal = 1; ad.block = 0;
    ad.scale = attn_scale;
    // Gemma-3 sliding-window layer over a sequence LONGER than the window (up to the window it is plain causal attention)
    ad.window = c.llm_window > 0 && T > c.llm_window && w->layer_local && w->layer_local[l] ? c.llm_window : 0;      // (any flavour: Gemma-3's local layers, every Mistral layer)
    return probe_skip(2) ? UVX_OK : attention_fwd(sx, dt, ad);
  };
  // second half: o_proj + residual, norm, gate|up (+ SwiGLU), down + residual.  compact (last layer of the training pair,
  // whole batch only): on the supervised rows gathered into the idle backward scratch.
  // MLP adapters (ABI 18).  gate_proj / up_proj: result += lora_B(lora_A(n2)) * scaling on the gate / up half of the interleaved gate|up tensor -
  // the rank-r kernels work on contiguous columns, so the term is formed in v.act (free until the GLU writes it) and added half by half
  auto mlp_in_adapters = [&](hipStream_t sx, const LlmWs& v, const LlmLayerStash& cur, int l) -> int {
    const uvx_enc_lora_layer_t& R = lora->layers[l];
    const int I = c.llm_inter, r = lora->r;
    for (int which = 0; which < 2; ++which) {
      const uvx_lora_proj_t& P = which ? R.u : R.g;
      if (!P.a) continue;
      void* t = at(cur.t3, 64 * which, dt);
      RC(lora_transpose(sx, dt, P.b, v.lbT, I, r));
      RC(lora_down(sx, dt, v.n, D, P.a, 0, t, 128, v.M, D, r, 1.0f));
      RC(lora_up(sx, dt, t, 128, v.lbT, 1, v.act, I, v.M, I, r, lora->scaling, 0));
      RC(gu_half(sx, dt, cur.gu, v.act, v.M, I, which, 1));
    }
    return UVX_OK;
  };
  auto layer_mlp = [&](hipStream_t sx, const LlmWs& v, int l, bool compact) -> int {
    const uvx_llm_layer_t& L = w->layers[l];
    const bool ad_in = lora && (lora->layers[l].g.a || lora->layers[l].u.a), ad_out = lora && lora->layers[l].d.a;
    const bool last = l + 1 == c.llm_layers;
    LlmLayerStash cur = llm_layer(v, slot_of(l));
    void* x_out = last ? v.x_final : llm_layer(v, slot_of(l + 1)).x_in;
    const int Mv = v.M;
    const int32_t* mdev = compact ? v.sup + Mv : nullptr;
    if (g3) {
      // Gemma3DecoderLayer: x_mid = x_in + post_attention_norm(o_proj(o));  x_out = x_mid + post_feedforward_norm(mlp(pre_feedforward_norm(x_mid)))
      // (the branch outputs before their post norms stay in the stash for the backward: o_pre, m_pre)
      RC(gemm(sx, dt, lin(cur.o, L.wo, cur.o_pre, Mv, D, s.OD)));
      if (lora && lora->layers[l].o.a)      // o_proj adapter: joins the branch before its post norm
        RC(lora_apply(sx, dt, cur.o, s.OD, lora->layers[l].o, cur.boT, at(cur.t2, 64, dt), cur.o_pre, D, Mv, s.OD, D, lora->r, lora->scaling));
      RC(rmsnorm_fwd(sx, dt, cur.o_pre, L.ln1_post, cur.x_mid, nullptr, Mv, D, c.rms_eps, fl, nullptr, cur.x_in));
      RC(rmsnorm_fwd(sx, dt, cur.x_mid, L.ln2, v.n, nullptr, Mv, D, c.rms_eps, fl));
      RC(gemm(sx, dt, lin(v.n, L.wgu, cur.gu, Mv, 2 * c.llm_inter, D)));
      if (ad_in) RC(mlp_in_adapters(sx, v, cur, l));
      RC(swiglu_fwd(sx, dt, cur.gu, v.act, Mv, c.llm_inter, /*layout=*/2, /*act=*/c.llm_act));
      RC(gemm(sx, dt, lin(v.act, L.wd, cur.m_pre, Mv, D, c.llm_inter)));
      if (ad_out)      // down_proj adapter: joins the branch before its post norm
        RC(lora_apply(sx, dt, v.act, c.llm_inter, lora->layers[l].d, v.lbT, cur.t4, cur.m_pre, D, Mv, c.llm_inter, D, lora->r, lora->scaling));
      return rmsnorm_fwd(sx, dt, cur.m_pre, L.ln2_post, x_out, nullptr, Mv, D, c.rms_eps, fl, nullptr, cur.x_mid);
    }
    // gather targets of the compact last layer: the idle backward scratch, or - a forward without stash (the KL teacher) - the other
    // layer slot's o / x_in, dead since the previous layer finished
    void* g_o = save_for_bwd ? v.d_o : llm_layer(v, slot_of(l) ^ 1).o;
    void* g_x = save_for_bwd ? v.dx : llm_layer(v, slot_of(l) ^ 1).x_in;
    if (compact) {   // gather the supervised rows of the attention output and of the residual stream
      if (rows) {    // (uvx_llm_fwd_rows: the caller's list)
        UVX_HIP(hipMemcpyAsync(v.sup, rows, sizeof(int32_t) * n_rows, hipMemcpyDeviceToDevice, sx));
        hipLaunchKernelGGL(set_i32_k, dim3(1), dim3(1), 0, sx, v.sup + Mv, n_rows);
        UVX_LAUNCH_CHECK();
      } else {
        RC(sup_rows(sx, labels, v.sup, B, T, c.vocab));
      }
      RC(gather_rows(sx, dt, cur.o, v.sup, Mv, g_o, s.OD));
      RC(gather_rows(sx, dt, cur.x_in, v.sup, Mv, g_x, D));
    }
    {
      GemmDesc g = lin(compact ? g_o : cur.o, L.wo, cur.x_mid, Mv, D, s.OD);
      g.residual = compact ? g_x : cur.x_in; g.ldr = D; g.m_dev = mdev;
      RC(gemm(sx, dt, g));
    }
    if (lora && lora->layers[l].o.a)        // o_proj adapter (never on the compact path: top is false under LoRA)
      RC(lora_apply(sx, dt, cur.o, s.OD, lora->layers[l].o, cur.boT, at(cur.t2, 64, dt), cur.x_mid, D, Mv, s.OD, D, lora->r, lora->scaling));
    if (!probe_skip(16)) RC(rmsnorm_fwd(sx, dt, cur.x_mid, L.ln2, probe_skip(256) && save_for_bwd ? v.d_n : v.n, nullptr, Mv, D, c.rms_eps, fl, mdev));
    {  // gate|up projection; wgu rows are packed as alternating 16-row gate / up blocks (weights.py)
      GemmDesc g = lin(v.n, L.wgu, cur.gu, Mv, 2 * c.llm_inter, D);
      const bool fused = dt == DT_BF16 && fl == UVX_LLM_LLAMA && !ad_in;   // SwiGLU fused into the epilogue (GeGLU, or adapters on gate / up: separate kernel)
      if (fused) { g.C2 = v.act; g.ldc2 = c.llm_inter; g.swiglu = 1; }
      g.m_dev = mdev;
      RC(gemm(sx, dt, g));
      if (ad_in) RC(mlp_in_adapters(sx, v, cur, l));
      if (!fused) RC(swiglu_fwd(sx, dt, cur.gu, v.act, Mv, c.llm_inter, /*layout=*/2, /*act=*/c.llm_act, mdev));
    }
    {
      GemmDesc g = lin(v.act, L.wd, x_out, Mv, D, c.llm_inter);
      g.residual = cur.x_mid; g.ldr = D; g.m_dev = mdev;
      RC(gemm(sx, dt, g));
    }
    if (ad_out)      // down_proj adapter (never on the compact path: top is false under LoRA)
      RC(lora_apply(sx, dt, v.act, c.llm_inter, lora->layers[l].d, v.lbT, cur.t4, x_out, D, Mv, c.llm_inter, D, lora->r, lora->scaling));
    return UVX_OK;
  };
  // schedule: one chain on the caller's stream, or (option 11) the batch slices on several streams - see Fork above.  The
  // chains advance in lockstep (same kernel at the same time): a staggered start was measured 3.3 ms per step slower
  // (profiles/r03_two_stream_stagger_and_tiles_ab.txt).
  const Chains ch = make_chains(st, s, c, B, T, dt == DT_BF16 && !lora);
  RC(chains_fork(ch));
  int rc_layers = UVX_OK;
  for (int l = 0; l < c.llm_layers && rc_layers == UVX_OK; ++l) {
    const bool compact = tc && l + 1 == c.llm_layers;
    for (int h = 0; h < ch.n && rc_layers == UVX_OK; ++h) {
      rc_layers = layer_attn(ch.st[h], ch.v[h], ch.b0[h + 1] - ch.b0[h], l);
      if (rc_layers == UVX_OK && !compact) rc_layers = layer_mlp(ch.st[h], ch.v[h], l, false);
    }
  }
  RC(chains_join(ch));   // (also after an error above: the side streams must not be left forked)
  RC(rc_layers);
  if (tc) RC(layer_mlp(st, s, c.llm_layers - 1, true));
  RC(rmsnorm_fwd(st, dt, s.x_final, w->norm, s.hn, nullptr, M, D, c.rms_eps, fl, tc ? s.sup + M : nullptr));   // (compact last layer: its rows only)
  if (rows) {
    if (!tc) {       // (compact last layer: the list is in place and s.hn holds its rows, in list order)
      if (n_rows > 0) UVX_HIP(hipMemcpyAsync(s.sup, rows, sizeof(int32_t) * n_rows, hipMemcpyDeviceToDevice, st));
      hipLaunchKernelGGL(set_i32_k, dim3(1), dim3(1), 0, st, s.sup + M, n_rows);
      UVX_LAUNCH_CHECK();
    }
    if (n_rows == 0) return UVX_OK;
    if (!tc) RC(gather_rows(st, dt, s.hn, s.sup, M, s.n, D));
    RC(gemm(st, dt, lin(tc ? s.hn : s.n, w->lm_head, s.logits, n_rows, c.vocab, D)));
    if (logits_rows) UVX_HIP(hipMemcpyAsync(logits_rows, s.logits, (size_t)n_rows * c.vocab * es, hipMemcpyDeviceToDevice, st));
    return UVX_OK;
  }
  if (labels && dt == DT_BF16 && g_options[3]) {
    // Loss path on the SUPERVISED rows only (positions whose next token carries a label): every other row of the
    // logits has zero weight in ForCausalLMLoss and a zero gradient, so the head GEMM, the CE and (uvx_llm_bwd) the
    // head dgrad run on the compacted rows - identical loss and gradients, ~T / n_supervised less head work.  The
    // row list is built on the device (no host sync): GEMMs are launched for M rows and clamp to the device count.
    if (logits) RC(gemm(st, dt, lin(s.hn, w->lm_head, logits, M, c.vocab, D)));   // the caller's full logits, if asked
    if (!tc) {   // (top_rows: the last layer already left s.hn compact, in the order of the row list)
      RC(sup_rows(st, labels, s.sup, B, T, c.vocab));
      RC(gather_rows(st, dt, s.hn, s.sup, M, s.n, D));
    }
    GemmDesc g = lin(tc ? s.hn : s.n, w->lm_head, s.logits, M, c.vocab, D);
    g.m_dev = s.sup + M;
    RC(gemm(st, dt, g));
    RC(ce_loss_fwd_bwd(st, dt, s.logits, labels, loss, s.ce_scratch, nullptr, B, T, c.vocab, c.vocab, 1.0f, s.sup));
    return UVX_OK;
  }
  RC(gemm(st, dt, lin(s.hn, w->lm_head, s.logits, M, c.vocab, D)));
  if (logits) UVX_HIP(hipMemcpyAsync(logits, s.logits, (size_t)M * c.vocab * es, hipMemcpyDeviceToDevice, st));
  if (labels) RC(ce_loss_fwd_bwd(st, dt, s.logits, labels, loss, s.ce_scratch, nullptr, B, T, c.vocab, c.vocab, 1.0f));
  return UVX_OK;
}

extern "C" int32_t uvx_llm_fwd(void* stream, const uvx_config_t* cfg, const uvx_llm_weights_t* w, const void* inputs_embeds,
                               const int64_t* attention_mask, const int64_t* labels, int32_t B, int32_t T, void* logits,
                               float* loss, int32_t save_for_bwd, void* workspace, size_t ws_bytes) {
  return llm_forward(stream, cfg, w, inputs_embeds, attention_mask, labels, B, T, logits, loss, save_for_bwd, workspace, ws_bytes,
                     nullptr, 0, nullptr);
}

extern "C" int32_t uvx_llm_fwd_rows(void* stream, const uvx_config_t* cfg, const uvx_llm_weights_t* w, const void* inputs_embeds,
                                    const int64_t* attention_mask, int32_t B, int32_t T, const int32_t* rows, int32_t n_rows,
                                    void* logits_rows, int32_t save_for_bwd, void* workspace, size_t ws_bytes) {
  UVX_CHECK(rows != nullptr, UVX_ERR_INVALID, "llm_fwd_rows: null row list");
  return llm_forward(stream, cfg, w, inputs_embeds, attention_mask, nullptr, B, T, nullptr, nullptr, save_for_bwd, workspace,
                     ws_bytes, rows, n_rows, logits_rows);
}

// KL loss on the compact student rows left by uvx_llm_fwd_rows(save_for_bwd = 1): pair [2][n_rows] = index into the teacher's
// compact rows (or -1), weights alike; d loss / d logits replaces the compact logits in place.
extern "C" int32_t uvx_llm_kl_loss_rows(void* stream, const uvx_config_t* cfg, const void* teacher_logits_rows,
                                        const int32_t* pair, const float* pair_w, int32_t B, int32_t T, int32_t n_rows,
                                        float temperature, float grad_scale, float* loss, void* workspace, size_t ws_bytes) {
  RC(check_cfg(cfg));
  UVX_CHECK(teacher_logits_rows && pair && pair_w && loss && workspace, UVX_ERR_INVALID, "llm_kl_loss_rows: null argument");
  const uvx_config_t& c = *cfg;
  if (B == 0 || T == 0) return UVX_OK;
  Arena a(workspace, ws_bytes);
  LlmWs s = llm_carve(a, c, B, T, 1);
  UVX_CHECK(a.fits(), UVX_ERR_WORKSPACE, "llm_kl_loss_rows: workspace %zu < %zu bytes", ws_bytes, a.off);
  UVX_CHECK(n_rows > 0 && n_rows <= s.M, UVX_ERR_SHAPE, "llm_kl_loss_rows: %d rows of %d", n_rows, s.M);
  return kl_loss_fwd_bwd((hipStream_t)stream, c.dtype, s.logits, teacher_logits_rows, pair, pair_w, loss, s.ce_scratch + 2,
                         s.logits, (long long)n_rows, c.vocab, c.vocab, c.vocab, temperature, grad_scale);
}

extern "C" int32_t uvx_llm_kl_loss(void* stream, const uvx_config_t* cfg, const void* teacher_logits, int64_t teacher_rows,
                                   const int32_t* pair_row, const float* pair_w, int32_t B, int32_t T, float temperature,
                                   float grad_scale, float* loss, void* workspace, size_t ws_bytes) {
  RC(check_cfg(cfg));
  UVX_CHECK(teacher_logits && pair_row && pair_w && loss && workspace, UVX_ERR_INVALID, "llm_kl_loss: null argument");
  UVX_CHECK(teacher_rows > 0, UVX_ERR_SHAPE, "llm_kl_loss: no teacher rows");
  const uvx_config_t& c = *cfg;
  if (B == 0 || T == 0) return UVX_OK;
  Arena a(workspace, ws_bytes);
  LlmWs s = llm_carve(a, c, B, T, 1);
  UVX_CHECK(a.fits(), UVX_ERR_WORKSPACE, "llm_kl_loss: workspace %zu < %zu bytes", ws_bytes, a.off);
  // student logits were left in the workspace by uvx_llm_fwd(save_for_bwd = 1); their gradient replaces them
  return kl_loss_fwd_bwd((hipStream_t)stream, c.dtype, s.logits, teacher_logits, pair_row, pair_w, loss, s.ce_scratch + 2,
                         s.logits, (long long)s.M, c.vocab, c.vocab, c.vocab, temperature, grad_scale);
}

// compact_in_place: the workspace holds d loss / d logits for the compact rows of its row list (uvx_llm_kl_loss_rows)
static int llm_backward(void* stream, const uvx_config_t* cfg, const uvx_llm_weights_t* w, const int64_t* labels,
                        int32_t B, int32_t T, float grad_scale, void* d_inputs_embeds, void* workspace, size_t ws_bytes,
                        bool compact_in_place, const uvx_encoder_lora_t* lora = nullptr,
                        const uvx_encoder_lora_grads_t* lgrads = nullptr, bool top_rows = false, int first_pos = 0) {
  RC(check_cfg(cfg));
  UVX_CHECK(w && d_inputs_embeds && workspace, UVX_ERR_INVALID, "llm_bwd: null argument");
  UVX_CHECK(first_pos >= 0 && first_pos <= T, UVX_ERR_INVALID, "llm_bwd: first_pos %d outside [0, T = %d]", first_pos, T);
  const uvx_config_t& c = *cfg;
  RC(llm_check(c, w, T));
  const bool wts = c.llm_wt_stream != 0;      // transposed weights made on the fly (include/uvx.h)
  UVX_CHECK(wts || w->lm_head_t != nullptr, UVX_ERR_INVALID, "llm_bwd: transposed weights (lm_head_t, *_t) are required (or llm_wt_stream)");
  hipStream_t st = (hipStream_t)stream;
  if (B == 0 || T == 0) return UVX_OK;
  Arena a(workspace, ws_bytes);
  LlmWs s = llm_carve(a, c, B, T, 1);
  UVX_CHECK(a.fits(), UVX_ERR_WORKSPACE, "llm_bwd: workspace %zu < %zu bytes", ws_bytes, a.off);
  const int dt = c.dtype, D = c.llm_d, M = s.M, dh = c.llm_head_dim, Hq = c.llm_heads, Hkv = c.llm_kv_heads;

  // llm_wt_stream: a side stream transposes lm_head and then, one layer ahead of the layer being differentiated, each layer's
  // four matrices into the alternating buffers s.wt[l & 1]; e_ready[b] = buffer b holds its layer, e_free[b] = the caller's stream
  // is done with buffer b.  (One chain only: the chains' side streams would each need the same waits.)
  // (neither resident copies nor the stream, bf16: the transposed pointers stay NULL and lin_dgrad takes the NN form on the forward weights)
  struct LayerT { const void *wqkv_t, *wo_t, *wgu_t, *wd_t; };
  WtStream* wt = nullptr;
  auto layer_t = [&](int l) -> LayerT {
    const uvx_llm_layer_t& L = w->layers[l];
    if (!wts) return LayerT{L.wqkv_t, L.wo_t, L.wgu_t, L.wd_t};
    const size_t es_ = esz(dt), nq = (size_t)s.QKV * D, no = (size_t)s.OD * D, ng = (size_t)2 * c.llm_inter * D;
    char* b = (char*)s.wt[l & 1];
    return LayerT{b, b + nq * es_, b + (nq + no) * es_, b + (nq + no + ng) * es_};
  };
  auto issue_layer_t = [&](int l) -> int {      // on the side stream: W^T of layer l into its buffer
    const uvx_llm_layer_t& L = w->layers[l];
    const LayerT t = layer_t(l);
    UVX_HIP(hipStreamWaitEvent(wt->side, wt->e_free[l & 1], 0));
    RC(transpose2d_streaming(wt->side, dt, L.wqkv, const_cast<void*>(t.wqkv_t), s.QKV, D, D, s.QKV));
    RC(transpose2d_streaming(wt->side, dt, L.wo, const_cast<void*>(t.wo_t), D, s.OD, s.OD, D));
    RC(transpose2d_streaming(wt->side, dt, L.wgu, const_cast<void*>(t.wgu_t), 2 * c.llm_inter, D, D, 2 * c.llm_inter));
    RC(transpose2d_streaming(wt->side, dt, L.wd, const_cast<void*>(t.wd_t), D, c.llm_inter, c.llm_inter, D));
    UVX_HIP(hipEventRecord(wt->e_ready[l & 1], wt->side));
    return UVX_OK;
  };
  const void* head_t = w->lm_head_t;
  if (wts) {
    wt = wt_stream_for_device();
    UVX_CHECK(wt != nullptr, UVX_ERR_RUNTIME, "llm_bwd: could not create the weight-transpose stream");
    UVX_HIP(hipEventRecord(wt->e_start, st));                 // everything issued before (an earlier backward's reads of the buffers)
    UVX_HIP(hipStreamWaitEvent(wt->side, wt->e_start, 0));
    RC(transpose2d_streaming(wt->side, dt, w->lm_head, s.head_t, c.vocab, D, D, c.vocab));
    UVX_HIP(hipEventRecord(wt->e_head, wt->side));
    RC(issue_layer_t(c.llm_layers - 1));
    UVX_HIP(hipStreamWaitEvent(st, wt->e_head, 0));
    head_t = s.head_t;
  }

  // d logits (in place over the saved logits), then the frozen head: d_hn = dlogits . W_head
  // (labels == NULL: uvx_llm_kl_loss already replaced the saved logits by their gradient)
  if (compact_in_place || (labels && dt == DT_BF16 && g_options[3])) {
    // compact supervised rows (see uvx_llm_fwd): d logits in place, head dgrad on those rows, scattered back
    if (!compact_in_place)
      RC(ce_loss_fwd_bwd(st, dt, s.logits, labels, nullptr, s.ce_scratch, s.logits, B, T, c.vocab, c.vocab, grad_scale, s.sup));
    UVX_HIP(hipMemsetAsync(s.d_hn, 0, (size_t)M * D * esz(dt), st));
    // d_hn[rows] = d logits_c . W_head: few rows x D outputs over K = vocab -> split K for parallelism on the first
    // `cap` compact rows (f32 partials in the not-yet-used d_gu scratch, summed in a fixed order), plain GEMM beyond
    const int nkt = c.vocab / 64;
    const int cap = M < 512 ? M : 512;
    const size_t room = (size_t)M * 2 * c.llm_inter * esz(dt) / ((size_t)cap * D * sizeof(float));
    int nsplit = 1;
    for (int d = 2; d <= 24 && (size_t)d <= room; ++d)
      if (c.vocab % 64 == 0 && nkt % d == 0 && nkt / d >= 16) nsplit = d;
    if (nsplit > 1) {
      float* partial = (float*)s.d_gu;
      const int Kc = c.vocab / nsplit;
      GemmDesc g = lin(s.logits, head_t, partial, cap, D, Kc);
      g.lda = c.vocab; g.ldb = c.vocab; g.batch = nsplit; g.sA = Kc; g.sB = Kc; g.sC = (long long)cap * D; g.out_f32 = 1;
      g.m_dev = s.sup + M;
      RC(gemm(st, dt, g));
      RC(splitk_reduce_scatter(st, dt, partial, nsplit, cap, s.sup, M, s.d_hn, D));
    }
    const int first = nsplit > 1 ? cap : 0;
    if (M > first) {
      GemmDesc g = lin(at(s.logits, (size_t)first * c.vocab, dt), head_t, s.d_n, M - first, D, c.vocab);
      g.m_dev = s.sup + M; g.m_dev_off = first;
      RC(gemm(st, dt, g));
      RC(scatter_rows(st, dt, s.d_n, s.sup, M, s.d_hn, D, first));
    }
  } else {
    if (labels) RC(ce_loss_fwd_bwd(st, dt, s.logits, labels, nullptr, s.ce_scratch, s.logits, B, T, c.vocab, c.vocab, grad_scale));
    RC(gemm(st, dt, lin(s.logits, head_t, s.d_hn, M, D, c.vocab)));
  }
  const int fl = c.llm_flavor;
  const bool g3 = fl == UVX_LLM_GEMMA3;
  const float attn_scale = c.llm_attn_scale > 0.f ? c.llm_attn_scale : 1.0f / sqrtf((float)dh);
  // top_rows (uvx_llm_bwd_train, after uvx_llm_fwd_train): the last layer's stash (x_final, x_mid, gate|up) holds the
  // supervised rows only; its MLP / o_proj gradients run on those rows and are scattered back before the attention backward
  UVX_CHECK(!top_rows || (!compact_in_place && labels && dt == DT_BF16), UVX_ERR_INVALID, "llm_bwd_train: labels are required, bf16 only");
  UVX_CHECK(!compact_in_place || dt == DT_BF16, UVX_ERR_UNSUPPORTED, "llm_bwd_rows: bf16 only");
  // (uvx_llm_bwd_rows after uvx_llm_fwd_rows: the same compact last-layer stash, keyed by the caller's row list)
  const bool tc = (top_rows || (compact_in_place && !lora)) && g_options[3] && !g3;
  if (top_rows || (compact_in_place && !lora)) RC(check_pair(workspace, tc));
  const int32_t* mdev_top = tc ? s.sup + M : nullptr;
  // first_pos (uvx_llm_bwd_train_from): the caller needs no gradient below that position of any sequence - the text prefix before the first audio
  // token: under the causal mask a position only feeds later ones, so nothing the adapter training updates is reachable from it.  Below the
  // row-compacted last layer every gradient tensor then holds the positions >= rs.skip only (sequence b at row b * rs.tc): the dgrad GEMMs, the SwiGLU
  // and norm backward run on B * rs.tc rows and read the stash through the map (kernels.h RowSkip), the fused attention backward takes the
  // compacted d o / d q|k|v (AttnBwdDesc::d_first; it still needs every key for d q).  Same arithmetic per remaining row: the audio rows of
  // d_inputs_embeds are bit-identical; its rows below rs.skip are zeros.  Conditions: the training step's entry points, the bf16 attention kernels on
  // natural-layout operands, one chain, no per-row stash reader outside the kernels that take the map (the LLM adapters' products do not).
  RowSkip rs;
  {
    AttnDesc f;
    f.B = B; f.T = T; f.D = dh; f.causal = 1; f.block = 0;
    const int s16 = first_pos / 16 * 16;
    if (s16 > 0 && s16 < T && (top_rows || compact_in_place) && !lora && g_options[11] < 2 && g_options[14] && attention_bwd_takes_d_first(dt, f)) { rs.skip = s16; rs.tc = T - s16; }      // (option 14: RoPE inverted inside the attention backward, by position - rope_k would take the row index)
  }
  auto rows_bwd = [&](const LlmWs& v) -> int { return rs.skip ? v.M / T * rs.tc : v.M; };
  if (rs.skip) UVX_HIP(hipMemsetAsync(d_inputs_embeds, 0, (size_t)M * D * esz(dt), st));      // (layer 0 writes the rows >= rs.skip of every sequence)
  if (tc) {
    RC(gather_rows(st, dt, s.d_hn, s.sup, M, s.d_n, D));                 // d_hn was scattered to full rows: back to compact
    RC(rmsnorm_bwd(st, dt, s.d_n, s.x_final, w->norm, nullptr, s.dx, nullptr, M, D, c.rms_eps, fl, mdev_top));
  } else if (rs.skip) {      // (no compact last layer - Gemma-3, option 3 = 0: the final norm's backward on every row, then the kept rows to the front)
    RC(rmsnorm_bwd(st, dt, s.d_hn, s.x_final, w->norm, nullptr, s.d_n, nullptr, M, D, c.rms_eps, fl));
    RC(take_rows_from(st, dt, s.d_n, s.dx, B * rs.tc, D, rs));
  } else {
    RC(rmsnorm_bwd(st, dt, s.d_hn, s.x_final, w->norm, nullptr, s.dx, nullptr, M, D, c.rms_eps, fl));
  }
  const size_t es = esz(dt);
  // MLP half of a layer's backward on the rows of the view v: v.dx (gradient of the layer's output) -> v.dx (gradient of
  // x_mid: residual + norm branch).  compact: the supervised rows of the last layer (whole batch, device-side count).
  // MLP adapters, backward.  down_proj: dy = the gradient of the down projection's output; its input act = GLU(gate|up) is recomputed into v.act;
  // d act += u . A_d BEFORE the GLU backward.
  auto mlp_out_adapter_bwd = [&](hipStream_t sx, const LlmWs& v, const LlmLayerStash& cur, int l, const void* dy) -> int {
    const uvx_lora_proj_t& P = lora->layers[l].d;
    const int I = c.llm_inter, r = lora->r;
    RC(swiglu_fwd(sx, dt, cur.gu, v.act, v.M, I, /*layout=*/2, /*act=*/c.llm_act));
    RC(lora_transpose(sx, dt, P.b, v.lbT, D, r));
    RC(lora_apply_bwd(sx, dt, v.act, I, dy, D, v.lbT, cur.t4, v.lu4, lgrads->layers[l].d, v.M, I, D, r, lora->scaling, v.lwg, llm_wg_floats(c, s.M)));
    return lora_up(sx, dt, v.lu4, 128, P.a, 1, v.d_act, I, v.M, I, r, 1.0f, 1);
  };
  // gate_proj / up_proj: dy = the gate / up half of d gate|up, extracted into v.d_act (free once the GLU backward has consumed it); their input
  // n2 = norm(x_mid) is recomputed into v.n; d n2 += u . A after the dgrad GEMM has written v.d_n
  auto mlp_in_adapters_bwd = [&](hipStream_t sx, const LlmWs& v, const LlmLayerStash& cur, int l, const void* ln2) -> int {
    const uvx_enc_lora_layer_t& R = lora->layers[l];
    const int I = c.llm_inter, r = lora->r;
    RC(rmsnorm_fwd(sx, dt, cur.x_mid, ln2, v.n, nullptr, v.M, D, c.rms_eps, c.llm_flavor));
    for (int which = 0; which < 2; ++which) {
      const uvx_lora_proj_t& P = which ? R.u : R.g;
      if (!P.a) continue;
      void* u = at(v.lu3, 64 * which, dt);
      RC(gu_half(sx, dt, v.d_gu, v.d_act, v.M, I, which, 0));
      RC(lora_transpose(sx, dt, P.b, v.lbT, I, r));
      RC(lora_apply_bwd(sx, dt, v.n, D, v.d_act, I, v.lbT, at(cur.t3, 64 * which, dt), u, which ? lgrads->layers[l].u : lgrads->layers[l].g, v.M, D, I, r,
                        lora->scaling, v.lwg, llm_wg_floats(c, s.M)));
      RC(lora_up(sx, dt, u, 128, P.a, 1, v.d_n, D, v.M, D, r, 1.0f, 1));
    }
    return UVX_OK;
  };
  auto layer_mlp_bwd = [&](hipStream_t sx, const LlmWs& v, int l, bool compact) -> int {
    const uvx_llm_layer_t& L = w->layers[l];
    LlmLayerStash cur = llm_layer(v, l);
    const int Mv = compact ? v.M : rows_bwd(v);      // (rs: the gradient tensors hold the positions >= rs.skip only; the stash is read through the map)
    const RowSkip map = compact ? RowSkip() : rs;
    const int32_t* mdev = compact ? v.sup + v.M : nullptr;
    const bool ad_in = lora && (lora->layers[l].g.a || lora->layers[l].u.a), ad_out = lora && lora->layers[l].d.a;
    if (g3) {
      // x_out = x_mid + post_ffw_norm(m_pre): d m_pre = norm'(dx) -> d act -> d gate|up -> d n2; d x_mid = dx + pre_ffw_norm'(d n2)
      RC(rmsnorm_bwd(sx, dt, v.dx, cur.m_pre, L.ln2_post, nullptr, v.d_n, nullptr, Mv, D, c.rms_eps, fl, nullptr, nullptr, map));
      RC(gemm(sx, dt, lin_dgrad(v.d_n, layer_t(l).wd_t, L.wd, v.d_act, Mv, c.llm_inter, D)));
      if (ad_out) RC(mlp_out_adapter_bwd(sx, v, cur, l, v.d_n));      // (d m_pre: behind the post norm)
      RC(swiglu_bwd(sx, dt, v.d_act, cur.gu, v.d_gu, Mv, c.llm_inter, /*layout=*/2, /*act=*/c.llm_act, nullptr, map));
      RC(gemm(sx, dt, lin_dgrad(v.d_gu, layer_t(l).wgu_t, L.wgu, v.d_n, Mv, D, 2 * c.llm_inter)));
      if (ad_in) RC(mlp_in_adapters_bwd(sx, v, cur, l, L.ln2));
      return rmsnorm_bwd(sx, dt, v.d_n, cur.x_mid, L.ln2, v.dx, v.dx, nullptr, Mv, D, c.rms_eps, fl, nullptr, nullptr, map);
    }
    if (dt == DT_BF16 && g_options[2] && fl == UVX_LLM_LLAMA && !ad_out && !map.skip) {   // d act = dx . W_down^T with the SwiGLU backward fused into the epilogue: writes d gate|up directly
      GemmDesc g = lin_dgrad(v.dx, layer_t(l).wd_t, L.wd, v.d_gu, Mv, c.llm_inter, D);
      g.ldc = 2 * c.llm_inter; g.C2 = cur.gu; g.ldc2 = 2 * c.llm_inter; g.swiglu = 2; g.m_dev = mdev;
      RC(gemm(sx, dt, g));
    } else {
      GemmDesc g = lin_dgrad(v.dx, layer_t(l).wd_t, L.wd, v.d_act, Mv, c.llm_inter, D);
      g.m_dev = mdev;
      RC(gemm(sx, dt, g));
      if (ad_out) RC(mlp_out_adapter_bwd(sx, v, cur, l, v.dx));
      if (!probe_skip(4)) RC(swiglu_bwd(sx, dt, v.d_act, cur.gu, v.d_gu, Mv, c.llm_inter, /*layout=*/2, /*act=*/c.llm_act, mdev, map));
    }
    {
      GemmDesc g = lin_dgrad(v.d_gu, layer_t(l).wgu_t, L.wgu, v.d_n, Mv, D, 2 * c.llm_inter);
      g.m_dev = mdev;
      RC(gemm(sx, dt, g));
    }
    if (ad_in) RC(mlp_in_adapters_bwd(sx, v, cur, l, L.ln2));
    return probe_skip(8) ? UVX_OK : rmsnorm_bwd(sx, dt, v.d_n, cur.x_mid, L.ln2, v.dx, v.dx, nullptr, Mv, D, c.rms_eps, fl, mdev, nullptr, map);
  };
  // attention half: v.dx (gradient of x_mid) -> dx_out (gradient of the layer's input).  d_o_ready: v.d_o and the residual
  // gradient `resid` were already produced for the whole batch (compact last layer), else d_o = dx . W_o^T here.
  auto layer_attn_bwd = [&](hipStream_t sx, const LlmWs& v, int Bv, int l, bool d_o_ready, const void* resid, void* dx_out, bool dx_full = false) -> int {
    const uvx_llm_layer_t& L = w->layers[l];
    LlmLayerStash cur = llm_layer(v, l);
    const int Mv = rows_bwd(v);      // (dx_full: dx_out keeps every row - layer 0 writes the caller's d_inputs_embeds through the map)
    if (g3) {      // x_mid = x_in + post_attention_norm(o_pre): d o_pre = norm'(d x_mid), then the o_proj dgrad
      RC(rmsnorm_bwd(sx, dt, v.dx, cur.o_pre, L.ln1_post, nullptr, v.d_n, nullptr, Mv, D, c.rms_eps, fl, nullptr, nullptr, rs));
      RC(gemm(sx, dt, lin_dgrad(v.d_n, layer_t(l).wo_t, L.wo, v.d_o, Mv, s.OD, D)));
    } else if (!d_o_ready) RC(gemm(sx, dt, lin_dgrad(v.dx, layer_t(l).wo_t, L.wo, v.d_o, Mv, s.OD, D)));
    const long long lwg_floats = llm_wg_floats(c, s.M);
    if (lora && lora->layers[l].o.a) {   // o_proj adapter: its gradients from d (o_proj output) - Gemma-3: behind the post norm -, and d o += u . A_o
      const void* d_y = g3 ? v.d_n : v.dx;
      RC(lora_apply_bwd(sx, dt, cur.o, s.OD, d_y, D, cur.boT, at(cur.t2, 64, dt), at(v.lu2, 64, dt), lgrads->layers[l].o, Mv, s.OD, D, lora->r, lora->scaling,
                        v.lwg, lwg_floats));
      RC(lora_up(sx, dt, at(v.lu2, 64, dt), 128, lora->layers[l].o.a, 1, v.d_o, s.OD, Mv, s.OD, lora->r, 1.0f, 1));
    }
    if (attention_needs_transposed_copies(dt)) RC(heads_transpose(sx, dt, cur.qkv, v.qT, Bv, T, s.Tp, Hq, dh, s.QKV));
    if (attention_needs_transposed_copies(dt)) RC(heads_transpose(sx, dt, at(cur.qkv, (size_t)Hq * dh, dt), v.kT, Bv, T, s.Tp, Hkv, dh, s.QKV));
    if (attention_needs_transposed_copies(dt)) RC(heads_transpose(sx, dt, v.d_o, v.doT, Bv, T, s.Tp, Hq, dh, s.OD));
    AttnBwdDesc bd;
    AttnDesc& ad = bd.f;
    ad.q = cur.qkv; ad.k = at(cur.qkv, (size_t)Hq * dh, dt); ad.v = at(cur.qkv, (size_t)(Hq + Hkv) * dh, dt);
    ad.o = cur.o; ad.lse = cur.lse;
    ad.kv_start = v.kvs; ad.kv_len = v.kvl;  // written by the forward pass
    ad.B = Bv; ad.T = T; ad.Tp = s.Tp; ad.Hq = Hq; ad.Hkv = Hkv; ad.D = dh;
    ad.ldq = ad.ldk = ad.ldv = s.QKV; ad.ldo = s.OD; ad.causal = 1; ad.block = 0;
    ad.scale = attn_scale;
    ad.window = c.llm_window > 0 && T > c.llm_window && w->layer_local && w->layer_local[l] ? c.llm_window : 0;      // (any flavour: Gemma-3's local layers, every Mistral layer)
    bd.dout = v.d_o; bd.qt = v.qT; bd.kt = v.kT; bd.dot = v.doT; bd.delta = v.delta; bd.dkv_part = v.dkv_part;
    bd.dq = v.d_qkv; bd.dk = at(v.d_qkv, (size_t)Hq * dh, dt); bd.dv = at(v.d_qkv, (size_t)(Hq + Hkv) * dh, dt);
    bd.lddq = bd.lddk = bd.lddv = s.QKV;
    bd.d_first = rs.skip;
    // the bf16 kernels write dq / dk RoPE-inverted (epilogue of the dQ kernel, GQA group reduction): no separate pass
    const bool rope_fused = attention_bwd_fuses_rope(dt) && g_options[14];
    const float* rope = g3 && w->layer_local && w->layer_local[l] ? w->rope_cos_sin_local : w->rope_cos_sin;
    if (rope_fused) bd.rope_cos_sin = rope;
    if (!probe_skip(1)) RC(attention_bwd(sx, dt, bd));
    if (!rope_fused) RC(rope_inplace(sx, dt, v.d_qkv, rope, nullptr, Mv, T, Hq + Hkv, dh, s.QKV, 1));
    if (c.llm_qk_norm) RC(qk_norm_bwd(sx, dt, v.d_qkv, cur.qk_raw, L.q_norm, L.k_norm, Mv, Hq, Hkv, dh, s.QKV, c.rms_eps, g3 ? 1 : 0, rs));
    RC(gemm(sx, dt, lin_dgrad(v.d_qkv, layer_t(l).wqkv_t, L.wqkv, v.d_n, Mv, D, s.QKV)));
    if (lora) {   // LoRA gradients of q_proj / k_proj (/ v_proj) and their contribution to d n1 (rank-r products, lora.hip)
      const uvx_enc_lora_layer_t& R = lora->layers[l];
      const uvx_enc_lora_layer_grads_t& G = lgrads->layers[l];
      const int r = lora->r, qc = Hq * dh, kc = Hkv * dh;
      void* dk = at(v.d_qkv, (size_t)qc, dt);
      void* dv = at(v.d_qkv, (size_t)(qc + kc), dt);
      if (R.q.a || R.k.a || R.v.a) RC(rmsnorm_fwd(sx, dt, cur.x_in, L.ln1, v.n, nullptr, Mv, D, c.rms_eps, fl));        // n1 recomputed
      if (R.q.a && R.k.a) {
        RC(lora_down(sx, dt, v.d_qkv, s.QKV, cur.bqT, 0, v.lu, 128, Mv, qc, r, lora->scaling));
        RC(lora_down(sx, dt, dk, s.QKV, cur.bkT, 0, at(v.lu, 64, dt), 128, Mv, kc, r, lora->scaling));
        const LoraWgradItem items[4] = {{v.n, D, v.lu, 128, G.q.a, D, 0, 1.0f}, {v.n, D, at(v.lu, 64, dt), 128, G.k.a, D, 0, 1.0f},
                                        {v.d_qkv, s.QKV, cur.t, 128, G.q.b, qc, 1, lora->scaling}, {dk, s.QKV, at(cur.t, 64, dt), 128, G.k.b, kc, 1, lora->scaling}};
        RC(lora_wgrad_batch(sx, dt, items, 4, Mv, r, v.lwg, lwg_floats));
      } else {
        if (R.q.a) RC(lora_apply_bwd(sx, dt, v.n, D, v.d_qkv, s.QKV, cur.bqT, cur.t, v.lu, G.q, Mv, D, qc, r, lora->scaling, v.lwg, lwg_floats));
        if (R.k.a) RC(lora_apply_bwd(sx, dt, v.n, D, dk, s.QKV, cur.bkT, at(cur.t, 64, dt), at(v.lu, 64, dt), G.k, Mv, D, kc, r, lora->scaling, v.lwg, lwg_floats));
      }
      if (R.v.a) RC(lora_apply_bwd(sx, dt, v.n, D, dv, s.QKV, cur.bvT, cur.t2, v.lu2, G.v, Mv, D, kc, r, lora->scaling, v.lwg, lwg_floats));
      if (R.q.a) RC(lora_up(sx, dt, v.lu, 128, R.q.a, 1, v.d_n, D, Mv, D, r, 1.0f, 1));
      if (R.k.a) RC(lora_up(sx, dt, at(v.lu, 64, dt), 128, R.k.a, 1, v.d_n, D, Mv, D, r, 1.0f, 1));
      if (R.v.a) RC(lora_up(sx, dt, v.lu2, 128, R.v.a, 1, v.d_n, D, Mv, D, r, 1.0f, 1));
    }
    return probe_skip(8) ? UVX_OK : rmsnorm_bwd(sx, dt, v.d_n, cur.x_in, L.ln1, resid, dx_out, nullptr, Mv, D, c.rms_eps, fl, nullptr, nullptr, rs, dx_full && rs.skip);
  };
  for (int l = 0; l < c.llm_layers; ++l) {
    const uvx_llm_layer_t& L = w->layers[l];
    UVX_CHECK(wts || dt == DT_BF16 || (L.wd_t && L.wgu_t && L.wo_t && L.wqkv_t), UVX_ERR_INVALID,
              "llm_bwd: layer %d lacks transposed weights (f32: the NN form of the dgrads exists on the bf16 path only)", l);
  }
  const int top = c.llm_layers - 1;
  if (wts) UVX_HIP(hipStreamWaitEvent(st, wt->e_ready[top & 1], 0));
  if (tc) {   // last layer of the training pair: MLP and o_proj gradients on the compact supervised rows (whole batch, this
              // stream), then d o and the residual-stream gradient go back to their full rows for the attention backward
    RC(layer_mlp_bwd(st, s, top, true));
    GemmDesc g = lin_dgrad(s.dx, layer_t(top).wo_t, w->layers[top].wo, s.doT, M, s.OD, D);     // doT ([B, Hq, Tp, dh] >= M * OD) is free until the transpose
    g.m_dev = mdev_top;
    RC(gemm(st, dt, g));
    const int32_t* rows_to = s.sup;
    if (rs.skip) {      // the supervised rows' places among the row-compacted gradients
      RC(compact_row_list(st, s.sup, s.sup_c, M, T, rs.skip));
      rows_to = s.sup_c;
    }
    UVX_HIP(hipMemsetAsync(s.d_o, 0, (size_t)M * s.OD * es, st));
    RC(scatter_rows(st, dt, s.doT, rows_to, M, s.d_o, s.OD));
    UVX_HIP(hipMemsetAsync(s.d_hn, 0, (size_t)M * D * es, st));
    RC(scatter_rows(st, dt, s.dx, rows_to, M, s.d_hn, D));
  }
  // schedule: one chain on the caller's stream, or (option 11) the batch slices on several streams - see Fork above
  const Chains ch = make_chains(st, s, c, B, T, dt == DT_BF16 && !lora && !wts);
  RC(chains_fork(ch));
  int rc_layers = UVX_OK;
  for (int l = top; l >= 0 && rc_layers == UVX_OK; --l) {
    const bool compact = tc && l == top;
    if (wts) {      // this layer's W^T must have landed; the next one's is started now, into the buffer layer l + 1 has released
      if (l != top && hipStreamWaitEvent(st, wt->e_ready[l & 1], 0) != hipSuccess) rc_layers = UVX_ERR_RUNTIME;
      if (l > 0 && rc_layers == UVX_OK) rc_layers = issue_layer_t(l - 1);
    }
    for (int h = 0; h < ch.n && rc_layers == UVX_OK; ++h) {
      const LlmWs& v = ch.v[h];
      if (!compact) rc_layers = layer_mlp_bwd(ch.st[h], v, l, false);
      void* dx_out = l == 0 ? (void*)((char*)d_inputs_embeds + (size_t)ch.b0[h] * T * D * es) : v.dx;
      if (rc_layers == UVX_OK) rc_layers = layer_attn_bwd(ch.st[h], v, ch.b0[h + 1] - ch.b0[h], l, compact, compact ? v.d_hn : v.dx, dx_out, l == 0);
    }
    if (wts && hipEventRecord(wt->e_free[l & 1], st) != hipSuccess && rc_layers == UVX_OK) rc_layers = UVX_ERR_RUNTIME;
  }
  RC(chains_join(ch));   // (also after an error above: the side streams must not be left forked)
  RC(rc_layers);
  if (fl == UVX_LLM_GEMMA) RC(scale_inplace(st, dt, d_inputs_embeds, (long long)M * D, gemma_normalizer(c)));   // d (x * normalizer)
  return UVX_OK;
}

extern "C" int32_t uvx_llm_bwd(void* stream, const uvx_config_t* cfg, const uvx_llm_weights_t* w, const int64_t* labels,
                               int32_t B, int32_t T, float grad_scale, void* d_inputs_embeds, void* workspace,
                               size_t ws_bytes) {
  return llm_backward(stream, cfg, w, labels, B, T, grad_scale, d_inputs_embeds, workspace, ws_bytes, false);
}

// The adapter-training step's own pair (include/uvx.h): identical loss and gradients to uvx_llm_fwd(save_for_bwd = 1, logits =
// NULL) + uvx_llm_bwd, with the last layer's o_proj / MLP / final norm and their gradients on the supervised rows only.
extern "C" int32_t uvx_llm_fwd_train(void* stream, const uvx_config_t* cfg, const uvx_llm_weights_t* w, const void* inputs_embeds,
                                     const int64_t* attention_mask, const int64_t* labels, int32_t B, int32_t T, float* loss,
                                     void* workspace, size_t ws_bytes) {
  return llm_forward(stream, cfg, w, inputs_embeds, attention_mask, labels, B, T, nullptr, loss, 1, workspace, ws_bytes, nullptr, 0,
                     nullptr, nullptr, true);
}
extern "C" int32_t uvx_llm_bwd_train(void* stream, const uvx_config_t* cfg, const uvx_llm_weights_t* w, const int64_t* labels,
                                     int32_t B, int32_t T, float grad_scale, void* d_inputs_embeds, void* workspace,
                                     size_t ws_bytes) {
  return llm_backward(stream, cfg, w, labels, B, T, grad_scale, d_inputs_embeds, workspace, ws_bytes, false, nullptr, nullptr, true);
}
// ... when the caller needs no gradient below position first_pos of any sequence (include/uvx.h)
extern "C" int32_t uvx_llm_bwd_train_from(void* stream, const uvx_config_t* cfg, const uvx_llm_weights_t* w, const int64_t* labels,
                                          int32_t B, int32_t T, int32_t first_pos, float grad_scale, void* d_inputs_embeds, void* workspace,
                                          size_t ws_bytes) {
  return llm_backward(stream, cfg, w, labels, B, T, grad_scale, d_inputs_embeds, workspace, ws_bytes, false, nullptr, nullptr, true, first_pos);
}

// LLM under LoRA training (text_model_lora_config.r > 0, apply_lora on the language model, ultravox_model.py:500-526):
// the forward adds the adapters to q_proj / k_proj, the backward also returns their gradients.
extern "C" int32_t uvx_llm_fwd_lora(void* stream, const uvx_config_t* cfg, const uvx_llm_weights_t* w, const uvx_encoder_lora_t* lora,
                                    const void* inputs_embeds, const int64_t* attention_mask, const int64_t* labels, int32_t B,
                                    int32_t T, void* logits, float* loss, int32_t save_for_bwd, void* workspace, size_t ws_bytes) {
  RC(check_cfg(cfg));
  RC(lora_check(lora, cfg->llm_layers, nullptr, "llm_fwd_lora"));
  return llm_forward(stream, cfg, w, inputs_embeds, attention_mask, labels, B, T, logits, loss, save_for_bwd, workspace, ws_bytes,
                     nullptr, 0, nullptr, lora);
}
extern "C" int32_t uvx_llm_bwd_lora(void* stream, const uvx_config_t* cfg, const uvx_llm_weights_t* w, const uvx_encoder_lora_t* lora,
                                    const int64_t* labels, int32_t B, int32_t T, float grad_scale, void* d_inputs_embeds,
                                    const uvx_encoder_lora_grads_t* grads, void* workspace, size_t ws_bytes) {
  RC(check_cfg(cfg));
  UVX_CHECK(grads && grads->layers, UVX_ERR_INVALID, "llm_bwd_lora: bad LoRA descriptor");
  RC(lora_check(lora, cfg->llm_layers, grads, "llm_bwd_lora"));
  return llm_backward(stream, cfg, w, labels, B, T, grad_scale, d_inputs_embeds, workspace, ws_bytes, false, lora, grads);
}

extern "C" int32_t uvx_llm_bwd_rows(void* stream, const uvx_config_t* cfg, const uvx_llm_weights_t* w, int32_t B, int32_t T,
                                    void* d_inputs_embeds, void* workspace, size_t ws_bytes) {
  return llm_backward(stream, cfg, w, nullptr, B, T, 1.0f, d_inputs_embeds, workspace, ws_bytes, true);
}
extern "C" int32_t uvx_llm_bwd_rows_from(void* stream, const uvx_config_t* cfg, const uvx_llm_weights_t* w, int32_t B, int32_t T, int32_t first_pos,
                                         void* d_inputs_embeds, void* workspace, size_t ws_bytes) {
  return llm_backward(stream, cfg, w, nullptr, B, T, 1.0f, d_inputs_embeds, workspace, ws_bytes, true, nullptr, nullptr, false, first_pos);
}

extern "C" int32_t uvx_adamw_clip_step(void* stream, int32_t state_dtype, void* param, float* master, const float* grad,
                                       void* m, void* v, int64_t n, float max_norm, float lr, float beta1, float beta2,
                                       float eps, float weight_decay, int32_t step, float* scratch) {
  hipStream_t st = (hipStream_t)stream;
  UVX_CHECK(param && grad && m && v && scratch, UVX_ERR_INVALID, "adamw: null argument");
  RC(uvx::grad_sq_norm(st, grad, n, scratch + 1, scratch));
  return uvx::adamw_clip_step(st, state_dtype, param, master, grad, m, v, n, scratch, max_norm, lr, beta1, beta2, eps,
                              weight_decay, step);
}
