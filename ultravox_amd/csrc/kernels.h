// Internal C++ launcher interface shared by the kernel translation units and the C-ABI layer.
// (Nothing here is exported; the exported surface is include/uvx.h.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/uvx.h"      // (the LoRA descriptor structs the helpers at the end of lora.hip take)

namespace uvx {

enum DType { DT_BF16 = 0, DT_F32 = 1 };

// ---- optional per-kernel timing with HIP events on the launch stream (bench.py's roofline leg) ----
enum ProfClass { PROF_GEMM = 0, PROF_ATTN = 1, PROF_OTHER = 2, PROF_NCLASS = 3 };
void prof_record_begin(hipStream_t st, int cls, double flops, double bytes);
void prof_record_end(hipStream_t st);
void prof_tag(int m, int n, int k, int batch, int variant);
bool prof_take(int cls, double flops, double bytes, hipEvent_t* a, hipEvent_t* b);
void prof_commit();
extern bool g_prof_on;
struct ProfScope {
  hipStream_t st; bool on;
  ProfScope(hipStream_t s, int cls, double flops, double bytes, bool enable = true) : st(s), on(g_prof_on && enable) { if (on) prof_record_begin(s, cls, flops, bytes); }
  ~ProfScope() { if (on) prof_record_end(st); }
};

struct GemmDesc {
  const void* A = nullptr;  // [M, K], row stride lda (activations)
  const void* B = nullptr;  // [N, K], row stride ldb (nn.Linear weight layout)
  void* C = nullptr;        // [M, N], row stride ldc
  const void* bias = nullptr;      // [N] or null
  const void* residual = nullptr;  // [M or res_mod, N] (row stride ldr) or null
  int M = 0, N = 0, K = 0;
  int lda = 0, ldb = 0, ldc = 0, ldr = 0;
  int res_mod = 0;   // >0: residual row = m % res_mod (positional-embedding add)
  int batch = 1;
  long long sA = 0, sB = 0, sC = 0, sR = 0;  // batch strides in elements
  int act = 0;       // 0 none, 1 exact-erf GELU; bf16 path, round 6 (the Whisper tower under LoRA training):
                     // 2 = GELU that KEEPS the pre-activation: C = round(acc * alpha + bias), C2 [M, N] (row stride ldc2) = round(gelu(C))
                     //     - what gemm + gelu_fwd produce, bit for bit;
                     // 3 = GELU BACKWARD: C = round(round(acc * alpha) * gelu'(C2)) with C2 [M, N] the saved pre-activation - what
                     //     gemm + gelu_bwd produce, bit for bit.  No residual / swiglu / f32 output / split-K with 2 and 3.
  int out_f32 = 0;   // bf16 path only: write f32 (wgrad)
  int accumulate = 0;
  float alpha = 1.0f;
  // bf16 path, round 6: B is stored [K, N] (row stride ldb) - C = A . B, the "NN" form: a dgrad d x = d y . W reads the forward weight
  // W [N_out, N_in] as it lies instead of a transposed copy.  Merged-phase tiles only (N % 8 == 0); bit-identical to the NT kernel on B^T.
  int b_kn = 0;
  // fused LlamaMLP activation (bf16 path): B's rows alternate 16 gate / 16 up rows; C gets gate|up in that
  // interleaved order, C2 [M, N/2] (row stride ldc2) gets silu(gate) * up
  void* C2 = nullptr;
  int ldc2 = 0;
  int swiglu = 0;
  // device-side row count: effective M = min(M, *m_dev) (tiles beyond it exit at once); M stays the launch bound
  const int32_t* m_dev = nullptr;
  int m_dev_off = 0;   // effective M = clamp(*m_dev - m_dev_off, 0, M): this launch covers compact rows [m_dev_off, ...)
  // split-K (round 5, bf16 path; gemm.hip "Split-K"): scratch the caller lends for f32 partial tiles (gemm_splitk_ws_bytes).  With it,
  // gemm_nt cuts a tile's K loop over several blocks when the tile count would leave most of the CUs idle (the prefill's M = 64..700
  // rows) and a reduce kernel runs the epilogue; null = never split.  splitk_force: 0 = the cost model decides, 1 = never, s > 1 = that factor.
  void* splitk_ws = nullptr;
  size_t splitk_ws_bytes = 0;
  int splitk_force = 0;
  // The RMSNorm that FOLLOWS this linear (bf16; decoder stack: o_proj -> post_attention_layernorm, down_proj -> the next layer's
  // input_layernorm): when the problem is split, the reduce kernel also writes norm_out [M, N] (row stride norm_ld) = RMSNorm(C row; norm_w,
  // norm_eps, flavor as rmsnorm_fwd) - bit for bit what rmsnorm_fwd computes from C - and sets *norm_done (host side, may be null).  A launch
  // that is not split ignores these fields and leaves *norm_done alone: the caller then runs rmsnorm_fwd itself.
  const void* norm_w = nullptr;
  void* norm_out = nullptr;
  int norm_ld = 0;
  float norm_eps = 0.f;
  int norm_flavor = 0;
  bool* norm_done = nullptr;
};
size_t gemm_splitk_ws_bytes(int M, int N);   // enough scratch for any split gemm_nt picks on an M x N output
int gemm_pick_split(int M, int N, int K, size_t ws_bytes, int* variant);   // host only: split factor (1 = none) and tile the model picks

extern int g_gemm_variant;
extern int g_gemm_split;
extern int g_options[32];
int gemm_pick_variant(int M, int N, int K, int batch);  // host only: the tile variant the cost model picks
int gemm_streamk_timeouts();   // stream-K waits that gave up since the last call (0 = healthy); synchronises
extern int g_gemm_ovr_n;
extern int g_gemm_ovr[32][4];
// bf16 operands, f32 accumulate (MFMA 16x16x32).
int gemm_nt(hipStream_t st, const GemmDesc& d);
// gemm_skinny.hip: weight-streaming kernel for M <= 16 rows (the decode step); gemm_nt dispatches to it
bool gemm_skinny_applicable(const GemmDesc& d);
int gemm_skinny_bf16(hipStream_t st, const GemmDesc& d);
// decode step: RMSNorm fused into the row-streaming kernel's input (M <= 2); UVX_ERR_UNSUPPORTED (nothing launched) when it does not apply
int gemm_skinny_rmsnorm_bf16(hipStream_t st, const GemmDesc& d, const void* norm_w, float eps, int flavor);
// f32 operands/outputs (parity mode; MFMA 16x16x4 f32).
int gemm_nt_f32(hipStream_t st, const GemmDesc& d);
// dispatch on dtype
inline int gemm(hipStream_t st, int dtype, const GemmDesc& d) {
  return dtype == DT_BF16 ? gemm_nt(st, d) : gemm_nt_f32(st, d);
}

// Row-compacted backward (model.hip: the LLM backward below the first position that needs a gradient): a gradient tensor holds only the positions
// >= skip of every sequence, sequence b at row b * tc (tc = T - skip), while the forward stash keeps all T rows per sequence.  Row r of the
// compact tensor is row r + (r / tc + 1) * skip of the stash.  skip == 0: no mapping.
struct RowSkip {
  int tc = 0, skip = 0;
};

// ---- norms.hip ----
int layernorm_fwd(hipStream_t st, int dtype, const void* x, const void* w, const void* b, void* y,
                  int rows, int cols, float eps);
// y = rmsnorm(x) * w ; optionally stores rstd[rows] (f32) for backward
// flavor 0 = LlamaRMSNorm: w * round(x * rstd) (two roundings, as HF computes it); flavor 1 = GemmaRMSNorm: the whole of
// x * rstd * (1 + w) in f32, ONE rounding ([3P] modeling_gemma.py GemmaRMSNorm.forward).
int rmsnorm_fwd(hipStream_t st, int dtype, const void* x, const void* w, void* y, float* rstd,
                int rows, int cols, float eps, int flavor = 0, const int32_t* rows_dev = nullptr, const void* resid = nullptr);
// (resid: y = round(rmsnorm(x)) + resid - Gemma-3's post norms feed the residual add directly)
// dx = d(rmsnorm)/dx (+ dx_add if given: residual-stream gradient), optional dw partial accumulation
// (f32 [cols], atomically accumulated; must be zeroed by the caller).
int rmsnorm_bwd(hipStream_t st, int dtype, const void* dy, const void* x, const void* w,
                const void* dx_add, void* dx, float* dw, int rows, int cols, float eps, int flavor = 0,
                const int32_t* rows_dev = nullptr, float* dw_part = nullptr, RowSkip x_map = RowSkip(), bool map_dx = false);
// x_map: dy / dx_add / dx are row-compacted, x is read through the map; map_dx: dx is written through the map as well (full-row output)
// dw_part: scratch of rmsnorm_bwd_dw_scratch_floats(rows, cols) floats - the weight gradient is then summed over the blocks in a fixed order
// (dw += the sum; bit-reproducible) instead of with atomics
long long rmsnorm_bwd_dw_scratch_floats(int rows, int cols);
// StackAudioFrames + RMSNorm (ultravox_model.py:722-730, 791): x [B, T, C] -> y [B, Tp/S, C*S]
int stack_rmsnorm_fwd(hipStream_t st, int dtype, const void* x, const void* w, void* y, void* stacked,
                      int B, int T, int C, int S, float eps);

// ---- elementwise.hip ----
// layout: 0 = [value | gate] halves (UltravoxProjector), 1 = [gate | up] halves, 2 = 16-wide gate/up blocks interleaved
// act: 0 = SiLU (SwiGLU: Llama MLP, UltravoxProjector), 1 = tanh-GELU (GeGLU: Gemma MLP, hidden_act gelu_pytorch_tanh),
// 2 = exact erf GELU (Gemma checkpoints whose config says hidden_act "gelu")
// rows_dev (here and in the rmsnorm entry points): device-side row count - rows at or beyond *rows_dev are skipped (the row-compacted last
// layer of the training pair: `rows` is only the launch bound)
int swiglu_fwd(hipStream_t st, int dtype, const void* in, void* out, int rows, int half, int gate_first, int act = 0,
               const int32_t* rows_dev = nullptr);
int swiglu_bwd(hipStream_t st, int dtype, const void* dout, const void* in, void* din, int rows, int half,
               int gate_first, int act = 0, const int32_t* rows_dev = nullptr, RowSkip in_map = RowSkip());
// in_map: dout / din are row-compacted, `in` (the stashed gate|up) is read through the map
// out[i] = rows[i] - (rows[i] / T + 1) * skip for i < rows[n] (RowSkip: the index of a full-layout row among the row-compacted gradients; a row of a
// position below `skip` has none and goes to the unused last row n - 1), out[n] = rows[n]
int compact_row_list(hipStream_t st, const int32_t* rows, int32_t* out, int n, int T, int skip);
// dst[r] = src[r + (r / map.tc + 1) * map.skip] for r < rows: the rows a RowSkip keeps, moved to the front (dst != src)
int take_rows_from(hipStream_t st, int dtype, const void* src, void* dst, int rows, int cols, RowSkip map);
// one half (0 gate / 1 up) of an interleaved gate|up tensor [M, 2 I] <-> contiguous [M, I]: add != 0: gu += flat (rounded once); else flat = gu
int gu_half(hipStream_t st, int dtype, void* gu, void* flat, long long M, int I, int which, int add);
// x[i] = round(x[i] * s) in place over n elements (Gemma: inputs_embeds * sqrt(hidden_size), and its gradient)
int scale_inplace(hipStream_t st, int dtype, void* x, long long n, float s);
int rope_inplace(hipStream_t st, int dtype, void* qkv, const float* cos_sin, const int32_t* pos, int rows,
                 int T, int n_heads_rot, int head_dim, int ld, int inverse);
// Qwen3 q_norm / k_norm (per-head RMSNorm, weights [head_dim]) fused with the rotary embedding, in place on the q | k columns of a
// [rows, ld] q|k|v buffer; raw (or null) receives the un-normalised q | k rows [rows, (Hq + Hkv) * head_dim] for qk_norm_bwd,
// which turns the gradient of the normalised rows (q | k columns of d_qkv) into the gradient of the raw ones, in place.
int qk_norm_rope(hipStream_t st, int dtype, void* qkv, const void* wq, const void* wk, void* raw, const float* cos_sin,
                 const int32_t* pos, int rows, int T, int Hq, int Hkv, int head_dim, int ld, float eps, int flavor = 0);
int qk_norm_bwd(hipStream_t st, int dtype, void* d_qkv, const void* raw, const void* wq, const void* wk, int rows, int Hq, int Hkv,
                int head_dim, int ld, float eps, int flavor = 0, RowSkip raw_map = RowSkip());
int embed_gather(hipStream_t st, int dtype, const void* table, const int64_t* ids, void* out, int rows,
                 int D, int vocab);
// owner[B*T] / item_batch[n_items] are int32 scratch filled by merge_owner and reused by the backward.
int merge_owner(hipStream_t st, int32_t* owner, int32_t* item_batch, const int64_t* audio_batch_size,
                const int64_t* start, const int32_t* len, int B, int n_items, int T, int Na);
// bwd = 0: embeds[b, start+j] = audio[a, j];  bwd = 1: daudio[a, j] = (d)embeds[b, start+j] (0 where not owner)
int merge_audio(hipStream_t st, int dtype, void* embeds, const void* audio, void* daudio, const int32_t* owner,
                const int32_t* item_batch, const int64_t* start, const int32_t* len, int n_items, int T, int D,
                int Na, int bwd);
int transpose2d(hipStream_t st, int dtype, const void* in, void* out, int rows, int cols, int ld_in,
                int ld_out, int batch, long long s_in, long long s_out);
// the same for one matrix that is read and written once and is much larger than the caches (non-temporal accesses)
int transpose2d_streaming(hipStream_t st, int dtype, const void* in, void* out, int rows, int cols, int ld_in, int ld_out);
int im2col_conv1(hipStream_t st, int dtype, const void* mel, int mel_is_f32, void* out, int B, int n_mels,
                 int F, int F_stride, int Kp);
int add_rows(hipStream_t st, int dtype, const void* a, const void* b, void* out, long long n);
int cast_f32_to(hipStream_t st, int dtype, const float* in, void* out, long long n);
int fill_zero(hipStream_t st, void* p, long long bytes);
// encoder LoRA training: GELU as a separate pass on the stashed pre-activation, its exact-derivative backward,
// LayerNorm backward for a frozen affine (dx only, + optional residual)
// un-gated activation y = act(x) and its backward (the projector with projector_act != "swiglu"): act 0 silu, 1 tanh-GELU, 2 exact GELU, 3 relu
int act_fwd(hipStream_t st, int dtype, const void* in, void* out, long long n, int act);
int act_bwd(hipStream_t st, int dtype, const void* dout, const void* in, void* din, long long n, int act);
int gelu_fwd(hipStream_t st, int dtype, const void* pre, void* out, long long n);
int gelu_bwd(hipStream_t st, int dtype, const void* dout, const void* pre, void* din, long long n);
int layernorm_bwd(hipStream_t st, int dtype, const void* dy, const void* x, const void* w, const void* dx_add, void* dx,
                  int rows, int cols, float eps);
// ---- lora.hip: rank-r products on the VALU (peft layouts: A [r, C], B [C, r]) ----
// Y[M, r] = round(alpha * sum_c X[m, c] * W(j, c));  W stored [r][C] (w_is_cr = 0) or [C][r] (1)
int lora_down(hipStream_t st, int dtype, const void* X, long long ldx, const void* W, int w_is_cr, void* Y, long long ldy,
              long long M, int C, int r, float alpha);
// Z[M, C] (+)= round(alpha * sum_j Y[m, j] * W(c, j));  W stored [C][r] (w_is_rc = 0) or [r][C] (1).  r <= 8: Y rows must be padded to 8
// columns (zeros beyond r, as lora_down leaves them) with ldy % 8 == 0 and 16-byte-aligned Y / W / Z - checked.
int lora_up(hipStream_t st, int dtype, const void* Y, long long ldy, const void* W, int w_is_rc, void* Z, long long ldz,
            long long M, int C, int r, float alpha, int accumulate);
// round 6: the q_proj / k_proj pair of a layer in ONE launch each (blockIdx.y = which; bit-identical to two calls; tuning option 22 = 1: two launches)
int lora_down2(hipStream_t st, int dtype, const void* X0, const void* X1, long long ldx, const void* W0, const void* W1, void* Y0, void* Y1,
               long long ldy, long long M, int C, int r, float alpha0, float alpha1);
int lora_up2(hipStream_t st, int dtype, const void* Y0, const void* Y1, long long ldy, const void* W0, const void* W1, void* Z0, void* Z1,
             long long ldz, long long M, int C0, int C1, int r, float alpha0, float alpha1);
// out = alpha * Y[M, r]^T . X[M, C]  as [r][C] or, transpose_out, [C][r]  (f32; scratch: lora_wgrad_scratch_floats)
long long lora_wgrad_scratch_floats(long long M, int C, int r);
int lora_transpose(hipStream_t st, int dtype, const void* in, void* out, int C, int r);   // [C, r] -> [r, C]
int lora_transpose2(hipStream_t st, int dtype, const void* in0, void* out0, int C0, const void* in1, void* out1, int C1, int r);   // two of them, one launch
// up to four lora_wgrad products over the same rows and rank with ONE reduce launch (bit-identical to separate lora_wgrad calls)
struct LoraWgradItem { const void* X; long long ldx; const void* Y; long long ldy; float* out; int C; int transpose_out; float alpha; };
int lora_wgrad_batch(hipStream_t st, int dtype, const LoraWgradItem* items, int n, long long M, int r, float* scratch, long long scratch_floats);
int lora_wgrad(hipStream_t st, int dtype, const void* X, long long ldx, const void* Y, long long ldy, float* out, long long M,
               int C, int r, int transpose_out, float alpha, float* scratch);

// whole adapted linears (peft Linear.forward / its autograd) built from the products above; descriptor check.  (The structs are include/uvx.h's.)
int lora_apply(hipStream_t st, int dt, const void* x, long long ldx, const uvx_lora_proj_t& P, void* bT, void* t, void* y, long long ldy,
               long long M, int cin, int cout, int r, float scale);
int lora_apply_bwd(hipStream_t st, int dt, const void* x, long long ldx, const void* dy, long long lddy, const void* bT, const void* t, void* u,
                   const uvx_lora_proj_grad_t& G, long long M, int cin, int cout, int r, float scale, float* scratch, long long scratch_floats);
int lora_check(const uvx_encoder_lora_t* lora, int n_layers, const uvx_encoder_lora_grads_t* grads, const char* who);

// ---- attention.hip ----
struct AttnDesc {
  const void* q = nullptr;   // [B, T, Hq, D] with row (token) stride ldq elements
  const void* k = nullptr;   // [B, T, Hkv, D] row stride ldk
  const void* v = nullptr;   // [B, T, Hkv, D] row stride ldv (backward only)
  const void* vt = nullptr;  // [B, Hkv, D, Tp]  (forward)
  void* o = nullptr;         // [B, T, Hq*D] row stride ldo
  float* lse = nullptr;      // [B, Hq, T]
  const int32_t* kv_start = nullptr;  // [B] first valid key (left padding) or null
  const int32_t* kv_len = nullptr;    // [B] one-past-last valid key or null
  int B = 0, T = 0, Tp = 0, Hq = 0, Hkv = 0, D = 0;
  int ldq = 0, ldk = 0, ldv = 0, ldo = 0;
  int causal = 0;
  int block = 0;  // >0: block-causal "latency" mask, block size in positions
  int window = 0; // >0 (causal): sliding window - a query sees keys in (q - window, q] only (Gemma-3's local layers)
  int q_begin = 0;  // forward only: query rows below this are not needed (chunked prefill over a cached prefix); blocks wholly below it exit
  float scale = 1.0f;
};
int attention_fwd(hipStream_t st, int dtype, const AttnDesc& d);
struct AttnBwdDesc {
  AttnDesc f;
  const void* dout = nullptr;  // [B, T, Hq*D] row stride ldo
  const void* qt = nullptr;    // [B, Hq, D, Tp]
  const void* kt = nullptr;    // [B, Hkv, D, Tp]
  const void* dot = nullptr;   // [B, Hq, D, Tp]
  float* delta = nullptr;      // [B, Hq, T] scratch
  float* dkv_part = nullptr;   // optional scratch (sized for [2, B, T, Hq, D] f32; the bf16 kernels store bf16): per-query-head dK/dV (GQA)
  void* dq = nullptr; void* dk = nullptr; void* dv = nullptr;  // same layouts/strides as q,k,v
  int lddq = 0, lddk = 0, lddv = 0;
  // bf16 kernels only (attention_bwd_fuses_rope): [T_table, D/2, 2] f32 cos / sin of the rotary embedding that produced q and k -
  // dq and dk are then written RoPE-INVERTED (gradients of the projections' outputs), saving the separate inverse-RoPE pass
  const float* rope_cos_sin = nullptr;
  // > 0 (a multiple of 16; attention_bwd_takes_d_first): dout / dq / dk / dv are ROW-COMPACTED - they hold the rows of positions
  // >= d_first only, sequence b at row b * (T - d_first); gradients of positions below it are neither read nor written
  int d_first = 0;
};
bool attention_bwd_is_fused(int dtype, const AttnDesc& f);
bool attention_bwd_takes_d_first(int dtype, const AttnDesc& f);
inline bool attention_bwd_fuses_rope(int dtype) { return dtype == DT_BF16; }
int attention_bwd(hipStream_t st, int dtype, const AttnBwdDesc& d);
// true: the attention kernels of this dtype read vt / qt / kt / dot (callers run heads_transpose first); false (bf16 with tuning
// option 12, the default): they read the natural q / k / v / dout through the transposing LDS read and ignore those pointers
bool attention_needs_transposed_copies(int dtype);
bool attention_tr_reads(int dtype);
int lds_tr_probe(hipStream_t st, const int32_t* addr /*[64] byte offsets*/, int32_t* out /*[256]*/);
// [B, T, H, D] (row stride ld) -> [B, H, D, Tp], zero padded in T
int heads_transpose(hipStream_t st, int dtype, const void* in, void* out, int B, int T, int Tp, int H,
                    int D, int ld);

// ---- loss.hip ----
// Shifted causal-LM cross entropy over bf16/f32 logits [rows=B*T, V] (ld = ldl).
// labels [B, T] int64 (ignore_index -100).  Writes loss (f32 scalar, mean over valid shifted tokens),
// n_valid, and (if dlogits) d loss / d logits in the logits dtype, IN PLACE allowed.
// scratch: 2 + B*T floats (scratch[0] = n_valid, scratch[1] = summed loss, then per-row losses).
int ce_loss_fwd_bwd(hipStream_t st, int dtype, const void* logits, const int64_t* labels, float* loss,
                    float* scratch, void* dlogits, int B, int T, int V, int ldl, float grad_scale,
                    const int32_t* sup = nullptr);
// rows[0..count) = positions (b*T + t) whose NEXT token carries a scorable label, in order; rest -1; rows[B*T] = count
int sup_rows(hipStream_t st, const int64_t* labels, int32_t* rows, int B, int T, int V);
// dst[c, :] = src[rows[c], :] for c < rows[n]; / dst[rows[c], :] = src[c, :] (dst pre-zeroed by the caller)
int gather_rows(hipStream_t st, int dtype, const void* src, const int32_t* rows, long long n, void* dst, int D);
int scatter_rows(hipStream_t st, int dtype, const void* src, const int32_t* rows, long long n, void* dst, int D,
                 int first = 0);   // compact rows [first, count): src row c - first -> dst row rows[c]
// dst[rows[c], :] = round(sum_z partial[z][c][:]) for c < min(count, cap); partial: f32 [nsplit][cap][D], fixed order
int splitk_reduce_scatter(hipStream_t st, int dtype, const float* partial, int nsplit, int cap, const int32_t* rows,
                          long long n, void* dst, int D);

// KL distillation: loss = sum_r sum_slot w[slot][r] * KL(softmax(teacher[pair_row[slot][r]]/tau) || softmax(student[r]/tau)),
// and (if dlogits, IN PLACE allowed) d loss / d student logits * grad_scale.  scratch: rows floats.
int kl_loss_fwd_bwd(hipStream_t st, int dtype, const void* student, const void* teacher, const int32_t* pair_row,
                    const float* pair_w, float* loss, float* scratch, void* dlogits, long long rows, int V, int ld_s, int ld_t,
                    float temperature, float grad_scale);

// ---- optim.hip ----
int grad_sq_norm(hipStream_t st, const float* g, long long n, float* partial /*>=1024 floats*/, float* out_sumsq);
int adamw_clip_step(hipStream_t st, int state_dtype, void* param, float* master, const float* grad,
                    void* m, void* v, long long n, const float* sumsq, float max_norm, float lr,
                    float beta1, float beta2, float eps, float wd, int step);

// ---- logmel.hip ----
int logmel(hipStream_t st, const float* pcm, const float* window, const float* tw_cos, const float* tw_sin,
           const float* mel_fb, float* out, float* scratch, int B, int L, int n_mels, int F_stride);

}  // namespace uvx
