// C-ABI: error reporting + single-op entry points (used by unit tests and by host code that wants
// one kernel at a time).  The coarse-grained model entry points live in model.hip.
#include <stdarg.h>
#include <string.h>
#include "common.h"
#include "kernels.h"
#include "../../include/uvx.h"
#ifdef UVX_PROBES
#include "../../include/uvx_probes.h"
#endif

static thread_local char g_err[512] = "";

// (internal: hidden visibility keeps it out of the dynamic symbol table - the exported set is exactly what include/uvx.h declares)
extern "C" __attribute__((visibility("hidden"))) void uvx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* uvx_last_error(void) { return g_err; }
extern "C" int32_t uvx_abi_version(void) { return UVX_ABI_VERSION; }
namespace uvx { extern int g_attn_qt; extern void* g_attn_tl; }
extern "C" int32_t uvx_attention_force_qt(int32_t qt) { uvx::g_attn_qt = qt; return UVX_OK; }
#ifdef UVX_PROBES
extern "C" int32_t uvx_probe_attn_timeline(void* stamps) {
  uvx::g_attn_tl = stamps;
  return UVX_OK;
}
extern "C" int32_t uvx_gemm_streamk_timeouts(void) { return uvx::gemm_streamk_timeouts(); }
#endif
extern "C" int32_t uvx_gemm_force_variant(int32_t v) {
  if (v <= -2) { uvx::g_gemm_variant = -1; uvx::g_gemm_split = 0; }  // -2: automatic variant, tail split off (A/B probes)
  else { uvx::g_gemm_variant = v; uvx::g_gemm_split = 1; }
  return UVX_OK;
}

extern "C" int32_t uvx_set_option(int32_t key, int32_t value) {
  UVX_CHECK(key > 0 && key < 32, UVX_ERR_INVALID, "uvx_set_option: unknown key %d", key);
  uvx::g_options[key] = value;
  return UVX_OK;
}

extern "C" int32_t uvx_get_option(int32_t key) { return key > 0 && key < 32 ? uvx::g_options[key] : -1; }

extern "C" int32_t uvx_probe_lds_tr(void* stream, const int32_t* addr, int32_t* out) {
  UVX_CHECK(addr && out, UVX_ERR_INVALID, "uvx_probe_lds_tr: null argument");
  return uvx::lds_tr_probe((hipStream_t)stream, addr, out);
}

extern "C" int32_t uvx_gemm_pick_variant(int32_t M, int32_t N, int32_t K, int32_t batch) { return uvx::gemm_pick_variant(M, N, K, batch); }

extern "C" int32_t uvx_gemm_override_variant(int32_t M, int32_t N, int32_t K, int32_t variant) {
  if (variant < 0) { uvx::g_gemm_ovr_n = 0; return UVX_OK; }
  UVX_CHECK(uvx::g_gemm_ovr_n < 32, UVX_ERR_INVALID, "gemm override table full");
  int* o = uvx::g_gemm_ovr[uvx::g_gemm_ovr_n++];
  o[0] = M; o[1] = N; o[2] = K; o[3] = variant;
  return UVX_OK;
}

extern "C" int32_t uvx_gemm(void* stream, int32_t dtype, const uvx_gemm_desc_t* g) {
  UVX_CHECK(g != nullptr, UVX_ERR_INVALID, "uvx_gemm: null descriptor");
  uvx::GemmDesc d;
  d.A = g->A; d.B = g->B; d.C = g->C; d.bias = g->bias; d.residual = g->residual;
  d.M = g->M; d.N = g->N; d.K = g->K;
  d.lda = g->lda; d.ldb = g->ldb; d.ldc = g->ldc; d.ldr = g->ldr;
  d.res_mod = g->res_mod; d.batch = g->batch;
  d.sA = g->stride_a; d.sB = g->stride_b; d.sC = g->stride_c; d.sR = g->stride_r;
  d.act = g->act; d.out_f32 = g->out_f32; d.accumulate = g->accumulate; d.alpha = g->alpha;
  d.C2 = g->C2; d.ldc2 = g->ldc2; d.swiglu = g->epilogue; d.b_kn = g->b_kn;
  UVX_CHECK(g->epilogue == 0 || dtype == uvx::DT_BF16, UVX_ERR_UNSUPPORTED, "uvx_gemm: fused SwiGLU epilogues are bf16 only");
  return uvx::gemm((hipStream_t)stream, dtype, d);
}

// uvx_gemm with split-K scratch (the prefill's few-hundred-row GEMMs; gemm.hip "Split-K"): same result contract as uvx_gemm
extern "C" size_t uvx_gemm_splitk_ws_bytes(int32_t M, int32_t N) { return M > 0 && N > 0 ? uvx::gemm_splitk_ws_bytes(M, N) : 0; }
extern "C" int32_t uvx_gemm_pick_split(int32_t M, int32_t N, int32_t K, size_t ws_bytes, int32_t* variant) {
  int v = 0;
  const int s = uvx::gemm_pick_split(M, N, K, ws_bytes, &v);
  if (variant) *variant = v;
  return s;
}
extern "C" int32_t uvx_gemm_splitk(void* stream, int32_t dtype, const uvx_gemm_desc_t* g, void* workspace, size_t ws_bytes, int32_t force_split) {
  UVX_CHECK(g != nullptr, UVX_ERR_INVALID, "uvx_gemm_splitk: null descriptor");
  UVX_CHECK(force_split >= 0, UVX_ERR_INVALID, "uvx_gemm_splitk: force_split %d (0 = automatic, 1 = never, s = that factor)", force_split);
  uvx::GemmDesc d;
  d.A = g->A; d.B = g->B; d.C = g->C; d.bias = g->bias; d.residual = g->residual;
  d.M = g->M; d.N = g->N; d.K = g->K;
  d.lda = g->lda; d.ldb = g->ldb; d.ldc = g->ldc; d.ldr = g->ldr;
  d.res_mod = g->res_mod; d.batch = g->batch;
  d.sA = g->stride_a; d.sB = g->stride_b; d.sC = g->stride_c; d.sR = g->stride_r;
  d.act = g->act; d.out_f32 = g->out_f32; d.accumulate = g->accumulate; d.alpha = g->alpha;
  d.C2 = g->C2; d.ldc2 = g->ldc2; d.swiglu = g->epilogue;
  d.splitk_ws = workspace; d.splitk_ws_bytes = workspace ? ws_bytes : 0; d.splitk_force = force_split;
  UVX_CHECK(g->epilogue == 0 || dtype == uvx::DT_BF16, UVX_ERR_UNSUPPORTED, "uvx_gemm_splitk: fused SwiGLU epilogues are bf16 only");
  return uvx::gemm((hipStream_t)stream, dtype, d);
}

// C = epilogue(RMSNorm(A; norm_w) . B^T): the decode step's fused norm + weight-streaming GEMV (bf16, at most 2 rows); any other
// problem runs as the two launches it replaces (norm_out: M x K scratch for that case).
extern "C" int32_t uvx_gemm_rmsnorm(void* stream, int32_t dtype, const uvx_gemm_desc_t* g, const void* norm_w, float eps, int32_t flavor,
                                    void* norm_out) {
  UVX_CHECK(g != nullptr && norm_w != nullptr, UVX_ERR_INVALID, "uvx_gemm_rmsnorm: null argument");
  uvx::GemmDesc d;
  d.A = g->A; d.B = g->B; d.C = g->C; d.bias = g->bias; d.residual = g->residual;
  d.M = g->M; d.N = g->N; d.K = g->K;
  d.lda = g->lda; d.ldb = g->ldb; d.ldc = g->ldc; d.ldr = g->ldr;
  d.res_mod = g->res_mod; d.batch = g->batch;
  d.sA = g->stride_a; d.sB = g->stride_b; d.sC = g->stride_c; d.sR = g->stride_r;
  d.act = g->act; d.out_f32 = g->out_f32; d.accumulate = g->accumulate; d.alpha = g->alpha;
  d.C2 = g->C2; d.ldc2 = g->ldc2; d.swiglu = g->epilogue;
  UVX_CHECK(g->epilogue == 0 || dtype == uvx::DT_BF16, UVX_ERR_UNSUPPORTED, "uvx_gemm_rmsnorm: fused SwiGLU epilogues are bf16 only");
  UVX_CHECK(d.lda == d.K, UVX_ERR_SHAPE, "uvx_gemm_rmsnorm: A rows must be contiguous (lda = K)");
  if (dtype == uvx::DT_BF16) {
    const int rc = uvx::gemm_skinny_rmsnorm_bf16((hipStream_t)stream, d, norm_w, eps, flavor);
    if (rc != UVX_ERR_UNSUPPORTED) return rc;
  }
  UVX_CHECK(norm_out != nullptr, UVX_ERR_INVALID, "uvx_gemm_rmsnorm: this problem runs as two launches and needs norm_out [M, K]");
  const int rc_norm = uvx::rmsnorm_fwd((hipStream_t)stream, dtype, d.A, norm_w, norm_out, nullptr, d.M, d.K, eps, flavor);
  if (rc_norm != UVX_OK) return rc_norm;
  d.A = norm_out;
  return uvx::gemm((hipStream_t)stream, dtype, d);
}

// ---- thin single-op wrappers ----
extern "C" int32_t uvx_layernorm(void* stream, int32_t dtype, const void* x, const void* w, const void* b, void* y,
                                 int32_t rows, int32_t cols, float eps) {
  return uvx::layernorm_fwd((hipStream_t)stream, dtype, x, w, b, y, rows, cols, eps);
}
extern "C" int32_t uvx_rmsnorm(void* stream, int32_t dtype, const void* x, const void* w, void* y, int32_t rows,
                               int32_t cols, float eps) {
  return uvx::rmsnorm_fwd((hipStream_t)stream, dtype, x, w, y, nullptr, rows, cols, eps);
}
extern "C" int32_t uvx_rmsnorm_bwd(void* stream, int32_t dtype, const void* dy, const void* x, const void* w,
                                   const void* dx_add, void* dx, float* dw, int32_t rows, int32_t cols, float eps) {
  return uvx::rmsnorm_bwd((hipStream_t)stream, dtype, dy, x, w, dx_add, dx, dw, rows, cols, eps);
}
extern "C" int32_t uvx_swiglu(void* stream, int32_t dtype, const void* in, void* out, int32_t rows, int32_t half,
                              int32_t gate_first) {
  return uvx::swiglu_fwd((hipStream_t)stream, dtype, in, out, rows, half, gate_first);
}
extern "C" int32_t uvx_swiglu_bwd(void* stream, int32_t dtype, const void* dout, const void* in, void* din, int32_t rows,
                                  int32_t half, int32_t gate_first) {
  return uvx::swiglu_bwd((hipStream_t)stream, dtype, dout, in, din, rows, half, gate_first);
}
extern "C" int32_t uvx_rope(void* stream, int32_t dtype, void* x, const float* cos_sin, int32_t rows, int32_t T,
                            int32_t n_heads, int32_t head_dim, int32_t ld, int32_t inverse) {
  return uvx::rope_inplace((hipStream_t)stream, dtype, x, cos_sin, nullptr, rows, T, n_heads, head_dim, ld, inverse);
}
extern "C" int32_t uvx_qk_norm_rope(void* stream, int32_t dtype, void* qkv, const void* wq, const void* wk, void* raw,
                                    const float* cos_sin, int32_t rows, int32_t T, int32_t Hq, int32_t Hkv, int32_t head_dim,
                                    int32_t ld, float eps, int32_t flavor) {
  return uvx::qk_norm_rope((hipStream_t)stream, dtype, qkv, wq, wk, raw, cos_sin, nullptr, rows, T, Hq, Hkv, head_dim, ld, eps, flavor);
}
extern "C" int32_t uvx_qk_norm_bwd(void* stream, int32_t dtype, void* d_qkv, const void* raw, const void* wq, const void* wk,
                                   int32_t rows, int32_t Hq, int32_t Hkv, int32_t head_dim, int32_t ld, float eps, int32_t flavor) {
  return uvx::qk_norm_bwd((hipStream_t)stream, dtype, d_qkv, raw, wq, wk, rows, Hq, Hkv, head_dim, ld, eps, flavor);
}
extern "C" int32_t uvx_ce_loss(void* stream, int32_t dtype, const void* logits, const int64_t* labels, float* loss,
                               void* dlogits, int32_t B, int32_t T, int32_t V, int32_t ld, float grad_scale,
                               float* scratch) {
  return uvx::ce_loss_fwd_bwd((hipStream_t)stream, dtype, logits, labels, loss, scratch, dlogits, B, T, V, ld, grad_scale);
}
extern "C" int32_t uvx_layernorm_bwd(void* stream, int32_t dtype, const void* dy, const void* x, const void* w, const void* dx_add,
                                     void* dx, int32_t rows, int32_t cols, float eps) {
  return uvx::layernorm_bwd((hipStream_t)stream, dtype, dy, x, w, dx_add, dx, rows, cols, eps);
}
extern "C" int32_t uvx_gelu(void* stream, int32_t dtype, const void* pre, void* out, int64_t n) {
  return uvx::gelu_fwd((hipStream_t)stream, dtype, pre, out, n);
}
extern "C" int32_t uvx_gelu_bwd(void* stream, int32_t dtype, const void* dout, const void* pre, void* din, int64_t n) {
  return uvx::gelu_bwd((hipStream_t)stream, dtype, dout, pre, din, n);
}
extern "C" int32_t uvx_kl_loss(void* stream, int32_t dtype, const void* student_logits, const void* teacher_logits,
                               const int32_t* pair_row, const float* pair_w, float* loss, void* dlogits, int64_t rows,
                               int32_t V, int32_t ld_student, int32_t ld_teacher, float temperature, float grad_scale,
                               float* scratch) {
  UVX_CHECK(student_logits && teacher_logits && pair_row && pair_w && scratch, UVX_ERR_INVALID, "kl_loss: null argument");
  return uvx::kl_loss_fwd_bwd((hipStream_t)stream, dtype, student_logits, teacher_logits, pair_row, pair_w, loss, scratch,
                              dlogits, rows, V, ld_student, ld_teacher, temperature, grad_scale);
}

// Attention with its transposed operand copies carved from the caller's workspace.
namespace {
struct AttnWs { void *vt, *qt, *kt, *dot; float* delta; float* part; int Tp; size_t bytes; };
AttnWs attn_carve(char* base, int dtype, const uvx_attn_desc_t& d, int backward) {
  AttnWs w = {};
  const size_t es = dtype == uvx::DT_BF16 ? 2 : 4;
  w.Tp = (d.T + 63) / 64 * 64;
  size_t off = 0;
  auto take = [&](size_t b) { size_t a = (off + 255) & ~(size_t)255; off = a + b; return base ? (void*)(base + a) : nullptr; };
  w.vt = take((size_t)d.B * d.Hkv * d.D * w.Tp * es);
  if (backward) {
    w.qt = take((size_t)d.B * d.Hq * d.D * w.Tp * es);
    w.kt = take((size_t)d.B * d.Hkv * d.D * w.Tp * es);
    w.dot = take((size_t)d.B * d.Hq * d.D * w.Tp * es);
    w.delta = (float*)take(sizeof(float) * (size_t)d.B * d.Hq * d.T);
    w.part = (float*)take(sizeof(float) * 2 * (size_t)d.B * d.T * d.Hq * d.D);
  }
  w.bytes = off + 256;
  return w;
}
uvx::AttnDesc to_desc(const uvx_attn_desc_t& d, const AttnWs& w) {
  uvx::AttnDesc a;
  a.q = d.q; a.k = d.k; a.v = d.v; a.vt = w.vt; a.o = d.o; a.lse = d.lse; a.kv_start = d.kv_start; a.kv_len = d.kv_len;
  a.B = d.B; a.T = d.T; a.Tp = w.Tp; a.Hq = d.Hq; a.Hkv = d.Hkv; a.D = d.D;
  a.ldq = d.ldq; a.ldk = d.ldk; a.ldv = d.ldv; a.ldo = d.ldo; a.causal = d.causal; a.block = d.block; a.window = d.window; a.scale = d.scale;
  return a;
}
}  // namespace

extern "C" size_t uvx_attention_ws_bytes(int32_t dtype, const uvx_attn_desc_t* d, int32_t backward) {
  return d ? attn_carve(nullptr, dtype, *d, backward).bytes : 0;
}
extern "C" int32_t uvx_attention_fwd(void* stream, int32_t dtype, const uvx_attn_desc_t* d, void* workspace,
                                     size_t ws_bytes) {
  UVX_CHECK(d && workspace, UVX_ERR_INVALID, "attention_fwd: null argument");
  AttnWs w = attn_carve((char*)workspace, dtype, *d, 0);
  UVX_CHECK(w.bytes <= ws_bytes + 256, UVX_ERR_WORKSPACE, "attention_fwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  if (uvx::attention_needs_transposed_copies(dtype)) {
    int rc = uvx::heads_transpose(st, dtype, d->v, w.vt, d->B, d->T, w.Tp, d->Hkv, d->D, d->ldv);
    if (rc) return rc;
  }
  return uvx::attention_fwd(st, dtype, to_desc(*d, w));
}
extern "C" int32_t uvx_attention_bwd(void* stream, int32_t dtype, const uvx_attn_desc_t* d, void* workspace,
                                     size_t ws_bytes) {
  UVX_CHECK(d && workspace && d->lse && d->dout, UVX_ERR_INVALID, "attention_bwd: null argument");
  AttnWs w = attn_carve((char*)workspace, dtype, *d, 1);
  UVX_CHECK(w.bytes <= ws_bytes + 256, UVX_ERR_WORKSPACE, "attention_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (uvx::attention_needs_transposed_copies(dtype)) {
    if ((rc = uvx::heads_transpose(st, dtype, d->q, w.qt, d->B, d->T, w.Tp, d->Hq, d->D, d->ldq))) return rc;
    if ((rc = uvx::heads_transpose(st, dtype, d->k, w.kt, d->B, d->T, w.Tp, d->Hkv, d->D, d->ldk))) return rc;
    if ((rc = uvx::heads_transpose(st, dtype, d->dout, w.dot, d->B, d->T, w.Tp, d->Hq, d->D, d->ldo))) return rc;
  }
  uvx::AttnBwdDesc b;
  b.f = to_desc(*d, w);
  b.dout = d->dout; b.qt = w.qt; b.kt = w.kt; b.dot = w.dot; b.delta = w.delta; b.dkv_part = w.part;
  b.dq = d->dq; b.dk = d->dk; b.dv = d->dv; b.lddq = d->lddq; b.lddk = d->lddk; b.lddv = d->lddv;
  return uvx::attention_bwd(st, dtype, b);
}
