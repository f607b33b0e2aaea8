// Shifted causal-LM cross entropy, forward + gradient in one pass over the logits.
//
// Reference: [3P] transformers ForCausalLMLoss / fixed_cross_entropy reached from
// ultravox_model.py:328-334 with labels; UltravoxModel.accepts_loss_kwargs = False (:50-53) so
// num_items_in_batch is None and the reduction is a MEAN over this rank's non-ignored shifted tokens:
//   logits.float(); labels padded with one -100 on the right and shifted left by one;
//   F.cross_entropy(ignore_index=-100, reduction="mean").
// Row r = (b, t) is scored against labels[b, t+1]; the last position of every sequence is ignored.
// dlogits = (softmax - onehot) * grad_scale / n_valid, written in the logits dtype (autograd casts the
// f32 gradient back through .float()); it may alias the logits buffer.
#include "common.h"
#include "kernels.h"

namespace {

// Same validity predicate as ce_rows_k / sup_rows_k: a label outside [0, V) other than -100 (torch raises on it) is
// treated as ignored by EVERY path - it must not deflate the mean of one path and not of the other.
__global__ void ce_count_k(const int64_t* __restrict__ labels, float* __restrict__ scratch, int B, int T, int V) {
  __shared__ float red[16];
  float c = 0.f;
  for (long long i = threadIdx.x; i < (long long)B * T; i += blockDim.x) {
    const int t = (int)(i % T);
    if (t + 1 < T) {
      const int64_t tgt = labels[i + 1];
      if (tgt != -100 && tgt >= 0 && tgt < V) c += 1.f;
    }
  }
  c = block_sum(c, red);
  if (threadIdx.x == 0) scratch[0] = c;
}

// Supervised rows, in order: r = (b, t) with t + 1 < T and a scorable label at (b, t + 1).  rows[0 .. count) are their
// indices, rows[count .. B*T) = -1; count goes to rows[B*T].  One block, ordered chunked scan (B*T is a few thousand).
__global__ void sup_rows_k(const int64_t* __restrict__ labels, int32_t* __restrict__ rows, int B, int Tlen, int V) {
  __shared__ int wsum[16];
  __shared__ int base;
  const long long n = (long long)B * Tlen;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  for (long long c0 = 0; c0 < n; c0 += blockDim.x) {
    const long long r = c0 + threadIdx.x;
    bool ok = false;
    if (r < n) {
      const int t = (int)(r % Tlen);
      if (t + 1 < Tlen) {
        const int64_t tgt = labels[r + 1];
        ok = tgt != -100 && tgt >= 0 && tgt < V;
      }
    }
    const unsigned long long bal = __ballot(ok);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[w] = __popcll(bal);
    __syncthreads();
    int off = base;
    for (int i = 0; i < w; ++i) off += wsum[i];
    if (ok) rows[off + before] = (int32_t)r;
    __syncthreads();
    if (threadIdx.x == 0) {
      int tot = 0;
      for (int i = 0; i < (int)(blockDim.x >> 6); ++i) tot += wsum[i];
      base += tot;
    }
    __syncthreads();
  }
  const int cnt = base;
  for (long long r = cnt + threadIdx.x; r < n; r += blockDim.x) rows[r] = -1;
  if (threadIdx.x == 0) rows[n] = cnt;
}

// rows == nullptr: logits row = sequence position (b, t).  rows != nullptr: logits are COMPACT, logits row c belongs to
// position rows[c] (c < count = rows[n_rows]); scratch[0] (n_valid) is then the count.
template <typename T>
__global__ void ce_rows_k(const T* __restrict__ logits, const int64_t* __restrict__ labels, float* __restrict__ scratch,
                          T* __restrict__ dlogits, int Tlen, int V, long long ldl, float grad_scale,
                          const int32_t* __restrict__ rows, long long n_rows) {
  __shared__ float red[16];
  const long long lrow = blockIdx.x;
  float* row_loss = scratch + 2;
  long long row = lrow;
  if (rows) {
    if (lrow >= rows[n_rows]) {                 // beyond the compact range: nothing was computed for this slot
      if (threadIdx.x == 0) row_loss[lrow] = 0.f;
      return;
    }
    row = rows[lrow];
  }
  const int t = (int)(row % Tlen);
  const int64_t tgt = (t + 1 < Tlen) ? labels[row + 1] : -100;
  const T* lr = logits + lrow * ldl;
  T* dr = dlogits ? dlogits + lrow * ldl : nullptr;
  if (tgt == -100 || tgt < 0 || tgt >= V) {
    if (threadIdx.x == 0) row_loss[lrow] = 0.f;
    if (dr) {
      float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int c = threadIdx.x * 8; c < V; c += blockDim.x * 8) st8<T>(dr + c, z);
    }
    return;
  }
  // online max / sum-exp, per thread then block
  float m = -__builtin_huge_valf(), s = 0.f;
  for (int c = threadIdx.x * 8; c < V; c += blockDim.x * 8) {
    float v[8];
    ld8<T>(lr + c, v);
    float mx = v[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) mx = fmaxf(mx, v[i]);
    const float mn = fmaxf(m, mx);
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) a += expf(v[i] - mn);
    s = s * expf(m - mn) + a;
    m = mn;
  }
  const float M = block_max(m, red);
  const float S = block_sum(s * expf(m - M), red);
  const float lse = M + logf(S);
  const float n_valid = scratch[0];
  if (threadIdx.x == 0) row_loss[lrow] = lse - ldf<T>(lr + tgt);
  if (dr) {
    const float g = grad_scale / n_valid;
    for (int c = threadIdx.x * 8; c < V; c += blockDim.x * 8) {
      float v[8];
      ld8<T>(lr + c, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float pr = expf(v[i] - lse);
        v[i] = (pr - ((int64_t)(c + i) == tgt ? 1.f : 0.f)) * g;
      }
      st8<T>(dr + c, v);
    }
  }
}

__global__ void ce_final_k(float* __restrict__ scratch, float* __restrict__ loss, long long rows) {
  __shared__ float red[16];
  float s = 0.f;
  for (long long i = threadIdx.x; i < rows; i += blockDim.x) s += scratch[2 + i];  // fixed order: deterministic
  s = block_sum(s, red);
  if (threadIdx.x == 0) {
    const float n = scratch[0];
    if (loss) loss[0] = n > 0.f ? s / n : __builtin_nanf("");  // torch: mean over zero elements = nan
    scratch[1] = s;
  }
}


// ---- KL-distillation loss (ultravox_model.py:157-256) ----
// Student row r is paired with up to two teacher rows: slot 0 = its partner among the prediction positions
// (F.kl_div over logits[pred_mask] vs alt logits[alt_pred_mask], "batchmean" -> weight 1 / n_pred), slot 1 = its
// partner among the end-of-turn positions (weight eot_loss_weight / n_eot).  Per pair
//   KL = sum_v softmax(t/tau)_v * (log_softmax(t/tau)_v - log_softmax(s/tau)_v)
//   d KL / d s_v = (softmax(s/tau)_v - softmax(t/tau)_v) / tau
// Everything is evaluated in f32 from the stored logits; rows without a partner get a zero gradient.
struct OnlineLse {
  float m = -__builtin_huge_valf(), s = 0.f;
  __device__ __forceinline__ void add8(const float* v) {
    float mx = v[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) mx = fmaxf(mx, v[i]);
    const float mn = fmaxf(m, mx);
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) a += expf(v[i] - mn);
    s = s * expf(m - mn) + a;
    m = mn;
  }
  __device__ __forceinline__ float finish(float* red) {
    const float M = block_max(m, red);
    const float S = block_sum(s * expf(m - M), red);
    return M + logf(S);
  }
};

template <typename T>
__global__ void kl_rows_k(const T* __restrict__ student, const T* __restrict__ teacher, const int32_t* __restrict__ pair_row,
                          const float* __restrict__ pair_w, float* __restrict__ row_loss, T* __restrict__ dlogits,
                          long long rows, int V, long long ld_s, long long ld_t, float inv_tau, float grad_scale) {
  __shared__ float red[16];
  const long long row = blockIdx.x;
  const int32_t t0 = pair_row[row], t1 = pair_row[rows + row];
  const float w0 = t0 >= 0 ? pair_w[row] : 0.f, w1 = t1 >= 0 ? pair_w[rows + row] : 0.f;
  const T* sr = student + row * ld_s;
  T* dr = dlogits ? dlogits + row * ld_s : nullptr;
  if (t0 < 0 && t1 < 0) {
    if (threadIdx.x == 0) row_loss[row] = 0.f;
    if (dr) {
      float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int c = threadIdx.x * 8; c < V; c += blockDim.x * 8) st8<T>(dr + c, z);
    }
    return;
  }
  const T* tr0 = teacher + (long long)(t0 >= 0 ? t0 : t1) * ld_t;
  const T* tr1 = teacher + (long long)(t1 >= 0 ? t1 : t0) * ld_t;
  const bool two = t0 >= 0 && t1 >= 0 && t0 != t1;   // the usual case pairs both slots with the same teacher row
  OnlineLse ls, l0, l1;
  for (int c = threadIdx.x * 8; c < V; c += blockDim.x * 8) {
    float v[8];
    ld8<T>(sr + c, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= inv_tau;
    ls.add8(v);
    ld8<T>(tr0 + c, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= inv_tau;
    l0.add8(v);
    if (two) {
      ld8<T>(tr1 + c, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] *= inv_tau;
      l1.add8(v);
    }
  }
  const float lse_s = ls.finish(red), lse_0 = l0.finish(red), lse_1 = two ? l1.finish(red) : lse_0;
  // slot weights: if both slots name the same teacher row they simply add up
  const float wa = two ? (t0 >= 0 ? w0 : 0.f) : (w0 + w1), wb = two ? w1 : 0.f;
  float kl_a = 0.f, kl_b = 0.f;
  const float g = grad_scale * inv_tau;
  for (int c = threadIdx.x * 8; c < V; c += blockDim.x * 8) {
    float a[8], b[8], o[8];
    ld8<T>(sr + c, a);
    ld8<T>(tr0 + c, b);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float la = a[i] * inv_tau - lse_s, lb = b[i] * inv_tau - lse_0;
      const float pt = expf(lb);
      kl_a += pt * (lb - la);
      a[i] = la;
      o[i] = wa * (expf(la) - pt);
    }
    if (two) {
      ld8<T>(tr1 + c, b);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float lb = b[i] * inv_tau - lse_1;
        const float pt = expf(lb);
        kl_b += pt * (lb - a[i]);
        o[i] += wb * (expf(a[i]) - pt);
      }
    }
    if (dr) {
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] *= g;
      st8<T>(dr + c, o);
    }
  }
  const float tot = block_sum(wa * kl_a + wb * kl_b, red);
  if (threadIdx.x == 0) row_loss[row] = tot;
}

__global__ void kl_final_k(const float* __restrict__ row_loss, float* __restrict__ loss, long long rows) {
  __shared__ float red[16];
  float s = 0.f;
  for (long long i = threadIdx.x; i < rows; i += blockDim.x) s += row_loss[i];  // fixed order: deterministic
  s = block_sum(s, red);
  if (threadIdx.x == 0) loss[0] = s;
}

__global__ void ce_count_from_rows_k(const int32_t* __restrict__ rows, float* __restrict__ scratch, long long n_rows) {
  if (threadIdx.x == 0 && blockIdx.x == 0) scratch[0] = (float)rows[n_rows];
}

}  // namespace

namespace uvx {

int sup_rows(hipStream_t st, const int64_t* labels, int32_t* rows, int B, int T, int V) {
  hipLaunchKernelGGL(sup_rows_k, dim3(1), dim3(1024), 0, st, labels, rows, B, T, V);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

// scratch: 2 + B*T floats.  scratch[0] = n_valid, scratch[1] = summed loss, scratch[2..] = per-row loss.
// sup != nullptr: `logits` / `dlogits` hold only the supervised rows, compacted in the order of sup_rows().
int ce_loss_fwd_bwd(hipStream_t st, int dtype, const void* logits, const int64_t* labels, float* loss, float* scratch,
                    void* dlogits, int B, int T, int V, int ldl, float grad_scale, const int32_t* sup) {
  UVX_CHECK(V % 8 == 0 && ldl % 8 == 0, UVX_ERR_SHAPE, "ce_loss: V=%d / ld=%d must be multiples of 8", V, ldl);
  const long long rows = (long long)B * T;
  UVX_CHECK(rows > 0, UVX_ERR_SHAPE, "ce_loss: empty batch");
  if (sup) hipLaunchKernelGGL(ce_count_from_rows_k, dim3(1), dim3(64), 0, st, sup, scratch, rows);
  else hipLaunchKernelGGL(ce_count_k, dim3(1), dim3(1024), 0, st, labels, scratch, B, T, V);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(ce_rows_k<bf16_t>, dim3(rows), dim3(256), 0, st, (const bf16_t*)logits, labels, scratch, (bf16_t*)dlogits, T, V, (long long)ldl, grad_scale, sup, rows);
  else
    hipLaunchKernelGGL(ce_rows_k<float>, dim3(rows), dim3(256), 0, st, (const float*)logits, labels, scratch, (float*)dlogits, T, V, (long long)ldl, grad_scale, sup, rows);
  hipLaunchKernelGGL(ce_final_k, dim3(1), dim3(1024), 0, st, scratch, loss, rows);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

// scratch: rows floats (per-row weighted KL).  pair_row: int32 [2][rows] teacher row or -1; pair_w: f32 [2][rows].
int kl_loss_fwd_bwd(hipStream_t st, int dtype, const void* student, const void* teacher, const int32_t* pair_row,
                    const float* pair_w, float* loss, float* scratch, void* dlogits, long long rows, int V, int ld_s, int ld_t,
                    float temperature, float grad_scale) {
  UVX_CHECK(V % 8 == 0 && ld_s % 8 == 0 && ld_t % 8 == 0, UVX_ERR_SHAPE, "kl_loss: V=%d / ld=%d,%d must be multiples of 8", V, ld_s, ld_t);
  UVX_CHECK(rows > 0, UVX_ERR_SHAPE, "kl_loss: empty batch");
  UVX_CHECK(temperature > 0.f, UVX_ERR_INVALID, "kl_loss: temperature must be positive");
  const float it = 1.0f / temperature;
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(kl_rows_k<bf16_t>, dim3(rows), dim3(256), 0, st, (const bf16_t*)student, (const bf16_t*)teacher, pair_row, pair_w, scratch, (bf16_t*)dlogits, rows, V, (long long)ld_s, (long long)ld_t, it, grad_scale);
  else
    hipLaunchKernelGGL(kl_rows_k<float>, dim3(rows), dim3(256), 0, st, (const float*)student, (const float*)teacher, pair_row, pair_w, scratch, (float*)dlogits, rows, V, (long long)ld_s, (long long)ld_t, it, grad_scale);
  if (loss) hipLaunchKernelGGL(kl_final_k, dim3(1), dim3(1024), 0, st, scratch, loss, rows);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

}  // namespace uvx
