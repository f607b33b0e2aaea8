// f32 "parity mode" attention (forward + backward): the same masks / GQA / log-sum-exp contract as
// attention.hip with every tensor f32 and exact expf.  One wave per (batch, head, row); deterministic (no
// atomics).  O(T^2 D) scalar-FMA work — meant for the parity configurations, not for throughput.
#include "common.h"
#include "kernels.h"

namespace {

struct AArgs {
  const float *q, *k, *v; float* o; float* lse;
  const int32_t *kv_start, *kv_len;
  int B, T, Hq, Hkv, D, ldq, ldk, ldv, ldo, causal, block, window;
  float scale;
  const float* dout; const float* delta_in; float* delta; float *dq, *dk, *dv; int lddq, lddk, lddv;
};

__device__ __forceinline__ bool ok_key(int key, int q, int lo, int hi, int causal, int block, int window = 0) {
  bool ok = key >= lo && key < hi;
  if (causal) ok = ok && key <= q;
  if (window > 0) ok = ok && key > q - window;      // sliding window (Gemma-3's local layers)
  if (block > 0) ok = ok && (key / block) <= (q / block);
  return ok;
}

// dot of two D-vectors split over the wave (D = 64: one element per lane; D = 128: two)
template <int D>
__device__ __forceinline__ float wdot(const float* a, const float* b, int lane) {
  float s = 0.f;
#pragma unroll
  for (int c = lane; c < D; c += 64) s += a[c] * b[c];
  return wave_sum(s);
}

template <int D>
__global__ void attn_fwd_f32_k(AArgs p) {
  const long long wid = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (wid >= (long long)p.B * p.Hq * p.T) return;
  const int q = (int)(wid % p.T), h = (int)((wid / p.T) % p.Hq), b = (int)(wid / ((long long)p.T * p.Hq));
  const int hk = h / (p.Hq / p.Hkv);
  const int lo = p.kv_start ? p.kv_start[b] : 0, hi = p.kv_len ? min(p.kv_len[b], p.T) : p.T;
  const float* qr = p.q + ((long long)b * p.T + q) * p.ldq + h * D;
  constexpr int E = D / 64;
  float qv[E], acc[E];
#pragma unroll
  for (int e = 0; e < E; ++e) { qv[e] = qr[lane + 64 * e]; acc[e] = 0.f; }
  // pass 1: row max;  pass 2: exp-sum and P.V  (two passes keep the arithmetic identical to softmax())
  float mx = -__builtin_huge_valf();
  for (int key = 0; key < p.T; ++key) {
    if (!ok_key(key, q, lo, hi, p.causal, p.block, p.window)) continue;
    const float* kr = p.k + ((long long)b * p.T + key) * p.ldk + hk * D;
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) s += qv[e] * kr[lane + 64 * e];
    s = wave_sum(s) * p.scale;
    mx = fmaxf(mx, s);
  }
  float l = 0.f;
  for (int key = 0; key < p.T; ++key) {
    if (!ok_key(key, q, lo, hi, p.causal, p.block, p.window)) continue;
    const float* kr = p.k + ((long long)b * p.T + key) * p.ldk + hk * D;
    const float* vr = p.v + ((long long)b * p.T + key) * p.ldv + hk * D;
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) s += qv[e] * kr[lane + 64 * e];
    s = wave_sum(s) * p.scale;
    const float pr = expf(s - mx);
    l += pr;
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] += pr * vr[lane + 64 * e];
  }
  float* orow = p.o + ((long long)b * p.T + q) * p.ldo + h * D;
  const float inv = l > 0.f ? 1.f / l : 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e) orow[lane + 64 * e] = acc[e] * inv;
  // same convention as the bf16 kernel: log2-domain log-sum-exp of the SCALED scores
  if (p.lse && lane == 0) p.lse[((long long)b * p.Hq + h) * p.T + q] = l > 0.f ? (mx + logf(l)) * 1.4426950408889634f : __builtin_huge_valf();
}

template <int D>
__global__ void attn_bwd_dq_f32_k(AArgs p) {  // also writes delta[b,h,q]
  const long long wid = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (wid >= (long long)p.B * p.Hq * p.T) return;
  const int q = (int)(wid % p.T), h = (int)((wid / p.T) % p.Hq), b = (int)(wid / ((long long)p.T * p.Hq));
  const int hk = h / (p.Hq / p.Hkv);
  const int lo = p.kv_start ? p.kv_start[b] : 0, hi = p.kv_len ? min(p.kv_len[b], p.T) : p.T;
  const float* qr = p.q + ((long long)b * p.T + q) * p.ldq + h * D;
  const float* dor = p.dout + ((long long)b * p.T + q) * p.ldo + h * D;
  const float* orow = p.o + ((long long)b * p.T + q) * p.ldo + h * D;
  const float lse = p.lse[((long long)b * p.Hq + h) * p.T + q] * 0.6931471805599453f;  // back to natural log
  const float dl = wdot<D>(dor, orow, lane);
  if (lane == 0) p.delta[((long long)b * p.Hq + h) * p.T + q] = dl;
  constexpr int E = D / 64;
  float acc[E];
#pragma unroll
  for (int e = 0; e < E; ++e) acc[e] = 0.f;
  for (int key = 0; key < p.T; ++key) {
    if (!ok_key(key, q, lo, hi, p.causal, p.block, p.window)) continue;
    const float* kr = p.k + ((long long)b * p.T + key) * p.ldk + hk * D;
    const float* vr = p.v + ((long long)b * p.T + key) * p.ldv + hk * D;
    const float s = wdot<D>(qr, kr, lane) * p.scale;
    const float pr = expf(s - lse);
    const float ds = pr * (wdot<D>(dor, vr, lane) - dl) * p.scale;
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] += ds * kr[lane + 64 * e];
  }
  float* dqr = p.dq + ((long long)b * p.T + q) * p.lddq + h * D;
#pragma unroll
  for (int e = 0; e < E; ++e) dqr[lane + 64 * e] = acc[e];
}

template <int D>
__global__ void attn_bwd_dkdv_f32_k(AArgs p) {  // one wave per (b, kv head, key); loops the GQA group and queries
  const long long wid = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (wid >= (long long)p.B * p.Hkv * p.T) return;
  const int key = (int)(wid % p.T), hk = (int)((wid / p.T) % p.Hkv), b = (int)(wid / ((long long)p.T * p.Hkv));
  const int grp = p.Hq / p.Hkv;
  const int lo = p.kv_start ? p.kv_start[b] : 0, hi = p.kv_len ? min(p.kv_len[b], p.T) : p.T;
  const float* kr = p.k + ((long long)b * p.T + key) * p.ldk + hk * D;
  const float* vr = p.v + ((long long)b * p.T + key) * p.ldv + hk * D;
  constexpr int E = D / 64;
  float dk[E], dv[E];
#pragma unroll
  for (int e = 0; e < E; ++e) { dk[e] = 0.f; dv[e] = 0.f; }
  for (int hh = 0; hh < grp; ++hh) {
    const int h = hk * grp + hh;
    for (int q = 0; q < p.T; ++q) {
      if (!ok_key(key, q, lo, hi, p.causal, p.block, p.window)) continue;
      const float* qr = p.q + ((long long)b * p.T + q) * p.ldq + h * D;
      const float* dor = p.dout + ((long long)b * p.T + q) * p.ldo + h * D;
      const float lse = p.lse[((long long)b * p.Hq + h) * p.T + q] * 0.6931471805599453f;
      const float dl = p.delta_in[((long long)b * p.Hq + h) * p.T + q];
      const float pr = expf(wdot<D>(qr, kr, lane) * p.scale - lse);
      const float ds = pr * (wdot<D>(dor, vr, lane) - dl) * p.scale;
#pragma unroll
      for (int e = 0; e < E; ++e) { dv[e] += pr * dor[lane + 64 * e]; dk[e] += ds * qr[lane + 64 * e]; }
    }
  }
  float* dkr = p.dk + ((long long)b * p.T + key) * p.lddk + hk * D;
  float* dvr = p.dv + ((long long)b * p.T + key) * p.lddv + hk * D;
#pragma unroll
  for (int e = 0; e < E; ++e) { dkr[lane + 64 * e] = dk[e]; dvr[lane + 64 * e] = dv[e]; }
}

AArgs mk(const uvx::AttnDesc& d) {
  AArgs a = {};
  a.q = (const float*)d.q; a.k = (const float*)d.k; a.v = (const float*)d.v; a.o = (float*)d.o; a.lse = d.lse;
  a.kv_start = d.kv_start; a.kv_len = d.kv_len;
  a.B = d.B; a.T = d.T; a.Hq = d.Hq; a.Hkv = d.Hkv; a.D = d.D;
  a.ldq = d.ldq; a.ldk = d.ldk; a.ldv = d.ldv; a.ldo = d.ldo; a.causal = d.causal; a.block = d.block; a.window = d.window; a.scale = d.scale;
  return a;
}

}  // namespace

namespace uvx {

int attention_fwd_f32(hipStream_t st, const AttnDesc& d) {
  UVX_CHECK(d.D == 64 || d.D == 128 || d.D == 256, UVX_ERR_UNSUPPORTED, "attention_f32: head_dim %d not supported", d.D);
  UVX_CHECK(d.v != nullptr, UVX_ERR_INVALID, "attention_f32: needs the natural-layout V");
  AArgs a = mk(d);
  const long long n = (long long)d.B * d.Hq * d.T;
  dim3 grid(cdiv(n, 4));
  if (d.D == 64) hipLaunchKernelGGL(attn_fwd_f32_k<64>, grid, dim3(256), 0, st, a);
  else if (d.D == 128) hipLaunchKernelGGL(attn_fwd_f32_k<128>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(attn_fwd_f32_k<256>, grid, dim3(256), 0, st, a);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

int attention_bwd_f32(hipStream_t st, const AttnBwdDesc& bd) {
  const AttnDesc& d = bd.f;
  UVX_CHECK(d.D == 64 || d.D == 128 || d.D == 256, UVX_ERR_UNSUPPORTED, "attention_f32: head_dim %d not supported", d.D);
  AArgs a = mk(d);
  a.dout = (const float*)bd.dout; a.delta = bd.delta; a.delta_in = bd.delta;
  a.dq = (float*)bd.dq; a.dk = (float*)bd.dk; a.dv = (float*)bd.dv; a.lddq = bd.lddq; a.lddk = bd.lddk; a.lddv = bd.lddv;
  const long long nq = (long long)d.B * d.Hq * d.T, nk = (long long)d.B * d.Hkv * d.T;
  if (d.D == 64) {
    hipLaunchKernelGGL(attn_bwd_dq_f32_k<64>, dim3(cdiv(nq, 4)), dim3(256), 0, st, a);
    hipLaunchKernelGGL(attn_bwd_dkdv_f32_k<64>, dim3(cdiv(nk, 4)), dim3(256), 0, st, a);
  } else if (d.D == 128) {
    hipLaunchKernelGGL(attn_bwd_dq_f32_k<128>, dim3(cdiv(nq, 4)), dim3(256), 0, st, a);
    hipLaunchKernelGGL(attn_bwd_dkdv_f32_k<128>, dim3(cdiv(nk, 4)), dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL(attn_bwd_dq_f32_k<256>, dim3(cdiv(nq, 4)), dim3(256), 0, st, a);
    hipLaunchKernelGGL(attn_bwd_dkdv_f32_k<256>, dim3(cdiv(nk, 4)), dim3(256), 0, st, a);
  }
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

}  // namespace uvx
