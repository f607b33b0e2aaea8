// K1: Whisper log-mel frontend on device.
//
// Reference: call site ultravox_processing.py:295-303; arithmetic [3P] transformers
// WhisperFeatureExtractor._torch_extract_fbank_features (pinned 4.51.3):
//   torch.stft(n_fft=400, hop=160, hann(400) periodic, center=True, pad_mode="reflect") -> |.|^2, drop the
//   last frame -> mel_filters[201 x n_mels]^T @ power -> log10(clamp(1e-10)) -> max(x, clip_max - 8) with
//   clip_max the per-clip maximum over all mels x frames -> (x + 4) / 4.
// The 400-point real DFT is evaluated as a dense f32 contraction against host-built twiddle tables
// (400 = 2^4 * 5^2 has no radix-2 FFT; the dense form keeps every product a plain fma in a fixed order,
// bit-reproducible across launches) - on the f32 matrix cores since round 2, same bits as the scalar loop.
// Pass 1: one block = 32 consecutive frames of one clip: windowed frames in LDS -> power -> mel -> log10,
//         written unnormalised, plus the block maximum.  Pass 2: per-clip max, floor and affine map.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int NFFT = 400, HOP = 160, NB = 201, NBP = 208, FR = 32, XS = 402, PS = 210;

__global__ __launch_bounds__(256) void logmel_pass1_k(const float* __restrict__ pcm, const float* __restrict__ window,
                                                      const float* __restrict__ tw_cos, const float* __restrict__ tw_sin,
                                                      const float* __restrict__ mel_fb, float* __restrict__ out,
                                                      float* __restrict__ blkmax, int L, int n_mels, int F, int F_stride) {
  extern __shared__ float sm[];
  float* xw = sm;                    // [FR][XS] windowed frames (row stride XS = 402 floats: conflict-free A-operand reads)
  float* pw = sm + FR * XS;          // [FR][PS] power spectrum (row stride 210 floats, same reason)
  __shared__ float red[16];
  const int b = blockIdx.y, f0 = blockIdx.x * FR;
  const float* x = pcm + (long long)b * L;
  for (int i = threadIdx.x; i < FR * NFFT; i += blockDim.x) {
    const int f = i / NFFT, n = i % NFFT;
    long long idx = (long long)(f0 + f) * HOP + n - NFFT / 2;
    if (idx < 0) idx = -idx;                       // reflect (no edge repeat)
    if (idx >= L) idx = 2LL * (L - 1) - idx;
    float v = 0.f;
    if (f0 + f < F && idx >= 0 && idx < L) v = x[idx] * window[n];
    xw[f * XS + n] = v;
  }
  __syncthreads();
  // DFT on the f32 matrix cores: [32 frames x 400] . [400 x (cos | sin) 208 bins].  v_mfma_f32_16x16x4_f32 is bitwise a
  // k-ordered fmaf chain (gemm_f32.hip), so re / im accumulate in the same order as the scalar loop `re = fmaf(x[n], cos[n][k], re)` over
  // n = 0..399 that this replaces (tests/test_kernels_gpu.py pins the output to the HF extractor fixture as before) (round 1: 650 us per 8 x 30 s at 12 TFLOP/s of VALU; the matrix pipe does the 7.7 GFLOP
  // in a tenth of that).  A = frames (row = frame lr, k = n), B = twiddles (col = bin lr): accumulator element e is
  // (frame 4 lk + e, bin lr) of the 16 x 16 tile.  Wave w owns bin tiles w, w + 4, w + 8, w + 12 (13 tiles: the last only
  // for wave 0), cos and sin of a tile in the same lanes so that the power needs no exchange.
  {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    constexpr int NBT = NBP / 16;
    f32x4_t re[4][2], im[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < 2; ++t) { re[j][t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; im[j][t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    const float* xa0 = xw + lr * XS + lk;
    const float* xa1 = xa0 + 16 * XS;
    // Branch-free main loop: every wave runs four tile pairs - waves 1..3 have no 13th tile and recompute tile 12 into
    // accumulators that are never stored (wave 0's four pairs are the critical path either way) - because a conditional load
    // makes hipcc drain the prefetch with s_waitcnt vmcnt(0).  The twiddles of the NEXT k-step (L2-resident tables, ~500
    // cycles away) are fetched before the MFMAs of the current one.
    int toff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) toff[j] = lk * NBP + min(w + 4 * j, NBT - 1) * 16 + lr;
    float cc[4], sc[4], cn[4], sn[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { cc[j] = tw_cos[toff[j]]; sc[j] = tw_sin[toff[j]]; }
    for (int kk = 0; kk < NFFT / 4; ++kk) {
      const float a0 = xa0[kk * 4], a1 = xa1[kk * 4];
      const int kn = (kk + 1 < NFFT / 4 ? kk + 1 : kk) * 4 * NBP;        // (last step: re-load, unused)
#pragma unroll
      for (int j = 0; j < 4; ++j) { cn[j] = tw_cos[kn + toff[j]]; sn[j] = tw_sin[kn + toff[j]]; }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        re[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, cc[j], re[j][0], 0, 0, 0);
        re[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, cc[j], re[j][1], 0, 0, 0);
        im[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, sc[j], im[j][0], 0, 0, 0);
        im[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, sc[j], im[j][1], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { cc[j] = cn[j]; sc[j] = sn[j]; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (w + 4 * j < NBT) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            pw[(t * 16 + lk * 4 + e) * PS + (w + 4 * j) * 16 + lr] = re[j][t][e] * re[j][t][e] + im[j][t][e] * im[j][t][e];
      }
    }
  }
  __syncthreads();
  // mel + log10, also on the f32 matrix cores: mel_fb [n_mels x 208] . power^T [208 x 32 frames], k ascending - the same
  // fmaf chain as the scalar loop it replaces (the 7 padded bins add fmaf(0, 0, a) = a).  A = filter rows (mel), B = power
  // (col = frame lr): accumulator element e is (mel 4 lk + e, frame lr), so a store instruction writes 16 consecutive
  // frames per mel row.  Wave w owns mel tiles w and w + 4 (n_mels = 80 or 128: 5 or 8 tiles).
  float mx = -__builtin_huge_valf();
  {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    const int n_mt = n_mels / 16;
    f32x4_t acc[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[j][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    int foff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) foff[j] = (min(w + 4 * j, n_mt - 1) * 16 + lr) * NBP + lk;   // (clamped: see the DFT loop)
    const float* pb0 = pw + lr * PS + lk;
    const float* pb1 = pb0 + 16 * PS;
    float fc[2], fn[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) fc[j] = mel_fb[foff[j]];
    for (int kk = 0; kk < NBP / 4; ++kk) {
      const float b0 = pb0[kk * 4], b1 = pb1[kk * 4];
      const int kn = (kk + 1 < NBP / 4 ? kk + 1 : kk) * 4;
#pragma unroll
      for (int j = 0; j < 2; ++j) fn[j] = mel_fb[foff[j] + kn];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fc[j], b0, acc[j][0], 0, 0, 0);
        acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fc[j], b1, acc[j][1], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) fc[j] = fn[j];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int mt = w + 4 * j;
      if (mt >= n_mt) continue;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int f = t * 16 + lr;
        if (f0 + f >= F) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int m = mt * 16 + lk * 4 + e;
          const float lg = log10f(fmaxf(acc[j][t][e], 1e-10f));
          out[((long long)b * n_mels + m) * F_stride + f0 + f] = lg;
          mx = fmaxf(mx, lg);
        }
      }
    }
  }
  mx = block_max(mx, red);
  if (threadIdx.x == 0) blkmax[b * gridDim.x + blockIdx.x] = mx;
}

__global__ void logmel_pass2_k(float* __restrict__ out, const float* __restrict__ blkmax, int nblk, int n_mels, int F,
                               int F_stride) {
  __shared__ float red[16];
  const int b = blockIdx.y;
  float mx = -__builtin_huge_valf();
  for (int i = threadIdx.x; i < nblk; i += blockDim.x) mx = fmaxf(mx, blkmax[b * nblk + i]);
  mx = block_max(mx, red);
  const float floorv = mx - 8.0f;
  const long long n = (long long)n_mels * F;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int m = (int)(i / F), f = (int)(i % F);
    float* p = out + ((long long)b * n_mels + m) * F_stride + f;
    *p = (fmaxf(*p, floorv) + 4.0f) / 4.0f;
  }
}

}  // namespace

namespace uvx {

// tables: window[400]; tw_cos/tw_sin [400][208] (k fastest, zero padded); mel_fb [n_mels][208].
// scratch: B * ceil(F/32) floats.
int logmel(hipStream_t st, const float* pcm, const float* window, const float* tw_cos, const float* tw_sin,
           const float* mel_fb, float* out, float* scratch, int B, int L, int n_mels, int F_stride) {
  UVX_CHECK(L % HOP == 0 && L >= 2 * HOP, UVX_ERR_SHAPE, "logmel: L=%d must be a multiple of 160 and >= 320", L);
  UVX_CHECK(L > NFFT / 2, UVX_ERR_SHAPE, "logmel: reflect padding needs L > 200");
  const int F = L / HOP;
  UVX_CHECK(F_stride >= F, UVX_ERR_SHAPE, "logmel: F_stride=%d < F=%d", F_stride, F);
  UVX_CHECK(n_mels % 16 == 0 && n_mels >= 16 && n_mels <= 128, UVX_ERR_SHAPE, "logmel: n_mels=%d must be a multiple of 16, at most 128", n_mels);
  if (B == 0) return UVX_OK;
  const int nblk = cdiv(F, FR);
  const size_t sh = sizeof(float) * (FR * XS + FR * PS);
  static PerDeviceOnce attr_set;
  UVX_SET_ATTR_ONCE(attr_set, logmel_pass1_k, sh);
  hipLaunchKernelGGL(logmel_pass1_k, dim3(nblk, B), dim3(256), sh, st, pcm, window, tw_cos, tw_sin, mel_fb, out, scratch,
                     L, n_mels, F, F_stride);
  hipLaunchKernelGGL(logmel_pass2_k, dim3(32, B), dim3(256), 0, st, out, scratch, nblk, n_mels, F, F_stride);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

}  // namespace uvx
