// K1: Whisper log-mel frontend on device.
//
// Reference: call site ultravox_processing.py:295-303; arithmetic [3P] transformers
// WhisperFeatureExtractor._torch_extract_fbank_features (pinned 4.51.3):
//   torch.stft(n_fft=400, hop=160, hann(400) periodic, center=True, pad_mode="reflect") -> |.|^2, drop the
//   last frame -> mel_filters[201 x n_mels]^T @ power -> log10(clamp(1e-10)) -> max(x, clip_max - 8) with
//   clip_max the per-clip maximum over all mels x frames -> (x + 4) / 4.
// The 400-point real DFT is evaluated as a dense f32 contraction against host-built twiddle tables
// (400 = 2^4 * 5^2 has no radix-2 FFT; at 0.35 GFLOP per 30 s clip the dense form costs microseconds and
// keeps every product a plain fma, bit-reproducible across launches).
// Pass 1: one block = 32 consecutive frames of one clip: windowed frames in LDS -> power -> mel -> log10,
//         written unnormalised, plus the block maximum.  Pass 2: per-clip max, floor and affine map.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int NFFT = 400, HOP = 160, NB = 201, NBP = 208, FR = 32;

__global__ __launch_bounds__(256) void logmel_pass1_k(const float* __restrict__ pcm, const float* __restrict__ window,
                                                      const float* __restrict__ tw_cos, const float* __restrict__ tw_sin,
                                                      const float* __restrict__ mel_fb, float* __restrict__ out,
                                                      float* __restrict__ blkmax, int L, int n_mels, int F, int F_stride) {
  extern __shared__ float sm[];
  float* xw = sm;                    // [FR][NFFT] windowed frames
  float* pw = sm + FR * NFFT;        // [FR][NBP] power spectrum
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
    xw[i] = v;
  }
  __syncthreads();
  // DFT: thread = (4 bins) x (8 frames)
  const int kg = threadIdx.x % 52, fg = threadIdx.x / 52;
  if (fg < 4) {
    float re[8][4], im[8][4];
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
      for (int k = 0; k < 4; ++k) { re[f][k] = 0.f; im[f][k] = 0.f; }
    const float* xr = xw + fg * 8 * NFFT;
    for (int n = 0; n < NFFT; ++n) {
      const float4 c = *reinterpret_cast<const float4*>(tw_cos + n * NBP + kg * 4);
      const float4 s = *reinterpret_cast<const float4*>(tw_sin + n * NBP + kg * 4);
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        const float v = xr[f * NFFT + n];
        re[f][0] = fmaf(v, c.x, re[f][0]); re[f][1] = fmaf(v, c.y, re[f][1]);
        re[f][2] = fmaf(v, c.z, re[f][2]); re[f][3] = fmaf(v, c.w, re[f][3]);
        im[f][0] = fmaf(v, s.x, im[f][0]); im[f][1] = fmaf(v, s.y, im[f][1]);
        im[f][2] = fmaf(v, s.z, im[f][2]); im[f][3] = fmaf(v, s.w, im[f][3]);
      }
    }
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
      for (int k = 0; k < 4; ++k) pw[(fg * 8 + f) * NBP + kg * 4 + k] = re[f][k] * re[f][k] + im[f][k] * im[f][k];
  }
  __syncthreads();
  // mel + log10: thread -> (mel m, frame f), frames fastest for coalesced stores
  float mx = -__builtin_huge_valf();
  for (int i = threadIdx.x; i < n_mels * FR; i += blockDim.x) {
    const int m = i / FR, f = i % FR;
    if (f0 + f >= F) continue;
    const float* fb = mel_fb + m * NBP;
    const float* pr = pw + f * NBP;
    float a = 0.f;
    for (int k = 0; k < NB; ++k) a = fmaf(fb[k], pr[k], a);
    const float lg = log10f(fmaxf(a, 1e-10f));
    out[((long long)b * n_mels + m) * F_stride + f0 + f] = lg;
    mx = fmaxf(mx, lg);
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
  if (B == 0) return UVX_OK;
  const int nblk = cdiv(F, FR);
  const size_t sh = sizeof(float) * (FR * NFFT + FR * NBP);
  static bool attr_set = false;
  if (!attr_set) {
    UVX_HIP(hipFuncSetAttribute((const void*)logmel_pass1_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    attr_set = true;
  }
  hipLaunchKernelGGL(logmel_pass1_k, dim3(nblk, B), dim3(256), sh, st, pcm, window, tw_cos, tw_sin, mel_fb, out, scratch,
                     L, n_mels, F, F_stride);
  hipLaunchKernelGGL(logmel_pass2_k, dim3(32, B), dim3(256), 0, st, out, scratch, nblk, n_mels, F, F_stride);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

}  // namespace uvx
