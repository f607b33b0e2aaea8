// bf16 GEMM for few rows (M <= 16; up to 64 through the LDS-staged kernel): the decode step of generate() (M = batch) and other few-row
// problems.  Three kernels, dispatched by gemm_skinny_bf16 below:
//   M <= 2            gemv_rows_bf16_k       row streaming (1 KiB contiguous non-temporal weight reads), v_dot2c arithmetic, optional fused RMSNorm
//   M = 3..64         gemm_skinny_bf16_k<.., STAGE>  MFMA, weights through a wave-private LDS tile (K % 2048 == 0)
//   otherwise M <= 16 gemm_skinny_bf16_k     MFMA, fragments straight from global memory (round 2's kernel, K-span mapping since round 4)
// Measured on Llama-3.3-70B's decode step (ms per token, B = 1 / 8 / 32): 36.0 / 37.9 / 87 before round 4, 23.0 / 25.6 / 39.6 after.
//
// C[M, N] = epilogue(A[M, K] . B[N, K]^T).  With so few rows the product is weight streaming: every element of B is read
// once and multiplied by M values, so the kernel is HBM-bound and the tiled kernels (128-row tiles, a handful of active
// CUs at N = 4096) are the wrong tool: 25.9 ms per decoded token at C2 against a 2 ms weight-streaming floor.
// Here a block of 8 waves owns 32 output columns (two 16 x 16 MFMA tiles: for the fused SwiGLU epilogue the gate block and
// its up block); the waves split K (k32 chunk i goes to wave i mod 8, so the 8 waves together read 512 contiguous bytes of
// each weight row per step), B fragments are loaded straight from global memory in MFMA operand layout (16 rows x 64 B per
// instruction), A (M x K, at most 448 KB, cache resident) likewise with rows >= M zeroed.  Partial accumulators are summed
// through LDS in a fixed order and wave 0 runs the shared epilogue arithmetic (bias, GELU, residual, SwiGLU; the same
// bf16 rounding points as store_tile in gemm.hip).
#include "common.h"
#include "kernels.h"

namespace {

struct SkinnyArgs {
  const bf16_t* A; const bf16_t* B; bf16_t* C; const bf16_t* bias; const bf16_t* residual; bf16_t* C2;
  int M, N, K, lda, ldb, ldc, ldr, ldc2, res_mod, act, swiglu;
  float alpha;
  // gemv_rows_bf16_k<.., NORM = true>: A is the RAW residual-stream row; RMSNorm(A; norm_w) is applied on the way in
  const bf16_t* norm_w; float norm_eps; int norm_flavor;
};

// TILES = 16-column MFMA tiles per block: 2 (needed by the fused SwiGLU: gate block + up block) or 1 (twice the blocks:
// N = 4096 gives 256 blocks instead of 128, one per CU)
// STAGE (round 4, M = 3..16): the weights reach the MFMA fragments through a wave-private LDS tile.  A fragment load straight from global
// memory touches 16 rows x 64 bytes; here a wave instruction reads 2 rows x 512 contiguous bytes (non-temporal), eight of them bring
// 16 rows x 256 k into the wave's own [16][256 + 8] bf16 tile, and the eight k32 fragments are read back from there - the HBM side sees
// the row-streaming pattern (profiles/r04_hbm_stream_patterns.txt), LDS traffic is twice the weight bytes (a few per cent of its bandwidth).
// The next step's loads are issued right after the tile has been written, so they fly during the fragment reads and MFMAs; no block
// barrier (LDS operations of one wave execute in order).  Needs K % 2048 == 0 (each wave's K span in whole 256-element steps).
constexpr int SKS_PITCH = 528;                    // bytes per staged row: 256 bf16 + 16 bytes of padding
// MT (STAGE only): 16-row tiles of the ACTIVATION matrix served by one block - M <= 16 MT.  Every staged weight fragment is multiplied with
// MT activation fragments, so batches up to 64 (M = 17..64 used to fall to the 128-row tiled kernels: a handful of active CUs at N = 8192)
// stream the weights once at the same rate.
// NORM (round 6, STAGE with MT = 1: the decode step at 3..16 rows; opt-in, option 24 = 1 - measured slower, see gemm_skinny_rmsnorm_bf16): A is the RAW residual-stream block and RMSNorm(A; norm_w) is applied on the
// way in, as gemv_rows_bf16_k<.., NORM> does for M <= 2 - the separate rmsnorm_fwd_k launch on 8 rows costs 8.7 us, 161 of them 1.4 ms of a
// 25 ms 70B token at B = 8.  A block reads every activation element once anyway (wave w: its K span of all 16 rows), so the prologue walks the
// same addresses first: lane (row, g) squares its 8 elements of every k32 chunk of the wave's span, the four lane groups of a row and then
// the eight waves are folded in a fixed order (LDS), rstd = rsqrt(sum / K + eps).  The main loop then normalises each activation fragment
// in registers with rmsnorm_fwd_k's arithmetic and rounding points (flavor 0: w * round(x * rstd); 1, Gemma: (x * rstd) * (1 + w)) before
// the MFMA - ~50 VALU instructions per fragment against the 16 KB of weights a wave streams per MFMA pair.  The first weight rows are already
// in flight (issue(0, 0)) while the prologue runs.  rstd may differ from the separate kernel's in the last bit (summation order).
template <int TILES, bool STAGE = false, int MT = 1, bool NORM = false>
__global__ __launch_bounds__(512) void gemm_skinny_bf16_k(SkinnyArgs p) {
  static_assert(STAGE || MT == 1, "several activation row tiles: staged kernel only");
  static_assert(!NORM || (STAGE && MT == 1), "fused RMSNorm: the staged kernel on one activation row tile");
  __shared__ float nss[NORM ? 8 * 16 : 1];
  extern __shared__ __attribute__((aligned(16))) unsigned char sk_dyn[];      // STAGE: 8 waves x 16 x SKS_PITCH bytes (then the reduction)
  __shared__ float red_static[STAGE ? 1 : 8 * 2 * 64 * 4];
  float (*red)[2][MT][64][4] = reinterpret_cast<float (*)[2][MT][64][4]>(STAGE ? reinterpret_cast<float*>(sk_dyn) : red_static);
  static_assert(!STAGE || 8 * 2 * MT * 64 * 4 * 4 <= 8 * 16 * SKS_PITCH, "reduction array fits the staging buffer");
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int frow = lane & 15, fg = lane >> 4;
  const int n0 = blockIdx.x * (16 * TILES);
  const bf16_t* b0 = p.B + (long long)min(n0 + frow, p.N - 1) * p.ldb + fg * 8;
  const bf16_t* b1 = p.B + (long long)min(n0 + 16 + frow, p.N - 1) * p.ldb + fg * 8;
  const bool m_ok = frow < p.M;
  const bf16_t* a0 = p.A + (long long)(m_ok ? frow : 0) * p.lda + fg * 8;
  f32x4_t acc[MT][2];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) { acc[mt][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[mt][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
  const bf16x8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
  const int nchunk = p.K / 32;
  // Which k32 chunks a wave takes.  Round 4: contiguous SPANS (wave w: chunks [w * n/8, (w + 1) * n/8)) instead of the interleave
  // w, w + 8, ...: every wave then walks its 16 rows front to back, which streams at 6.0 instead of 4.9 TB/s on the 70B gate|up
  // matrix (profiles/r04_hbm_stream_patterns.txt, patterns 4 / 1).  The interleave remains for K that does not split evenly.
  if constexpr (STAGE) {
    typedef unsigned int su32x4_t __attribute__((ext_vector_type(4)));
    unsigned char* buf = sk_dyn + w * (16 * SKS_PITCH);
    const int kspan = p.K / 8, k_begin = w * kspan, nst = kspan / 256;
    const int lrow = lane >> 5, lk = (lane & 31) * 8;            // load mapping: instruction i brings rows 2 i + lrow, elements lk .. lk + 7
    // one (step, tile) pair at a time: its 8 row-pair loads were issued while the previous pair was being multiplied
    su32x4_t ld[8];
    auto issue = [&](int st, int t) {
#pragma unroll
      for (int i8 = 0; i8 < 8; ++i8) {
        const bf16_t* src = p.B + (long long)min(n0 + 16 * t + 2 * i8 + lrow, p.N - 1) * p.ldb + k_begin + st * 256 + lk;
        ld[i8] = __builtin_nontemporal_load(reinterpret_cast<const su32x4_t*>(src));
      }
    };
    issue(0, 0);
    // activation fragments: MT <= 2: the 8 chunks of a step stay in registers for both weight tiles; MT = 4: four chunks at a time,
    // re-read (L1 / L2) for the second weight tile.  (Two chunks at a time at MT = 2 - 126 instead of 164 VGPRs, two blocks per CU - was
    // measured SLOWER: 44.3 vs 39.6 ms per 70B token at B = 32: the re-reads cost more than the occupancy buys.)
    constexpr int CH = MT <= 2 ? 8 : 4;
    const bf16_t* am[MT];
    bool mok[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      mok[mt] = mt * 16 + frow < p.M;
      am[mt] = p.A + (long long)(mok[mt] ? mt * 16 + frow : 0) * p.lda + fg * 8;
    }
    bf16x8_t xf[MT][CH];
    float rstd = 0.f;
    if constexpr (NORM) {
      float ss = 0.f;
      for (int c0 = 0; c0 < kspan / 32; c0 += 8) {           // (kspan is a multiple of 256)
        u16x8_t v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = *reinterpret_cast<const u16x8_t*>(am[0] + k_begin + (c0 + c) * 32);
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float f = bf2f(v[c][e]); ss += f * f; }
      }
      ss += __shfl_xor(ss, 16, 64);
      ss += __shfl_xor(ss, 32, 64);
      if (fg == 0) nss[w * 16 + frow] = ss;
      __syncthreads();
      float tot = 0.f;
#pragma unroll
      for (int ww = 0; ww < 8; ++ww) tot += nss[ww * 16 + frow];
      rstd = rsqrtf(tot / p.K + p.norm_eps);
    }
    // (NORM) one activation fragment: 8 raw elements of this lane's row -> RMSNorm'ed bf16 operand
    auto norm_frag = [&](const bf16_t* src, int k) {
      const u16x8_t raw = *reinterpret_cast<const u16x8_t*>(src + k);
      const u16x8_t wv = *reinterpret_cast<const u16x8_t*>(p.norm_w + k + fg * 8);
      u16x8_t o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x = bf2f(raw[e]) * rstd, g = bf2f(wv[e]);
        o[e] = f2bf(p.norm_flavor ? x * (1.0f + g) : g * bf2f(f2bf(x)));
      }
      return __builtin_bit_cast(bf16x8_t, o);
    };
    for (int st = 0; st < nst; ++st) {
      const int k0 = k_begin + st * 256;
#pragma unroll
      for (int t = 0; t < TILES; ++t) {          // (compile-time tile index: the accumulators are addressed statically)
#pragma unroll
        for (int i8 = 0; i8 < 8; ++i8) *reinterpret_cast<su32x4_t*>(buf + (2 * i8 + lrow) * SKS_PITCH + lk * 2) = ld[i8];
        // the registers are free again: the next pair's rows
        if (t + 1 < TILES) issue(st, t + 1);
        else if (st + 1 < nst) issue(st + 1, 0);
#pragma unroll
        for (int h = 0; h < 8 / CH; ++h) {
          if (CH < 8 || t == 0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int c = 0; c < CH; ++c) {
                if constexpr (NORM) xf[mt][c] = mok[mt] ? norm_frag(am[mt], k0 + (h * CH + c) * 32) : zero;
                else xf[mt][c] = mok[mt] ? *reinterpret_cast<const bf16x8_t*>(am[mt] + k0 + (h * CH + c) * 32) : zero;
              }
          }
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            const bf16x8_t wf = *reinterpret_cast<const bf16x8_t*>(buf + frow * SKS_PITCH + (h * CH + c) * 64 + fg * 16);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[mt][c], acc[mt][t], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();            // every wave is done with its tile: the buffer becomes the reduction array
  }
  if constexpr (!STAGE) {
  const bool span = nchunk % 8 == 0;
  const int stride = span ? 1 : 8, end = span ? (w + 1) * (nchunk / 8) : nchunk;
  int i = span ? w * (nchunk / 8) : w;
  // UN chunks per iteration: 2-3 x UN independent 16-byte loads in flight per lane (the kernel lives on memory-level parallelism)
  constexpr int UN = TILES == 2 ? 4 : 8;
  for (; i + stride * (UN - 1) < end; i += stride * UN) {
    bf16x8_t x[UN], u[UN], v[UN];
#pragma unroll
    for (int t = 0; t < UN; ++t) {
      const int kk = (i + stride * t) * 32;
      u[t] = *reinterpret_cast<const bf16x8_t*>(b0 + kk);
      if (TILES == 2) v[t] = *reinterpret_cast<const bf16x8_t*>(b1 + kk);
      x[t] = *reinterpret_cast<const bf16x8_t*>(a0 + kk);
    }
#pragma unroll
    for (int t = 0; t < UN; ++t) {
      const bf16x8_t xx = m_ok ? x[t] : zero;
      acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u[t], xx, acc[0][0], 0, 0, 0);
      if (TILES == 2) acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v[t], xx, acc[0][1], 0, 0, 0);
    }
  }
  for (; i < end; i += stride) {
    const int kk = i * 32;
    const bf16x8_t u = *reinterpret_cast<const bf16x8_t*>(b0 + kk);
    bf16x8_t v = zero;
    if (TILES == 2) v = *reinterpret_cast<const bf16x8_t*>(b1 + kk);
    const bf16x8_t x = *reinterpret_cast<const bf16x8_t*>(a0 + kk);
    const bf16x8_t xx = m_ok ? x : zero;
    acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u, xx, acc[0][0], 0, 0, 0);
    if (TILES == 2) acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v, xx, acc[0][1], 0, 0, 0);
  }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[w][0][mt][lane][e] = acc[mt][0][e]; red[w][1][mt][lane][e] = acc[mt][1][e]; }
  __syncthreads();
  if (w >= MT) return;          // wave mt finishes activation row tile mt
  const int mt = w;
  float s[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = 0.f;
#pragma unroll
      for (int ww = 0; ww < 8; ++ww) t += red[ww][j][mt][lane][e];     // fixed order: deterministic
      s[j][e] = t * p.alpha;
    }
  const int m = mt * 16 + frow;
  if (m >= p.M) return;
  if (p.swiglu) {    // tile 0 = 16 gate columns, tile 1 = the matching up columns (interleaved packing, weights.py)
    const int n = n0 + fg * 4;
    if (n >= p.N) return;
    u16x4_t og, ou, oa;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      og[e] = f2bf(s[0][e]);
      ou[e] = f2bf(s[1][e]);
      const float g = bf2f(og[e]);
      oa[e] = f2bf(bf2f(f2bf(g / (1.0f + __expf(-g)))) * bf2f(ou[e]));
    }
    bf16_t* crow = p.C + (long long)m * p.ldc + n;
    *reinterpret_cast<u16x4_t*>(crow) = og;
    *reinterpret_cast<u16x4_t*>(crow + 16) = ou;
    *reinterpret_cast<u16x4_t*>(p.C2 + (long long)m * p.ldc2 + n0 / 2 + fg * 4) = oa;
    return;
  }
#pragma unroll
  for (int j = 0; j < TILES; ++j) {
    const int n = n0 + 16 * j + fg * 4;
    if (n >= p.N) continue;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = s[j][e] + (p.bias ? bf2f(p.bias[n + e]) : 0.f);
      t = bf2f(f2bf(t));
      if (p.act == 1) t = bf2f(f2bf(gelu_fast(t)));
      v[e] = t;
    }
    if (p.residual) {
      const int rm = p.res_mod > 0 ? (m % p.res_mod) : m;
      const u16x4_t r4 = *reinterpret_cast<const u16x4_t*>(p.residual + (long long)rm * p.ldr + n);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += bf2f(r4[e]);
    }
    u16x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
    *reinterpret_cast<u16x4_t*>(p.C + (long long)m * p.ldc + n) = o;
  }
}


// ---- round 4: row-streaming kernel for M <= 2 (the decode step at batch 1 and 2) ----
// The MFMA mapping above reads 16 weight rows x 64 bytes per wave instruction (row stride K * 2 bytes): on Llama-3.3-70B's
// gate|up matrix (940 MB) that pattern streams at 4.8 TB/s, while the same bytes read as whole rows - 1 KiB contiguous per wave
// instruction, non-temporal - stream at 6.5-7.1 TB/s (tools/probes/hbm_stream_probe.hip, profiles/r04_hbm_stream_patterns.txt).
// Here a block of 8 waves owns RB consecutive weight rows and walks them R at a time; for each row the 8 waves together read
// 8 KiB contiguous per step (wave w takes the 512-element steps w, w + 8, ...), so a block streams its RB x K region front to back.
// A lane holds 8 consecutive k of a row; the matching 8 k of each of the M activation rows come from L1 / L2 (M x K x 2 bytes
// in all, re-read per R rows); products are exact (bf16 x bf16 in f32), accumulation f32 (v_dot2c_f32_bf16), lanes and waves are
// folded in a fixed order.  Epilogue arithmetic and rounding points as the kernel above.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

// (the components are copied to scalars first: __builtin_bit_cast applied directly to a vector-element lvalue - bit_cast(a.y) -
//  reads element 0 for every component under hipcc 7.2, which also shrinks the 16-byte loads to 4 bytes)
__device__ __forceinline__ float dot2(unsigned a, unsigned b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
}
__device__ __forceinline__ float dot8(const u32x4_t a, const u32x4_t b, float c) {
  const unsigned a0 = a.x, a1 = a.y, a2 = a.z, a3 = a.w, b0 = b.x, b1 = b.y, b2 = b.z, b3 = b.w;
  return dot2(a3, b3, dot2(a2, b2, dot2(a1, b1, dot2(a0, b0, c))));
}

// Sums NV per-lane values over the 64 lanes in NV - 1 + (6 - log2 NV) shuffles instead of 6 NV: at every stage a lane keeps one half of
// its values and hands the other half to its partner (xor 32, 16, ...), until one value is left, which the remaining butterfly stages
// finish.  Afterwards v[0] of lane L is the total of value index idx(L) = the top log2(NV) bits of L read as a number (bit 5 = the index's
// most significant bit); all 64 >> log2(NV) lanes of a group hold the same total.  Fixed order: deterministic.
template <int NV>
__device__ __forceinline__ void wave_sum_multi(float (&v)[NV], int lane) {
  static_assert(NV >= 1 && NV <= 64 && (NV & (NV - 1)) == 0, "power of two");
  int d = 32;
#pragma unroll
  for (int h = NV / 2; h >= 1; h /= 2, d >>= 1) {
    const bool upper = (lane & d) != 0;
#pragma unroll
    for (int i = 0; i < h; ++i) {
      const float keep = upper ? v[h + i] : v[i], send = upper ? v[i] : v[h + i];
      v[i] = keep + __shfl_xor(send, d, 64);
    }
  }
#pragma unroll
  for (; d >= 1; d >>= 1) v[0] += __shfl_xor(v[0], d, 64);
}
template <int NV> __device__ __forceinline__ int wave_sum_multi_index(int lane) {
  int bits = 0;
  for (int n = NV; n > 1; n >>= 1) ++bits;
  return bits == 0 ? 0 : (lane >> (6 - bits));
}

// NORM: the activation rows are RMS-normalised on the way in (the decode step's input_layernorm -> q|k|v and post_attention_layernorm ->
// gate|up: the separate rmsnorm_fwd_k launch on ONE row of 8192 elements cost 6.2 us, 161 of them 1.0 ms of a 26 ms 70B token).  Every
// block normalises the M rows itself - they are M x K x 2 bytes, read by every block anyway - into LDS (xs, dynamic: MB x K bf16) with
// rmsnorm_fwd_k's arithmetic and rounding points (flavor 0: w * round(x * rstd); 1, Gemma: (x * rstd) * (1 + w)), and the main loop reads
// its activation vectors from there.  The sum of squares is folded in this block's own order (per-thread strided partial sums, wave
// butterfly, waves in order): rstd may differ from the separate kernel's in the last bit.
template <int MB /*activation rows served: 1, 2*/, int R /*weight rows in flight per wave*/, int RB /*weight rows per block: 8, 16 or 32*/,
          bool NORM = false, int KSTEPS = (MB <= 2 ? 2 : 1) /*512-element k-steps in flight per trip*/>
__global__ __launch_bounds__(512) void gemv_rows_bf16_k(SkinnyArgs p) {
  __shared__ float part[8][RB][MB];
  __shared__ float nred[16];
  extern __shared__ __attribute__((aligned(16))) unsigned char xs_raw[];      // NORM: [MB][K] bf16
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n0 = blockIdx.x * RB;
  const int nsteps = (p.K + 511) / 512;
  const u32x4_t zero = {0u, 0u, 0u, 0u};
  const bf16_t* arow[MB];
  if (NORM) {
    bf16_t* xs = reinterpret_cast<bf16_t*>(xs_raw);
    constexpr int MAXV = 4;                       // 8-element vectors per thread: K <= 512 * 8 * MAXV = 16384
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      const bf16_t* xr = p.A + (long long)min(m, p.M - 1) * p.lda;
      float v[MAXV][8];
      float ss = 0.f;
#pragma unroll
      for (int q = 0; q < MAXV; ++q) {
        const int c = (threadIdx.x + q * 512) * 8;
        if (c < p.K) {
          ld8<bf16_t>(xr + c, v[q]);
#pragma unroll
          for (int e = 0; e < 8; ++e) ss += v[q][e] * v[q][e];
        }
      }
      const float rstd = rsqrtf(block_sum(ss, nred) / p.K + p.norm_eps);
#pragma unroll
      for (int q = 0; q < MAXV; ++q) {
        const int c = (threadIdx.x + q * 512) * 8;
        if (c < p.K) {
          float wv[8], o[8];
          ld8<bf16_t>(p.norm_w + c, wv);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = p.norm_flavor ? (v[q][e] * rstd) * (1.0f + wv[e]) : wv[e] * rnd<bf16_t>(v[q][e] * rstd);
          st8<bf16_t>(xs + (long long)m * p.K + c, o);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MB; ++m) arow[m] = xs + (long long)m * p.K + lane * 8;
  } else {
#pragma unroll
    for (int m = 0; m < MB; ++m) arow[m] = p.A + (long long)min(m, p.M - 1) * p.lda + lane * 8;
  }
  for (int rb = 0; rb < RB; rb += R) {
    float acc[R][MB];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int m = 0; m < MB; ++m) acc[r][m] = 0.f;
    const bf16_t* brow[R];
#pragma unroll
    for (int r = 0; r < R; ++r) brow[r] = p.B + (long long)min(n0 + rb + r, p.N - 1) * p.ldb + lane * 8;
    // KSTEPS k-steps per trip: KSTEPS x R weight loads (HBM, non-temporal) + KSTEPS x MB activation loads (cache / LDS) in flight per lane
    for (int j = w; j < nsteps; j += 8 * KSTEPS) {
      u32x4_t wv[KSTEPS][R], xv[KSTEPS][MB];
#pragma unroll
      for (int q = 0; q < KSTEPS; ++q) {
        const int kq = (j + 8 * q) * 512;
        const bool ok = j + 8 * q < nsteps && kq + lane * 8 < p.K;
#pragma unroll
        for (int r = 0; r < R; ++r) wv[q][r] = ok ? __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(brow[r] + kq)) : zero;
#pragma unroll
        for (int m = 0; m < MB; ++m) xv[q][m] = ok ? *reinterpret_cast<const u32x4_t*>(arow[m] + kq) : zero;
      }
#pragma unroll
      for (int q = 0; q < KSTEPS; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int m = 0; m < MB; ++m) acc[r][m] = dot8(wv[q][r], xv[q][m], acc[r][m]);
    }
    // fold the lanes: all R x MB values in one halving butterfly (value index = r * MB + m)
    float flat[R * MB];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int m = 0; m < MB; ++m) flat[r * MB + m] = acc[r][m];
    wave_sum_multi<R * MB>(flat, lane);
    if ((lane & (64 / (R * MB) - 1)) == 0) {
      const int idx = wave_sum_multi_index<R * MB>(lane);
      part[w][rb + idx / MB][idx % MB] = flat[0];
    }
  }
  __syncthreads();
  // ---- epilogue (one thread per output element; the sums over the waves in a fixed order) ----
  const int t = threadIdx.x;
  if (p.swiglu) {      // RB = 32: rows 0..15 = gate block, 16..31 = the matching up block (interleaved packing, weights.py)
    const int m = t / 16, c = t % 16;
    if (m >= p.M || m >= MB || n0 + c >= p.N) return;
    float g = 0.f, u = 0.f;
#pragma unroll
    for (int ww = 0; ww < 8; ++ww) { g += part[ww][c][m]; u += part[ww][16 + c][m]; }
    const bf16_t og = f2bf(g * p.alpha), ou = f2bf(u * p.alpha);
    const float gf = bf2f(og);
    bf16_t* crow = p.C + (long long)m * p.ldc + n0;
    crow[c] = og;
    crow[16 + c] = ou;
    p.C2[(long long)m * p.ldc2 + n0 / 2 + c] = f2bf(bf2f(f2bf(gf / (1.0f + __expf(-gf)))) * bf2f(ou));
    return;
  }
  const int m = t / RB, c = t % RB, n = n0 + c;
  if (m >= p.M || m >= MB || n >= p.N) return;
  float v = 0.f;
#pragma unroll
  for (int ww = 0; ww < 8; ++ww) v += part[ww][c][m];
  v = v * p.alpha + (p.bias ? bf2f(p.bias[n]) : 0.f);
  v = bf2f(f2bf(v));
  if (p.act == 1) v = bf2f(f2bf(gelu_fast(v)));
  if (p.residual) v += bf2f(p.residual[(long long)(p.res_mod > 0 ? (m % p.res_mod) : m) * p.ldr + n]);
  p.C[(long long)m * p.ldc + n] = f2bf(v);
}

}  // namespace

namespace uvx {

// Weight rows per block: 32 for the SwiGLU epilogue (a gate block + its up block); otherwise whichever of 32 / 16 / 8 spreads the blocks most
// evenly over the CUs by more than 5 % (q|k|v of Llama-3.3-70B, N = 10240: 640 blocks of 16 rows = 2.5 per CU - some CUs stream 3 blocks while others
// stream 2 - against 1280 blocks of 8 rows = 5 per CU); ties go to the larger block (fewer re-reads of the activation rows).
static int gemv_rows_per_block(const uvx::GemmDesc& d) {
  if (d.swiglu) return 32;
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
  }();
  int best = 32;
  double best_cost = 1e30;
  for (int rb : {32, 16, 8}) {
    const int blocks = (d.N + rb - 1) / rb;
    if (rb == 32 && d.N < 16384) continue;              // (few large blocks leave CUs idle at the small N of the attention projections)
    const double per_cu = (double)blocks / cus, cost = ceil(per_cu) / per_cu;
    if (cost < 0.95 * best_cost) { best_cost = cost; best = rb; }      // a smaller block only for a clear (> 5 %) balance gain
  }
  return best;
}

// C = epilogue(RMSNorm(A; norm_w, eps, flavor) . B^T) for the decode step's few rows: one launch instead of rmsnorm_fwd + gemm.
// Returns UVX_ERR_UNSUPPORTED (no launch, no error text) when the fused kernel does not serve the problem: the caller then runs the two.
int gemm_skinny_rmsnorm_bf16(hipStream_t st, const GemmDesc& d, const void* norm_w, float eps, int flavor) {
  const bool common = d.M > 0 && d.batch <= 1 && !d.out_f32 && !d.accumulate && !d.m_dev && d.swiglu != 2 && d.act < 2 && !d.b_kn && d.lda % 8 == 0 &&
                      d.ldb % 8 == 0 && (!d.swiglu || (d.N % 32 == 0 && d.C2)) && norm_w && uvx::g_options[4] == 1;
  const bool ok = common && d.M <= 2 && d.K % 8 == 0 && d.K <= 16384 && (size_t)d.M * d.K * 2 <= 64 * 1024;
  // 3..16 rows: the staged MFMA kernel normalises its activation fragments in registers - OPT-IN (option 24 = 1): measured 19-32 % SLOWER per token than
  // the two launches (profiles/r06_decode_norm_in_skinny_ab.txt): the kernel lives on loads in flight per wave and the ~70 VALU instructions per fragment
  // lengthen every wave's step
  const bool ok_staged = common && d.M >= 3 && d.M <= 16 && d.K % 2048 == 0 && d.N % 4 == 0 && d.ldc % 4 == 0 && (!d.swiglu || d.ldc2 % 4 == 0) &&
                         (!d.residual || d.ldr % 4 == 0) && uvx::g_options[24] == 1;
  if (!ok && !ok_staged) return UVX_ERR_UNSUPPORTED;
  SkinnyArgs a;
  a.A = (const bf16_t*)d.A; a.B = (const bf16_t*)d.B; a.C = (bf16_t*)d.C; a.bias = (const bf16_t*)d.bias;
  a.residual = (const bf16_t*)d.residual; a.C2 = (bf16_t*)d.C2;
  a.M = d.M; a.N = d.N; a.K = d.K; a.lda = d.lda; a.ldb = d.ldb; a.ldc = d.ldc; a.ldr = d.ldr; a.ldc2 = d.ldc2;
  a.res_mod = d.res_mod; a.act = d.act; a.swiglu = d.swiglu; a.alpha = d.alpha;
  a.norm_w = (const bf16_t*)norm_w; a.norm_eps = eps; a.norm_flavor = flavor;
  uvx::ProfScope prof(st, uvx::PROF_GEMM, 2.0 * d.M * d.N * (double)d.K,
                      ((double)d.M * d.K + (double)d.N * d.K) * 2.0 + (double)d.M * d.N * 2.0);
  if (uvx::g_prof_on) uvx::prof_tag(d.M, d.N, d.K, 1, ok ? 201 : 202);
  if (!ok) {
    constexpr size_t shs = 8 * 16 * SKS_PITCH;
    const bool two = d.swiglu || d.N >= 16384;
    const dim3 sgrid(two ? (d.N + 31) / 32 : (d.N + 15) / 16);
    if (two) {
      static PerDeviceOnce attr_n2;
      UVX_SET_ATTR_ONCE(attr_n2, (gemm_skinny_bf16_k<2, true, 1, true>), shs);
      hipLaunchKernelGGL((gemm_skinny_bf16_k<2, true, 1, true>), sgrid, dim3(512), shs, st, a);
    } else {
      static PerDeviceOnce attr_n1;
      UVX_SET_ATTR_ONCE(attr_n1, (gemm_skinny_bf16_k<1, true, 1, true>), shs);
      hipLaunchKernelGGL((gemm_skinny_bf16_k<1, true, 1, true>), sgrid, dim3(512), shs, st, a);
    }
    UVX_LAUNCH_CHECK();
    return UVX_OK;
  }
  const int rb = gemv_rows_per_block(d);
  const dim3 grid((d.N + rb - 1) / rb);
  const int mb = d.M;
  const size_t sh = (size_t)mb * d.K * 2;
#define UVX_GEMVN(MB, RBV) do { \
    static PerDeviceOnce attr; \
    UVX_SET_ATTR_ONCE(attr, (gemv_rows_bf16_k<MB, 4, RBV, true>), 64 * 1024); \
    hipLaunchKernelGGL((gemv_rows_bf16_k<MB, 4, RBV, true>), grid, dim3(512), sh, st, a); } while (0)
#define UVX_GEMVN_R(RR, RBV) do { \
    static PerDeviceOnce attr; \
    UVX_SET_ATTR_ONCE(attr, (gemv_rows_bf16_k<1, RR, RBV, true>), 64 * 1024); \
    hipLaunchKernelGGL((gemv_rows_bf16_k<1, RR, RBV, true>), grid, dim3(512), sh, st, a); } while (0)
  if (mb == 1 && uvx::g_options[26] == 8) { if (rb == 32) UVX_GEMVN_R(8, 32); else if (rb == 16) UVX_GEMVN_R(8, 16); else UVX_GEMVN_R(8, 8); }
  else if (mb == 1 && uvx::g_options[26] == 0) { if (rb == 32) UVX_GEMVN_R(2, 32); else if (rb == 16) UVX_GEMVN_R(2, 16); else UVX_GEMVN_R(2, 8); }
  else if (mb == 1 && uvx::g_options[26] == 16) { if (rb == 32) UVX_GEMVN_R(16, 32); else if (rb == 16) UVX_GEMVN_R(16, 16); else UVX_GEMVN_R(8, 8); }
  else
  if (rb == 32) { if (mb == 1) UVX_GEMVN(1, 32); else UVX_GEMVN(2, 32); }
  else if (rb == 16) { if (mb == 1) UVX_GEMVN(1, 16); else UVX_GEMVN(2, 16); }
  else { if (mb == 1) UVX_GEMVN(1, 8); else UVX_GEMVN(2, 8); }
#undef UVX_GEMVN
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

bool gemm_skinny_applicable(const GemmDesc& d) {
  // (M = 17..64: only the staged kernel serves several activation row tiles)
  const bool rows_ok = d.M <= 16 || (d.M <= 64 && d.K % 2048 == 0 && uvx::g_options[4] != 2);
  return d.M > 0 && rows_ok && d.batch <= 1 && !d.out_f32 && !d.accumulate && !d.m_dev && d.swiglu != 2 && d.act < 2 && !d.b_kn &&
         d.K % 32 == 0 && d.lda % 8 == 0 && d.ldb % 8 == 0 && d.N % 4 == 0 && d.ldc % 4 == 0 &&
         (!d.swiglu || (d.N % 32 == 0 && d.ldc2 % 4 == 0)) && (!d.residual || d.ldr % 4 == 0) && uvx::g_options[4];
}

int gemm_skinny_bf16(hipStream_t st, const GemmDesc& d) {
  SkinnyArgs a;
  a.A = (const bf16_t*)d.A; a.B = (const bf16_t*)d.B; a.C = (bf16_t*)d.C; a.bias = (const bf16_t*)d.bias;
  a.residual = (const bf16_t*)d.residual; a.C2 = (bf16_t*)d.C2;
  a.M = d.M; a.N = d.N; a.K = d.K; a.lda = d.lda; a.ldb = d.ldb; a.ldc = d.ldc; a.ldr = d.ldr; a.ldc2 = d.ldc2;
  a.res_mod = d.res_mod; a.act = d.act; a.swiglu = d.swiglu; a.alpha = d.alpha;
  a.norm_w = nullptr; a.norm_eps = 0.f; a.norm_flavor = 0;
  uvx::ProfScope prof(st, uvx::PROF_GEMM, 2.0 * d.M * d.N * (double)d.K,
                      ((double)d.M * d.K + (double)d.N * d.K) * 2.0 + (double)d.M * d.N * 2.0);
  if (uvx::g_prof_on) uvx::prof_tag(d.M, d.N, d.K, 1, 200);
  // M <= 8: the row-streaming kernel (option 4 = 2 keeps the MFMA mapping for every M: same-box A/B)
  // (more activation rows cost more than the mapping gains - every weight vector then needs M activation vectors from L1 / LDS and the
  //  registers halve the occupancy: 70B decode, ms per token, row kernel / MFMA mapping: B = 2: 27.0 / 28.8, B = 3: 34.1 / 29.1,
  //  B = 4: 34.4 / 29.3, B = 8: 53 / 31 - profiles/r04_decode_gemv_batch_ab.txt - so the row-streaming kernel serves M <= 2.
  //  With the halving-butterfly lane reduction and one k-step per trip the picture is the same: B = 4: 37.4 / 29.3, B = 8: 89 / 30.6.)
  if (d.M <= 2 && uvx::g_options[4] != 2 && d.K % 8 == 0 && (!d.swiglu || d.N % 32 == 0)) {
    const int rb = gemv_rows_per_block(d);
    const dim3 grid((d.N + rb - 1) / rb);
#define UVX_GEMV(MB, RR) do { if (rb == 32) hipLaunchKernelGGL((gemv_rows_bf16_k<MB, RR, 32>), grid, dim3(512), 0, st, a); \
                          else if (rb == 16) hipLaunchKernelGGL((gemv_rows_bf16_k<MB, RR, 16>), grid, dim3(512), 0, st, a); \
                          else hipLaunchKernelGGL((gemv_rows_bf16_k<MB, RR, 8>), grid, dim3(512), 0, st, a); } while (0)
    // Weight rows in flight per wave at M = 1 (tuning option 26; profiles/r06_decode_gemv_rows_in_flight_ab.txt, ms per decoded token, 8B / 70B): 16 rows 4.53 / 28.98,
    // 8 rows 3.70 / 24.41, 4 rows (rounds 4-5) 3.27 / 22.79, 2 rows 3.25 / 22.65 - the default since round 6; 0 = 2 rows, 4 / 8 / 16 force (the lane butterfly
    // folds R values at once, so the forms agree to rounding, not bitwise)
    if (d.M == 1 && uvx::g_options[26] == 8) UVX_GEMV(1, 8);
    else if (d.M == 1 && uvx::g_options[26] == 0) UVX_GEMV(1, 2);
    else if (d.M == 1 && uvx::g_options[26] == 16) { if (rb == 32) hipLaunchKernelGGL((gemv_rows_bf16_k<1, 16, 32>), grid, dim3(512), 0, st, a);
                                                      else if (rb == 16) hipLaunchKernelGGL((gemv_rows_bf16_k<1, 16, 16>), grid, dim3(512), 0, st, a);
                                                      else hipLaunchKernelGGL((gemv_rows_bf16_k<1, 8, 8>), grid, dim3(512), 0, st, a); }
    else if (d.M == 1) UVX_GEMV(1, 4); else UVX_GEMV(2, 4);
#undef UVX_GEMV
    UVX_LAUNCH_CHECK();
    return UVX_OK;
  }
  if (d.K % 2048 == 0 && uvx::g_options[4] != 2) {      // weights through a wave-private LDS tile (option 4 = 2: straight fragment loads, A/B)
    constexpr size_t sh = 8 * 16 * SKS_PITCH;             // 67.6 KB: two blocks per CU
    const bool two = d.swiglu || d.N >= 16384;
    const dim3 grid(two ? (d.N + 31) / 32 : (d.N + 15) / 16);
#define UVX_SKS(TT, MTT) do { \
      static PerDeviceOnce attr2; \
      UVX_SET_ATTR_ONCE(attr2, (gemm_skinny_bf16_k<TT, true, MTT>), sh); \
      hipLaunchKernelGGL((gemm_skinny_bf16_k<TT, true, MTT>), grid, dim3(512), sh, st, a); } while (0)
    if (d.M <= 16) { if (two) UVX_SKS(2, 1); else UVX_SKS(1, 1); }
    else if (d.M <= 32) { if (two) UVX_SKS(2, 2); else UVX_SKS(1, 2); }
    else { if (two) UVX_SKS(2, 4); else UVX_SKS(1, 4); }
#undef UVX_SKS
    UVX_LAUNCH_CHECK();
    return UVX_OK;
  }
  if (d.swiglu || d.N >= 16384) hipLaunchKernelGGL((gemm_skinny_bf16_k<2, false>), dim3((d.N + 31) / 32), dim3(512), 0, st, a);
  else hipLaunchKernelGGL((gemm_skinny_bf16_k<1, false>), dim3((d.N + 15) / 16), dim3(512), 0, st, a);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

}  // namespace uvx
