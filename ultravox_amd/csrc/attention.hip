// Flash attention (forward + backward) on MFMA 16x16x32 bf16, LDS-staged K / V^T tiles.
//
// Replaces the [3P] SDPA calls under WhisperAttention (non-causal, additive key-padding mask built by
// the reference at ultravox_model.py:915-926, optional block-causal latency mask :834-863,:928-936) and
// LlamaAttention (causal GQA + key-padding mask from the collator, ultravox_processing.py:36).
//
// Layout trick ("swapped" products): scores are computed TRANSPOSED, S^T[key][q] = K . Q^T, so in the
// MFMA accumulator layout (row = 4*(lane>>4)+reg, col = lane&15) every lane owns ONE query row
// (q = lane&15) and 4 consecutive keys per 16-key subtile.  Softmax statistics are then lane-local
// (plus two xor-shuffles across the 4 lane groups), and the probabilities are ALREADY in the B-operand
// layout of the second product O^T[d][q] = V^T[d][key] . P^T[key][q] — no LDS round trip for P.
// The contraction index of an MFMA is free to permute, so B-slot (g, s) is mapped to key
// 32*kp + 16*(s>>2) + 4*g + (s&3) and the V^T A-operand is read with the same map (two 8-byte reads).
// V^T ([B, Hkv, D, Tp], keys contiguous) is produced by a transpose kernel right after the QKV GEMM.
//
// The backward pass is two kernels (dK/dV per key block, dQ per query block), each recomputing the
// probabilities from the saved log-sum-exp; transposed operand copies (Q^T, dO^T, K^T) are made by the
// same transpose kernel so that every MFMA operand is a contiguous 8/16-byte LDS read.
#include "common.h"
#include "kernels.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float NEG_INF = -__builtin_huge_valf();

struct AttnArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; const bf16_t* vt;
  bf16_t* o; float* lse;
  const int32_t* kv_start; const int32_t* kv_len;
  int B, T, Tp, Hq, Hkv;
  int ldq, ldk, ldv, ldo;
  int causal, block, q_begin;
  int window;   // > 0 (with causal): a query sees the last `window` positions only, key > q - window (Gemma-3's sliding-window layers)
  float sc;  // softmax scale * log2(e)
  // backward
  const bf16_t* dout; const bf16_t* qt; const bf16_t* kt; const bf16_t* dot;
  float* delta;
  bf16_t* dq; bf16_t* dk; bf16_t* dv;
  int lddq, lddk, lddv;
  float scale;
  float* dkv_part;  // scratch for [2][B, T, Hq, D] per-query-head results (GQA; stored as bf16) or null
  const float* rope;  // backward: [T_table, D/2, 2] f32 cos / sin - dQ and dK leave the kernels RoPE-INVERTED (the gradient of q, k
                      // before the rotary embedding), position = the row's index in its sequence; null = none
  unsigned long long* tl;  // probes build: per-wave s_memtime stamps of the fused backward kernel (uvx_probe_attn_timeline); else null
  int dfirst;  // fused backward only (AttnBwdDesc::d_first): dout / dq / dk / dv hold the rows of positions >= dfirst, sequence b at row b * (T - dfirst)
};
// Timeline stamps (libuvx_probes.so only): lane 0 of every wave writes slot `s` of its 16-slot record.
#ifdef UVX_PROBES
#define TL_STAMP(s) do { if (p.tl && lane == 0) p.tl[((long long)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + w) * 16 + (s)] = __builtin_readcyclecounter(); } while (0)
#define TL_NOW() (p.tl ? __builtin_readcyclecounter() : 0ULL)
#define TL_PUT(s, v) do { if (p.tl && lane == 0) p.tl[((long long)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + w) * 16 + (s)] = (v); } while (0)
#define TLF_IDX ((((long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + w) * 16
#define TLF_PUT(s, v) do { if (p.tl && lane == 0) p.tl[TLF_IDX + (s)] = (v); } while (0)
#else
#define TLF_PUT(s, v) do {} while (0)
#define TL_STAMP(s) do {} while (0)
#define TL_NOW() 0ULL
#define TL_PUT(s, v) do {} while (0)
#endif

// Inverse rotary embedding of one gradient row held as DT accumulator tiles (element e of tile dt <-> d = 16 dt + 4 g + e):
// the pair (d, d + D/2) sits in tiles dt and dt + DT/2 of the same lane.  Same arithmetic and rounding points as rope_k
// (elementwise.hip) applied to the bf16-rounded gradient: every product rounded to bf16, the sum rounded by the store.
template <int DT>
__device__ __forceinline__ void rope_inverse_tiles(float (&v)[DT][4], const float* __restrict__ cs, int pos, int g) {
  constexpr int HALF = DT / 2;
  const float* t = cs + (long long)pos * (HALF * 16) * 2;
#pragma unroll
  for (int dt = 0; dt < HALF; ++dt) {
    const float4 a = *reinterpret_cast<const float4*>(t + (dt * 16 + 4 * g) * 2);
    const float4 b = *reinterpret_cast<const float4*>(t + (dt * 16 + 4 * g) * 2 + 4);
    const float co[4] = {a.x, a.z, b.x, b.z}, si[4] = {a.y, a.w, b.y, b.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float c = bf2f(f2bf(co[e])), sn = -bf2f(f2bf(si[e]));
      const float lo = v[dt][e], hi = v[dt + HALF][e];
      v[dt][e] = bf2f(f2bf(lo * c)) + bf2f(f2bf(-hi * sn));
      v[dt + HALF][e] = bf2f(f2bf(hi * c)) + bf2f(f2bf(lo * sn));
    }
  }
}

// Reductions over the four lanes that share a query column (lanes l, l ^ 16, l ^ 32, l ^ 48) on the VALU: v_permlane16_swap exchanges
// the odd 16-lane rows of its first operand with the even rows of its second, v_permlane32_swap the upper half of the first with the
// lower half of the second - with both operands = x the two results are x's two halves broadcast, so one max / add of them is the
// pairwise reduction.  __shfl_xor(x, 16 / 32) compiles to ds_bpermute_b32 + s_waitcnt lgkmcnt(0): an LDS round trip that also
// waits for every fragment read in flight (twice per 16 query rows and key tile in the forward loop).  Same values bit for bit:
// max is exact, and a + b == b + a.
// (the two results are copied into scalars before the casts: __builtin_bit_cast applied to an ELEMENT of the returned ext-vector reads
//  element 0 whichever index is written - hipcc 7.2 - which silently turns the max / add into a no-op)
__device__ __forceinline__ float u2f(unsigned u) { return __builtin_bit_cast(float, u); }
// one v_max_f32 (fmaxf on values that come out of a cross-lane op is preceded by two canonicalising v_max x, x, x)
__device__ __forceinline__ float vmax1(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float quad_max(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const unsigned a0 = a[0], a1 = a[1];
  const unsigned w = __builtin_bit_cast(unsigned, vmax1(u2f(a0), u2f(a1)));
  const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  const unsigned b0 = b[0], b1 = b[1];
  return vmax1(u2f(b0), u2f(b1));
}
__device__ __forceinline__ float quad_sum(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const unsigned a0 = a[0], a1 = a[1];
  const unsigned w = __builtin_bit_cast(unsigned, u2f(a0) + u2f(a1));
  const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  const unsigned b0 = b[0], b1 = b[1];
  return u2f(b0) + u2f(b1);
}

__device__ __forceinline__ bf16x8_t lds_b128(const char* p) { return *reinterpret_cast<const bf16x8_t*>(p); }

// two 8-byte LDS reads -> one 8 x bf16 fragment
__device__ __forceinline__ bf16x8_t lds_2xb64(const char* p0, const char* p1) {
  typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
  bf16x4_t a = *reinterpret_cast<const bf16x4_t*>(p0);
  bf16x4_t b = *reinterpret_cast<const bf16x4_t*>(p1);
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ bf16x8_t pack8(const float* lo, const float* hi) {
  u16x8_t r;
#pragma unroll
  for (int i = 0; i < 4; ++i) { r[i] = f2bf(lo[i]); r[4 + i] = f2bf(hi[i]); }
  return __builtin_bit_cast(bf16x8_t, r);
}

// (bitwise, not short-circuit: hipcc then emits compares + selects instead of a chain of exec-mask branches per element;
// the latency-block test - two integer divisions - sits behind a wave-uniform branch)
__device__ __forceinline__ bool key_ok(int key, int q, int k_lo, int k_hi, int causal, int block, int window = 0) {
  bool ok = (key >= k_lo) & (key < k_hi) & ((causal == 0) | (key <= q)) & ((window <= 0) | (key > q - window));
  if (block > 0) ok = ok & ((key / block) <= (q / block));
  return ok;
}

// "natural" tile: R rows x D cols bf16, 16-byte chunks XOR-swizzled with the row index.
template <int D>
__device__ __forceinline__ int nat_off(int row, int chunk) {
  constexpr int NCH = D / 8;
  return row * (D * 2) + ((chunk ^ (row & (NCH - 1))) << 4);
}
// "transposed" tile: D rows of W keys (W = 64 or 32) = W*2 bytes per row, addressed in 8-byte chunks c8 and
// XOR-swizzled with tr_sw(row).  The swizzle is chosen to be conflict-free for BOTH forms hipcc may emit
// for the paired 8-byte fragment reads: plain ds_read_b64 (2 x 32 lanes, bank = (addr/4) mod 64) and the
// merged ds_read2st64_b64 (contiguous 16-lane groups, bank = (addr/4) mod 32).  An odd swizzle swaps the two
// 8-byte halves of a 16-byte chunk, which the 16-byte staging store mirrors in registers.
template <int W>
__device__ __forceinline__ int tr_sw(int row) { return W == 64 ? (row & 15) : ((row >> 1) & 7); }
template <int W>
__device__ __forceinline__ int tr_off8(int row, int c8) {
  constexpr int M8 = W / 4 - 1;  // chunk index mask
  return row * (W * 2) + (((c8 ^ tr_sw<W>(row)) & M8) << 3);
}
// A-operand fragment [row = d0 + (lane & 15)][k-slot (g, s)] of the TRANSPOSE of a natural tile (rows = contraction index r,
// columns = d), with the permuted contraction order used throughout this file: slot (g, s) <-> r = r0 + 16*(s>>2) + 4*g + (s&3).
// gfx950's transposing LDS read (ds_read_b64_tr_b16) makes the transposed global copies (V^T, Q^T, K^T, dO^T: four HBM round
// trips per layer) unnecessary: every lane supplies the address of ONE 8-byte chunk (4 consecutive d of one row), and within
// each group of 16 lanes result element j of lane i is element (i & 3) of the chunk supplied by lane 4*j + (i >> 2).  So
// lane i points at row r0 + 4*g + (i >> 2), columns d0 + 4*(i & 3) ..+3, and receives rows r0 + 4*g + j (j = 0..3) of column
// d0 + i; a second read 16 rows further down completes the 8-slot fragment.  (Semantics pinned on the GPU by
// tests/test_kernels_gpu.py::test_lds_transpose_read_semantics through uvx_probe_lds_tr.)
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
__device__ __forceinline__ s16x4_t lds_tr_b64(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
}
template <int D>
__device__ __forceinline__ bf16x8_t tr_frag(const char* tile, int r0, int d0, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const int row = r0 + 4 * g + (i >> 2), col = d0 + 4 * (i & 3);
  const char* p0 = tile + nat_off<D>(row, col >> 3) + ((col & 4) << 1);
  const char* p1 = tile + nat_off<D>(row + 16, col >> 3) + ((col & 4) << 1);
  const s16x4_t a = lds_tr_b64(p0), b = lds_tr_b64(p1);
  return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

// Register-staged tiles (async-STAGE split): the global loads of tile j+1 are ISSUED before the MFMA work
// of tile j and written to the other LDS buffer after it, so HBM/L2 latency hides under compute and
// there is one barrier per tile.
// natural tile: rows [r0, r0+R) of a [*, ld] matrix, D columns (rows clamped to rmax-1).
template <int D, int R, int NT = 256>
struct NatRegs { u16x8_t v[R * (D / 8) / NT]; };
template <int D, int R, int NT = 256>
__device__ __forceinline__ void load_nat(NatRegs<D, R, NT>& g, const bf16_t* base, long long ld, int r0, int rmax, int tid) {
  constexpr int NCH = D / 8, N = R * NCH / NT;
  static_assert(N >= 1 && N * NT == R * NCH, "tile does not divide over the block's threads");
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const int i = tid + k * NT, r = i / NCH, c = i % NCH;
    // UNCONDITIONAL load of a clamped row: a fixed number of VMEM ops per iteration lets hipcc place counted
    // s_waitcnt vmcnt(N) instead of draining the prefetch with vmcnt(0); rows >= rmax repeat row rmax-1 and are
    // always masked out by the callers (they are finite, so 0 * x stays 0)
    g.v[k] = *reinterpret_cast<const u16x8_t*>(base + (long long)min(r0 + r, rmax - 1) * ld + c * 8);
  }
}
// the same with rows clamped to [rmin, rmax - 1]: a matrix whose rows below rmin do not exist (the fused backward's row-compacted dO)
template <int D, int R, int NT = 256>
__device__ __forceinline__ void load_nat_from(NatRegs<D, R, NT>& g, const bf16_t* base, long long ld, int r0, int rmin, int rmax, int tid) {
  constexpr int NCH = D / 8, N = R * NCH / NT;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const int i = tid + k * NT, r = i / NCH, c = i % NCH;
    g.v[k] = *reinterpret_cast<const u16x8_t*>(base + (long long)max(min(r0 + r, rmax - 1), rmin) * ld + c * 8);
  }
}
template <int D, int R, int NT = 256>
__device__ __forceinline__ void store_nat(char* lds, const NatRegs<D, R, NT>& g, int tid) {
  constexpr int NCH = D / 8, N = R * NCH / NT;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const int i = tid + k * NT, r = i / NCH, c = i % NCH;
    *reinterpret_cast<u16x8_t*>(lds + nat_off<D>(r, c)) = g.v[k];
  }
}
// transposed tile: D rows, W keys starting at t0 of a [D, Tp] matrix (zero padded in memory).
template <int D, int W, int NT = 256>
struct TrRegs { u16x8_t v[D * (W / 8) / NT]; };
template <int D, int W, int NT = 256>
__device__ __forceinline__ void load_tr(TrRegs<D, W, NT>& g, const bf16_t* base, int Tp, int t0, int tid) {
  constexpr int NC = W / 8, N = D * NC / NT;
  static_assert(N >= 1 && N * NT == D * NC, "tile does not divide over the block's threads");
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const int i = tid + k * NT, r = i / NC, c = i % NC;
    g.v[k] = *reinterpret_cast<const u16x8_t*>(base + (long long)r * Tp + t0 + c * 8);
  }
}
template <int D, int W, int NT = 256>
__device__ __forceinline__ void store_tr(char* lds, const TrRegs<D, W, NT>& g, int tid) {
  constexpr int NC = W / 8, N = D * NC / NT;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const int i = tid + k * NT, r = i / NC, c = i % NC;
    const int sw = tr_sw<W>(r);
    u16x8_t v = g.v[k];
    if (sw & 1) v = __builtin_shufflevector(v, v, 4, 5, 6, 7, 0, 1, 2, 3);
    *reinterpret_cast<u16x8_t*>(lds + r * (W * 2) + (((c ^ (sw >> 1)) & (NC - 1)) << 4)) = v;
  }
}

// Stores of DT accumulator tiles held "transposed" (element e of tile dt <-> column 16 dt + 4 g + e of row fr = lane & 15).
// Written as they are, every store instruction touches 16 rows with 32 contiguous bytes each - partial lines, measured at ~5
// cycles per 32-byte piece in the fused backward kernel (a quarter of its run time went into issuing its dK / dV / dQ stores,
// profiles/r03_attn_timeline.txt).  store_rows_staged passes them through a 16 x (64 + 8)-column bf16 LDS tile of the wave, 64
// columns at a time, so that a row leaves as 128 contiguous bytes from four lanes: head_dim 128 kernels 7 - 15 % faster at the
// LLM's shape; at head_dim 64 (128-byte rows) it is neutral (forward) to 4 % slower (backward pair), so store_rows keeps the direct
// form there (profiles/r03_attn_staged_stores_ab.txt).  rows r >= rows_valid are not written.
constexpr int STAGE_ROW = 144, STAGE_BYTES = 16 * STAGE_ROW;
template <int DT>
__device__ __forceinline__ void store_rows_staged(char* stage, const u16x4_t (&v)[DT], bf16_t* row0, long long ld, int rows_valid, int lane) {
  static_assert(DT % 4 == 0, "64 columns per round");
  const int fr = lane & 15, g = lane >> 4, r = lane >> 2, seg = lane & 3;
#pragma unroll
  for (int h = 0; h < DT / 4; ++h) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<u16x4_t*>(stage + fr * STAGE_ROW + (dt * 16 + g * 4) * 2) = v[h * 4 + dt];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const u16x8_t a = *reinterpret_cast<const u16x8_t*>(stage + r * STAGE_ROW + seg * 32);
    const u16x8_t b = *reinterpret_cast<const u16x8_t*>(stage + r * STAGE_ROW + seg * 32 + 16);
    if (r < rows_valid) {
      *reinterpret_cast<u16x8_t*>(row0 + (long long)r * ld + h * 64 + seg * 16) = a;
      *reinterpret_cast<u16x8_t*>(row0 + (long long)r * ld + h * 64 + seg * 16 + 8) = b;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}
template <int DT>
__device__ __forceinline__ void store_rows(char* stage, const u16x4_t (&v)[DT], bf16_t* row0, long long ld, int rows_valid, int lane) {
  if constexpr (DT >= 8) {
    store_rows_staged<DT>(stage, v, row0, ld, rows_valid, lane);
  } else {
    const int fr = lane & 15, g = lane >> 4;
    if (fr < rows_valid) {
#pragma unroll
      for (int d = 0; d < DT; ++d) *reinterpret_cast<u16x4_t*>(row0 + (long long)fr * ld + d * 16 + g * 4) = v[d];
    }
  }
}

// =================================== forward ===================================
// TR: V is staged in its natural [key][d] layout and read through the transposing LDS read (no V^T copy in global memory)
// PL: the cross-lane max / sum of a query row through v_permlane*_swap (quad_max) instead of two ds_bpermute shuffles
// (tried, round 6: the tile's V fragments read into registers BEFORE the softmax - 184 registers, occupancy 2: 7 % slower, profiles/r06_attn_fwd64_ab.txt)
// (tried, round 6: s_setprio 1 around the two MFMA clusters of a tile - 111.6 / 114.9 us against 112.4 / 113.6: neutral, same file)
// (tried, round 6: the softmax row sums from a fifth "d-tile" of ones in the P.V product instead of 16 v_add_f32 per 16 scores - 111.3 / 113.0 us
//  against 113.9 / 113.4: within noise, and the normaliser then sums bf16-rounded probabilities (max |d O| 2e-3); not kept)
// GQ (round 6, short causal sequences under grouped-query attention): the block's four waves take the SAME QT query tiles of FOUR query heads of one KV
// head instead of four query ranges of one head - the K / V tiles a block stages serve four heads (a quarter of the L2 -> LDS traffic and of the blocks'
// prologues per head), the grid is (Hq / 4, B, T / (16 QT)).  Per (head, query row) the key tiles arrive in the same order: bit-identical results.
template <int D, int QT, bool TR, bool PL = true, bool GQ = false>
__global__ __launch_bounds__(256) void attn_fwd_k(AttnArgs p) {
  constexpr int BQ = (GQ ? 1 : 4) * QT * 16;
  constexpr int KS = D / 32;   // k-steps of the QK^T product
  constexpr int DT = D / 16;   // 16-row tiles of O^T
  constexpr int TILE = 64 * D * 2;  // bytes of one K tile == one V^T tile
  __shared__ __attribute__((aligned(16))) char ldsKV[4 * TILE];  // [buf][K | V^T]

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  // grid = (heads, batch, query blocks): the query block is the slowest index and, under a causal mask, the LAST block -
  // the one with the most key tiles - is dispatched first (longest-first: the short blocks fill the tail of the launch)
  const int b = blockIdx.y, h = GQ ? (int)blockIdx.x * 4 + w : (int)blockIdx.x, hk = h / (p.Hq / p.Hkv);      // (GQ: Hq / Hkv is a multiple of 4 - hk is block-uniform)
  const int qb0 = (p.causal ? (int)(gridDim.z - 1 - blockIdx.z) : (int)blockIdx.z) * BQ;
  if (qb0 + BQ <= p.q_begin) return;   // chunked prefill: these query rows belong to the cached prefix (block-uniform exit)
  const int q0 = GQ ? qb0 : qb0 + w * QT * 16;
  const int k_lo = p.kv_start ? p.kv_start[b] : 0;
  const int k_hi = p.kv_len ? min(p.kv_len[b], p.T) : p.T;

  const unsigned long long tlf0 = TL_NOW();
  unsigned long long tl_s = 0, tl_sm = 0, tl_pv = 0, tl_cm = 0, tl_it = 0;
  (void)tlf0; (void)tl_s; (void)tl_sm; (void)tl_pv; (void)tl_cm; (void)tl_it;
  // Q fragments (B operand): Q[q = fr][d = ks*32 + g*8 ..]
  bf16x8_t qf[QT][KS];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int q = q0 + t * 16 + fr;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      u16x8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (q < p.T) v = *reinterpret_cast<const u16x8_t*>(p.q + ((long long)b * p.T + q) * p.ldq + h * D + ks * 32 + g * 8);
      qf[t][ks] = __builtin_bit_cast(bf16x8_t, v);
    }
  }

  f32x4_t acc_o[QT][DT];
  float m_run[QT], l_run[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    m_run[t] = NEG_INF; l_run[t] = 0.f;
#pragma unroll
    for (int d = 0; d < DT; ++d) acc_o[t][d] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }

  int kb_end = p.T;
  if (p.causal) kb_end = min(kb_end, qb0 + BQ);
  if (p.block > 0) kb_end = min(kb_end, ((qb0 + BQ - 1) / p.block + 1) * p.block);
  kb_end = min(kb_end, k_hi);
  const bf16_t* kbase = p.k + (long long)b * p.T * p.ldk + hk * D;
  const bf16_t* vtbase = TR ? nullptr : p.vt + ((long long)b * p.Hkv + hk) * D * p.Tp;
  const bf16_t* vbase = TR ? p.v + (long long)b * p.T * p.ldv + hk * D : nullptr;

  NatRegs<D, 64> kreg;
  TrRegs<D, 64> vreg;      // (!TR)
  NatRegs<D, 64> vnreg;    // (TR)
  // (sliding window: the block's first query sees nothing before qb0 - window + 1)
  const int kb_begin = (max(k_lo, p.window > 0 ? qb0 - p.window + 1 : 0) / 64) * 64;
  if (kb_begin < kb_end) {
    load_nat<D, 64>(kreg, kbase, p.ldk, kb_begin, p.T, tid);
    store_nat<D, 64>(ldsKV, kreg, tid);
    if constexpr (TR) {
      load_nat<D, 64>(vnreg, vbase, p.ldv, kb_begin, p.T, tid);
      store_nat<D, 64>(ldsKV + TILE, vnreg, tid);
    } else {
      load_tr<D, 64>(vreg, vtbase, p.Tp, kb_begin, tid);
      store_tr<D, 64>(ldsKV + TILE, vreg, tid);
    }
  }
  __syncthreads();
  const unsigned long long tlf1 = TL_NOW();
  (void)tlf1;
  int cur = 0;
  for (int kb = kb_begin; kb < kb_end; kb += 64, cur ^= 1) {
    const unsigned long long tla = TL_NOW();
    const char* ldsK = ldsKV + cur * 2 * TILE;
    const char* ldsV = ldsK + TILE;
    // issue the next tile's global loads now; they land while this tile computes (the last iteration re-loads
    // its own tile: branch-free, so the VMEM count per iteration is static)
    const int kn = kb + 64 < kb_end ? kb + 64 : kb;
    load_nat<D, 64>(kreg, kbase, p.ldk, kn, p.T, tid);
    if constexpr (TR) load_nat<D, 64>(vnreg, vbase, p.ldv, kn, p.T, tid);
    else load_tr<D, 64>(vreg, vtbase, p.Tp, kn, tid);

    // ---- S^T = K . Q^T ----
    f32x4_t s[QT][4];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) s[t][kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8_t kf = lds_b128(ldsK + nat_off<D>(kt * 16 + fr, ks * 4 + g));
#pragma unroll
        for (int t = 0; t < QT; ++t) s[t][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[t][ks], s[t][kt], 0, 0, 0);
      }

    const unsigned long long tlb = TL_NOW();
    // ---- online softmax (per lane: q = fr; keys kb + kt*16 + g*4 + e) ----
    // A tile needs per-element masking only at the edges (padding, diagonal, latency-block boundary); the
    // test is wave-uniform, so interior tiles skip all of the integer mask arithmetic.
    bool need_mask = kb < k_lo || kb + 64 > k_hi;
    if (p.causal) need_mask = need_mask || kb + 63 > q0;
    if (p.block > 0) need_mask = need_mask || (kb + 63) / p.block > q0 / p.block;
    if (p.window > 0) need_mask = need_mask || kb <= q0 + QT * 16 - 1 - p.window;    // the tile's first key lies outside some row's window
    bf16x8_t pf[QT][2];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      const int q = q0 + t * 16 + fr;
      float mx = NEG_INF;
      // scores stay RAW (unscaled) in s[][]; sc > 0, so the row max commutes with the scale and the scale
      // is folded into the exp2 argument as one fma per element
      if (need_mask) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int key = kb + kt * 16 + g * 4 + e;
            const float v = key_ok(key, q, k_lo, k_hi, p.causal, p.block, p.window) ? s[t][kt][e] : NEG_INF;
            s[t][kt][e] = v;
            mx = fmaxf(mx, v);
          }
      } else {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int e = 0; e < 4; ++e) mx = fmaxf(mx, s[t][kt][e]);
      }
      if constexpr (PL) mx = quad_max(mx);
      else { mx = fmaxf(mx, __shfl_xor(mx, 16, 64)); mx = fmaxf(mx, __shfl_xor(mx, 32, 64)); }
      mx *= p.sc;  // (-inf stays -inf)
      const float m_new = fmaxf(m_run[t], mx);
      const float m_use = (m_new == NEG_INF) ? 0.f : m_new;
      const bool changed = m_new != m_run[t];
      const float alpha = __builtin_amdgcn_exp2f(m_run[t] - m_use);
      m_run[t] = m_new;
      float ps = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float pv = __builtin_amdgcn_exp2f(fmaf(s[t][kt][e], p.sc, -m_use));  // raw v_exp_f32: args <= 0
          s[t][kt][e] = pv;
          ps += pv;
        }
      // (the two-wide forms of this scale + row sum - v_pk_fma_f32 / v_pk_add_f32, 29 VALU instructions fewer per 64-key tile - were
      //  measured neutral on the encoder shape: profiles/r03_attn_fwd_timeline.txt)
      l_run[t] = l_run[t] * alpha + ps;
      if (__any(changed)) {  // wave-uniform: once the running max has settled the O rescale is skipped (alpha == 1 exactly)
#pragma unroll
        for (int d = 0; d < DT; ++d) acc_o[t][d] *= alpha;
      }
      float lo[4], hi[4];
#pragma unroll
      for (int kp = 0; kp < 2; ++kp) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { lo[e] = s[t][2 * kp][e]; hi[e] = s[t][2 * kp + 1][e]; }
        pf[t][kp] = pack8(lo, hi);
      }
    }

    const unsigned long long tlc = TL_NOW();
    // ---- O^T += V^T . P^T ----
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int kp = 0; kp < 2; ++kp) {
        const int row = d * 16 + fr;
        bf16x8_t vf;
        if constexpr (TR) vf = tr_frag<D>(ldsV, kp * 32, d * 16, lane);
        else vf = lds_2xb64(ldsV + tr_off8<64>(row, kp * 8 + g), ldsV + tr_off8<64>(row, kp * 8 + 4 + g));
#pragma unroll
        for (int t = 0; t < QT; ++t) acc_o[t][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[t][kp], acc_o[t][d], 0, 0, 0);
      }
    const unsigned long long tld = TL_NOW();
    // the other buffer was last read one iteration ago, before the previous barrier
    store_nat<D, 64>(ldsKV + (cur ^ 1) * 2 * TILE, kreg, tid);
    if constexpr (TR) store_nat<D, 64>(ldsKV + (cur ^ 1) * 2 * TILE + TILE, vnreg, tid);
    else store_tr<D, 64>(ldsKV + (cur ^ 1) * 2 * TILE + TILE, vreg, tid);
    __syncthreads();
    const unsigned long long tle = TL_NOW();
    tl_s += tlb - tla; tl_sm += tlc - tlb; tl_pv += tld - tlc; tl_cm += tle - tld; tl_it += 1;
  }
  const unsigned long long tlf2 = TL_NOW();
  (void)tlf2;

  // ---- epilogue ----
  // (after the loop's last barrier nobody reads the K / V tiles any more: each wave stages its O rows through a private piece
  //  of them and stores whole 128-byte row segments - store_rows_staged)
  char* stage = ldsKV + w * STAGE_BYTES;
  static_assert(4 * STAGE_BYTES <= 4 * TILE, "stage fits the tile buffers");
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int q = q0 + t * 16 + fr;
    float l = l_run[t];
    if constexpr (PL) l = quad_sum(l);
    else { l += __shfl_xor(l, 16, 64); l += __shfl_xor(l, 32, 64); }
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    u16x4_t o4[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int e = 0; e < 4; ++e) o4[d][e] = f2bf(acc_o[t][d][e] * inv);
    if (q0 + t * 16 < p.T)   // wave-uniform
      store_rows<DT>(stage, o4, p.o + ((long long)b * p.T + q0 + t * 16) * p.ldo + h * D, p.ldo, p.T - (q0 + t * 16), lane);
    if (p.lse && g == 0 && q < p.T) p.lse[((long long)b * p.Hq + h) * p.T + q] = l > 0.f ? m_run[t] + log2f(l) : __builtin_huge_valf();
  }
  TLF_PUT(0, tlf0); TLF_PUT(1, tlf1); TLF_PUT(2, tlf2); TLF_PUT(3, TL_NOW());
  TLF_PUT(4, tl_s); TLF_PUT(5, tl_sm); TLF_PUT(6, tl_pv); TLF_PUT(7, tl_cm); TLF_PUT(8, tl_it);
}

// =================================== backward: dK, dV ===================================
// Block = (key block of NT/4, kv head, batch): NT/64 waves, wave w owns keys kb0 + w*16 .. +16.  Loops over the query
// heads of the GQA group and over ST-query steps (ST = 32 or 64).  NT = 512 (head_dim 128: the LLM): every staged Q / dO
// tile feeds 128 keys instead of 64.  ST = 64 halves the number of steps: a step is a chain of dependent latencies (LDS
// reads -> 16 MFMAs -> exp / pack -> LDS reads -> 16 MFMAs -> staging stores -> barrier) that two waves per SIMD do not
// hide (PMC, round 2: 8 000 wave cycles per 32-query step for ~400 instructions), so the fixed part is paid half as often.
// DEEP: two-deep register prefetch (two staging sets, alternating) instead of one step ahead.
// TR: only the natural Q / dO tiles are staged; their transposes come from the transposing LDS read (no Q^T / dO^T copies).
template <int D, int NT, int ST, bool DEEP, bool TR, int WT = 1>
__global__ __launch_bounds__(NT, 1) void attn_bwd_dkdv_k(AttnArgs p) {
  constexpr int KB = NT / 4 * WT;   // keys per block
  constexpr int KS = D / 32, DT = D / 16, QT = ST / 16, KP = ST / 32;
  constexpr int TILE = ST * D * 2;
  constexpr int NTILE = TR ? 2 : 4;   // tiles per buffer
  __shared__ __attribute__((aligned(16))) char ldsAll[2 * NTILE * TILE + 4 * ST * 4];  // [buf][Q | dO (| Q^T | dO^T)], then [buf][lse | delta]
  float* ldsStat = reinterpret_cast<float*>(ldsAll + 2 * NTILE * TILE);

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  // With `dkv_part` every block handles ONE query head (grid.x = Hq: 4x the parallelism under GQA) and
  // writes f32 partials that gqa_reduce_k sums in a fixed order; without it the block loops over its group.
  const int grp_all = p.Hq / p.Hkv;
  const bool split = p.dkv_part != nullptr;
  // grid = (heads, batch, key blocks): the key block is the SLOWEST index, so under a causal mask the blocks with the most
  // query steps (key block 0 sees every query) are dispatched first and the short ones fill the tail (longest-first; with
  // the key block fastest a CU could draw two of the longest blocks and the launch ran 28 steps deep instead of 18)
  const int b = blockIdx.y, kb0 = blockIdx.z * KB;
  const int hk = split ? blockIdx.x / grp_all : blockIdx.x;
  const int h_first = split ? blockIdx.x : hk * grp_all;
  const int grp = split ? 1 : grp_all;
  const int k_lo = p.kv_start ? p.kv_start[b] : 0;
  const int k_hi = p.kv_len ? min(p.kv_len[b], p.T) : p.T;
  const int key_w0 = kb0 + w * 16 * WT;   // this wave's 16 WT keys: tile wt holds key_w0 + 16 wt + (lane & 15) (B-operand column)

  // K, V fragments for this wave's WT 16-key tiles: B operand, [key = fr][d = ks*32 + g*8 ..]
  bf16x8_t kf[WT][KS], vf[WT][KS];
#pragma unroll
  for (int wt = 0; wt < WT; ++wt)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int key = key_w0 + wt * 16 + fr;
      u16x8_t a = {0, 0, 0, 0, 0, 0, 0, 0}, c = {0, 0, 0, 0, 0, 0, 0, 0};
      if (key < p.T) {
        a = *reinterpret_cast<const u16x8_t*>(p.k + ((long long)b * p.T + key) * p.ldk + hk * D + ks * 32 + g * 8);
        c = *reinterpret_cast<const u16x8_t*>(p.v + ((long long)b * p.T + key) * p.ldv + hk * D + ks * 32 + g * 8);
      }
      kf[wt][ks] = __builtin_bit_cast(bf16x8_t, a);
      vf[wt][ks] = __builtin_bit_cast(bf16x8_t, c);
    }
  f32x4_t acc_dk[WT][DT], acc_dv[WT][DT];
#pragma unroll
  for (int wt = 0; wt < WT; ++wt)
#pragma unroll
    for (int d = 0; d < DT; ++d) { acc_dk[wt][d] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc_dv[wt][d] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }

  int q_begin = 0;
  if (p.causal) q_begin = kb0;
  if (p.block > 0) q_begin = max(q_begin, (kb0 / p.block) * p.block);
  q_begin = (q_begin / ST) * ST;
  // (sliding window: the block's last key is seen by queries up to kb0 + KB - 1 + window - 1)
  const int q_stop = p.window > 0 ? min(p.T, kb0 + KB - 1 + p.window) : p.T;

  // flattened (head-in-group, ST-query step) iteration space, software pipelined: the global loads of a later step are
  // issued before the MFMA work of the current one and written to the other LDS buffer after it (one barrier per step)
  const int nq = q_begin < q_stop ? (q_stop - q_begin + ST - 1) / ST : 0;
  const int n_it = grp * nq;
  struct StepRegs { NatRegs<D, ST, NT> q, d_o; TrRegs<D, TR ? 8 * NT / D : ST, NT> qt, dot; float l, dl; };   // (TR: qt / dot unused, minimal)
  StepRegs r0, r1;
  // (head, query step) cursors advance by increments: an integer division per step is ~25 scalar instructions
  const int q_last = q_begin + (nq - 1) * ST, h_last = h_first + grp - 1;
  int ih = h_first, iqs = q_begin;     // next step to load
  int cqs = q_begin;                   // step being computed
  auto issue = [&](StepRegs& r) {
    const int h = ih, qs = iqs;
    // advance; past the end: stay on the last step (re-loaded branch-free, static VMEM count; never committed to a live buffer)
    if (iqs < q_last) iqs += ST;
    else if (ih < h_last) { iqs = q_begin; ++ih; }
    load_nat<D, ST, NT>(r.q, p.q + (long long)b * p.T * p.ldq + h * D, p.ldq, qs, p.T, tid);
    load_nat_from<D, ST, NT>(r.d_o, p.dout + ((long long)b * (p.T - p.dfirst) - p.dfirst) * p.ldo + h * D, p.ldo, qs, p.dfirst, p.T, tid);      // (dfirst: see attn_bwd_fused_k)
    if constexpr (!TR) {
      load_tr<D, ST, NT>(r.qt, p.qt + ((long long)b * p.Hq + h) * D * p.Tp, p.Tp, qs, tid);
      load_tr<D, ST, NT>(r.dot, p.dot + ((long long)b * p.Hq + h) * D * p.Tp, p.Tp, qs, tid);
    }
    {  // every thread loads (clamped index, branch-free: keeps the per-iteration VMEM count static);
       // queries >= T are masked by the consumers
      const int q = min(qs + (tid & (ST - 1)), p.T - 1);
      r.l = p.lse[((long long)b * p.Hq + h) * p.T + q];
      r.dl = p.delta[((long long)b * p.Hq + h) * p.T + q];
    }
  };
  auto commit = [&](const StepRegs& r, int buf) {
    char* base = ldsAll + buf * NTILE * TILE;
    store_nat<D, ST, NT>(base, r.q, tid);
    store_nat<D, ST, NT>(base + TILE, r.d_o, tid);
    if constexpr (!TR) {
      store_tr<D, ST, NT>(base + 2 * TILE, r.qt, tid);
      store_tr<D, ST, NT>(base + 3 * TILE, r.dot, tid);
    }
    if (tid < ST) { ldsStat[buf * 2 * ST + tid] = r.l; ldsStat[buf * 2 * ST + ST + tid] = r.dl; }
  };
  auto compute = [&](int cur) {
    const int qs = cqs;
    cqs = cqs < q_last ? cqs + ST : q_begin;
    constexpr int KW = 16 * WT;       // this wave's keys: key_w0 .. key_w0 + KW - 1
    // nothing to add when every (query, key) pair of this wave's tiles is masked: its keys lie after the step's last query
    // (the first steps of a causal block) or outside the valid key range
    if ((p.causal && key_w0 > qs + ST - 1) || key_w0 >= k_hi || key_w0 + KW <= k_lo || (p.window > 0 && key_w0 + KW - 1 <= qs - p.window)) return;
    const char* ldsQ = ldsAll + cur * NTILE * TILE;
    const char* ldsDO = ldsQ + TILE;
    const char* ldsQT = ldsQ + 2 * TILE;      // (!TR)
    const char* ldsDOT = ldsQ + 3 * TILE;     // (!TR)
    const float* ldsL = ldsStat + cur * 2 * ST;
    const float* ldsDl = ldsL + ST;
    // S[q][key] and dP[q][key] for the QT 16-query tiles x WT 16-key tiles: A = Q / dO rows (read ONCE for all WT key tiles),
    // B = K / V fragments
    f32x4_t s[WT][QT], dp[WT][QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
#pragma unroll
      for (int wt = 0; wt < WT; ++wt) { s[wt][t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dp[wt][t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8_t a = lds_b128(ldsQ + nat_off<D>(t * 16 + fr, ks * 4 + g));
        const bf16x8_t c = lds_b128(ldsDO + nat_off<D>(t * 16 + fr, ks * 4 + g));
#pragma unroll
        for (int wt = 0; wt < WT; ++wt) {
          s[wt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, kf[wt][ks], s[wt][t], 0, 0, 0);
          dp[wt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c, vf[wt][ks], dp[wt][t], 0, 0, 0);
        }
      }
    }
    // accumulator element e of tile (wt, t): q = qs + t*16 + g*4 + e, key = key_w0 + 16 wt + fr
    float pr[WT][QT][4], ds[WT][QT][4];
    bool need_mask = key_w0 < k_lo || key_w0 + KW > k_hi || qs + ST > p.T;
    if (p.causal) need_mask = need_mask || key_w0 + KW - 1 > qs;
    if (p.block > 0) need_mask = need_mask || (key_w0 + KW - 1) / p.block > qs / p.block;
    if (p.window > 0) need_mask = need_mask || key_w0 <= qs + ST - 1 - p.window;
    unsigned okbits[WT];              // bit t*4 + e: the (query, key) pair of that accumulator element takes part
#pragma unroll
    for (int wt = 0; wt < WT; ++wt) okbits[wt] = 0xffffu;
    if (need_mask) {
#pragma unroll
      for (int wt = 0; wt < WT; ++wt) {
        const int key = key_w0 + wt * 16 + fr;
        okbits[wt] = 0u;
#pragma unroll
        for (int t = 0; t < QT; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int q = qs + t * 16 + g * 4 + e;
            okbits[wt] |= (unsigned)((q < p.T) & key_ok(key, q, k_lo, k_hi, p.causal, p.block, p.window)) << (t * 4 + e);
          }
      }
    }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      const float4 l4 = *reinterpret_cast<const float4*>(ldsL + t * 16 + g * 4);      // this lane's four queries of the tile
      const float4 d4 = *reinterpret_cast<const float4*>(ldsDl + t * 16 + g * 4);
      const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int wt = 0; wt < WT; ++wt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float ev = __builtin_amdgcn_exp2f(s[wt][t][e] * p.sc - lq[e]);
          const float pv = (okbits[wt] >> (t * 4 + e)) & 1u ? ev : 0.f;
          pr[wt][t][e] = pv;
          ds[wt][t][e] = pv * (dp[wt][t][e] - dq[e]);
        }
    }
    // B operands per 32-query chunk kp: slot (g, s) <-> q = qs + 32*kp + 16*(s>>2) + 4*g + (s&3)
    bf16x8_t pB[WT][KP], dsB[WT][KP];
#pragma unroll
    for (int wt = 0; wt < WT; ++wt)
#pragma unroll
      for (int kp = 0; kp < KP; ++kp) { pB[wt][kp] = pack8(pr[wt][2 * kp], pr[wt][2 * kp + 1]); dsB[wt][kp] = pack8(ds[wt][2 * kp], ds[wt][2 * kp + 1]); }
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      const int row = d * 16 + fr;
#pragma unroll
      for (int kp = 0; kp < KP; ++kp) {
        bf16x8_t a, c;
        if constexpr (TR) {
          a = tr_frag<D>(ldsDO, kp * 32, d * 16, lane);
          c = tr_frag<D>(ldsQ, kp * 32, d * 16, lane);
        } else {
          a = lds_2xb64(ldsDOT + tr_off8<ST>(row, kp * 8 + g), ldsDOT + tr_off8<ST>(row, kp * 8 + 4 + g));
          c = lds_2xb64(ldsQT + tr_off8<ST>(row, kp * 8 + g), ldsQT + tr_off8<ST>(row, kp * 8 + 4 + g));
        }
#pragma unroll
        for (int wt = 0; wt < WT; ++wt) {
          acc_dv[wt][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pB[wt][kp], acc_dv[wt][d], 0, 0, 0);
          acc_dk[wt][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c, dsB[wt][kp], acc_dk[wt][d], 0, 0, 0);
        }
      }
    }
  };
  if (n_it > 0) {
    issue(r0);
    commit(r0, 0);
    if (DEEP) issue(r1);
  }
  __syncthreads();
  if (DEEP) {
    for (int it = 0; it < n_it; it += 2) {
      issue(r0);                   // buffer 0 holds step it; r1 holds step it + 1 (in flight); r0 <- step it + 2
      compute(0);
      commit(r1, 1);
      __syncthreads();
      if (it + 1 >= n_it) break;
      issue(r1);                   // buffer 1 holds step it + 1; r0 holds step it + 2 (in flight); r1 <- step it + 3
      compute(1);
      commit(r0, 0);
      __syncthreads();
    }
  } else {                         // one register set, one step ahead
    for (int it = 0; it < n_it; ++it) {
      issue(r0);
      compute(it & 1);
      commit(r0, (it & 1) ^ 1);
      __syncthreads();
    }
  }
  // accumulators hold dV^T / dK^T: row d = dt*16 + g*4 + e, col key = fr.  After the loop's last barrier the staging buffers are
  // free: every wave passes its rows through a private piece of them and stores whole 128-byte segments (store_rows_staged).
  char* stage = ldsAll + w * STAGE_BYTES;
  static_assert((NT / 64) * STAGE_BYTES <= 2 * NTILE * TILE, "stage fits the staging buffers");
#pragma unroll
  for (int wt = 0; wt < WT; ++wt) {
    const int key0 = key_w0 + wt * 16, key = key0 + fr;
    if (key0 >= p.T || key0 < p.dfirst) continue;    // wave-uniform (dfirst: row-compacted gradients - positions below it are not stored)
    const long long drow0 = (long long)b * (p.T - p.dfirst) - p.dfirst;
    u16x4_t ok[DT], ov[DT];
    if (split) {
      // per-query-head dK / dV, rounded to bf16 like the un-grouped result below: that is where the reference rounds too (SDPA
      // returns bf16 gradients for the repeat_kv-expanded heads and autograd sums the group afterwards) - and half the bytes
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int e = 0; e < 4; ++e) { ok[d][e] = f2bf(acc_dk[wt][d][e] * p.scale); ov[d][e] = f2bf(acc_dv[wt][d][e]); }
      bf16_t* dkp = reinterpret_cast<bf16_t*>(p.dkv_part) + (((long long)b * p.T + key0) * p.Hq + h_first) * D;
      bf16_t* dvp = dkp + (long long)p.B * p.T * p.Hq * D;
      store_rows<DT>(stage, ok, dkp, (long long)p.Hq * D, p.T - key0, lane);
      store_rows<DT>(stage, ov, dvp, (long long)p.Hq * D, p.T - key0, lane);
    } else {
      float dkv[DT][4];
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int e = 0; e < 4; ++e) dkv[d][e] = bf2f(f2bf(acc_dk[wt][d][e] * p.scale));
      if (p.rope) rope_inverse_tiles<DT>(dkv, p.rope, min(key, p.T - 1), g);
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int e = 0; e < 4; ++e) { ok[d][e] = f2bf(dkv[d][e]); ov[d][e] = f2bf(acc_dv[wt][d][e]); }
      store_rows<DT>(stage, ok, p.dk + (drow0 + key0) * p.lddk + hk * D, p.lddk, p.T - key0, lane);
      store_rows<DT>(stage, ov, p.dv + (drow0 + key0) * p.lddv + hk * D, p.lddv, p.T - key0, lane);
    }
  }
}

// =================================== backward: dQ ===================================
// Block = (query block of NT/4, head, batch): NT/64 waves, wave w owns queries qb0 + w*16 .. +16; loops over ST-key steps
// (NT, ST, DEEP as in the dK/dV kernel).
template <int D, int NT, int ST, bool DEEP, bool TR, int WT = 1>
__global__ __launch_bounds__(NT, 1) void attn_bwd_dq_k(AttnArgs p) {
  constexpr int QB = NT / 4 * WT;   // queries per block (WT 16-query tiles per wave)
  constexpr int KS = D / 32, DT = D / 16, KT = ST / 16, KP = ST / 32;
  constexpr int TILE = ST * D * 2;
  constexpr int NTILE = TR ? 2 : 3;   // tiles per buffer
  __shared__ __attribute__((aligned(16))) char ldsAll[2 * NTILE * TILE];  // [buf][K | V (| K^T)]

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  // grid = (heads, batch, query blocks), the query block slowest and - under a causal mask - the LAST one first: it has
  // the most key steps (longest-first dispatch, as in the dK/dV kernel)
  const int b = blockIdx.y, h = blockIdx.x, hk = h / (p.Hq / p.Hkv);
  const int qb0 = (p.causal ? (int)(gridDim.z - 1 - blockIdx.z) : (int)blockIdx.z) * QB;
  const int q_w0 = qb0 + w * 16 * WT;  // this wave's 16 WT queries: tile wt holds q_w0 + 16 wt + (lane & 15)
  const int k_lo = p.kv_start ? p.kv_start[b] : 0;
  const int k_hi = p.kv_len ? min(p.kv_len[b], p.T) : p.T;

  // delta[q] = sum_d dO[q, d] * O[q, d] is computed HERE (this kernel runs first): the four lanes that share a query row hold
  // its dO fragments, so the row dot product costs one extra O load per fragment and two shuffles - and the separate
  // delta kernel (one launch per layer, 20 us at T = 316) is gone.  The result is also written out for the dK/dV kernel.
  bf16x8_t qf[WT][KS], dof[WT][KS];
  float dl[WT], lse[WT];
#pragma unroll
  for (int wt = 0; wt < WT; ++wt) {
    const int q = q_w0 + wt * 16 + fr;
    dl[wt] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      u16x8_t a = {0, 0, 0, 0, 0, 0, 0, 0}, c = {0, 0, 0, 0, 0, 0, 0, 0}, o8 = {0, 0, 0, 0, 0, 0, 0, 0};
      if (q < p.T) {
        a = *reinterpret_cast<const u16x8_t*>(p.q + ((long long)b * p.T + q) * p.ldq + h * D + ks * 32 + g * 8);
        c = *reinterpret_cast<const u16x8_t*>(p.dout + ((long long)b * (p.T - p.dfirst) - p.dfirst + max(q, p.dfirst)) * p.ldo + h * D + ks * 32 + g * 8);
        o8 = *reinterpret_cast<const u16x8_t*>(p.o + ((long long)b * p.T + q) * p.ldo + h * D + ks * 32 + g * 8);
      }
      qf[wt][ks] = __builtin_bit_cast(bf16x8_t, a);
      dof[wt][ks] = __builtin_bit_cast(bf16x8_t, c);
#pragma unroll
      for (int e = 0; e < 8; ++e) dl[wt] += bf2f(c[e]) * bf2f(o8[e]);
    }
    dl[wt] += __shfl_xor(dl[wt], 16, 64);
    dl[wt] += __shfl_xor(dl[wt], 32, 64);
    if (g == 0 && q < p.T) p.delta[((long long)b * p.Hq + h) * p.T + q] = dl[wt];
    lse[wt] = q < p.T ? p.lse[((long long)b * p.Hq + h) * p.T + q] : __builtin_huge_valf();
  }

  f32x4_t acc[WT][DT];
#pragma unroll
  for (int wt = 0; wt < WT; ++wt)
#pragma unroll
    for (int d = 0; d < DT; ++d) acc[wt][d] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  int kend = p.T;
  if (p.causal) kend = min(kend, qb0 + QB);
  if (p.block > 0) kend = min(kend, ((qb0 + QB - 1) / p.block + 1) * p.block);
  kend = min(kend, k_hi);
  const bf16_t* kbase = p.k + (long long)b * p.T * p.ldk + hk * D;
  const bf16_t* vbase = p.v + (long long)b * p.T * p.ldv + hk * D;
  const bf16_t* ktbase = TR ? nullptr : p.kt + ((long long)b * p.Hkv + hk) * D * p.Tp;

  // key steps of ST, software pipelined like the dK/dV kernel
  struct StepRegs { NatRegs<D, ST, NT> k, v; TrRegs<D, TR ? 8 * NT / D : ST, NT> kt; };   // (TR: kt unused, minimal)
  StepRegs r0, r1;
  const int k_begin = (max(k_lo, p.window > 0 ? qb0 - p.window + 1 : 0) / ST) * ST;     // (sliding window: nothing before the block's first query's window)
  const int n_it = k_begin < kend ? (kend - k_begin + ST - 1) / ST : 0;
  int iks = k_begin, cks = k_begin;    // next key step to load / key step being computed
  const int k_last = k_begin + (n_it - 1) * ST;
  auto issue = [&](StepRegs& r) {
    const int ks0 = iks;
    if (iks < k_last) iks += ST;                        // past the end: re-load the last step (static VMEM count)
    load_nat<D, ST, NT>(r.k, kbase, p.ldk, ks0, p.T, tid);
    load_nat<D, ST, NT>(r.v, vbase, p.ldv, ks0, p.T, tid);
    if constexpr (!TR) load_tr<D, ST, NT>(r.kt, ktbase, p.Tp, ks0, tid);
  };
  auto commit = [&](const StepRegs& r, int buf) {
    store_nat<D, ST, NT>(ldsAll + buf * NTILE * TILE, r.k, tid);
    store_nat<D, ST, NT>(ldsAll + buf * NTILE * TILE + TILE, r.v, tid);
    if constexpr (!TR) store_tr<D, ST, NT>(ldsAll + buf * NTILE * TILE + 2 * TILE, r.kt, tid);
  };
  auto compute = [&](int cur) {
    const int ks0 = cks;
    cks += ST;
    constexpr int QW = 16 * WT;     // this wave's queries: q_w0 .. q_w0 + QW - 1
    // every pair of this wave's tiles masked (its queries lie before the step's first key, or past the sequence): nothing to add
    if ((p.causal && ks0 > q_w0 + QW - 1) || q_w0 >= p.T || (p.window > 0 && ks0 + ST - 1 <= q_w0 - p.window)) return;
    const char* ldsK = ldsAll + cur * NTILE * TILE;
    const char* ldsV = ldsK + TILE;
    const char* ldsKT = ldsK + 2 * TILE;      // (!TR)
    f32x4_t s[WT][KT], dp[WT][KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) {
#pragma unroll
      for (int wt = 0; wt < WT; ++wt) { s[wt][t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dp[wt][t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8_t a = lds_b128(ldsK + nat_off<D>(t * 16 + fr, ks * 4 + g));     // (read ONCE for all WT query tiles)
        const bf16x8_t c = lds_b128(ldsV + nat_off<D>(t * 16 + fr, ks * 4 + g));
#pragma unroll
        for (int wt = 0; wt < WT; ++wt) {
          s[wt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[wt][ks], s[wt][t], 0, 0, 0);     // S^T[key][q]
          dp[wt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c, dof[wt][ks], dp[wt][t], 0, 0, 0);  // dP^T[key][q]
        }
      }
    }
    float ds[WT][KT][4];
    bool need_mask = ks0 < k_lo || ks0 + ST > k_hi || q_w0 + QW > p.T;
    if (p.causal) need_mask = need_mask || ks0 + ST - 1 > q_w0;
    if (p.block > 0) need_mask = need_mask || (ks0 + ST - 1) / p.block > q_w0 / p.block;
    if (p.window > 0) need_mask = need_mask || ks0 <= q_w0 + QW - 1 - p.window;
    unsigned okbits[WT];              // bit t*4 + e: the (key, query) pair of that accumulator element takes part
#pragma unroll
    for (int wt = 0; wt < WT; ++wt) okbits[wt] = 0xffffu;
    if (need_mask) {
#pragma unroll
      for (int wt = 0; wt < WT; ++wt) {
        const int q = q_w0 + wt * 16 + fr;
        okbits[wt] = 0u;
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int key = ks0 + t * 16 + g * 4 + e;
            okbits[wt] |= (unsigned)((q < p.T) & key_ok(key, q, k_lo, k_hi, p.causal, p.block, p.window)) << (t * 4 + e);
          }
      }
    }
#pragma unroll
    for (int wt = 0; wt < WT; ++wt)
#pragma unroll
      for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float ev = __builtin_amdgcn_exp2f(s[wt][t][e] * p.sc - lse[wt]);
          const float pv = (okbits[wt] >> (t * 4 + e)) & 1u ? ev : 0.f;
          ds[wt][t][e] = pv * (dp[wt][t][e] - dl[wt]);
        }
    bf16x8_t dsB[WT][KP];
#pragma unroll
    for (int wt = 0; wt < WT; ++wt)
#pragma unroll
      for (int kp = 0; kp < KP; ++kp) dsB[wt][kp] = pack8(ds[wt][2 * kp], ds[wt][2 * kp + 1]);
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      const int row = d * 16 + fr;
#pragma unroll
      for (int kp = 0; kp < KP; ++kp) {
        bf16x8_t a;
        if constexpr (TR) a = tr_frag<D>(ldsK, kp * 32, d * 16, lane);
        else a = lds_2xb64(ldsKT + tr_off8<ST>(row, kp * 8 + g), ldsKT + tr_off8<ST>(row, kp * 8 + 4 + g));
#pragma unroll
        for (int wt = 0; wt < WT; ++wt) acc[wt][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, dsB[wt][kp], acc[wt][d], 0, 0, 0);  // dQ^T[d][q]
      }
    }
  };
  if (n_it > 0) {
    issue(r0);
    commit(r0, 0);
    if (DEEP) issue(r1);
  }
  __syncthreads();
  if (DEEP) {
    for (int it = 0; it < n_it; it += 2) {
      issue(r0);                   // buffer 0 holds step it; r1 holds step it + 1 (in flight); r0 <- step it + 2
      compute(0);
      commit(r1, 1);
      __syncthreads();
      if (it + 1 >= n_it) break;
      issue(r1);                   // buffer 1 holds step it + 1; r0 holds step it + 2 (in flight); r1 <- step it + 3
      compute(1);
      commit(r0, 0);
      __syncthreads();
    }
  } else {                         // one register set, one step ahead
    for (int it = 0; it < n_it; ++it) {
      issue(r0);
      compute(it & 1);
      commit(r0, (it & 1) ^ 1);
      __syncthreads();
    }
  }
#pragma unroll
  for (int wt = 0; wt < WT; ++wt) {
    const int q0 = q_w0 + wt * 16, q = q0 + fr;
    if (q0 >= p.T || q0 < p.dfirst) continue;   // wave-uniform; rows leave through the (now free) staging buffers as 128-byte segments
    char* stage = ldsAll + w * STAGE_BYTES;
    float dqv[DT][4];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int e = 0; e < 4; ++e) dqv[d][e] = bf2f(f2bf(acc[wt][d][e] * p.scale));
    if (p.rope) rope_inverse_tiles<DT>(dqv, p.rope, min(q, p.T - 1), g);
    u16x4_t oq[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int e = 0; e < 4; ++e) oq[d][e] = f2bf(dqv[d][e]);
    store_rows<DT>(stage, oq, p.dq + ((long long)b * (p.T - p.dfirst) - p.dfirst + q0) * p.lddq + h * D, p.lddq, p.T - q0, lane);
  }
}

// =================================== backward, fused (short causal sequences) ===================================
// One block = ONE (batch, query head): the whole S = Q K^T triangle of a sequence of at most TMAX positions is worked off on one
// CU, so S and dP are computed ONCE (5 matrix products instead of the 7 of the dQ + dK/dV kernel pair), and the pair's second
// launch, its second staging stream and the `delta` round trip through global memory disappear.
//   prologue  delta[q] = rowsum(dO * O) and the saved log-sum-exp into LDS.
//   phase 1   waves own 16-key tiles (K / V fragments in registers, dK^T / dV^T accumulators in registers); the Q / dO rows arrive
//             in 64-query chunks (natural layout, staged once per pass for all eight waves).  Per (32 queries x 16 keys):
//             S, dP -> P, dS -> dV^T += dO^T P, dK^T += Q^T dS (transposed operands by ds_read_b64_tr_b16) - and dS goes, as
//             bf16, into a triangular LDS array of 16 x 16 tiles [q][key] (105 KB at 320 positions).
//   phase 2   waves own 16-query tiles (dQ^T accumulators in registers); K arrives in 64-key chunks (double buffered):
//             dQ^T += K^T dS^T with dS^T read back as the B operand (8 bytes per lane and tile).
// Key tiles are dealt to the waves in three passes (j = w, 15 - w, 16 + w) - one key tile's accumulators at a time fit the
// register file; a pass only walks the query chunks at or below its first key (causality).  Same arithmetic per element as
// the kernel pair (P and dS rounded to bf16 at the same points, f32 accumulation); the summation ORDER over queries / keys
// differs, so results agree to rounding, not bitwise.  Fixed order: repeated launches are bit-identical.
template <int D, int TMAX, int NQT>
__global__ __launch_bounds__(512, 1) void attn_bwd_fused_k(AttnArgs p) {
  constexpr int KS = D / 32, DT = D / 16, NT16 = TMAX / 16;
  constexpr int QS = NQT * 16;                    // queries per step (32 or 64): one dependent LDS -> MFMA -> exp -> MFMA chain per step
  static_assert(NQT == 2 || NQT == 4, "a step covers one or two 32-query contraction chunks");
  constexpr int CH = 64;                          // rows per staged chunk
  constexpr int TILE = CH * D * 2;                // bytes of one staged chunk
  constexpr int DS_BYTES = NT16 * (NT16 + 1) / 2 * 512;
  static_assert(TMAX % 64 == 0 && NT16 <= 20, "three passes of eight key tiles cover at most 20 tiles (the last pass takes four)");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ldsDS = smem;                              // [tile (i, j <= i)][q 16][key 16] bf16
  char* ldsBuf = smem + DS_BYTES;                  // phase 1: [Q chunk | dO chunk]; phase 2: [K chunk] x 2
  float* ldsLse = reinterpret_cast<float*>(smem + DS_BYTES + 2 * TILE);
  float* ldsDl = ldsLse + TMAX;
  char* ldsStage = smem + DS_BYTES + 2 * TILE + 2 * TMAX * 4;   // [wave][16 rows][64 + 8 columns] bf16: store_rows_staged

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int h = blockIdx.x, b = blockIdx.y, grp = p.Hq / p.Hkv, hk = h / grp;
  const int T = p.T;
  const int k_lo = p.kv_start ? p.kv_start[b] : 0;
  const int k_hi = p.kv_len ? min(p.kv_len[b], T) : T;
  char* stage = ldsStage + w * STAGE_BYTES;
  const bf16_t* qbase = p.q + (long long)b * T * p.ldq + h * D;
  // Row-compacted gradients (p.dfirst = s > 0, a multiple of 16): dout / dq / dk / dv hold positions >= s only, sequence b at row b * (T - s) - the
  // caller needs no gradient below position s (model.hip: the text prefix before the first audio token; under the causal mask nothing trainable is
  // reachable from it).  Position t of this sequence is row drow0 + t; rows of positions < s do not exist: dO is read clamped to position s there
  // (finite copies - a query below s only meets keys below s, whose dK / dV, like its own dQ, are never stored) and the stores skip those tiles.
  const int s0 = p.dfirst;
  const long long drow0 = (long long)b * (T - s0) - s0;
  const bf16_t* dobase = p.dout + drow0 * p.ldo + h * D;
  const bf16_t* obase = p.o + (long long)b * T * p.ldo + h * D;
  const bf16_t* kbase = p.k + (long long)b * T * p.ldk + hk * D;
  const bf16_t* vbase = p.v + (long long)b * T * p.ldv + hk * D;
  const int nch = (T + CH - 1) / CH;               // query / key chunks that hold rows
  const int nt = (T + 15) / 16;                    // 16-row tiles that hold rows

  TL_STAMP(0);
  unsigned long long tl_step = 0, tl_wait = 0, tl_n = 0, tl_p2 = 0;
  (void)tl_step; (void)tl_wait; (void)tl_n; (void)tl_p2;
  // ---- prologue: the saved log-sum-exp.  delta[q] = rowsum(dO * O) is computed chunk by chunk while pass 0 stages its Q / dO
  // chunks (the O rows ride along with that prefetch): as a prologue of its own it cost 12-20K cycles of pure HBM time - every
  // block of the launch asking for its 160 KB of dO and O at once (profiles/r03_attn_timeline.txt)
  for (int r = tid; r < TMAX; r += 512) ldsLse[r] = r < T ? p.lse[((long long)b * p.Hq + h) * T + r] : __builtin_huge_valf();

  TL_STAMP(1);
  // ---- phase 1: dK, dV, dS ----
  NatRegs<D, CH, 512> rq, rdo, ro;
  static_assert(sizeof(rdo.v) / sizeof(rdo.v[0]) == 2 && D / 8 == 16, "delta: a row's 16 chunks sit in 16 consecutive threads");
#pragma unroll 1
  for (int pass = 0; pass < 3; ++pass) {
    const int j = pass == 0 ? w : (pass == 1 ? 15 - w : 16 + w);          // this wave's key tile
    const bool have = j < nt && (pass < 2 || w < 4) && j * 16 < k_hi && j * 16 + 16 > k_lo;
    const int j_min = pass == 0 ? 0 : (pass == 1 ? 8 : 16);                // first key tile of the pass (block-uniform)
    if (j_min >= nt) break;
    const int key = j * 16 + fr;
    bf16x8_t kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      u16x8_t a = {0, 0, 0, 0, 0, 0, 0, 0}, c = {0, 0, 0, 0, 0, 0, 0, 0};
      if (have && key < T) {
        a = *reinterpret_cast<const u16x8_t*>(kbase + (long long)key * p.ldk + ks * 32 + g * 8);
        c = *reinterpret_cast<const u16x8_t*>(vbase + (long long)key * p.ldv + ks * 32 + g * 8);
      }
      kf[ks] = __builtin_bit_cast(bf16x8_t, a);
      vf[ks] = __builtin_bit_cast(bf16x8_t, c);
    }
    f32x4_t acc_dk[DT], acc_dv[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) { acc_dk[d] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc_dv[d] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    const int c_first = (j_min * 16) / CH;         // causal: queries below the pass's first key see none of its keys
    load_nat<D, CH, 512>(rq, qbase, p.ldq, c_first * CH, T, tid);
    load_nat_from<D, CH, 512>(rdo, dobase, p.ldo, c_first * CH, s0, T, tid);
    if (pass == 0) load_nat<D, CH, 512>(ro, obase, p.ldo, 0, T, tid);
#pragma unroll 1
    for (int c = c_first; c < nch; ++c) {
      const unsigned long long tl_a = TL_NOW();
      __syncthreads();                              // every wave is done with the previous chunk (and, first, the prologue)
      store_nat<D, CH, 512>(ldsBuf, rq, tid);
      store_nat<D, CH, 512>(ldsBuf + TILE, rdo, tid);
      if (pass == 0) {                              // delta of this chunk's rows (pass 0 visits every chunk)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          float dl = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) dl += bf2f(rdo.v[k][e]) * bf2f(ro.v[k][e]);
          dl += __shfl_xor(dl, 1, 64); dl += __shfl_xor(dl, 2, 64); dl += __shfl_xor(dl, 4, 64); dl += __shfl_xor(dl, 8, 64);
          const int row = c * CH + (tid >> 4) + 32 * k;
          if ((tid & 15) == 0) ldsDl[row] = row < T ? dl : 0.f;   // (rows >= T were loaded as copies of row T - 1)
        }
      }
      __syncthreads();
      tl_wait += TL_NOW() - tl_a;
      if (c + 1 < nch) {                            // next chunk's loads fly under this chunk's products
        load_nat<D, CH, 512>(rq, qbase, p.ldq, (c + 1) * CH, T, tid);
        load_nat_from<D, CH, 512>(rdo, dobase, p.ldo, (c + 1) * CH, s0, T, tid);
        if (pass == 0) load_nat<D, CH, 512>(ro, obase, p.ldo, (c + 1) * CH, T, tid);
      }
      if (!have) continue;
#pragma unroll 1
      for (int t2 = 0; t2 < CH / QS; ++t2) {
        const int qs = c * CH + t2 * QS;            // QS queries: tiles i0 .. i0 + NQT - 1
        if (qs + QS - 1 < j * 16 || qs >= T) continue;  // wholly above the diagonal / past the sequence (wave-uniform)
        const unsigned long long tl_b = TL_NOW();
        const char* ldsQ = ldsBuf + 0;
        const char* ldsDO = ldsBuf + TILE;
        f32x4_t s[NQT], dp[NQT];
#pragma unroll
        for (int t = 0; t < NQT; ++t) {
          s[t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dp[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const bf16x8_t a = lds_b128(ldsQ + nat_off<D>(t2 * QS + t * 16 + fr, ks * 4 + g));
            const bf16x8_t cc = lds_b128(ldsDO + nat_off<D>(t2 * QS + t * 16 + fr, ks * 4 + g));
            s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, kf[ks], s[t], 0, 0, 0);
            dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cc, vf[ks], dp[t], 0, 0, 0);
          }
        }
        // accumulator element e of tile t: q = qs + t*16 + g*4 + e, key = this lane's key
        float pr[NQT][4], ds[NQT][4];
#pragma unroll
        for (int t = 0; t < NQT; ++t) {
          const float4 l4 = *reinterpret_cast<const float4*>(ldsLse + qs + t * 16 + g * 4);
          const float4 d4 = *reinterpret_cast<const float4*>(ldsDl + qs + t * 16 + g * 4);
          const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int q = qs + t * 16 + g * 4 + e;
            const bool ok = (q < T) & (key >= k_lo) & (key < k_hi) & (key <= q);
            const float ev = __builtin_amdgcn_exp2f(s[t][e] * p.sc - lq[e]);
            const float pv = ok ? ev : 0.f;
            pr[t][e] = pv;
            ds[t][e] = pv * (dp[t][e] - dq4[e]);
          }
          const int i = (qs >> 4) + t;              // query tile; stored only on / below the diagonal (the rest is never read)
          if (i >= j && i < NT16) {
            bf16_t* tile = reinterpret_cast<bf16_t*>(ldsDS + (i * (i + 1) / 2 + j) * 512);
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[(g * 4 + e) * 16 + fr] = f2bf(ds[t][e]);
          }
        }
#pragma unroll
        for (int kp = 0; kp < NQT / 2; ++kp) {
          const bf16x8_t pB = pack8(pr[2 * kp], pr[2 * kp + 1]), dsB = pack8(ds[2 * kp], ds[2 * kp + 1]);
#pragma unroll
          for (int d = 0; d < DT; ++d) {
            const bf16x8_t a = tr_frag<D>(ldsDO, t2 * QS + kp * 32, d * 16, lane);
            const bf16x8_t cc = tr_frag<D>(ldsQ, t2 * QS + kp * 32, d * 16, lane);
            acc_dv[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pB, acc_dv[d], 0, 0, 0);
            acc_dk[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cc, dsB, acc_dk[d], 0, 0, 0);
          }
        }
        tl_step += TL_NOW() - tl_b; tl_n += 1;
      }
    }
    TL_STAMP(2 + 2 * pass);
    // dK^T / dV^T of this wave's key tile: row d = dt*16 + g*4 + e, col key = fr (per query head under GQA: gqa_reduce_k sums);
    // stored through the wave's LDS stage as whole 128-byte row segments
    if (j < nt && (pass < 2 || w < 4) && j * 16 >= s0) {
      u16x4_t ok[DT], ov[DT];
      if (p.dkv_part) {
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
          for (int e = 0; e < 4; ++e) { ok[d][e] = f2bf(acc_dk[d][e] * p.scale); ov[d][e] = f2bf(acc_dv[d][e]); }
        bf16_t* dkp = reinterpret_cast<bf16_t*>(p.dkv_part) + (((long long)b * T + j * 16) * p.Hq + h) * D;
        bf16_t* dvp = dkp + (long long)p.B * T * p.Hq * D;
        store_rows_staged<DT>(stage, ok, dkp, (long long)p.Hq * D, T - j * 16, lane);
        store_rows_staged<DT>(stage, ov, dvp, (long long)p.Hq * D, T - j * 16, lane);
      } else {
        float dkv[DT][4];
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
          for (int e = 0; e < 4; ++e) dkv[d][e] = bf2f(f2bf(acc_dk[d][e] * p.scale));
        if (p.rope) rope_inverse_tiles<DT>(dkv, p.rope, min(key, T - 1), g);
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
          for (int e = 0; e < 4; ++e) { ok[d][e] = f2bf(dkv[d][e]); ov[d][e] = f2bf(acc_dv[d][e]); }
        store_rows_staged<DT>(stage, ok, p.dk + (drow0 + j * 16) * p.lddk + hk * D, p.lddk, T - j * 16, lane);
        store_rows_staged<DT>(stage, ov, p.dv + (drow0 + j * 16) * p.lddv + hk * D, p.lddv, T - j * 16, lane);
      }
    }
    TL_STAMP(3 + 2 * pass);
  }

  // ---- phase 2: dQ ----
  // query tiles of this wave: w, 19 - w and (w < 4) 8 + w  ->  slots 0..2
  int qi[3] = {w, NT16 - 1 - w, w < 4 ? 8 + w : NT16};
  f32x4_t acc[3][DT];
#pragma unroll
  for (int z = 0; z < 3; ++z)
#pragma unroll
    for (int d = 0; d < DT; ++d) acc[z][d] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  NatRegs<D, CH, 512> rk;
  load_nat<D, CH, 512>(rk, kbase, p.ldk, 0, T, tid);
  __syncthreads();                                  // phase 1 is over: its chunk buffer is free, every dS tile is written
  store_nat<D, CH, 512>(ldsBuf, rk, tid);
#pragma unroll 1
  for (int kc = 0; kc < nch; ++kc) {
    const char* ldsK = ldsBuf + (kc & 1) * TILE;
    if (kc + 1 < nch) load_nat<D, CH, 512>(rk, kbase, p.ldk, (kc + 1) * CH, T, tid);
    __syncthreads();                                // chunk kc is in place (and chunk kc - 1's buffer is no longer read)
    const unsigned long long tl_c = TL_NOW();
#pragma unroll
    for (int z = 0; z < 3; ++z) {
      const int i = qi[z];
      if (i >= nt) continue;
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const int j0 = kc * 4 + k2 * 2;             // key tiles j0, j0 + 1 (32 keys)
        if (j0 > i) continue;
        typedef __attribute__((ext_vector_type(4))) unsigned short u16x4v;
        // (a key tile wholly outside the valid key range was never visited in phase 1: its dS tiles do not exist - zeros)
        const bool ok0 = j0 * 16 < k_hi && j0 * 16 + 16 > k_lo, ok1 = j0 + 1 <= i && (j0 + 1) * 16 < k_hi && (j0 + 1) * 16 + 16 > k_lo;
        if (!ok0 && !ok1) continue;
        u16x4v lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
        if (ok0) lo = *reinterpret_cast<const u16x4v*>(ldsDS + (i * (i + 1) / 2 + j0) * 512 + fr * 32 + g * 8);
        if (ok1) hi = *reinterpret_cast<const u16x4v*>(ldsDS + (i * (i + 1) / 2 + j0 + 1) * 512 + fr * 32 + g * 8);
        const bf16x8_t dsB = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          const bf16x8_t a = tr_frag<D>(ldsK, k2 * 32, d * 16, lane);
          acc[z][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, dsB, acc[z][d], 0, 0, 0);   // dQ^T[d][q]
        }
      }
    }
    tl_p2 += TL_NOW() - tl_c;
    if (kc + 1 < nch) store_nat<D, CH, 512>(ldsBuf + ((kc + 1) & 1) * TILE, rk, tid);
  }
  TL_STAMP(8);
#pragma unroll
  for (int z = 0; z < 3; ++z) {
    if (qi[z] >= nt || qi[z] * 16 < s0) continue;   // wave-uniform
    const int q = qi[z] * 16 + fr;
    float dqv[DT][4];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int e = 0; e < 4; ++e) dqv[d][e] = bf2f(f2bf(acc[z][d][e] * p.scale));
    if (p.rope) rope_inverse_tiles<DT>(dqv, p.rope, min(q, T - 1), g);
    u16x4_t oq[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int e = 0; e < 4; ++e) oq[d][e] = f2bf(dqv[d][e]);
    store_rows_staged<DT>(stage, oq, p.dq + (drow0 + qi[z] * 16) * p.lddq + h * D, p.lddq, T - qi[z] * 16, lane);
  }
  TL_STAMP(9);
  TL_PUT(10, tl_step); TL_PUT(11, tl_wait); TL_PUT(12, tl_n); TL_PUT(13, tl_p2);
}

// dk/dv[b, t, hk, :] = sum over the GQA group (fixed order, f32) of the bf16 per-query-head results; with `rope` the summed dK is
// then RoPE-inverted (same arithmetic as rope_k on the bf16-rounded sum).  One thread = 8 columns c..c+7 of the first half of
// the head AND their partners c + D/2.. (the rotary pairs) of one (b, t, hk).
// dfirst > 0 (AttnBwdDesc::d_first): dk / dv are row-compacted - positions below dfirst are skipped, (b, t) is row b * (T - dfirst) + t - dfirst.
__global__ void gqa_reduce_k(const bf16_t* __restrict__ part, bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, int B, int T,
                             int Hq, int Hkv, int D, int lddk, int lddv, const float* __restrict__ rope, int dfirst) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int dv16 = D / 16;
  const long long n = (long long)B * T * Hkv * dv16;
  if (i >= n) return;
  const int c = (int)(i % dv16) * 8, hk = (int)((i / dv16) % Hkv);
  const long long bt = i / ((long long)dv16 * Hkv);
  if ((int)(bt % T) < dfirst) return;
  const long long drow = bt - (bt / T + 1) * dfirst;
  const int grp = Hq / Hkv, half_d = D / 2;
  const long long half = (long long)B * T * Hq * D;
  float sk[2][8], sv[2][8];
#pragma unroll
  for (int z = 0; z < 2; ++z)
#pragma unroll
    for (int e = 0; e < 8; ++e) { sk[z][e] = 0.f; sv[z][e] = 0.f; }
  for (int gq = 0; gq < grp; ++gq) {
    const bf16_t* pk = part + (bt * Hq + hk * grp + gq) * D + c;
#pragma unroll
    for (int z = 0; z < 2; ++z) {
      const u16x8_t a = *reinterpret_cast<const u16x8_t*>(pk + z * half_d);
      const u16x8_t v = *reinterpret_cast<const u16x8_t*>(pk + z * half_d + half);
#pragma unroll
      for (int e = 0; e < 8; ++e) { sk[z][e] += bf2f(a[e]); sv[z][e] += bf2f(v[e]); }
    }
  }
#pragma unroll
  for (int z = 0; z < 2; ++z)
#pragma unroll
    for (int e = 0; e < 8; ++e) sk[z][e] = bf2f(f2bf(sk[z][e]));
  if (rope) {
    const float* t = rope + ((long long)(bt % T) * half_d + c) * 2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float co = bf2f(f2bf(t[2 * e])), sn = -bf2f(f2bf(t[2 * e + 1]));
      const float lo = sk[0][e], hi = sk[1][e];
      sk[0][e] = bf2f(f2bf(lo * co)) + bf2f(f2bf(-hi * sn));
      sk[1][e] = bf2f(f2bf(hi * co)) + bf2f(f2bf(lo * sn));
    }
  }
#pragma unroll
  for (int z = 0; z < 2; ++z) {
    u16x8_t ok, ov;
#pragma unroll
    for (int e = 0; e < 8; ++e) { ok[e] = f2bf(sk[z][e]); ov[e] = f2bf(sv[z][e]); }
    *reinterpret_cast<u16x8_t*>(dk + drow * lddk + hk * D + z * half_d + c) = ok;
    *reinterpret_cast<u16x8_t*>(dv + drow * lddv + hk * D + z * half_d + c) = ov;
  }
}

// probe: out[lane*4 + j] = element j that ds_read_b64_tr_b16 returns to `lane` when lane l supplies the byte address addr[l] of
// an LDS image whose 16-bit element e holds the value e (pins the semantics tr_frag relies on)
__global__ void lds_tr_probe_k(const int32_t* __restrict__ addr, int32_t* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) unsigned short img[4096];
  for (int e = threadIdx.x; e < 4096; e += 64) img[e] = (unsigned short)e;
  __syncthreads();
  const s16x4_t v = lds_tr_b64(reinterpret_cast<const char*>(img) + addr[threadIdx.x]);
#pragma unroll
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (int32_t)(unsigned short)v[j];
}

AttnArgs make_args(const uvx::AttnDesc& d) {
  AttnArgs a = {};
  a.q = (const bf16_t*)d.q; a.k = (const bf16_t*)d.k; a.v = (const bf16_t*)d.v; a.vt = (const bf16_t*)d.vt;
  a.o = (bf16_t*)d.o; a.lse = d.lse; a.kv_start = d.kv_start; a.kv_len = d.kv_len;
  a.B = d.B; a.T = d.T; a.Tp = d.Tp; a.Hq = d.Hq; a.Hkv = d.Hkv;
  a.ldq = d.ldq; a.ldk = d.ldk; a.ldv = d.ldv; a.ldo = d.ldo;
  a.causal = d.causal; a.block = d.block; a.q_begin = d.q_begin; a.window = d.causal ? d.window : 0;
  a.sc = d.scale * LOG2E; a.scale = d.scale;
  return a;
}

int check_desc(const uvx::AttnDesc& d) {
  UVX_CHECK(d.D == 64 || d.D == 128 || d.D == 256, UVX_ERR_UNSUPPORTED, "attention: head_dim %d not supported (64, 128 or 256)", d.D);
  UVX_CHECK(d.Hkv > 0 && d.Hq % d.Hkv == 0, UVX_ERR_SHAPE, "attention: Hq=%d not a multiple of Hkv=%d", d.Hq, d.Hkv);
  UVX_CHECK(d.Tp % 64 == 0 && d.Tp >= d.T, UVX_ERR_SHAPE, "attention: Tp=%d must be a multiple of 64 and >= T=%d", d.Tp, d.T);
  UVX_CHECK(d.ldq % 8 == 0 && d.ldk % 8 == 0 && d.ldo % 8 == 0, UVX_ERR_SHAPE, "attention: row strides must be multiples of 8");
  return UVX_OK;
}

}  // namespace

namespace uvx {

int g_attn_qt = 0;  // probes: force the forward kernel's q-tile count (0 = automatic)
void* g_attn_tl = nullptr;  // probes build: stamp buffer of the fused backward kernel (uvx_probe_attn_timeline)
// bf16 kernels read transposed operands out of the NATURAL tiles with ds_read_b64_tr_b16 (tuning option 12, default on):
// callers then skip heads_transpose and may leave vt / qt / kt / dot null
bool attention_tr_reads(int dtype) { return dtype == DT_BF16 && g_options[12] != 0; }
bool attention_needs_transposed_copies(int dtype) { return dtype == DT_BF16 && g_options[12] == 0; }
int attention_fwd_f32(hipStream_t st, const AttnDesc& d);
int attention_bwd_f32(hipStream_t st, const AttnBwdDesc& d);

int attention_fwd(hipStream_t st, int dtype, const AttnDesc& d) {
  if (d.B == 0 || d.T == 0) return UVX_OK;
  if (dtype != DT_BF16) return attention_fwd_f32(st, d);
  int rc = check_desc(d);
  if (rc) return rc;
  AttnArgs a = make_args(d);
  a.tl = (unsigned long long*)g_attn_tl;
  // q rows per block = 64 * QT.  Long sequences (the encoder's 1500 frames) want QT = 2 for K/V reuse; short
  // ones (the LLM's few hundred tokens) are latency-bound and want more, smaller blocks and fewer registers.
  // (the probe override applies to the head_dim-64 kernels only - the encoder's; the others have one or two instantiations)
  const int qt = (uvx::g_attn_qt > 0 && d.D == 64) ? (uvx::g_attn_qt > 4 ? 4 : uvx::g_attn_qt) : (d.T >= 1024 ? 2 : 1);
  dim3 grid(d.Hq, d.B, cdiv(d.T, 4 * qt * 16));
  const bool tr = attention_tr_reads(dtype);   // V natural + transposing LDS reads (no V^T copy) - tuning option 12
  UVX_CHECK(tr ? d.v != nullptr : d.vt != nullptr, UVX_ERR_INVALID, "attention_fwd: %s is null", tr ? "v" : "vt");
#define FWD(DD, Q) do { if (tr) hipLaunchKernelGGL((attn_fwd_k<DD, Q, true>), grid, dim3(256), 0, st, a); \
                        else hipLaunchKernelGGL((attn_fwd_k<DD, Q, false>), grid, dim3(256), 0, st, a); } while (0)
  if (d.D == 64) {
    // (the grouped-query form of the head_dim-128 branch below, for the LLMs with 64-wide heads - Llama-3.2-1B: 32 query / 8 KV heads; same rule, option 25)
    const bool gq64_ok = tr && d.causal && d.block == 0 && d.T <= 512 && d.Hkv > 0 && d.Hq != d.Hkv && (d.Hq / d.Hkv) % 4 == 0 && d.q_begin == 0 && uvx::g_attn_qt == 0;
    const int gq64 = !gq64_ok || g_options[25] == 5 ? 0 : g_options[25] ? g_options[25] : ((long long)(d.Hq / 4) * d.B * cdiv(d.T, 32) >= 512 ? 1 : 0);
    if (gq64 == 1 || gq64 == 3) { grid = dim3(d.Hq / 4, d.B, cdiv(d.T, 32)); hipLaunchKernelGGL((attn_fwd_k<64, 2, true, true, true>), grid, dim3(256), 0, st, a); }
    else if (gq64 == 4) { grid = dim3(d.Hq / 4, d.B, cdiv(d.T, 16)); hipLaunchKernelGGL((attn_fwd_k<64, 1, true, true, true>), grid, dim3(256), 0, st, a); }
    else
    if (qt == 4) FWD(64, 4);
    else if (qt == 3) FWD(64, 3);
    else if (qt == 2) {
      // tuning option 20 = 1: the rounds 1-5 form of the row max (ds_bpermute shuffles) - A/B only, bit-identical
      if (g_options[20] == 1 && tr) hipLaunchKernelGGL((attn_fwd_k<64, 2, true, false>), grid, dim3(256), 0, st, a);
      else FWD(64, 2);
    }
    else FWD(64, 1);
  } else if (d.D == 128) {
    // The grouped-query form (GQ above) for short causal sequences: two query tiles per wave, taken when the launch still has two blocks per CU
    // (profiles/r06_attn_fwd_gq_probe.txt: C2's 8 x 32 heads x 316 positions 30.3 -> 26.3 us, Llama-3.3-70B's 8 x 64 heads 57.7 -> 39.9; one prompt's
    // 160 blocks are 6 % slower and keep the default form).  Tuning option 25: 0 = this rule, 1 / 3 / 4 = force two / three / one query tiles per wave
    // (four spill), 5 = never.  Bit-identical in every form.
    const bool gq_ok = tr && d.causal && d.block == 0 && d.T <= 512 && d.Hkv > 0 && (d.Hq / d.Hkv) % 4 == 0 && d.q_begin == 0;
    const int gq = !gq_ok || g_options[25] == 5 ? 0 : g_options[25] ? g_options[25] : ((long long)(d.Hq / 4) * d.B * cdiv(d.T, 32) >= 512 ? 1 : 0);
    if (gq == 1) { grid = dim3(d.Hq / 4, d.B, cdiv(d.T, 32)); hipLaunchKernelGGL((attn_fwd_k<128, 2, true, true, true>), grid, dim3(256), 0, st, a); }
    else if (gq == 4) { grid = dim3(d.Hq / 4, d.B, cdiv(d.T, 16)); hipLaunchKernelGGL((attn_fwd_k<128, 1, true, true, true>), grid, dim3(256), 0, st, a); }
    else if (gq == 3) { grid = dim3(d.Hq / 4, d.B, cdiv(d.T, 48)); hipLaunchKernelGGL((attn_fwd_k<128, 3, true, true, true>), grid, dim3(256), 0, st, a); }
    else if (qt == 2) FWD(128, 2);
    else FWD(128, 1);
  } else {   // head_dim 256 (Gemma): one q tile per wave keeps the O accumulators (64 registers) + Q fragments in budget
    grid = dim3(d.Hq, d.B, cdiv(d.T, 64));
    FWD(256, 1);
  }
#undef FWD
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

int lds_tr_probe(hipStream_t st, const int32_t* addr, int32_t* out) {
  hipLaunchKernelGGL(lds_tr_probe_k, dim3(1), dim3(64), 0, st, addr, out);
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

// head_dim 128, causal, at most 320 positions (the LLM's training sequences): ONE fused kernel per (batch, query head)
constexpr int FUSED_TMAX = 320;
bool attention_bwd_is_fused(int dtype, const AttnDesc& f) {
  return dtype == DT_BF16 && attention_tr_reads(dtype) && f.D == 128 && f.causal && f.block == 0 && f.T <= FUSED_TMAX && g_options[13] &&
         (f.window <= 0 || f.window >= f.T);   // (a window that covers the sequence is plain causal attention)
}

// row-compacted gradients (AttnBwdDesc::d_first): the bf16 kernels that read the natural-layout operands (no transposed copy of d o), causal masks
// only - "a query below d_first meets keys below d_first only" is what makes the clamped d o reads harmless
bool attention_bwd_takes_d_first(int dtype, const AttnDesc& f) {
  return dtype == DT_BF16 && attention_tr_reads(dtype) && f.causal && f.block == 0 && (f.D == 64 || f.D == 128 || f.D == 256);
}

int attention_bwd(hipStream_t st, int dtype, const AttnBwdDesc& d) {
  if (d.f.B == 0 || d.f.T == 0) return UVX_OK;
  if (dtype != DT_BF16) return attention_bwd_f32(st, d);
  int rc = check_desc(d.f);
  if (rc) return rc;
  AttnArgs a = make_args(d.f);
  a.dout = (const bf16_t*)d.dout; a.qt = (const bf16_t*)d.qt; a.kt = (const bf16_t*)d.kt; a.dot = (const bf16_t*)d.dot;
  a.delta = d.delta; a.dq = (bf16_t*)d.dq; a.dk = (bf16_t*)d.dk; a.dv = (bf16_t*)d.dv;
  a.lddq = d.lddq; a.lddk = d.lddk; a.lddv = d.lddv;
  UVX_CHECK(d.lddq % 8 == 0 && d.lddk % 8 == 0 && d.lddv % 8 == 0, UVX_ERR_SHAPE, "attention_bwd: gradient row strides must be multiples of 8");
  a.dkv_part = (d.f.Hq != d.f.Hkv) ? d.dkv_part : nullptr;
  a.o = (bf16_t*)d.f.o;
  a.rope = d.rope_cos_sin;
  a.tl = (unsigned long long*)g_attn_tl;
  // dQ first: it computes delta = rowsum(dO * O) on the fly and leaves it in d.delta for the dK/dV kernel
  const int kv_heads = a.dkv_part ? d.f.Hq : d.f.Hkv;
  // head_dim 128 (the LLM): 128 queries / keys per block (8 waves), 32-row steps, two-deep prefetch.  64-row steps (ST = 64,
  // with or without the second staging set) were measured within 3 % of this at the C2 shape in round 2 and again on the
  // transposing-read kernels in round 3 (85.15 vs 84.96 ms per step, profiles/r03_call5_ab.txt): not instantiated.
  const bool tr = attention_tr_reads(dtype);   // natural tiles + transposing LDS reads (no Q^T / K^T / dO^T copies) - option 12
  UVX_CHECK(tr || (d.qt && d.kt && d.dot), UVX_ERR_INVALID, "attention_bwd: transposed operand copies are null");
  UVX_CHECK(d.f.v != nullptr, UVX_ERR_INVALID, "attention_bwd: v is null");
#define BWD(DD, NTH, STEP, DEEP, WQ, WK) do { \
    const dim3 GQ(d.f.Hq, d.f.B, cdiv(d.f.T, (NTH) / 4 * (WQ))), GK(kv_heads, d.f.B, cdiv(d.f.T, (NTH) / 4 * (WK))); \
    if (tr) { hipLaunchKernelGGL((attn_bwd_dq_k<DD, NTH, STEP, DEEP, true, WQ>), GQ, dim3(NTH), 0, st, a); \
              hipLaunchKernelGGL((attn_bwd_dkdv_k<DD, NTH, STEP, DEEP, true, WK>), GK, dim3(NTH), 0, st, a); } \
    else { hipLaunchKernelGGL((attn_bwd_dq_k<DD, NTH, STEP, DEEP, false, WQ>), GQ, dim3(NTH), 0, st, a); \
           hipLaunchKernelGGL((attn_bwd_dkdv_k<DD, NTH, STEP, DEEP, false, WK>), GK, dim3(NTH), 0, st, a); } } while (0)
  // (attention_bwd_is_fused above) S and dP once, dS through LDS (tuning option 13, default on)
  const bool fused = attention_bwd_is_fused(dtype, d.f);
  UVX_CHECK(d.d_first == 0 || (attention_bwd_takes_d_first(dtype, d.f) && d.d_first > 0 && d.d_first % 16 == 0 && d.d_first < d.f.T), UVX_ERR_UNSUPPORTED,
            "attention_bwd: d_first = %d needs the bf16 causal kernels on natural-layout operands and a multiple of 16 below T", d.d_first);
  a.dfirst = d.d_first;
  if (fused) {
    constexpr int smem = (FUSED_TMAX / 16) * (FUSED_TMAX / 16 + 1) / 2 * 512 + 2 * 64 * 128 * 2 + 2 * FUSED_TMAX * 4 + 8 * STAGE_BYTES;
    static_assert(smem <= 160 * 1024, "LDS of one CU");
    static PerDeviceOnce attr_set;
    UVX_SET_ATTR_ONCE(attr_set, (attn_bwd_fused_k<128, FUSED_TMAX, 2>), smem);
    // (64-query steps, NQT = 4: 87.9 vs 87.5 ms per step - profiles/r03_call13_probes.txt - not instantiated)
    hipLaunchKernelGGL((attn_bwd_fused_k<128, FUSED_TMAX, 2>), dim3(d.f.Hq, d.f.B), dim3(512), smem, st, a);
  }
  else if (d.f.D == 64) {
    // head_dim 64 (the Whisper tower under LoRA training, 1500 frames): WQ / WK = 16-row tiles per wave of the dQ / dK,dV kernel.  Two
    // tiles per wave read every staged K / V (Q / dO) fragment once for twice the MFMAs and halve the steps + barriers per output row
    // (bit-identical: each output element still sums its steps in the same order).  Measured (profiles/r06_attn_bwd64_ab.txt, B = 8, 16
    // heads, 1500 frames, pair per layer): one tile each 393.9 us, dQ x2 388.9, both x2 391.0, dK,dV x2 only 399.4, 64-row steps 556,
    // eight-wave blocks 450-469 - a 1 % effect: the pair is latency-, not bandwidth-bound.  Tuning option 19 picks the form (A/B).
    switch (g_options[19]) {
      case 1: BWD(64, 256, 32, true, 1, 1); break;     // rounds 1-5: one tile per wave
      case 2: BWD(64, 256, 32, true, 2, 2); break;
      case 3: BWD(64, 256, 32, true, 1, 2); break;
      case 4: BWD(64, 256, 64, true, 2, 2); break;     // 64-row steps
      case 5: BWD(64, 512, 64, true, 1, 1); break;     // eight waves, 128 rows per block
      case 6: BWD(64, 512, 64, true, 2, 1); break;     // (two dK,dV tiles per wave of a 512-thread block spill)
      default: BWD(64, 256, 32, true, 2, 1); break;
    }
  }
  else if (d.f.D == 128) BWD(128, 512, 32, true, 1, 1);
  else BWD(256, 256, 32, false, 1, 1);
#undef BWD
  if (a.dkv_part) {
    const long long n16 = (long long)d.f.B * d.f.T * d.f.Hkv * (d.f.D / 16);
    hipLaunchKernelGGL(gqa_reduce_k, dim3(cdiv(n16, 256)), dim3(256), 0, st, (const bf16_t*)a.dkv_part, a.dk, a.dv, d.f.B, d.f.T, d.f.Hq,
                       d.f.Hkv, d.f.D, a.lddk, a.lddv, a.rope, a.dfirst);
  }
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}

}  // namespace uvx
