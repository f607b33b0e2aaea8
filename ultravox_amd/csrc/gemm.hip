// bf16 MFMA GEMM for gfx950:  C[M,N] = act(alpha * A[M,K] . B[N,K]^T + bias) (+ residual)
//
// Every linear layer on the Ultravox hot path is this "NT" form (activations [tokens, K] times an
// nn.Linear weight [N, K], reference ultravox_model.py:793,798 and the [3P] Whisper/Llama layers).
// The frozen-weight dgrad GEMMs (dX = dY . W) are ALSO run through this kernel against a transposed
// copy of W that is materialised once at load time (288 GB HBM makes the 2x weight copy free), and
// the projector wgrad (dW = dY^T . X) runs on transposed activations.  One kernel to tune.
//
// Tile: 128(M) x 128(N) x 64(K), 256 threads = 4 waves in 2(M) x 2(N), each wave 64x64 as 4x4
// v_mfma_f32_16x16x32_bf16.  Both operands are K-contiguous so each tile row is one 128-byte line:
// staged global->LDS with global_load_lds_dwordx4 (16 B/lane, 1 KiB per wave instruction, LDS image
// lane-linear).  LDS rows are XOR-swizzled at 16-byte granularity (chunk ^= row & 7) by permuting
// the per-lane SOURCE address inside the same 128-byte line (coalescing unchanged) and applying the
// same involution on the ds_read_b128 fragment reads, which makes them bank-conflict free.
//
// The MFMA "A" operand is the weight tile and "B" the activation tile, so the accumulator fragment
// (rows = 4 consecutive n per lane, col = m) stores 4 consecutive output columns per lane: 8-byte
// bf16x4 / 16-byte f32x4 stores, bias as one vector load.
#include <math.h>
#include <mutex>
#include <type_traits>
#include <unordered_map>
#include <utility>
#include <hip/hip_ext.h>
#include "common.h"
#include "kernels.h"

namespace uvx {
int g_gemm_variant = -1; int g_gemm_split = 1;
// probe hook: per-shape variant overrides (in-situ A/B of the tile choice inside bench.py)
int g_gemm_ovr_n = 0; int g_gemm_ovr[32][4];
// probe hook (uvx_set_option): [1] epilogue access: 0 = 8-byte fragment layout, 1 = 16 bytes via v_permlane16_swap
// (-0.5 ms/step at C2), 2 = row-contiguous through LDS (default), [2] SwiGLU backward fused into the
// dgrad GEMM (off: measured neutral at C2 - the separate elementwise kernel runs at 6.7 TB/s, the fused epilogue is
// serialised behind each tile's main loop), [3] loss head on the supervised rows only (on), [4] weight-streaming kernel for
// M <= 16 (on), [5] residual rows of a wave tile fetched up front in the GEMM epilogue (off: same-box A/B at C2, round 2:
// 98.2 / 98.6 ms per step with it, 97.5 / 97.6 without - the 2 * MI extra live uint4 per lane cost more than the hidden latency),
// [6] tile picker uses the merged-phase kernels 31..34 (on; 0 = the round-1 four-phase set 11, 15..19),
// [7] m-major tile order when the activation matrix is the larger operand (M > N: the encoder's GEMMs; on)
// [8] (probe builds) let the picker choose the stream-K variants 39..42 (off: measured slower, see the kernel),
// [10] tail-split threshold in per cent of the whole launch's modelled cost (0 = 88),
// [9] stream-K flavour: data-parallel rounds before the stream-K part: 1 = all but the last full round ("two-tile"), 0 = none
int g_options[32] = {0, 2, 0, 1, 1, 0, 1, 1, 0, 1, 0, 1, 1, 1, 1, 0, /*16*/ 1, 1, 0, 0, 0, 0, 0, 0, /*24*/ 0, 0, 0, 0, 0, 0, 0, 0};   // keys: include/uvx.h uvx_set_option
}  // -1 = automatic; probes may force a tile variant / disable the tail split

namespace {

struct GemmArgs {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  const bf16_t* bias;
  const bf16_t* residual;
  int M, N, K;
  int lda, ldb, ldc, ldr;
  int res_mod;
  long long sA, sB, sC, sR;
  int act;
  int out_f32;
  int accumulate;  // f32 output only: C += result
  float alpha;
  int tiles_m, tiles_n;
  bf16_t* C2;      // swiglu mode: activation output [M, N/2]
  int ldc2;
  int m_major;       // XCD-contiguous tile runs share an activation panel instead of a weight panel (option 7, M > N)
  int swiglu;      // 1: columns alternate 16-wide gate / up blocks; also write silu(gate) * up to C2
  int sw_stage;    // swiglu == 2: whole-line epilogue through the LDS stage (0 = the fragment-layout form)
  int wide_io;     // 16-byte epilogue loads / stores (probe switch; on by default)
  const int32_t* m_dev;  // device-side row count (nullptr: M is exact)
  int m_dev_off;         // rows of the compact list handled by earlier launches
  // stream-K launches (SK kernels): data-parallel rounds before the stream-K part, partial-accumulator slots
  // (one per block), publish flags (one per block, + a timeout counter at [gridDim.x]) and this launch's flag value
  int sk_full;
  float* sk_ws;
  unsigned* sk_flags;
  unsigned sk_epoch;
  // split-K launches (round 5): blockIdx.y = z of ksplit takes K-tiles [z nk / ksplit, (z + 1) nk / ksplit) and writes its f32 partial
  // tile to slab z of C (stride sC); sA = sB = 0.  splitk_reduce_k sums the slabs in a fixed order and runs the epilogue.
  int ksplit;
  int b_kn;        // B is stored [K, N] (row stride ldb): the "NN" form - a dgrad reads the forward weight [N_out, N_in] as it lies (GemmDesc::b_kn)
};

constexpr int BM = 128, BN = 128, BK = 64;

__device__ __forceinline__ void glds16(const bf16_t* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}


// Epilogue memory access in 16-byte pieces.  The accumulator layout gives a lane 4 consecutive output columns of a row
// (8 bytes of bf16) and the store / load path is issue-bound (32 accesses per lane per operand for a 128 x 64 wave
// tile).  Two row fragments (i, i+1) are therefore exchanged between the lane groups fg and fg ^ 1 with
// v_permlane16_swap: afterwards a lane holds 8 consecutive columns of ONE row (fragment i for even fg, i+1 for odd fg)
// = one dwordx4.  The exchange is an involution, so 16-byte LOADS are brought into fragment layout the same way.
// a = packed columns (0-1, 2-3) of fragment i, b = of fragment i+1.  Every lane of the wave must take part.
__device__ __forceinline__ void frag_pair_swap(uint32_t (&a)[2], uint32_t (&b)[2]) {
  const auto s0 = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
  const auto s1 = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
  a[0] = s0[0]; b[0] = s0[1]; a[1] = s1[0]; b[1] = s1[1];
}
__device__ __forceinline__ uint32_t pack2(float lo, float hi) { return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16); }
__device__ __forceinline__ float unpack_lo(uint32_t v) { return bf2f((bf16_t)(v & 0xffffu)); }
__device__ __forceinline__ float unpack_hi(uint32_t v) { return bf2f((bf16_t)(v >> 16)); }
// act == 3 (GELU backward): two bf16 gradients x gelu'(two bf16 pre-activations), rounded
__device__ __forceinline__ uint32_t mul_gelu_grad2(uint32_t d, uint32_t x) {
  return pack2(unpack_lo(d) * gelu_fast_grad(unpack_lo(x)), unpack_hi(d) * gelu_fast_grad(unpack_hi(x)));
}
// fragment-layout pair -> one 16-byte store at `dst` (the lane's row / 8-column slot), if ok
__device__ __forceinline__ void store_pair16(bf16_t* dst, bool ok, uint32_t (&a)[2], uint32_t (&b)[2]) {
  frag_pair_swap(a, b);
  if (ok) *reinterpret_cast<uint4*>(dst) = make_uint4(a[0], a[1], b[0], b[1]);
}
// one 16-byte load from `src` (zeros if !ok) -> fragment-layout pair
__device__ __forceinline__ void load_pair16(const bf16_t* src, bool ok, uint32_t (&a)[2], uint32_t (&b)[2]) {
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (ok) v = *reinterpret_cast<const uint4*>(src);
  a[0] = v.x; a[1] = v.y; b[0] = v.z; b[1] = v.w;
  frag_pair_swap(a, b);
}

// Shared epilogue.  acc[j][i] is the 16x16 accumulator fragment whose rows are output columns
// n_base + 16 j + 4 fg .. +3 (4 consecutive per lane) and whose column is output row m_base + 16 i + frow.
// The reference's bf16 rounding points are restated: round(acc*alpha + bias), round(act(.)), then + residual.
// Row-contiguous output through LDS.  A store instruction in fragment layout touches 32 rows x 32 bytes: HBM sees
// quarter-line writes and the fixed cost of a launch is (output bytes) / ~2 TB/s (tools/gpu_gemm_overhead_probe.py).
// With `stage` (this wave's MI*16 x 128-byte slice of the block's LDS, free after the main loop) the finished bf16
// fragments are written to LDS (16-byte chunks XOR-swizzled by row), read back row-wise - 8 lanes = one 128-byte row of
// the wave tile - and stored / residual-added as full lines.  Same values, same rounding order.
template <int ROWS, int CH /*16-byte chunks per row*/>
__device__ __forceinline__ char* stage_slot(char* stage, int row, int chunk) {
  return stage + row * (CH * 16) + ((chunk ^ (row & (CH - 1))) << 4);
}

// GI = row fragments staged per pass: MI (the whole wave tile, MI*16 x 128 B of LDS per wave) or 1 (16 rows = 2 KiB per
// wave: the persistent kernels, whose operand buffers are already being refilled for the next tile).
// ACT23: the instantiation that also carries the GELU epilogues which keep / consume the pre-activation (act 2 / 3, round 6).  They are a
// build of their own: compiled into the one epilogue, their live values cost the 256-row tile its spill-free register budget (276 bytes of
// scratch per lane in EVERY launch of that tile, with or without act 2 / 3 - the first form of this round, caught by tools/isa_resources.py).
template <int NJ, int MI, int GI = MI, bool ACT23 = false>
__device__ __forceinline__ void store_tile(const GemmArgs& p, f32x4_t (&acc)[NJ][MI], int m_base, int n_base, int frow,
                                           int fg, long long z, char* stage = nullptr) {
  const bool bf16_out = !p.out_f32;
  const int lane = fg * 16 + frow;
  if (NJ == 4 && stage && p.wide_io == 2 && bf16_out && p.swiglu == 2 && p.sw_stage && (p.N & 15) == 0 && (p.ldc & 7) == 0 &&
      (p.ldc2 & 7) == 0 && ((uintptr_t)p.C & 15) == 0 && ((uintptr_t)p.C2 & 15) == 0) {
    // Fused SwiGLU BACKWARD through the LDS stage (round 3): the tile is d act = dY . W_down^T.  Pass 1 parks the bf16 d act
    // rows in this wave's slice; pass 2 walks the matching rows of gate|up / d gate|d up - 256 contiguous bytes per wave-tile
    // row in the interleaved layout [g0 (16 cols) | u0 | g1 | u1 | ...] - with SIXTEEN lanes per row, so every load and store
    // instruction moves whole 128-byte lines (the fragment-layout version below moves half lines and measured neutral
    // against the separate swiglu_bwd kernel).  A lane owns one 16-byte chunk: gate or up columns 8 h .. 8 h + 7 of block b;
    // the other kind comes from lane ^ 2 by a cross-lane exchange; each lane then produces its own kind of gradient (the
    // sigmoid is evaluated by both partners: VALU is idle here).  Arithmetic and rounding points of swiglu_bwd_k, bit for bit.
    constexpr int ROWS = GI * 16;
    static_assert(MI % GI == 0, "row fragments per pass must divide the wave tile");
    const int c16 = lane & 15, blk = c16 >> 2, is_up = (c16 >> 1) & 1, hf = c16 & 1;
#pragma unroll
    for (int i0 = 0; i0 < MI; i0 += GI) {
      if (i0 > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the previous pass's reads are done (WAR on the slice)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int ii = 0; ii < GI; ++ii) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[j][i0 + ii][e] * p.alpha;
          *reinterpret_cast<uint2*>(stage_slot<ROWS, 8>(stage, ii * 16 + frow, 2 * j + (fg >> 1)) + (fg & 1) * 8) =
              make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int n16 = n_base + blk * 16;                                   // first d-act column of this lane's block
      const long long goff = (long long)(n16 >> 4) * 32 + is_up * 16 + hf * 8;   // the lane's chunk in the interleaved row
#pragma unroll 2     // (fully unrolled, hipcc hoists every iteration's loads and spills around the 128 live accumulators)
      for (int it = 0; it < ROWS / 4; ++it) {
        const int row = it * 4 + (lane >> 4), m = m_base + i0 * 16 + row;
        const uint4 dv = *reinterpret_cast<const uint4*>(stage_slot<ROWS, 8>(stage, row, blk * 2 + hf));
        const bool ok = m < p.M && n16 < p.N;
        uint4 mine = make_uint4(0u, 0u, 0u, 0u);
        if (ok) mine = *reinterpret_cast<const uint4*>(p.C2 + (long long)m * p.ldc2 + goff);
        uint4 other;
        other.x = __shfl_xor(mine.x, 2, 64); other.y = __shfl_xor(mine.y, 2, 64);
        other.z = __shfl_xor(mine.z, 2, 64); other.w = __shfl_xor(mine.w, 2, 64);
        const uint4 gq = is_up ? other : mine, uq = is_up ? mine : other;
        const uint32_t dw[4] = {dv.x, dv.y, dv.z, dv.w}, gw[4] = {gq.x, gq.y, gq.z, gq.w}, uw[4] = {uq.x, uq.y, uq.z, uq.w};
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = (e & 1) ? unpack_hi(dw[e >> 1]) : unpack_lo(dw[e >> 1]);
          const float g = (e & 1) ? unpack_hi(gw[e >> 1]) : unpack_lo(gw[e >> 1]);
          const float u = (e & 1) ? unpack_hi(uw[e >> 1]) : unpack_lo(uw[e >> 1]);
          const float sg = 1.0f / (1.0f + expf(-g));
          o[e] = is_up ? d * bf2f(f2bf(g * sg)) : d * u * (sg * (1.0f + g * (1.0f - sg)));
        }
        if (ok)
          *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + z * p.sC + (long long)m * p.ldc + goff) =
              make_uint4(pack2(o[0], o[1]), pack2(o[2], o[3]), pack2(o[4], o[5]), pack2(o[6], o[7]));
      }
    }
    return;
  }
  if (NJ == 4 && stage && p.wide_io == 2 && bf16_out && p.swiglu != 2 && (p.N & 7) == 0 && (p.ldc & 7) == 0 && (p.sC & 7) == 0 &&
      ((uintptr_t)p.C & 15) == 0 &&
      ((p.swiglu == 0 && p.act < 2) || ((p.ldc2 & 7) == 0 && ((uintptr_t)p.C2 & 15) == 0)) &&
      (!p.residual || ((p.ldr & 7) == 0 && (p.sR & 7) == 0 && ((uintptr_t)p.residual & 15) == 0))) {
    constexpr int ROWS = GI * 16;
    static_assert(MI % GI == 0, "row fragments per pass must divide the wave tile");
    float bv[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = n_base + j * 16 + fg * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[j][e] = 0.f;
      if (p.bias && n < p.N) {
        u16x4_t b4 = *reinterpret_cast<const u16x4_t*>(p.bias + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[j][e] = bf2f(b4[e]);
      }
    }
    // (round 2's option 5 - the residual rows of the whole wave tile prefetched into a register array ahead of the staging
    // pass - measured slower in situ and was removed in round 3: its dynamically indexed array also gave every kernel of the
    // family a private (scratch) segment, used or not)
#pragma unroll
    for (int i0 = 0; i0 < MI; i0 += GI) {
      if (i0 > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the previous pass's reads are done (WAR on the slice)
      // ---- pass 1: [ROWS x 64] of C ----
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int ii = 0; ii < GI; ++ii) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float t = bf2f(f2bf(acc[j][i0 + ii][e] * p.alpha + bv[j][e]));
            if (p.act == 1) t = bf2f(f2bf(gelu_fast(t)));      // (act 2: C keeps the pre-activation, pass 2 writes gelu; act 3: below)
            v[e] = t;
          }
          *reinterpret_cast<uint2*>(stage_slot<ROWS, 8>(stage, ii * 16 + frow, 2 * j + (fg >> 1)) + (fg & 1) * 8) =
              make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      {
        const int c8 = lane & 7, n8 = n_base + c8 * 8;
#pragma unroll
        for (int it = 0; it < ROWS / 8; ++it) {
          const int row = it * 8 + (lane >> 3), m = m_base + i0 * 16 + row;
          uint4 o = *reinterpret_cast<const uint4*>(stage_slot<ROWS, 8>(stage, row, c8));
          if (m < p.M && n8 < p.N) {
            if (ACT23 && p.act == 3) {       // GELU backward: the saved pre-activation arrives as whole lines, like a residual
              const uint4 x = *reinterpret_cast<const uint4*>(p.C2 + (long long)m * p.ldc2 + n8);
              o.x = mul_gelu_grad2(o.x, x.x); o.y = mul_gelu_grad2(o.y, x.y); o.z = mul_gelu_grad2(o.z, x.z); o.w = mul_gelu_grad2(o.w, x.w);
            }
            if (p.residual) {
              const int rm = p.res_mod > 0 ? (m % p.res_mod) : m;
              const uint4 r = *reinterpret_cast<const uint4*>(p.residual + z * p.sR + (long long)rm * p.ldr + n8);
              o.x = pack2(unpack_lo(o.x) + unpack_lo(r.x), unpack_hi(o.x) + unpack_hi(r.x));
              o.y = pack2(unpack_lo(o.y) + unpack_lo(r.y), unpack_hi(o.y) + unpack_hi(r.y));
              o.z = pack2(unpack_lo(o.z) + unpack_lo(r.z), unpack_hi(o.z) + unpack_hi(r.z));
              o.w = pack2(unpack_lo(o.w) + unpack_lo(r.w), unpack_hi(o.w) + unpack_hi(r.w));
            }
            *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + z * p.sC + (long long)m * p.ldc + n8) = o;
          }
        }
      }
      if (ACT23 && p.act == 2) {
        // ---- pass 2 (GELU that keeps its pre-activation): round(gelu(round(acc * alpha + bias))), [ROWS x 64] -> C2 ----
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // pass-1 reads done before the slice is overwritten
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
          for (int ii = 0; ii < GI; ++ii) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_fast(bf2f(f2bf(acc[j][i0 + ii][e] * p.alpha + bv[j][e])));
            *reinterpret_cast<uint2*>(stage_slot<ROWS, 8>(stage, ii * 16 + frow, 2 * j + (fg >> 1)) + (fg & 1) * 8) =
                make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int c8 = lane & 7, n8 = n_base + c8 * 8;
#pragma unroll
        for (int it = 0; it < ROWS / 8; ++it) {
          const int row = it * 8 + (lane >> 3), m = m_base + i0 * 16 + row;
          const uint4 o = *reinterpret_cast<const uint4*>(stage_slot<ROWS, 8>(stage, row, c8));
          if (m < p.M && n8 < p.N) *reinterpret_cast<uint4*>(p.C2 + (long long)m * p.ldc2 + n8) = o;
        }
        continue;
      }
      if (p.swiglu != 1) continue;
      // ---- pass 2 (fused SwiGLU): act = round(silu(gate)) * up, [ROWS x 32] -> C2; gate = fragment j (even), up = j + 1 ----
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // pass-1 reads done before the slice is overwritten
#pragma unroll
      for (int j = 0; j < NJ; j += 2) {
#pragma unroll
        for (int ii = 0; ii < GI; ++ii) {
          float a[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float g = bf2f(f2bf(acc[j][i0 + ii][e] * p.alpha));
            a[e] = bf2f(f2bf(g / (1.0f + __expf(-g)))) * bf2f(f2bf(acc[j + 1][i0 + ii][e] * p.alpha));
          }
          *reinterpret_cast<uint2*>(stage_slot<ROWS, 4>(stage, ii * 16 + frow, j + (fg >> 1)) + (fg & 1) * 8) =
              make_uint2(pack2(a[0], a[1]), pack2(a[2], a[3]));
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      {
        const int c4 = lane & 3, n8 = n_base / 2 + c4 * 8;
#pragma unroll
        for (int it = 0; it < ROWS / 16; ++it) {
          const int row = it * 16 + (lane >> 2), m = m_base + i0 * 16 + row;
          const uint4 o = *reinterpret_cast<const uint4*>(stage_slot<ROWS, 4>(stage, row, c4));
          if (m < p.M && 2 * n8 < p.N) *reinterpret_cast<uint4*>(p.C2 + (long long)m * p.ldc2 + n8) = o;
        }
      }
    }
    return;
  }
  if (NJ == 4 && GI >= 2 && stage && p.wide_io == 2 && p.out_f32 && !p.accumulate && !p.bias && !p.residual && p.act == 0 && p.swiglu == 0 &&
      (p.N & 3) == 0 && (p.ldc & 3) == 0 && (p.sC & 3) == 0 && ((uintptr_t)p.C & 15) == 0) {
    // f32 output through the stage (round 5: the split-K partial tiles - twice the bytes of a bf16 tile, and in fragment layout a store
    // instruction would touch 16 rows x 64 bytes).  One row fragment per pass: the wave parks 16 rows x 64 columns x 4 bytes = 4 KiB
    // (256-byte rows, 16-byte chunks XOR-swizzled by row) in its slice, reads them back with SIXTEEN lanes per row and stores two whole
    // 128-byte lines per row.  Same values as the fragment-layout path below.
    float* Cf = reinterpret_cast<float*>(p.C) + z * p.sC;
    const int c16 = lane & 15, n4 = n_base + c16 * 4;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if (i > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the previous pass's reads are done (WAR on the slice)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        *reinterpret_cast<float4*>(stage + frow * 256 + (((j * 4 + fg) ^ frow) << 4)) =
            make_float4(acc[j][i][0] * p.alpha, acc[j][i][1] * p.alpha, acc[j][i][2] * p.alpha, acc[j][i][3] * p.alpha);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 4 + (lane >> 4), m = m_base + i * 16 + row;
        const float4 o = *reinterpret_cast<const float4*>(stage + row * 256 + ((c16 ^ row) << 4));
        if (m < p.M && n4 < p.N) *reinterpret_cast<float4*>(Cf + (long long)m * p.ldc + n4) = o;
      }
    }
    return;
  }
  constexpr int MI2 = MI & ~1;
  // row / column slot of this lane in the 16-byte layout (see frag_pair_swap): row fragment i + (fg & 1), columns 8 (fg >> 1)
  const int prow = (fg & 1) * 16 + frow, pcol = (fg >> 1) * 8;
  if (p.swiglu == 1) {
    // fused LlamaMLP activation: fragment j (even) holds 16 gate columns, fragment j+1 the matching up columns
    // (weights are packed that way at load time); gate|up is stored for the backward pass and
    // act = round(silu(round(gate))) * round(up) — the reference's rounding points — goes to C2.
    const bool wide = p.wide_io && (p.ldc & 7) == 0 && (p.ldc2 & 7) == 0 && ((uintptr_t)p.C & 15) == 0 && ((uintptr_t)p.C2 & 15) == 0;
#pragma unroll
    for (int j = 0; j < NJ; j += 2) {
      const int n = n_base + j * 16 + fg * 4;
#pragma unroll
      for (int i = 0; i < MI; i += 2) {
        const bool pair = wide && i + 1 < MI;          // compile-time per i, wave-uniform
        uint32_t pg[2][2], pu[2][2], pa[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (i + h >= MI) break;
          float a[4];
          bf16_t og[4], ou[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            og[e] = f2bf(acc[j][i + h][e] * p.alpha);
            ou[e] = f2bf(acc[j + 1][i + h][e] * p.alpha);
            const float g = bf2f(og[e]);
            a[e] = bf2f(f2bf(g / (1.0f + __expf(-g)))) * bf2f(ou[e]);
          }
          pg[h][0] = (uint32_t)og[0] | ((uint32_t)og[1] << 16); pg[h][1] = (uint32_t)og[2] | ((uint32_t)og[3] << 16);
          pu[h][0] = (uint32_t)ou[0] | ((uint32_t)ou[1] << 16); pu[h][1] = (uint32_t)ou[2] | ((uint32_t)ou[3] << 16);
          pa[h][0] = pack2(a[0], a[1]); pa[h][1] = pack2(a[2], a[3]);
        }
        if (pair) {
          const int m = m_base + i * 16 + prow, n8 = n_base + j * 16 + pcol;
          const bool ok = m < p.M && n8 < p.N;
          bf16_t* crow = reinterpret_cast<bf16_t*>(p.C) + z * p.sC + (long long)m * p.ldc + n8;
          store_pair16(crow, ok, pg[0], pg[1]);
          store_pair16(crow + 16, ok, pu[0], pu[1]);
          store_pair16(p.C2 + (long long)m * p.ldc2 + (n_base + j * 16) / 2 + pcol, ok, pa[0], pa[1]);
        } else {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            if (i + h >= MI) break;
            const int m = m_base + (i + h) * 16 + frow;
            if (m >= p.M || n >= p.N) continue;
            bf16_t* crow = reinterpret_cast<bf16_t*>(p.C) + z * p.sC + (long long)m * p.ldc + n;
            *reinterpret_cast<uint2*>(crow) = make_uint2(pg[h][0], pg[h][1]);
            *reinterpret_cast<uint2*>(crow + 16) = make_uint2(pu[h][0], pu[h][1]);
            *reinterpret_cast<uint2*>(p.C2 + (long long)m * p.ldc2 + (n_base + j * 16) / 2 + fg * 4) = make_uint2(pa[h][0], pa[h][1]);
          }
        }
      }
    }
    return;
  }
  if (p.swiglu == 2) {
    // fused LlamaMLP activation BACKWARD: the tile holds d act = dY . W_down^T; with gate|up (C2, interleaved 16-column
    // blocks as above) it becomes d gate | d up in the same interleaved layout (C, ldc = 2 N) - the arithmetic and the
    // bf16 rounding points of swiglu_bwd_k (elementwise.hip), which this replaces on the bf16 path.
    const bool wide = p.wide_io && (p.ldc & 7) == 0 && (p.ldc2 & 7) == 0 && ((uintptr_t)p.C & 15) == 0 && ((uintptr_t)p.C2 & 15) == 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = n_base + j * 16 + fg * 4;
      const int goff = (n >> 4) * 32 + (n & 15);
      const int nb = n_base + j * 16;                      // first column of the fragment
      const int goff8 = (nb >> 4) * 32 + pcol;             // the lane's 8-column slot in the interleaved layout
#pragma unroll
      for (int i = 0; i < MI; i += 2) {
        const bool pair = wide && i + 1 < MI;
        uint32_t g2[2][2], u2[2][2], dg2[2][2], du2[2][2];
        if (pair) {
          const int m = m_base + i * 16 + prow;
          const bool ok = m < p.M && nb + pcol < p.N;
          const bf16_t* gu = p.C2 + (long long)m * p.ldc2 + goff8;
          load_pair16(gu, ok, g2[0], g2[1]);
          load_pair16(gu + 16, ok, u2[0], u2[1]);
        } else {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            if (i + h >= MI) break;
            const int m = m_base + (i + h) * 16 + frow;
            uint2 gv = make_uint2(0u, 0u), uv = make_uint2(0u, 0u);
            if (m < p.M && n < p.N) {
              const bf16_t* gu = p.C2 + (long long)m * p.ldc2 + goff;
              gv = *reinterpret_cast<const uint2*>(gu);
              uv = *reinterpret_cast<const uint2*>(gu + 16);
            }
            g2[h][0] = gv.x; g2[h][1] = gv.y; u2[h][0] = uv.x; u2[h][1] = uv.y;
          }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (i + h >= MI) break;
          float dgv[4], duv[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float d = bf2f(f2bf(acc[j][i + h][e] * p.alpha));
            const uint32_t gw = g2[h][e >> 1], uw = u2[h][e >> 1];
            const float g = (e & 1) ? unpack_hi(gw) : unpack_lo(gw), u = (e & 1) ? unpack_hi(uw) : unpack_lo(uw);
            const float sg = 1.0f / (1.0f + expf(-g));
            duv[e] = d * bf2f(f2bf(g * sg));
            dgv[e] = d * u * (sg * (1.0f + g * (1.0f - sg)));
          }
          dg2[h][0] = pack2(dgv[0], dgv[1]); dg2[h][1] = pack2(dgv[2], dgv[3]);
          du2[h][0] = pack2(duv[0], duv[1]); du2[h][1] = pack2(duv[2], duv[3]);
        }
        if (pair) {
          const int m = m_base + i * 16 + prow;
          const bool ok = m < p.M && nb + pcol < p.N;
          bf16_t* crow = reinterpret_cast<bf16_t*>(p.C) + z * p.sC + (long long)m * p.ldc + goff8;
          store_pair16(crow, ok, dg2[0], dg2[1]);
          store_pair16(crow + 16, ok, du2[0], du2[1]);
        } else {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            if (i + h >= MI) break;
            const int m = m_base + (i + h) * 16 + frow;
            if (m >= p.M || n >= p.N) continue;
            bf16_t* crow = reinterpret_cast<bf16_t*>(p.C) + z * p.sC + (long long)m * p.ldc + goff;
            *reinterpret_cast<uint2*>(crow) = make_uint2(dg2[h][0], dg2[h][1]);
            *reinterpret_cast<uint2*>(crow + 16) = make_uint2(du2[h][0], du2[h][1]);
          }
        }
      }
    }
    return;
  }
  // bf16 output: 16-byte stores (and residual loads) for the paired row fragments; the rest takes the 8-byte path below
  const bool wide_store = p.wide_io && bf16_out && p.act < 2 && (p.N & 7) == 0 && (p.ldc & 7) == 0 && (p.sC & 7) == 0 && ((uintptr_t)p.C & 15) == 0;
  const bool wide_res = p.residual && p.res_mod == 0 && (p.ldr & 7) == 0 && (p.sR & 7) == 0 && ((uintptr_t)p.residual & 15) == 0;
  if (wide_store) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = n_base + j * 16 + fg * 4;
      const bool n_ok = n < p.N;
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      if (p.bias && n_ok) {
        u16x4_t b4 = *reinterpret_cast<const u16x4_t*>(p.bias + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = bf2f(b4[e]);
      }
#pragma unroll
      for (int i = 0; i < MI2; i += 2) {
        const int mp = m_base + i * 16 + prow, n8 = n_base + j * 16 + pcol;
        const bool ok = mp < p.M && n8 < p.N;
        uint32_t r2[2][2] = {{0u, 0u}, {0u, 0u}};
        if (wide_res) load_pair16(p.residual + z * p.sR + (long long)mp * p.ldr + n8, ok, r2[0], r2[1]);
        uint32_t pk[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int m = m_base + (i + h) * 16 + frow;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float t = bf2f(f2bf(acc[j][i + h][e] * p.alpha + bv[e]));
            if (p.act == 1) t = bf2f(f2bf(gelu_fast(t)));
            v[e] = t;
          }
          if (wide_res) {
            v[0] += unpack_lo(r2[h][0]); v[1] += unpack_hi(r2[h][0]); v[2] += unpack_lo(r2[h][1]); v[3] += unpack_hi(r2[h][1]);
          } else if (p.residual && n_ok && m < p.M) {
            const int rm = p.res_mod > 0 ? (m % p.res_mod) : m;
            u16x4_t r4 = *reinterpret_cast<const u16x4_t*>(p.residual + z * p.sR + (long long)rm * p.ldr + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bf2f(r4[e]);
          }
          pk[h][0] = pack2(v[0], v[1]); pk[h][1] = pack2(v[2], v[3]);
        }
        store_pair16(reinterpret_cast<bf16_t*>(p.C) + z * p.sC + (long long)mp * p.ldc + n8, ok, pk[0], pk[1]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = n_base + j * 16 + fg * 4;
    if (n >= p.N) continue;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
      u16x4_t b4 = *reinterpret_cast<const u16x4_t*>(p.bias + n);
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[e] = bf2f(b4[e]);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if (wide_store && i < MI2) continue;             // already stored above
      const int m = m_base + i * 16 + frow;
      if (m >= p.M) continue;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = acc[j][i][e] * p.alpha + bv[e];
        if (bf16_out) t = bf2f(f2bf(t));
        if (p.act == 1) {
          t = gelu_fast(t);
          if (bf16_out) t = bf2f(f2bf(t));
        }
        v[e] = t;
      }
      if (ACT23 && p.act >= 2) {       // (bf16 output only - gemm_nt checks; fragment-layout form of the staged passes above)
        bf16_t* x2 = p.C2 + (long long)m * p.ldc2 + n;
        if (p.act == 2) {
          u16x4_t g4;
#pragma unroll
          for (int e = 0; e < 4; ++e) g4[e] = f2bf(gelu_fast(v[e]));
          *reinterpret_cast<u16x4_t*>(x2) = g4;
        } else {
          const u16x4_t x4 = *reinterpret_cast<const u16x4_t*>(x2);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= gelu_fast_grad(bf2f(x4[e]));
        }
      }
      if (p.residual) {
        const int rm = p.res_mod > 0 ? (m % p.res_mod) : m;
        u16x4_t r4 = *reinterpret_cast<const u16x4_t*>(p.residual + z * p.sR + (long long)rm * p.ldr + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += bf2f(r4[e]);
      }
      const long long off = z * p.sC + (long long)m * p.ldc + n;
      if (bf16_out) {
        u16x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
        *reinterpret_cast<u16x4_t*>(reinterpret_cast<bf16_t*>(p.C) + off) = o;
      } else {
        float4* dst = reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + off);
        if (p.accumulate) {
          float4 c = *dst;
          v[0] += c.x; v[1] += c.y; v[2] += c.z; v[3] += c.w;
        }
        *dst = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  }
}

// Epilogue of the 32 x 32 x 16 kernels (M32): acc[nb][im] is the 32 x 32 accumulator of output columns n_base + 32 nb + .. and rows
// m_base + 32 im + ..; element e of lane l: column 8 (e >> 2) + 4 (l >> 5) + (e & 3), row l & 31 (the MFMA's D rows are the W rows).
// Plain epilogues only - bias, GELU (act 1), residual, bf16 output, 16-byte-aligned operands (launch_variant sends everything else to the
// 16 x 16 x 32 twin): the wave parks its IM * 32 rows x 64 columns as bf16 in its LDS slice and stores whole 128-byte lines, with
// store_tile's arithmetic and rounding points.
template <int IM>
__device__ __forceinline__ void store_tile32(const GemmArgs& p, f32x16_t (&acc)[2][IM], int m_base, int n_base, int lane, long long z, char* stage) {
  constexpr int ROWS = IM * 32;
  const int r32 = lane & 31, h32 = lane >> 5;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n_base + nb * 32 + q * 8 + h32 * 4;
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      if (p.bias && n < p.N) {
        const u16x4_t b4 = *reinterpret_cast<const u16x4_t*>(p.bias + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = bf2f(b4[e]);
      }
#pragma unroll
      for (int im = 0; im < IM; ++im) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = bf2f(f2bf(acc[nb][im][q * 4 + e] * p.alpha + bv[e]));
          if (p.act == 1) t = bf2f(f2bf(gelu_fast(t)));
          v[e] = t;
        }
        *reinterpret_cast<uint2*>(stage_slot<ROWS, 8>(stage, im * 32 + r32, nb * 4 + q) + h32 * 8) = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
      }
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int c8 = lane & 7, n8 = n_base + c8 * 8;
#pragma unroll
  for (int it = 0; it < ROWS / 8; ++it) {
    const int row = it * 8 + (lane >> 3), m = m_base + row;
    uint4 o = *reinterpret_cast<const uint4*>(stage_slot<ROWS, 8>(stage, row, c8));
    if (m < p.M && n8 < p.N) {
      if (p.residual) {
        const int rm = p.res_mod > 0 ? (m % p.res_mod) : m;
        const uint4 r = *reinterpret_cast<const uint4*>(p.residual + z * p.sR + (long long)rm * p.ldr + n8);
        o.x = pack2(unpack_lo(o.x) + unpack_lo(r.x), unpack_hi(o.x) + unpack_hi(r.x));
        o.y = pack2(unpack_lo(o.y) + unpack_lo(r.y), unpack_hi(o.y) + unpack_hi(r.y));
        o.z = pack2(unpack_lo(o.z) + unpack_lo(r.z), unpack_hi(o.z) + unpack_hi(r.z));
        o.w = pack2(unpack_lo(o.w) + unpack_lo(r.w), unpack_hi(o.w) + unpack_hi(r.w));
      }
      *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + z * p.sC + (long long)m * p.ldc + n8) = o;
    }
  }
}

template <bool ACT23 = false>
__global__ __launch_bounds__(256, 2) void gemm_nt_bf16_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) char lds[2 * BM * BK * 2];
  char* ldsX = lds;
  char* ldsW = lds + BM * BK * 2;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wr = w >> 1, wc = w & 1;

  // XCD-aware bijective remap: block b is dispatched to XCD b % 8; give every XCD a contiguous run
  // of tile ids so that neighbouring tiles (same weight panel) share that XCD's L2.
  const int nwg = gridDim.x;
  const int orig = blockIdx.x;
  const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  const int tm = wg % p.tiles_m, tn = wg / p.tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;
  if (p.m_dev) {   // rows known only on the device: clamp M, drop tiles beyond it (before any barrier)
    const int me = min(p.M, max(*p.m_dev - p.m_dev_off, 0));
    if (m0 >= me) return;
    p.M = me;
  }
  const long long z = blockIdx.y;
  const bf16_t* A = p.A + z * p.sA;
  const bf16_t* B = p.B + z * p.sB;
  int k_len = p.K;
  if (p.ksplit > 1) {   // this block's K range (whole K-tiles; the ranges differ by at most one K-tile)
    const int nk = p.K / BK, kb = (int)(z * nk / p.ksplit), ke = (int)((z + 1) * nk / p.ksplit);
    A += kb * BK; B += kb * BK; k_len = (ke - kb) * BK;
  }

  // ---- staging addresses: wave w issues instructions (i*4 + w), each covering 8 tile rows ----
  const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
  const bf16_t* ap[4];
  const bf16_t* bp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rr = (i * 4 + w) * 8 + srow;
    const int am = min(m0 + rr, p.M - 1);
    const int bn = min(n0 + rr, p.N - 1);
    ap[i] = A + (long long)am * p.lda + schunk * 8;
    bp[i] = B + (long long)bn * p.ldb + schunk * 8;
  }

  // ---- fragment read offsets (bytes inside a tile) ----
  const int frow = lane & 15, fg = lane >> 4;
  int xoff[2], woff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int pos = (ks * 4 + fg) ^ (frow & 7);
    xoff[ks] = (wr * 64 + frow) * 128 + pos * 16;
    woff[ks] = (wc * 64 + frow) * 128 + pos * 16;
  }

  f32x4_t acc[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < k_len; k0 += BK) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(ap[i] + k0, ldsX + (i * 4 + w) * 1024);
      glds16(bp[i] + k0, ldsW + (i * 4 + w) * 1024);
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): LDS-DMA landed (this wave)
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t xa[4], wa[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        xa[i] = *reinterpret_cast<const bf16x8_t*>(ldsX + xoff[ks] + i * 16 * 128);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        wa[j] = *reinterpret_cast<const bf16x8_t*>(ldsW + woff[ks] + j * 16 * 128);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[j], xa[i], acc[j][i], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue ----
  __syncthreads();   // every wave is done with the staged operand tiles: the LDS becomes the output stage
  store_tile<4, 4, 4, ACT23>(p, acc, m0 + wr * 64, n0 + wc * 64, frow, fg, z, lds + w * (4 * 16 * 128));
}


#ifdef UVX_PROBES
#include "gemm_probe_kernels.inc"
#endif

// ------------------------------------------------------------------------------------------------
// Eight-phase kernel: 256 x 256 x 64 tile, 8 waves (2 x 4, 128 x 64 per wave), full 128-byte rows in LDS (one
// L2 request per cache line, half the requests of the k32 ring above), four phases per K-tile.  The K-tile is
// staged as four 16 KiB half-tiles chosen so that each is read in exactly ONE phase by every wave:
//   XA = the first 64 rows of each wave-row's 128 X rows,  XB = the other 64;
//   WA = the first 32 rows of each wave-column's 64 W rows, WB = the other 32.
// Phase      ds_read (per wave)           MFMA (16 each: 4 X frags x 2 W frags x 2 k-halves)   LDS-DMA issued (2 per wave)
//   1        XA (8) + WA (4)              XA x WA                                              XB of tile t+1
//   2        WB (4)                       XA x WB                                              XA of tile t+2
//   3        XB (8)                       XB x WB                                              WA of tile t+2
//   4        -                            XB x WA  (WA kept in registers)                      WB of tile t+2
// Two buffer sets (tile parity) x 4 half-tiles = 128 KiB.  The only DMA wait is a counted vmcnt(6) in phase 4
// (the three half-tiles issued in phases 2-4 stay in flight): everything of tile t+1 has then landed and is read
// from the next phase on.  Every phase is [reads + DMA issue, lgkmcnt(0), barrier, MFMAs under setprio, barrier];
// wave-row 1 runs one barrier behind wave-row 0, so on every SIMD one wave is in its MFMA section while the
// other is in its load section.
//   RAW: a half-tile is read >= one full phase after the counted wait (+ barrier) that retired its DMA.
//   WAR: a half-tile is re-staged >= one phase after its last read, whose lgkmcnt(0) precedes that phase's
//        first barrier in BOTH wave groups.
// Generalised to BM x 256 tiles (BM = 32 MI): each wave-row owns MI fragments of X rows, split MA = ceil(MI/2) in
// half A and MB = MI - MA in half B; phases 1/2 issue 2 MA x 2 MFMAs, phases 3/4 issue 2 MB x 2.  The DMA schedule is
// an ordered list of 1 KiB instructions (8 rows x 128 B): [XB of t+1] in phase 1 and [XA | WA | WB of t+2] spread
// evenly over phases 2-4; every wave issues the same count per phase (surplus slots go to a dummy 1 KiB target),
// so the counted wait in phase 4 is exact.
// (Tried, no measurable effect on the K-tile time (tools/gpu_gemm_overhead_probe.py slope, +-1 %): dropping s_setprio;
// waiting for the phase-2/3 fragment reads after the barrier instead of before it.)
// (Tried and dropped: reading the next tile's WA fragments in phase 4 to balance the per-phase LDS reads 8/4/8/4
// instead of 12/4/8/0 - 3-10 % slower: the extra counted wait it needs in phase 3 shortens the DMA window.)
// NS = number of buffer sets (K-tiles resident in LDS): 2, or 3 where 3 sets fit the 160 KiB (BM <= 160); with NS sets
// the DMA runs NS-1 tiles ahead: phase 1 issues XB of tile t+NS-1, phases 2-4 [XA | WA | WB] of tile t+NS, and the
// counted wait of phase 4 leaves (NS-1) N234 + (NS-2) N1 instructions in flight.
// MODE (probe builds): 0 = the kernel; 1 = operand delivery only (MFMAs skipped); 2 = arithmetic only (no DMA after the prologue);
// 3 = no epilogue; 4 = timeline: every wave stamps s_memtime at the start of each phase's load section, at the start of its
// MFMA section and at its end (3 x 4 stamps per K-tile, low 32 bits, first 64 K-tiles; stamps are taken into SGPRs and
// written to LDS at the start of the NEXT phase so that no wait lands inside a section) and the first four blocks dump
// them into C instead of the output tile: [block][wave][K-tile][phase][load start | MFMA start | MFMA end] as u32
// (tools/gpu_gemm_timeline.py decodes them)
// PERSIST: the grid is one block per CU and every block walks tiles b, b + grid, b + 2 grid, ... (the order the hardware
// would dispatch them).  When a tile's K loop ends, the DMA for the first NS K-tiles of the NEXT tile is issued before the
// epilogue of the current one, which stages through its own 2 KiB per wave: the output stores (HBM-write bound, 8 us of
// a 110 us tile at K = 4096) and the pipeline fill of the next tile overlap instead of adding up.
// Measured (profiles/r01_gemm_cold_probe_ph8.txt, tools/gpu_gemm_overhead_probe.py): the fixed cost of a 4.4-round launch
// drops 47.5 -> 39.6 us, +0.5...1.5 % on the multi-round shapes - far less than the 33 us a free epilogue would give
// (MODE 3 probe): all CUs finish a round together, so the 30-50 MB of output per round is one HBM-write burst that
// throttles store ISSUE on every CU at once, overlapped or not.  Kept as probe variants (23..26), not selected."
// PH = phases per K-tile.  4: the schedule above.  2 ("four-phase" kernel, round 2): phases 1+2 and 3+4 merged - [read XA, WA,
// WB | 32 MFMAs: XA x WA, XA x WB] and [read XB | 32 MFMAs: XB x WB, XB x WA].  Why: with 16 MFMAs per section (256 cycles) the
// LOAD section of the other wave row is the longer one (phase 1 reads 12 fragments per wave = 48 KB per row through a
// 256 B/clk LDS, plus 2 LDS-DMA issues, plus the barrier hand-off: ~380 cycles), so the matrix pipe waits for loads and the
// K-tile costs ~3000 cycles against the 2048 of its 128 MFMAs per SIMD (profiles/r02_gemm_timeline.txt: 8 hand-offs per
// K-tile at ~60 cycles, phase-1 excess, MFMA sections at 280).  With 32 MFMAs per section (512 cycles) every load section
// (<= 16 fragment reads + <= 6 DMA issues) hides under the other row's MFMAs and there are 4 hand-offs per K-tile.
// DMA schedule (NS = 2): XB(t+1) is issued in the first load section of tile t, [XA | WA | WB](t+2) in the second;
// each load section ends with ONE counted wait that leaves exactly one XB piece set and one REST piece set in flight, i.e.
// retires what was issued a whole K-tile earlier (XB(t) before the section that reads it next; REST(t+1) before tile t+1).
// Three buffer sets (NS = 3, round 5; BM <= 160): XB(t + 2) is issued in the first load section of tile t, [XA | WA | WB](t + 3) in the second,
// and each counted wait leaves TWO XB and TWO REST piece sets in flight - a piece has two K-tile times to arrive instead of one.  For the
// prefill's few-hundred-row problems every weight byte comes from HBM (a panel is shared by two row tiles, not sixteen) and one K-tile
// time (~1 us) does not cover that latency; the last NS - 1 tiles drain with vmcnt(0).
// RAW / WAR: same rules as above - a region is read one section after the wait + barrier that retired it, and restaged
// no earlier than two sections after its last read by EITHER row (XB(t-1)'s set: read in the second load section of tile
// t-1, restaged in the first of tile t; [XA | WA | WB](t): read in the first load section of tile t, restaged in the second).
// SK (stream-K, round 2): the grid is one block per CU (a multiple of 8) and the launch's work - tiles x K-tiles - is cut
// into equal shares instead of whole tiles.  Block b (XCD b & 7 under round-robin dispatch, local index b >> 3) first
// computes p.sk_full whole tiles like a persistent data-parallel kernel (tile b + r * grid: the K loops of a round stay
// aligned across the chip, which is what lets concurrent tiles share operand panels in L2), then its 1/32 of the K-tiles
// of the tiles its XCD has left: a contiguous range that starts and ends anywhere inside a tile.  A piece that does
// not start at K-tile 0 ("contributor": at most one per block, and it is the block's FIRST stream-K piece) is written as
// f32 accumulators to the block's slot and published with a release store of the launch's epoch; the block that owns the
// tile's first K-tile ("owner": its LAST piece) adds the slots of the blocks that continue the tile, in a fixed order, and
// runs the normal epilogue.  Every block gets the same number of K-tiles (+-1), so no CU idles through a partial last
// round (2528 x 4096 x 28672: 160 tiles of 256 x 256 on 256 CUs; 2528 x 28672 x 4096: 4.375 rounds) - the chip is
// power-limited, so the gain could only be a fraction of the idle share it removes.
// MEASURED (profiles/r02_gemm_streamk_probe.txt): correct and deterministic, but 5...55 % SLOWER than the data-parallel twin
// on every C2 shape, so these are probe builds (39..42, libuvx_probes.so), never picked.  Two costs the model above leaves
// out: (1) blocks whose K loops are not aligned stop sharing operand panels in L2 - a 256 x 256 tile has 128 flop per
// operand byte, i.e. 10 TB/s at 1.3 PFLOP/s unless concurrent tiles hit each other's panels; (2) every block writes and
// another re-reads a 256 KB f32 partial through device-scope release / acquire (L2 write-back + invalidate): 128 MB per
// launch, six times the output of 2528 x 4096.
// Progress: owners wait only on blocks of higher index, a contribution never waits, and blocks are dispatched in index
// order, so the wait ends as soon as the needed blocks are resident; a bounded spin (then a counted timeout and a wrong
// tile, never a hung queue) guards the case of a device with < 9 free CUs.  Deterministic: fixed ranges, fixed sum order.
// M32 (round 6, variants 61 / 62): the merged-phase kernel on v_mfma_f32_32x32x16_bf16 - the same tile, wave grid, DMA schedule and LDS
// regions, 8 accumulators of 32 x 32 per wave instead of 32 of 16 x 16: half the MFMA issues and half the operand-register reads per flop
// (a power experiment on a clock-capped matrix pipe; round 1 measured a 32 x 32 flavour of a kernel two generations older 10-20 % slower).
// A 32-row operand read has 16 lanes of one LDS lane group on 16 rows of the SAME 16-byte chunk column; rows are 128 bytes, so with the
// production swizzle (chunk ^ (row & 7)) rows r and r + 8 collide - two-way conflicts on every fragment read.  The M32 image is therefore
// swizzled with (row >> 1) & 7, which is distinct over the 8 even and the 8 odd rows of every ds_read_b128 lane group
// ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and their upper-half twins).  k order inside a K-tile = ascending 16-column steps: the
// summation order differs from the 16 x 16 x 32 kernels (two 32-column steps), results agree to f32 rounding, not bitwise.
// NN (round 6): B is stored [K, N] row-major - C = A . B instead of A . B^T - so that the frozen towers' dgrads (d x = d y . W) read the forward
// weight W [N_out, N_in] as it lies and the transposed copies (16 GB at C2; streamed at 70B for 6 % of the step) are not needed.  Only the W
// side changes: a K-tile's WA / WB regions are staged as [k 64][region column 128] images (256-byte rows; one DMA instruction = 4 k-rows x
// 16 chunks; a row's region columns are the four waves' 32-column pieces, 64 contiguous bytes each in global memory), and a wave's A-operand
// fragment [16 n][32 k] comes out of the image through the transposing LDS read: ds_read_b64_tr_b16 hands lane (i, g) the four consecutive
// k-rows r0 .. r0 + 3 of column c0 + i when lane i of the group points at row r0 + (i >> 2), columns c0 + 4 (i & 3) - two reads (r0 = 32 kh +
// 8 g and + 4) fill the fragment's 8 slots in NATURAL k order, so the X fragments, the MFMA sequence and therefore every result bit are those
// of the NT kernel on the transposed copy.  Rows are 256 bytes apart (all on the same banks): 16-byte chunk c of k-row r sits at position
// c ^ f(r), f(r) = 2 ((r & 3) | ((r >> 1) & 4)) - distinct over the 8 rows {r0..r0+3, r0+8..r0+11} one 32-lane half of a read touches.
template <int BM, int NS = 2, int MODE = 0, bool PERSIST = false, int PH = 4, bool SK = false, bool M32 = false, bool NN = false, bool ACT23 = false>
__global__ __launch_bounds__(512, 1) void gemm_nt_bf16_ph8_kernel(GemmArgs p) {
  static_assert(!NN || (PH == 2 && !PERSIST && !SK && !M32 && MODE == 0), "the NN form exists for the merged-phase production kernels");
  static_assert(PH == 4 || (PH == 2 && (NS == 2 || NS == 3) && MODE == 0), "the merged-phase schedule: two or three buffer sets");
  static_assert(!M32 || (PH == 2 && !PERSIST && !SK && (BM / 32) % 4 == 0), "32 x 32 MFMAs: merged-phase schedule, 32-row halves");
  static_assert(!SK || (PH == 2 && !PERSIST), "stream-K is built on the merged-phase kernel");
  constexpr int BNW = 256;
  constexpr int MI = BM / 32, MA = (MI + 1) / 2, MB = MI - MA;
  constexpr int XA_ROWS = 2 * MA * 16, XB_ROWS = 2 * MB * 16;
  constexpr int XA_I = XA_ROWS / 8, XB_I = XB_ROWS / 8, W_I = 16;          // DMA instructions per region
  constexpr int O_XA = 0, O_XB = XA_ROWS * 128, O_WA = O_XB + XB_ROWS * 128, O_WB = O_WA + 128 * 128;
  constexpr int SET = O_WB + 128 * 128;
  constexpr int REST = XA_I + 2 * W_I;
  constexpr int N1 = (XB_I + 7) / 8;                                       // per wave, phase 1
  constexpr int N234 = (REST + 7) / 8;                                     // per wave, phases 2-4 together
  constexpr int N2 = (N234 + 2) / 3, N3 = (N234 - N2 + 1) / 2, N4 = N234 - N2 - N3;
  static_assert(8 * N2 <= XA_I + W_I, "WB must not be re-staged in the phase that reads it");
  constexpr int PRIO = MODE == 5 ? 1 : MODE == 6 ? 2 : MODE == 7 ? 0 : 3;  // production: 3 (static); MODE 5..7: probes of 1, 2, 0
  constexpr int TL_TILES = 64, TL_WAVE = TL_TILES * 12 * 4;                // MODE 4: stamp bytes per wave
  constexpr int O_DUMMY = NS * SET, O_STAGE = O_DUMMY + 1024, LDS_BYTES = O_STAGE + (PERSIST ? 8 * 2048 : 0) + (MODE == 4 ? 8 * TL_WAVE : 0);
  static_assert(LDS_BYTES <= 160 * 1024, "buffer sets exceed the LDS");
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
  constexpr int INFLIGHT = (NS - 1) * N234 + (NS - 2) * N1;

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 2, wc = w & 3;
  const int ntiles = p.tiles_m * p.tiles_n;            // == gridDim.x unless PERSIST
  if (p.m_dev) p.M = min(p.M, max(*p.m_dev - p.m_dev_off, 0));   // rows known only on the device: clamp M
  int m0, n0;
  // tile `o` in dispatch order -> its origin (XCD-aware bijective remap); false if o is past the end
  auto tile_origin = [&](int o, int& m0_, int& n0_) {
    if (o >= ntiles) return false;
    const int xcd = o & 7, q8 = ntiles >> 3, r8 = ntiles & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (o >> 3);
    // which operand an XCD's contiguous run of tiles SHARES: n-major (default: consecutive tiles walk the M panels of one
    // weight panel - every XCD streams its own weights once and re-fetches the activations) or m-major (p.m_major: the
    // activation panel is the big operand - encoder GEMMs, M = 12000 rows against 1024...4096 weight rows - so consecutive
    // tiles walk the weight panels of one activation panel and it is the small weight matrix that gets re-fetched per XCD)
    if (p.m_major) { m0_ = (wg / p.tiles_n) * BM; n0_ = (wg % p.tiles_n) * BNW; }
    else { m0_ = (wg % p.tiles_m) * BM; n0_ = (wg / p.tiles_m) * BNW; }
    return true;
  };
  // first / next tile with rows to compute (tiles beyond a device-side row count are skipped); uniform over the block
  auto next_tile = [&](int& o, int& m0_, int& n0_) {
    for (;;) {
      if (!tile_origin(o, m0_, n0_)) return false;
      if (m0_ < p.M) return true;
      if (!PERSIST) return false;
      o += gridDim.x;
    }
  };
  int nk = p.K / BK;
  int ks_first = 0;               // split-K launch (p.ksplit > 1): first K-tile of this block's range; folded into A / B below
  if (!SK && p.ksplit > 1) {
    ks_first = (int)((long long)blockIdx.y * nk / p.ksplit);
    nk = (int)((long long)(blockIdx.y + 1) * nk / p.ksplit) - ks_first;
  }
  int kbase = 0, nks = nk;        // this piece: K-tiles [kbase, kbase + nks) of the tile at (m0, n0)  (SK only: else the whole K)
  // stream-K work list of this block (see the template comment)
  const int sk_x = blockIdx.x & 7, sk_wl = blockIdx.x >> 3, sk_per = gridDim.x >> 3;   // XCD, index / blocks within the XCD
  int sk_round = 0, sk_it = 0, sk_hi = 0, sk_lt = 0, sk_I = 0;
  if (SK) {
    const int n_x = (ntiles >> 3) + (sk_x < (ntiles & 7));       // tiles in this XCD's run of the dispatch order
    sk_I = (n_x - p.sk_full * sk_per) * nk;                        // K-tiles left to this XCD after the data-parallel rounds
    sk_it = (int)((long long)sk_wl * sk_I / sk_per);
    sk_hi = (int)((long long)(sk_wl + 1) * sk_I / sk_per);
  }
  auto sk_next = [&]() {          // next piece of this block: sets m0, n0, kbase, nks
    if (sk_round < p.sk_full) {
      kbase = 0; nks = nk;
      tile_origin(sk_round * (int)gridDim.x + (int)blockIdx.x, m0, n0);
      ++sk_round;
      return true;
    }
    if (sk_it >= sk_hi) return false;
    sk_lt = sk_it / nk;
    kbase = sk_it - sk_lt * nk;
    nks = min(nk - kbase, sk_hi - sk_it);
    sk_it += nks;
    tile_origin(((p.sk_full * sk_per + sk_lt) << 3) + sk_x, m0, n0);
    return true;
  };
  int orig = blockIdx.x;
  if (SK) { if (!sk_next()) return; }
  else if (!next_tile(orig, m0, n0)) return;   // (before any barrier)
  const long long z = blockIdx.y;
  const bf16_t* A = p.A + z * p.sA + ks_first * BK;
  const bf16_t* B = p.B + z * p.sB + (NN ? (long long)ks_first * BK * p.ldb : (long long)ks_first * BK);

  // one DMA instruction = rows 8q .. 8q+7 of a region; lane -> row 8q + (lane >> 3), LDS chunk position lane & 7
  // holds source chunk (lane & 7) ^ (row & 7)
  // (M32: the image is swizzled with (region row >> 1) & 7 instead of row & 7 - see the template comment; region row = 8 q + srow)
  const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
  auto sch = [&](int q) { return M32 ? ((lane & 7) ^ (((q * 8 + srow) >> 1) & 7)) : schunk; };
  struct Slot { const bf16_t* g; int off; };
  constexpr int W_FLAG = 1 << 30;      // (NN) set in `off` of a W slot: its K-tiles are BK ROWS (BK * ldb elements) apart, not BK elements
  const long long wstep = NN ? (long long)BK * p.ldb : (long long)BK;
  auto nn_f = [](int r) { return 2 * ((r & 3) | ((r >> 1) & 4)); };   // (NN) chunk swizzle of k-row r
  auto x_slot = [&](bool half_b, int q) {
    const int per = (half_b ? MB : MA) * 16;
    const int hr = q * 8 + srow, wrr = hr / per, rem = hr - wrr * per;
    const int row = wrr * (BM / 2) + (half_b ? MA * 16 : 0) + rem;
    return Slot{A + (long long)min(m0 + row, p.M - 1) * p.lda + sch(q) * 8, (half_b ? O_XB : O_XA) + q * 1024};
  };
  auto w_slot = [&](bool half_b, int q) {
    if constexpr (NN) {
      // instruction q = k-rows 4 q .. 4 q + 3 of the region; lane -> row 4 q + (lane >> 4), LDS chunk position lane & 15 holds region chunk
      // (lane & 15) ^ f(row); region chunk c' = columns 64 (c' >> 2) + (half_b ? 32 : 0) + 8 (c' & 3) .. + 7 of the tile
      const int row = q * 4 + (lane >> 4), cc = (lane & 15) ^ nn_f(row);
      const int n = (cc >> 2) * 64 + (half_b ? 32 : 0) + (cc & 3) * 8;
      return Slot{B + (long long)row * p.ldb + min(n0 + n, p.N - 8), ((half_b ? O_WB : O_WA) + q * 1024) | W_FLAG};
    }
    const int hr = q * 8 + srow;
    const int n = (hr >> 5) * 64 + (half_b ? 32 : 0) + (hr & 31);
    return Slot{B + (long long)min(n0 + n, p.N - 1) * p.ldb + sch(q) * 8, (half_b ? O_WB : O_WA) + q * 1024};
  };
  Slot sxb[N1], srest[N234];
  auto fill_slots = [&](Slot (&sxb_)[N1], Slot (&srest_)[N234]) {   // for the tile at (m0, n0)
    const Slot dummy = Slot{A + (long long)min(m0, p.M - 1) * p.lda + schunk * 8, -1};
#pragma unroll
    for (int k = 0; k < N1; ++k) {
      const int g = k * 8 + w;
      sxb_[k] = g < XB_I ? x_slot(true, g) : dummy;
    }
#pragma unroll
    for (int k = 0; k < N234; ++k) {
      const int g = k * 8 + w;
      srest_[k] = g < XA_I ? x_slot(false, g) : g < XA_I + W_I ? w_slot(false, g - XA_I) : g < REST ? w_slot(true, g - XA_I - W_I) : dummy;
    }
  };
  fill_slots(sxb, srest);
  const int frow = lane & 15, fg = lane >> 4;
  int xo[2], wo[2];   // fragment byte offsets (k-half 0 / 1) relative to the region base
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) {
    const int c = ((fg + 4 * kh) ^ (frow & 7)) * 16;
    xo[kh] = frow * 128 + c;
    wo[kh] = (wc * 32 + frow) * 128 + c;
  }
  const int xa_base = O_XA + wr * MA * 16 * 128, xb_base = O_XB + wr * MB * 16 * 128;
  // (NN) A-operand fragment (16 region columns wc * 32 + j * 16 + .., k = 32 kh + 8 fg + 0..7) of a [64 k][128 columns] image: two transposing reads
  auto nn_frag = [&](const char* region, int j, int kh) {
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    const int i = lane & 15, r0 = kh * 32 + fg * 8 + (i >> 2);
    const int c = wc * 4 + j * 2 + ((i & 3) >> 1), half = (i & 1) * 8;
    const char* p0 = region + r0 * 256 + ((c ^ nn_f(r0)) << 4) + half;
    const char* p1 = region + (r0 + 4) * 256 + ((c ^ nn_f(r0 + 4)) << 4) + half;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
    return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
  };

  auto dma = [&](const Slot& sl, int t, int set_idx) {   // set_idx = t % NS, tracked by the caller
    if (MODE == 2 && t >= NS) return;
    if constexpr (NN) {
      const bool isw = sl.off >= 0 && (sl.off & W_FLAG);
      glds16(sl.g + (long long)t * (isw ? wstep : (long long)BK), lds + (sl.off < 0 ? O_DUMMY : (sl.off & ~W_FLAG) + set_idx * SET));
    } else {
      glds16(sl.g + (SK ? kbase + t : t) * BK, lds + (sl.off < 0 ? O_DUMMY : sl.off + set_idx * SET));
    }
  };
#define UVX_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
// PRIO: who gets issue priority on a SIMD shared by a wave in its MFMA section and one in its load section.
// 3 (production) = static: the second-dispatched wave row runs at s_setprio 1 throughout, no per-section flips
// (MI355X_MICROARCH.md "two waves per SIMD", item 4); 0 = the MFMA section at priority 1 (the 8-phase template's choice);
// 1 = no priorities; 2 = the LOAD section at priority 1.  Cold-weight probe, same box (profiles/r02_gemm_prio_probe.txt):
// static is +1.5 ... +6 % over 0 on six of the seven LLM shapes and never slower than -0.3 %; 1 and 2 sit in between.
#define UVX_PHASE_SYNC()                                   \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
  __builtin_amdgcn_sched_barrier(0);                       \
  __builtin_amdgcn_s_barrier();                            \
  __builtin_amdgcn_sched_barrier(0);                       \
  if (PRIO == 0) __builtin_amdgcn_s_setprio(1);            \
  if (PRIO == 2) __builtin_amdgcn_s_setprio(0)
// MODE 4 stamps: UVX_TL_TAKE(k) reads the clock into tl[k]; UVX_TL_FLUSH(ph) writes the previous phase's three stamps
#define UVX_TL_TAKE(k)                                                             \
  if (MODE == 4) {                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                             \
    tl[k] = __builtin_readcyclecounter();                                          \
  }
#define UVX_TL_FLUSH(tt, ph)                                                       \
  if (MODE == 4 && (tt) >= 0 && (tt) < TL_TILES) {                                 \
    lds_u32* d_ = tl_base + ((tt) * 4 + (ph)) * 3;                                 \
    d_[0] = (unsigned)tl[0]; d_[1] = (unsigned)tl[1]; d_[2] = (unsigned)tl[2];     \
  }
#define UVX_PHASE_END()                                    \
  if (PRIO == 0) __builtin_amdgcn_s_setprio(0);            \
  __builtin_amdgcn_sched_barrier(0);                       \
  __builtin_amdgcn_s_barrier();                            \
  if (PRIO == 2) __builtin_amdgcn_s_setprio(1);            \
  __builtin_amdgcn_sched_barrier(0)

  // prologue: REST(0), XB(0), REST(1), XB(1), ..., REST(NS-1)  (the steady-state issue order)
  auto issue_prologue = [&](const Slot (&sxb_)[N1], const Slot (&srest_)[N234]) {
#pragma unroll
    for (int tt = 0; tt < NS; ++tt) {
      if (tt < nks) {
#pragma unroll
        for (int k = 0; k < N234; ++k) dma(srest_[k], tt, tt);
        if (tt < NS - 1) {
#pragma unroll
          for (int k = 0; k < N1; ++k) dma(sxb_[k], tt, tt);
        }
      }
    }
  };
  issue_prologue(sxb, srest);

  for (;;) {   // one iteration per output tile (exactly one unless PERSIST)
  // K-tile 0 has landed when at most INFLIGHT DMA instructions are outstanding (in PERSIST mode the previous tile's
  // output stores sit in the same counter behind them: the bound on the total still bounds the loads)
  if (nks >= NS) UVX_VMCNT(INFLIGHT);
  else UVX_VMCNT(0);
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();  // stagger: this half runs one barrier behind
  if (PRIO == 3 && wr == 1) __builtin_amdgcn_s_setprio(1);
  if (PRIO == 2) __builtin_amdgcn_s_setprio(1);
  __builtin_amdgcn_sched_barrier(0);

  f32x4_t acc[4][MI];                    // (unused - and removed by the compiler - in the M32 build)
  f32x16_t acc32[2][(MI + 1) / 2];       // (M32) [WA | WB][32-row block of X: MA / 2 from half A, then MB / 2 from half B]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < (MI + 1) / 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc32[j][i][e] = 0.f;
  // (M32) fragment byte offsets: row r32 = lane & 31 of a 32-row block, 16-byte chunk 2 ks + (lane >> 5) of its K-tile row
  const int r32 = lane & 31, h32 = lane >> 5;
  int wo32[4], xo32[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int rw = wc * 32 + r32;                       // row inside the WA / WB region
    wo32[ks] = rw * 128 + (((2 * ks + h32) ^ ((rw >> 1) & 7)) << 4);
    xo32[ks] = r32 * 128 + (((2 * ks + h32) ^ ((r32 >> 1) & 7)) << 4);     // (+ a multiple of 32 rows: the swizzle key is unchanged)
  }

  int cs = 0;   // t % NS
  typedef __attribute__((address_space(3))) unsigned lds_u32;
  unsigned long long tl[3] = {0ull, 0ull, 0ull};
  lds_u32* tl_base = (lds_u32*)(lds + O_STAGE + (PERSIST ? 8 * 2048 : 0) + w * TL_WAVE);
  for (int t = 0; t < nks; ++t) {
    const char* set = lds + cs * SET;
    const int ps = cs == 0 ? NS - 1 : cs - 1;   // (t + NS - 1) % NS
    bf16x8_t xa[MA][2], wa[2][2], wb[2][2];
    if constexpr (M32) {
      constexpr int XH = MA / 2;      // 32-row blocks of X per half and wave row
      bf16x8_t wa3[4], wb3[4], x3[XH][4];
      // ---- section A: read XA, WA, WB; DMA: XB of tile t+1; wait: XB(t) ----
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        wa3[ks] = *reinterpret_cast<const bf16x8_t*>(set + O_WA + wo32[ks]);
        wb3[ks] = *reinterpret_cast<const bf16x8_t*>(set + O_WB + wo32[ks]);
      }
#pragma unroll
      for (int i = 0; i < XH; ++i)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) x3[i][ks] = *reinterpret_cast<const bf16x8_t*>(set + xa_base + xo32[ks] + i * 32 * 128);
      if (t + NS - 1 < nks) {
#pragma unroll
        for (int k = 0; k < N1; ++k) dma(sxb[k], t + NS - 1, ps);
        UVX_VMCNT((NS - 1) * (N234 + N1));
      } else {
        UVX_VMCNT(0);
      }
      UVX_PHASE_SYNC();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int i = 0; i < XH; ++i) acc32[0][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa3[ks], x3[i][ks], acc32[0][i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < XH; ++i) acc32[1][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb3[ks], x3[i][ks], acc32[1][i], 0, 0, 0);
      }
      UVX_PHASE_END();
      // ---- section B: read XB (WA, WB stay in registers); DMA: [XA | WA | WB] of tile t+2; wait: REST(t+1) ----
#pragma unroll
      for (int i = 0; i < XH; ++i)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) x3[i][ks] = *reinterpret_cast<const bf16x8_t*>(set + xb_base + xo32[ks] + i * 32 * 128);
      if (t + NS < nks) {
#pragma unroll
        for (int k = 0; k < N234; ++k) dma(srest[k], t + NS, cs);
        UVX_VMCNT((NS - 1) * (N234 + N1));
      } else {
        UVX_VMCNT(0);
      }
      UVX_PHASE_SYNC();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int i = 0; i < XH; ++i) acc32[1][XH + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb3[ks], x3[i][ks], acc32[1][XH + i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < XH; ++i) acc32[0][XH + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa3[ks], x3[i][ks], acc32[0][XH + i], 0, 0, 0);
      }
      UVX_PHASE_END();
      cs = cs == NS - 1 ? 0 : cs + 1;
      continue;
    }
    if (PH == 2) {
      // ---- section A: read XA, WA, WB; DMA: XB of tile t+1; wait: XB(t) (issued one K-tile ago) ----
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
          if constexpr (NN) {
            wa[j][kh] = nn_frag(set + O_WA, j, kh);
            wb[j][kh] = nn_frag(set + O_WB, j, kh);
          } else {
            wa[j][kh] = *reinterpret_cast<const bf16x8_t*>(set + O_WA + wo[kh] + j * 16 * 128);
            wb[j][kh] = *reinterpret_cast<const bf16x8_t*>(set + O_WB + wo[kh] + j * 16 * 128);
          }
        }
#pragma unroll
      for (int i = 0; i < MA; ++i)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) xa[i][kh] = *reinterpret_cast<const bf16x8_t*>(set + xa_base + xo[kh] + i * 16 * 128);
      if (t + NS - 1 < nks) {
#pragma unroll
        for (int k = 0; k < N1; ++k) dma(sxb[k], t + NS - 1, ps);
        UVX_VMCNT((NS - 1) * (N234 + N1));
      } else {
        UVX_VMCNT(0);
      }
      UVX_PHASE_SYNC();
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < MA; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[j][kh], xa[i][kh], acc[j][i], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < MA; ++i) acc[2 + j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j][kh], xa[i][kh], acc[2 + j][i], 0, 0, 0);
      }
      UVX_PHASE_END();
      // ---- section B: read XB (WA, WB stay in registers); DMA: [XA | WA | WB] of tile t+2; wait: REST(t+1) ----
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) xa[i][kh] = *reinterpret_cast<const bf16x8_t*>(set + xb_base + xo[kh] + i * 16 * 128);
      if (t + NS < nks) {
#pragma unroll
        for (int k = 0; k < N234; ++k) dma(srest[k], t + NS, cs);
        UVX_VMCNT((NS - 1) * (N234 + N1));
      } else {
        UVX_VMCNT(0);
      }
      UVX_PHASE_SYNC();
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < MB; ++i) acc[2 + j][MA + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j][kh], xa[i][kh], acc[2 + j][MA + i], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < MB; ++i) acc[j][MA + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[j][kh], xa[i][kh], acc[j][MA + i], 0, 0, 0);
      }
      UVX_PHASE_END();
      cs = cs == NS - 1 ? 0 : cs + 1;
      continue;
    }
    // ---- phase 1: XA x WA; DMA: XB of tile t+1 ----
    UVX_TL_FLUSH(t - 1, 3);
    UVX_TL_TAKE(0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) wa[j][kh] = *reinterpret_cast<const bf16x8_t*>(set + O_WA + wo[kh] + j * 16 * 128);
#pragma unroll
    for (int i = 0; i < MA; ++i)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) xa[i][kh] = *reinterpret_cast<const bf16x8_t*>(set + xa_base + xo[kh] + i * 16 * 128);
    if (t + NS - 1 < nk) {
#pragma unroll
      for (int k = 0; k < N1; ++k) dma(sxb[k], t + NS - 1, ps);
    }
    UVX_PHASE_SYNC();
    UVX_TL_TAKE(1);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < MA; ++i)
          if (MODE != 1) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[j][kh], xa[i][kh], acc[j][i], 0, 0, 0);
          else asm volatile("" ::"v"(wa[j][kh]), "v"(xa[i][kh]));
    UVX_TL_TAKE(2);
    UVX_PHASE_END();
    // ---- phase 2: XA x WB; DMA: first third of [XA | WA | WB] of tile t+2 ----
    UVX_TL_FLUSH(t, 0);
    UVX_TL_TAKE(0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) wb[j][kh] = *reinterpret_cast<const bf16x8_t*>(set + O_WB + wo[kh] + j * 16 * 128);
    if (t + NS < nk) {
#pragma unroll
      for (int k = 0; k < N2; ++k) dma(srest[k], t + NS, cs);
    }
    UVX_PHASE_SYNC();
    UVX_TL_TAKE(1);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < MA; ++i)
          if (MODE != 1) acc[2 + j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j][kh], xa[i][kh], acc[2 + j][i], 0, 0, 0);
          else asm volatile("" ::"v"(wb[j][kh]), "v"(xa[i][kh]));
    UVX_TL_TAKE(2);
    UVX_PHASE_END();
    // ---- phase 3: XB x WB ----
    UVX_TL_FLUSH(t, 1);
    UVX_TL_TAKE(0);
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) xa[i][kh] = *reinterpret_cast<const bf16x8_t*>(set + xb_base + xo[kh] + i * 16 * 128);
    if (t + NS < nk) {
#pragma unroll
      for (int k = 0; k < N3; ++k) dma(srest[N2 + k], t + NS, cs);
    }
    UVX_PHASE_SYNC();
    UVX_TL_TAKE(1);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < MB; ++i)
          if (MODE != 1) acc[2 + j][MA + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j][kh], xa[i][kh], acc[2 + j][MA + i], 0, 0, 0);
          else asm volatile("" ::"v"(wb[j][kh]), "v"(xa[i][kh]));
    UVX_TL_TAKE(2);
    UVX_PHASE_END();
    // ---- phase 4: XB x WA (WA kept in registers); the counted wait that retires tile t+1 ----
    UVX_TL_FLUSH(t, 2);
    UVX_TL_TAKE(0);
    if (t + NS < nk) {
#pragma unroll
      for (int k = 0; k < N4; ++k) dma(srest[N2 + N3 + k], t + NS, cs);
      UVX_VMCNT(INFLIGHT);
    } else {
      UVX_VMCNT(0);
    }
    UVX_PHASE_SYNC();
    UVX_TL_TAKE(1);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < MB; ++i)
          if (MODE != 1) acc[j][MA + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[j][kh], xa[i][kh], acc[j][MA + i], 0, 0, 0);
          else asm volatile("" ::"v"(wa[j][kh]), "v"(xa[i][kh]));
    UVX_TL_TAKE(2);
    UVX_PHASE_END();
    cs = cs == NS - 1 ? 0 : cs + 1;
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();  // balance the other half's extra barrier
  if (PRIO >= 2) __builtin_amdgcn_s_setprio(0);
  __builtin_amdgcn_sched_barrier(0);

  if (MODE == 4) {   // probe: dump the stamps of the first four blocks instead of the tile
    UVX_TL_FLUSH(nk - 1, 3);
    float keep = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < MI; ++i) keep += acc[j][i][0] + acc[j][i][1] + acc[j][i][2] + acc[j][i][3];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (blockIdx.x < 4) {
      unsigned* out = reinterpret_cast<unsigned*>(p.C) + ((long long)blockIdx.x * 8 + w) * (TL_TILES * 12);
      for (int i = lane; i < TL_TILES * 12; i += 64) out[i] = i < nk * 12 ? tl_base[i] : 0u;
      if (keep == 12345.678f) out[0] = 1u;
    }
    return;
  }
  if (MODE == 3) {   // probe: no epilogue (keeps the accumulators alive with one conditional store)
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < MI; ++i) t += acc[j][i][0] + acc[j][i][1] + acc[j][i][2] + acc[j][i][3];
    if (t == 12345.678f) *reinterpret_cast<float*>(p.C) = t;
    return;
  }
  if (SK) {
    constexpr int SLOT_V4 = 4 * MI * 512;                       // float4 per block slot: [j][i][thread]
    float4* slots = reinterpret_cast<float4*>(p.sk_ws);
    if (kbase != 0) {   // contributor: partial sums to this block's slot, then publish
      float4* slot = slots + (size_t)blockIdx.x * SLOT_V4 + tid;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) slot[(j * MI + i) * 512] = make_float4(acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]);
      __threadfence();                                            // each thread's stores are visible device-wide ...
      __syncthreads();                                            // ... before thread 0 says so
      if (tid == 0) __hip_atomic_store(p.sk_flags + blockIdx.x, p.sk_epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (nks != nk) {  // owner of a split tile: the blocks that continue it are the next ones of this XCD's list
        const int tile_end = (sk_lt + 1) * nk;
        for (int c = sk_wl + 1; c < sk_per; ++c) {
          const int clo = (int)((long long)c * sk_I / sk_per), chi = (int)((long long)(c + 1) * sk_I / sk_per);
          if (clo >= tile_end) break;
          if (chi == clo) continue;                               // (an empty share contributes nothing)
          const int g = (c << 3) + sk_x;
          if (tid == 0) {
            int spins = 0;
            while (__hip_atomic_load(p.sk_flags + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.sk_epoch) {
              __builtin_amdgcn_s_sleep(8);
              if (++spins > (1 << 21)) { atomicAdd(p.sk_flags + gridDim.x, 1u); break; }   // ~1 s: count it, do not hang
            }
          }
          __syncthreads();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // every wave: nothing below is served from a stale line
          const float4* slot = slots + (size_t)g * SLOT_V4 + tid;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
              const float4 v = slot[(j * MI + i) * 512];
              acc[j][i][0] += v.x; acc[j][i][1] += v.y; acc[j][i][2] += v.z; acc[j][i][3] += v.w;
            }
            __builtin_amdgcn_sched_barrier(0);   // MI loads in flight at a time: the accumulators already fill half the file
          }
        }
      }
      __syncthreads();   // every wave is done with the staged operand tiles: the LDS becomes the output stage
      store_tile<4, MI>(p, acc, m0 + wr * (BM / 2), n0 + wc * 64, frow, fg, z, lds + w * (MI * 16 * 128));
    }
    if (!sk_next()) break;
    __syncthreads();     // the output stage is the next piece's operand buffer
    fill_slots(sxb, srest);
    issue_prologue(sxb, srest);
    continue;
  }
  if (!PERSIST) {
    __syncthreads();   // every wave is done with the staged operand tiles: the LDS becomes the output stage
    if constexpr (M32) store_tile32<(MI + 1) / 2>(p, acc32, m0 + wr * (BM / 2), n0 + wc * 64, lane, z, lds + w * (MI * 16 * 128));
    else store_tile<4, MI, MI, ACT23>(p, acc, m0 + wr * (BM / 2), n0 + wc * 64, frow, fg, z, lds + w * (MI * 16 * 128));
    break;
  }
  // every wave has passed its last fragment read (the barriers above): the operand buffers are free.  Start the next
  // tile's pipeline fill, THEN write this tile out through the per-wave 2 KiB stage.
  const int m0_cur = m0, n0_cur = n0;
  int next = orig + gridDim.x;
  const bool has_next = next_tile(next, m0, n0);
  if (has_next) {
    fill_slots(sxb, srest);
    issue_prologue(sxb, srest);
  }
  __builtin_amdgcn_sched_barrier(0);
  store_tile<4, MI, 1>(p, acc, m0_cur + wr * (BM / 2), n0_cur + wc * 64, frow, fg, z, lds + O_STAGE + w * 2048);
  if (!has_next) break;
  orig = next;
  }
#undef UVX_VMCNT
#undef UVX_PHASE_SYNC
#undef UVX_PHASE_END
#undef UVX_TL_TAKE
#undef UVX_TL_FLUSH
}

#ifdef UVX_PROBES
// ------------------------------------------------------------------------------------------------
// Four-wave kernel with a hand-scheduled K loop (round 4): (32 MI) x 256 x 64 tile, 256 threads = 4 waves as 2 x 2, one wave
// per SIMD, (16 MI) x 128 per wave = 8 x MI accumulator fragments in AGPRs (a[0 : 32 MI)), all 160 KiB of LDS as a ring of
// five 32 KiB operand slots filled by LDS-DMA.  The whole K loop - pipeline fill, steady state, tails, drain - is ONE
// inline-asm statement generated by tools/gen_gemm_a4.py (gemm_a4_loop.inc): the 8-wave kernels above spend ~2800 cycles
// per K-tile against the 2048 of its MFMAs because each wave's load section (fragment reads + DMA issue + barrier hand-off)
// has to hide under 32 MFMAs of its SIMD partner; here a wave reads (MI + 8) x 2 fragments for 16 MI MFMAs (half the LDS
// bytes per MFMA of the 2 x 4 geometry), every ds_read / DMA issue sits in an MFMA's shadow by construction, and there is one
// barrier per K-tile.  hipcc cannot schedule this loop (the q4 probe kernel: 570 TF/s, 256 accumulators shuffled through
// v_accvgpr moves), hence the asm.  Same MFMA, operand roles, swizzle and k order as the rest of the family: results are
// bit-identical across tile variants.  What stays in C++: tile mapping, address set-up, and the epilogue, which reads the
// accumulators out of the named AGPRs (a4_acc) in passes of two row fragments through a per-wave LDS stage.
// The asm addresses LDS absolutely: the kernel's single __shared__ array is the ring (its base is folded into the operands).
#include "gemm_a4_loop.inc"

template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// accumulator fragment at a[R : R + 3] (the K loop's literal registers; volatile keeps the reads behind the loop statement)
template <int R>
__device__ __forceinline__ f32x4_t a4_acc() {
  float x, y, z, w;
  asm volatile("v_accvgpr_read_b32 %0, a[%c4]\n\tv_accvgpr_read_b32 %1, a[%c5]\n\tv_accvgpr_read_b32 %2, a[%c6]\n\tv_accvgpr_read_b32 %3, a[%c7]"
               : "=v"(x), "=v"(y), "=v"(z), "=v"(w) : "n"(R), "n"(R + 1), "n"(R + 2), "n"(R + 3));
  return f32x4_t{x, y, z, w};
}

// Epilogue of the hand-scheduled kernels: NJ x MI fragments per wave, fragment (j, i) = output columns n_base + 16 j + 4 fg .. + 3,
// output row m_base + 16 i + frow (the family's convention, store_tile), accumulators read out of the AGPRs two row fragments at
// a time.  bf16 output goes through the wave's LDS stage (32 rows x 32 NJ bytes): fragments in (16-byte chunks XOR-swizzled by
// row), whole rows out - 2 NJ lanes = one row of the wave tile, contiguous in memory - with the residual added on the way; rounding
// points as store_tile (after bias, after the activation, after the residual add).  The host side (a4_applicable) only routes
// launches here whose pointers / strides allow the 16-byte accesses.
template <int NJ, int MI>
__device__ __forceinline__ void store_tile_a4(const GemmArgs& p, int m_base, int n_base, int lane, long long z, char* stage) {
  constexpr int CH = 2 * NJ, RPI = 64 / CH;     // 16-byte chunks per stage row; rows per read instruction
  const int frow = lane & 15, fg = lane >> 4;
  if (p.out_f32) {     // f32 (wgrad / split-K partials): fragment layout is already 16 bytes per lane
    static_for<NJ>([&](auto J) {
      constexpr int j = decltype(J)::value;
      const int n = n_base + j * 16 + fg * 4;
      static_for<MI>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const f32x4_t a = a4_acc<(j * MI + i) * 4>();
        const int m = m_base + i * 16 + frow;
        if (m < p.M && n < p.N) {
          float4* dst = reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + z * p.sC + (long long)m * p.ldc + n);
          float4 v = make_float4(a[0] * p.alpha, a[1] * p.alpha, a[2] * p.alpha, a[3] * p.alpha);
          if (p.accumulate) { const float4 c = *dst; v.x += c.x; v.y += c.y; v.z += c.z; v.w += c.w; }
          *dst = v;
        }
      });
    });
    return;
  }
  float bv[NJ][4];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = n_base + j * 16 + fg * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[j][e] = 0.f;
    if (p.bias && n < p.N) {
      const u16x4_t b4 = *reinterpret_cast<const u16x4_t*>(p.bias + n);
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[j][e] = bf2f(b4[e]);
    }
  }
  constexpr int NP = (MI + 1) / 2;
  static_for<NP>([&](auto P) {
    constexpr int i0 = 2 * decltype(P)::value;
    constexpr int gi = (MI - i0) < 2 ? (MI - i0) : 2;            // row fragments in this pass
    if (i0 > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the previous pass's reads are done (WAR on the stage)
    // ---- pass 1: [16 gi x 16 NJ] of C into the stage ----
    static_for<NJ>([&](auto J) {
      constexpr int j = decltype(J)::value;
      static_for<gi>([&](auto II) {
        constexpr int ii = decltype(II)::value;
        const f32x4_t a = a4_acc<(j * MI + i0 + ii) * 4>();
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = bf2f(f2bf(a[e] * p.alpha + bv[j][e]));
          if (p.act == 1) t = bf2f(f2bf(gelu_fast(t)));
          v[e] = t;
        }
        *reinterpret_cast<uint2*>(stage_slot<32, CH>(stage, ii * 16 + frow, 2 * j + (fg >> 1)) + (fg & 1) * 8) =
            make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
      });
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    {
      const int cc = lane & (CH - 1), n8 = n_base + cc * 8;
#pragma unroll
      for (int it = 0; it < gi * 16 / RPI; ++it) {
        const int row = it * RPI + lane / CH, m = m_base + i0 * 16 + row;
        uint4 o = *reinterpret_cast<const uint4*>(stage_slot<32, CH>(stage, row, cc));
        if (m < p.M && n8 < p.N) {
          if (p.residual) {
            const int rm = p.res_mod > 0 ? (m % p.res_mod) : m;
            const uint4 r = *reinterpret_cast<const uint4*>(p.residual + z * p.sR + (long long)rm * p.ldr + n8);
            o.x = pack2(unpack_lo(o.x) + unpack_lo(r.x), unpack_hi(o.x) + unpack_hi(r.x));
            o.y = pack2(unpack_lo(o.y) + unpack_lo(r.y), unpack_hi(o.y) + unpack_hi(r.y));
            o.z = pack2(unpack_lo(o.z) + unpack_lo(r.z), unpack_hi(o.z) + unpack_hi(r.z));
            o.w = pack2(unpack_lo(o.w) + unpack_lo(r.w), unpack_hi(o.w) + unpack_hi(r.w));
          }
          *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + z * p.sC + (long long)m * p.ldc + n8) = o;
        }
      }
    }
    if (p.swiglu == 1) {
      // ---- pass 2 (fused SwiGLU): act = round(silu(gate)) * up, [16 gi x 8 NJ] -> C2; gate = fragment j (even), up = j + 1 ----
      constexpr int CH2 = NJ, RPI2 = 64 / CH2;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // pass-1 reads done before the stage is overwritten
      static_for<NJ / 2>([&](auto J2) {
        constexpr int j = 2 * decltype(J2)::value;
        static_for<gi>([&](auto II) {
          constexpr int ii = decltype(II)::value;
          const f32x4_t ag = a4_acc<(j * MI + i0 + ii) * 4>();
          const f32x4_t au = a4_acc<((j + 1) * MI + i0 + ii) * 4>();
          float a[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float g = bf2f(f2bf(ag[e] * p.alpha));
            a[e] = bf2f(f2bf(g / (1.0f + __expf(-g)))) * bf2f(f2bf(au[e] * p.alpha));
          }
          *reinterpret_cast<uint2*>(stage_slot<32, CH2>(stage, ii * 16 + frow, j + (fg >> 1)) + (fg & 1) * 8) =
              make_uint2(pack2(a[0], a[1]), pack2(a[2], a[3]));
        });
      });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int cc = lane & (CH2 - 1), n8 = n_base / 2 + cc * 8;
#pragma unroll
      for (int it = 0; it < gi * 16 / RPI2; ++it) {
        const int row = it * RPI2 + lane / CH2, m = m_base + i0 * 16 + row;
        const uint4 o = *reinterpret_cast<const uint4*>(stage_slot<32, CH2>(stage, row, cc));
        if (m < p.M && 2 * n8 < p.N) *reinterpret_cast<uint4*>(p.C2 + (long long)m * p.ldc2 + n8) = o;
      }
    }
  });
}

#define UVX_A4_OPERANDS                                                                                                      \
  [xo0] "v"(xo[0]), [xo1] "v"(xo[1]), [xo2] "v"(xo[2]), [xo3] "v"(xo[3]), [xo4] "v"(xo[4]), [xo5] "v"(xo[5]),                 \
  [xo6] "v"(xo[6]), [xo7] "v"(xo[7]), [wo0] "v"(wo[0]), [wo1] "v"(wo[1]), [wo2] "v"(wo[2]), [wo3] "v"(wo[3]),                 \
  [wo4] "v"(wo[4]), [wo5] "v"(wo[5]), [wo6] "v"(wo[6]), [wo7] "v"(wo[7]), [sAlo] "s"(a_lo), [sAhi] "s"(a_hi),                 \
  [sBlo] "s"(b_lo), [sBhi] "s"(b_hi), [nk] "s"(nk), [vx0] "v"(vx[0]), [vx1] "v"(vx[1]), [vw0] "v"(vw[0]), [vw1] "v"(vw[1]),   \
  [wofs] "s"(wofs)
#define UVX_A8_OPERANDS                                                                                                      \
  [xo0] "v"(xo[0]), [xo1] "v"(xo[1]), [xo2] "v"(xo[2]), [xo3] "v"(xo[3]), [wo0] "v"(wo[0]), [wo1] "v"(wo[1]),                 \
  [wo2] "v"(wo[2]), [wo3] "v"(wo[3]), [sAlo] "s"(a_lo), [sAhi] "s"(a_hi),                                                     \
  [sBlo] "s"(b_lo), [sBhi] "s"(b_hi), [nk] "s"(nk), [vx0] "v"(vx[0]), [vx1] "v"(vx[1]), [vw0] "v"(vw[0]), [vw1] "v"(vw[1]),   \
  [wofs] "s"(wofs)

// NWN = waves along N: 2 = four waves (one per SIMD, (16 MI) x 128 each), 4 = eight waves (two per SIMD, (16 MI) x 64 each).
// SCHED selects the generated interleave (gemm_a4_loop.inc; 0 = production, others: probe builds).
template <int NWN, int MI, int SCHED = 0>
__global__ __launch_bounds__(128 * NWN, NWN / 2) void gemm_nt_bf16_a4_kernel(GemmArgs p) {
  static_assert(MI == 8 && (NWN == 2 || NWN == 4), "generated loops exist for MI = 8");
  constexpr int BMT = 32 * MI, BNW = 256, WAVES = 2 * NWN, NJ = 16 / NWN, NX = 4 * MI / WAVES, NW = 32 / WAVES;
  __shared__ __attribute__((aligned(16))) char lds[5 * 32768];

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w / NWN, wc = w % NWN;
  if (p.m_dev) p.M = min(p.M, max(*p.m_dev - p.m_dev_off, 0));   // rows known only on the device: clamp M
  // XCD-aware bijective remap of the dispatch order (see the eight-phase kernel's tile_origin)
  const int ntiles = p.tiles_m * p.tiles_n, o = blockIdx.x;
  const int xcd = o & 7, q8 = ntiles >> 3, r8 = ntiles & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (o >> 3);
  int m0, n0;
  if (p.m_major) { m0 = (wg / p.tiles_n) * BMT; n0 = (wg % p.tiles_n) * BNW; }
  else { m0 = (wg % p.tiles_m) * BMT; n0 = (wg / p.tiles_m) * BNW; }
  if (m0 >= p.M) return;   // (before any barrier)
  const long long z = blockIdx.y;
  const char* A = reinterpret_cast<const char*>(p.A + z * p.sA + (long long)m0 * p.lda);
  const char* B = reinterpret_cast<const char*>(p.B + z * p.sB + (long long)n0 * p.ldb);
  const unsigned a_lo = (unsigned)(uintptr_t)A, a_hi = (unsigned)((uintptr_t)A >> 32);
  const unsigned b_lo = (unsigned)(uintptr_t)B, b_hi = (unsigned)((uintptr_t)B >> 32);

  // DMA instruction i of wave w covers unit rows 8 (w + WAVES i) .. + 7: lane -> row + (lane >> 3), LDS chunk position lane & 7
  // holds source chunk (lane & 7) ^ (row & 7); rows past the matrix edge re-read its last row (results are masked at the store)
  const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
  unsigned xo[NX], wo[NW];
#pragma unroll
  for (int i = 0; i < NX; ++i) xo[i] = (unsigned)min(8 * (w + WAVES * i) + srow, p.M - 1 - m0) * (unsigned)(p.lda * 2) + schunk * 16;
#pragma unroll
  for (int i = 0; i < NW; ++i) wo[i] = (unsigned)min(8 * (w + WAVES * i) + srow, p.N - 1 - n0) * (unsigned)(p.ldb * 2) + schunk * 16;
  // fragment reads: X fragment i = slot + vx[kh] + 2048 i, W fragment j = slot + vw[kh] + 2048 j
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const int frow = lane & 15, fg = lane >> 4;
  unsigned vx[2], vw[2];
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) {
    const unsigned c = ((fg + 4 * kh) ^ (frow & 7)) * 16;
    vx[kh] = lds_base + (wr * (16 * MI) + frow) * 128 + c;
    vw[kh] = lds_base + (wc * (16 * NJ) + frow) * 128 + c;
  }
  const unsigned wofs = lds_base + w * 1024;
  const int nk = p.K / 64;
#define UVX_AX_LOOP(G, S) asm volatile(UVX_##G##_MI8_LOOP_S##S : : UVX_##G##_OPERANDS : UVX_##G##_MI8_CLOBBER)
  if constexpr (NWN == 2) {
    if constexpr (SCHED == 0) UVX_AX_LOOP(A4, 1);
#ifdef UVX_PROBES
    if constexpr (SCHED == 1) UVX_AX_LOOP(A4, 0);
    if constexpr (SCHED == 2) UVX_AX_LOOP(A4, 2);
    if constexpr (SCHED == 3) UVX_AX_LOOP(A4, 3);
    if constexpr (SCHED == 4) UVX_AX_LOOP(A4, 4);
    if constexpr (SCHED == 5) UVX_AX_LOOP(A4, 5);
#endif
  } else {
    if constexpr (SCHED == 0) UVX_AX_LOOP(A8, 0);
#ifdef UVX_PROBES
    if constexpr (SCHED == 1) UVX_AX_LOOP(A8, 1);
    if constexpr (SCHED == 2) UVX_AX_LOOP(A8, 2);
    if constexpr (SCHED == 3) UVX_AX_LOOP(A8, 3);
    if constexpr (SCHED == 4) UVX_AX_LOOP(A8, 4);
    if constexpr (SCHED == 5) UVX_AX_LOOP(A8, 5);
#endif
  }
#undef UVX_AX_LOOP
  store_tile_a4<NJ, MI>(p, m0 + wr * (16 * MI), n0 + wc * (16 * NJ), lane, z, lds + w * (32 * 32 * NJ));
}

// Eight waves, ping-pong by wave row (gen_gemm_a4.py "a8pp"): same tile / ring / fragments as the eight-wave kernel above, but a
// wave alternates a PURE MFMA phase (64 MFMAs = one K-tile, fragments of the whole K-tile in registers) with a load phase (24
// fragment reads + 8 LDS-DMA instructions), the two waves of a SIMD run half a period apart, and the DMA is split by operand:
// wave row 0 issues the activation units, wave row 1 the weight units.  One barrier per K-tile.
template <int SCHED = 0>
__global__ __launch_bounds__(512, 2) void gemm_nt_bf16_a8pp_kernel(GemmArgs p) {
  constexpr int MI = 8, NJ = 4, BMT = 256, BNW = 256;
  __shared__ __attribute__((aligned(16))) char lds[5 * 32768];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 2, wc = w & 3;
  if (p.m_dev) p.M = min(p.M, max(*p.m_dev - p.m_dev_off, 0));   // rows known only on the device: clamp M
  const int ntiles = p.tiles_m * p.tiles_n, o = blockIdx.x;
  const int xcd = o & 7, q8 = ntiles >> 3, r8 = ntiles & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (o >> 3);
  int m0, n0;
  if (p.m_major) { m0 = (wg / p.tiles_n) * BMT; n0 = (wg % p.tiles_n) * BNW; }
  else { m0 = (wg % p.tiles_m) * BMT; n0 = (wg / p.tiles_m) * BNW; }
  if (m0 >= p.M) return;   // (before any barrier)
  const long long z = blockIdx.y;
  // this wave row's DMA operand: row 0 = activations (A rows m0 ..), row 1 = weights (B rows n0 ..)
  const char* G = wr == 0 ? reinterpret_cast<const char*>(p.A + z * p.sA + (long long)m0 * p.lda)
                          : reinterpret_cast<const char*>(p.B + z * p.sB + (long long)n0 * p.ldb);
  const unsigned gp_lo = (unsigned)(uintptr_t)G, gp_hi = (unsigned)((uintptr_t)G >> 32);
  const int lim = wr == 0 ? p.M - 1 - m0 : p.N - 1 - n0;
  const unsigned ldg = (unsigned)((wr == 0 ? p.lda : p.ldb) * 2);
  // DMA instruction i of wave wc (of its row) covers unit rows 8 (wc + 4 i) .. + 7 (see the four-wave kernel)
  const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
  unsigned go[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) go[i] = (unsigned)min(8 * (wc + 4 * i) + srow, lim) * ldg + schunk * 16;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const int frow = lane & 15, fg = lane >> 4;
  unsigned vx[2], vw[2];
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) {
    const unsigned c = ((fg + 4 * kh) ^ (frow & 7)) * 16;
    vx[kh] = lds_base + (wr * 128 + frow) * 128 + c;
    vw[kh] = lds_base + (wc * 64 + frow) * 128 + c;
  }
  const unsigned wofs = lds_base + wc * 1024;
  const int nk = p.K / 64;
#define UVX_PP_LOOP(S)                                                                                                         \
  asm volatile(UVX_A8PP_MI8_LOOP_S##S : : [go0] "v"(go[0]), [go1] "v"(go[1]), [go2] "v"(go[2]), [go3] "v"(go[3]), [go4] "v"(go[4]), \
               [go5] "v"(go[5]), [go6] "v"(go[6]), [go7] "v"(go[7]), [gplo] "s"(gp_lo), [gphi] "s"(gp_hi), [nk] "s"(nk),       \
               [vx0] "v"(vx[0]), [vx1] "v"(vx[1]), [vw0] "v"(vw[0]), [vw1] "v"(vw[1]), [wofs] "s"(wofs), [wr] "s"(wr)            \
               : UVX_A8PP_MI8_CLOBBER)
  if constexpr (SCHED == 0) UVX_PP_LOOP(0);
#ifdef UVX_PROBES
  if constexpr (SCHED == 1) UVX_PP_LOOP(1);
  if constexpr (SCHED == 2) UVX_PP_LOOP(2);
  if constexpr (SCHED == 3) UVX_PP_LOOP(3);
#endif
#undef UVX_PP_LOOP
  store_tile_a4<NJ, MI>(p, m0 + wr * 128, n0 + wc * 64, lane, z, lds + w * 4096);
}
#endif  // UVX_PROBES (hand-scheduled K loops: a record of round 4, never picked; the product library does not carry them)

// Tile choice.  Every CU works through ~tiles/256 rounds of tiles (see variant_cost for the partial last
// round); a tile costs BM x BN / speed(variant), speeds measured on
// MI355X (profiles/r01_gemm_variants.txt).  variant 0 = 128x128 narrow; 1..4 = {128,160,192,256} x 256 wide.
// (A 32x32x16-MFMA flavour of the wide kernel was measured 10-20 % SLOWER than 16x16x32 and dropped.)
struct Variant { int bm, bn; double speed; double c; };
// Tile variants: 0 = 128x128 (4 waves, several blocks per CU); 1..4 = double-buffered {128,160,192,256} x 256;
// 5..8 = ping-pong k32 ring {128,160,192,256} x 256; 9, 10 = three-buffer {128,160} x 256; 11, 15, 16, 17 = eight-phase
// {256,160,192,128} x 256; 12 = 4-wave kernel with 128x128 wave tiles (probe only: hipcc shuffles its 256
// accumulators through v_accvgpr moves, 570 TF); 13, 14 = probe modes of 8.  (Also measured and dropped,
// profiles/r01_gemm_variants.txt: a 32x32x16-MFMA flavour, 10-20 % slower.)
//
// Cost model.  time ~ rounds(tiles) * bm * bn * (K/64 + c) / speed: `speed` is the asymptotic TF/s of a full round of
// tiles, `c` the fixed per-tile cost (pipeline fill, epilogue) in K-tiles.  Both are fitted to
// tools/gpu_gemm_cold_probe.py (profiles/r01_gemm_cold_probe*.txt), where every launch reads a DIFFERENT weight matrix
// as the training step does: 16 GB of frozen weights per pass never sit in the 256 MB Infinity Cache, and a
// back-to-back probe on one weight buffer overstates the shallow-prefetch kernels by 10-25 % and ranks them wrongly.
// speed 0 = probe only.
constexpr int kNumVariants = 63;
const Variant kVariants[kNumVariants] = {
    {128, 128, 880., 2.},   {128, 256, 935., 4.75}, {160, 256, 1020., 4.75}, {192, 256, 1024., 4.75}, {256, 256, 1250., 8.7},
    {128, 256, 980., 9.},   {160, 256, 1106., 9.},  {192, 256, 1118., 9.},   {256, 256, 1283., 9.3},  {128, 256, 0., 9.},
    {160, 256, 1162., 8.9}, {256, 256, 1380., 8.5}, {256, 256, 0., 9.},      {256, 256, 0., 9.},      {256, 256, 0., 9.},
    {160, 256, 1230., 9.},  {192, 256, 1390., 12.}, {128, 256, 1116., 6.},
    {160, 256, 1245., 9.},  {128, 256, 0., 6.},   {256, 256, 0., 9.},   {256, 256, 0., 9.},   {256, 256, 0., 9.},   // 20..22 = probe modes of 11
    {256, 256, 0., 4.},     {192, 256, 0., 6.},     {160, 256, 0., 5.},     {128, 256, 0., 4.},
    {256, 256, 0., 9.},    // 27 = timeline probe of 11 (MODE 4)
    {256, 256, 0., 9.},     {256, 256, 0., 9.},     {256, 256, 0., 9.},    // 28..30 = issue-priority probes of 11 (MODE 5..7: none / load section / MFMA section at priority 1)
    // 31..34 = merged-phase (PH = 2) {256,192,160,128} x 256: the production set since round 2.  Speeds = their four-phase
    // twins' x the same-box cold-probe ratio (profiles/r02_gemm_ph2_probe.txt: 256: +3...+10 %, 192: +0...+4 %, 160: +0...+4 %
    // over the three-buffer 18, 128: +2...+5 %); fixed costs as the twins' except 192 (10.5 instead of 12: the in-situ table of the
    // first merged-phase run had 12000 x 3072 x 1024 on the 256 tile at 108 us where the 192 tile takes 87).
    // (160-row tile: 1285 -> 1305 after the in-situ A/B of profiles/r02_gemm_k28672_tile_ab.txt - at 2528 x 4096 x 28672 the model
    //  had the 192-row tile (224 tiles, a 7/8-filled round) 0.1 % ahead, in the step the 160-row tile (256 tiles) is 10 % faster
    //  on that shape: 91.75 -> 90.2 ms per step; the nudge flips only the K >= 24576 dgrad shapes of C2 / C4 / C5)
    {256, 256, 1470., 8.5}, {192, 256, 1430., 10.5}, {160, 256, 1305., 9.},  {128, 256, 1160., 6.},
    {256, 256, 0., 6.},     {192, 256, 0., 8.},      {160, 256, 0., 7.},     {128, 256, 0., 5.},    // 35..38 = PERSISTENT merged-phase (probe)
    // 39..42 = STREAM-K merged-phase {256,192,160,128} x 256 (probe builds; see the kernel).  Cost: sk_cost() below.
    {256, 256, 1470., 8.5}, {192, 256, 1430., 10.5}, {160, 256, 1285., 9.},  {128, 256, 1160., 6.},
    // 43 = FOUR-WAVE kernel with the hand-scheduled K loop, 256 x 256 (round 4; see gemm_nt_bf16_a4_kernel); 44, 45 = other
    // interleaves of the same loop (probe builds)
    {256, 256, 0., 7.},     {256, 256, 0., 7.},      {256, 256, 0., 7.},
    {256, 256, 0., 7.},     {256, 256, 0., 7.},      {256, 256, 0., 7.},    // 46..48 = timing probes of 43: no in-loop DMA / nor fragment reads / no MFMAs
    // 49 = EIGHT-WAVE kernel with the hand-scheduled K loop (two free-running waves per SIMD), 256 x 256; 50, 51 = other interleaves;
    // 52..54 = its timing probes
    {256, 256, 0., 7.},     {256, 256, 0., 7.},      {256, 256, 0., 7.},
    {256, 256, 0., 7.},     {256, 256, 0., 7.},      {256, 256, 0., 7.},
    // 55 = eight waves, PING-PONG by wave row (pure MFMA phase / load phase, DMA split by operand); 56..58 = its timing probes
    {256, 256, 0., 7.},     {256, 256, 0., 7.},      {256, 256, 0., 7.},     {256, 256, 0., 7.},
    // 59, 60 = merged-phase {160,128} x 256 with THREE buffer sets (round 5): twice the DMA look-ahead, for problems whose weights all come
    // from HBM (the prefill).  Measured within +-2 % of their two-set twins 33 / 34 at every prefill shape and split factor
    // (profiles/r05_gemm_splitk_probe_three_buffer_sets.txt), so NEITHER picker takes them: speed 0 keeps them out of pick_variant, and
    // pick_split's filter admits only 0 and 31..34 unless a variant is forced.  They stay in the product library as forced / override-table
    // choices (uvx_gemm_force_variant, uvx_gemm_override_variant) with their bit-identity test against the twins.
    {160, 256, 0., 9.},     {128, 256, 0., 6.},
    // 61, 62 = merged-phase {256,128} x 256 on 32 x 32 x 16 MFMAs (round 6, M32; plain epilogues - anything else runs the twin 31 / 34).
    // Speed 0: forced / override-table choices only (profiles/r06_gemm_mfma32_probe.txt).
    {256, 256, 0., 8.5},    {128, 256, 0., 6.}};
// (round 5, tried: the merged-phase kernel at 320 x 256 - both 160-row tiles of a 316-row prompt in ONE block, every weight byte staged once
//  per K-tile for all rows.  160 accumulator + 72 fragment registers per wave leave hipcc 35 spills at two waves per SIMD, and the spilled
//  registers are the DMA source pointers: each reload sits behind an s_waitcnt vmcnt(0) inside the K loop, which drains the LDS-DMA
//  pipeline the kernel lives on.  Not built.)
   // 23..26 = persistent eight-phase {256,192,160,128} x 256   // 18, 19 = eight-phase {160,128} x 256 with three buffer sets (+1-2 % on single-round shapes)
// The production set: 128x128 (0) and the eight-phase kernels (11, 15..19).  Everything else is a superseded family or a
// probe build of the eight-phase kernel and exists only in libuvx_probes.so (-DUVX_PROBES); the picker never selects it.
// (59, 60: built, never picked - see the table)
constexpr bool is_production(int v) { return v == 0 || v == 11 || (v >= 15 && v <= 19) || (v >= 31 && v <= 34) || (v >= 59 && v <= 62); }
constexpr bool is_a4(int v) { return v >= 43 && v <= 58; }
constexpr bool is_streamk(int v) { return v >= 39 && v <= 42; }
bool variant_available(int v) {
#ifdef UVX_PROBES
  return v >= 0 && v < kNumVariants;
#else
  return v >= 0 && v < kNumVariants && is_production(v);
#endif
}
// `gelu`: the launch carries the bias + GELU epilogue (the encoder's fc1).  Its cost grows with the tile's area and is paid by every CU at
// the same time; in situ the 192-row tile is 13 % faster than the 256-row tile on 12000 x 4096 x 1024 (128 vs 147-150 us, and the fc2 GEMM
// behind it 86 vs 97 us; whole step -0.6 ms: profiles/r04_gemm_fc1_tile_ab.txt) where the plain model had the 256-row tile 10 % ahead:
// three more K-tiles of fixed cost for the 256-row tiles under that epilogue.
double variant_cost(int v, int M, int N, int K, int batch, bool gelu = false) {
  const double tiles = (double)cdiv(M, kVariants[v].bm) * cdiv(N, kVariants[v].bn) * batch;
  // Rounds of tiles over the 256 CUs.  A partly filled last round is cheaper than a full one (the kernels are bound
  // by operand delivery through the shared L2 / fabric, so fewer active CUs each run faster) but never cheaper
  // than ~0.55 of one: measured behaviour is well described by max(0.55, frac^0.6).
  const double r = tiles / 256.0, frac = r - floor(r);
  const double rounds = floor(r) + (frac > 0. ? fmax(0.55, pow(frac, 0.6)) : 0.);
  const double c = kVariants[v].c + (gelu && kVariants[v].bm == 256 ? 3.0 : 0.0);
  return rounds * kVariants[v].bm * kVariants[v].bn * (K / 64.0 + c) / (kVariants[v].speed > 0. ? kVariants[v].speed : 1300.);   // (speed 0: forced probe tiles)
}
// Stream-K launch geometry: one block per CU; `full` data-parallel rounds, the rest of the tiles shared out by K-tiles.
int sk_grid() {
  static int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
    return n;
  }();
  return cus >= 64 && cus % 8 == 0 ? cus : 0;     // 0: stream-K unavailable (the XCD-local work lists need a multiple of 8)
}
int sk_full_rounds(long long tiles, int grid) {
  if (tiles < grid) return 0;
  const int rounds = (int)(tiles / grid);
  return uvx::g_options[9] ? rounds - 1 : 0;
}
// time ~ (every block's share of the K-tiles + per-piece fixed costs) x tile area / speed.  A block runs `full` whole
// tiles plus a share that touches 2 (sometimes 3) tiles: one more pipeline fill than the data-parallel kernel, plus writing /
// re-reading one f32 partial (~2.5 K-tiles' worth of time on a 256 x 256 tile).  `kSkSlow`: with every CU busy the chip
// clocks lower than through a partly filled round (power limit) - fitted to profiles/r02_gemm_streamk_probe.txt.
constexpr double kSkSlow = 1.06, kSkFix = 2.5;
double sk_cost(int v, int M, int N, int K, int grid) {
  const Variant& V = kVariants[v];
  const double tiles = (double)cdiv(M, V.bm) * cdiv(N, V.bn), nk = K / 64.0;
  const double full = sk_full_rounds((long long)tiles, grid);
  const double share = (tiles - full * grid) * nk / grid;          // K-tiles of the stream-K part, per block
  const double pieces = share / nk + 1.0;                          // ~ pieces per block in the stream-K part
  return (full * (nk + V.c) + share + pieces * V.c + kSkFix) * V.bm * V.bn / V.speed * kSkSlow;
}
// probes: the tile variant forced for this shape (uvx_gemm_force_variant / uvx_gemm_override_variant), or -1
int forced_variant(int M, int N, int K) {
  int forced = uvx::g_gemm_variant;
  for (int i = 0; i < uvx::g_gemm_ovr_n; ++i)
    if (uvx::g_gemm_ovr[i][0] == M && uvx::g_gemm_ovr[i][1] == N && uvx::g_gemm_ovr[i][2] == K) forced = uvx::g_gemm_ovr[i][3];
  return forced;
}
int pick_variant(int M, int N, int K, int batch, double* cost_out = nullptr, bool gelu = false) {
  const int forced = forced_variant(M, N, K);
  double best = 1e30;
  int best_v = 0;
  if (forced >= 0 && forced < kNumVariants) {   // probe-only variants carry speed 0: never let the cost model veto them
    if (cost_out) *cost_out = kVariants[forced].speed > 0. ? variant_cost(forced, M, N, K, batch, gelu) : 0.;
    return forced;
  }
#ifdef UVX_PROBES
  const int skg = batch == 1 && uvx::g_options[8] && uvx::g_options[6] ? sk_grid() : 0;
#else
  const int skg = 0;
#endif
  for (int v = 0; v < kNumVariants; ++v) {
    if (kVariants[v].speed <= 0. || !(is_production(v) || (skg && is_streamk(v)))) continue;
    // option 6 (default 1): merged-phase kernels 31..34 replace their four-phase twins 11, 15..19 (0 = the round-1 set, for A/B)
    if (v != 0 && ((v >= 31) != (uvx::g_options[6] != 0))) continue;
    double cost;
    if (is_streamk(v)) {
      // (a launch whose tiles fill whole rounds has nothing to share out; tiny launches stay data-parallel)
      const long long tiles = (long long)cdiv(M, kVariants[v].bm) * cdiv(N, kVariants[v].bn);
      if (!skg || tiles % skg == 0 || tiles * (K / 64) < 4LL * skg) continue;
      cost = sk_cost(v, M, N, K, skg);
    } else {
      cost = variant_cost(v, M, N, K, batch, gelu);
    }
    if (cost < best) { best = cost; best_v = v; }
  }
  if (cost_out) *cost_out = best;
  return best_v;
}

// Stream-K scratch: per stream (launches on one stream are ordered, so one set of slots per stream is enough; two streams
// never share one), allocated on first use and kept: grid x 512 threads x 128 f32 of partial sums (64 MB at 256 CUs) and
// grid + 1 flags.  `epoch` is the value a contributor publishes in its flag: one per launch, never reused on the stream.
struct SkScratch { float* ws = nullptr; unsigned* flags = nullptr; unsigned epoch = 0; };
std::mutex g_sk_mu;
std::unordered_map<hipStream_t, SkScratch> g_sk_scratch;
bool sk_acquire(hipStream_t st, int grid, GemmArgs& a) {
  std::lock_guard<std::mutex> lock(g_sk_mu);
  SkScratch& sc = g_sk_scratch[st];
  if (!sc.ws) {
    const size_t slot_bytes = (size_t)512 * 128 * sizeof(float);
    void *w = nullptr, *f = nullptr;
    if (hipMalloc(&w, slot_bytes * grid) != hipSuccess) return false;
    if (hipMalloc(&f, sizeof(unsigned) * (grid + 1)) != hipSuccess || hipMemset(f, 0, sizeof(unsigned) * (grid + 1)) != hipSuccess) {
      (void)hipFree(w);
      return false;
    }
    sc.ws = (float*)w; sc.flags = (unsigned*)f;
  }
  a.sk_ws = sc.ws; a.sk_flags = sc.flags; a.sk_epoch = ++sc.epoch;
  return true;
}

// bench.py's live timing (prof.hip): the start / stop events ride on the kernel's own dispatch packet (hipExtLaunchKernelGGL) instead of
// being recorded as two extra barrier packets around it - separate hipEventRecord calls cost ~2.8 us each on the stream, 2 ms per C2
// step for its 354 GEMM launches (profiles/r03_prof_event_overhead.txt).  Set by gemm_nt around launch_variant; null = plain launch.
struct LaunchEvents { hipEvent_t start = nullptr, stop = nullptr; };
thread_local LaunchEvents g_launch_ev;
#define UVX_GEMM_LAUNCH(KERNEL, GRID, BLOCK, ST, ARG)                                                                      \
  do {                                                                                                                      \
    if (g_launch_ev.start || g_launch_ev.stop)                                                                              \
      hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, 0, ST, g_launch_ev.start, g_launch_ev.stop, 0, ARG);                       \
    else                                                                                                                    \
      hipLaunchKernelGGL(KERNEL, GRID, BLOCK, 0, ST, ARG);                                                                  \
  } while (0)

void launch_variant(hipStream_t st, int variant, GemmArgs a, int M, int N, int batch) {
  if (a.b_kn && (variant < 31 || variant > 34)) variant = 34;     // (the NN form exists for the merged-phase tiles: a tail launch's small-tile pick)
  if (a.act >= 2 && variant != 0 && (variant < 32 || variant > 34)) variant = variant == 31 || variant == 11 || variant == 16 ? 32 : variant == 15 || variant == 18 || variant == 59 ? 33 : 34;   // (act 2 / 3: builds of 0 and 32..34 - the 256-row tile would spill)
  a.M = M; a.N = N;
  a.tiles_m = cdiv(M, kVariants[variant].bm); a.tiles_n = cdiv(N, kVariants[variant].bn);
  dim3 grid(a.tiles_m * a.tiles_n, batch);
#ifdef UVX_PROBES
  if (is_streamk(variant)) {
    const int g = sk_grid();
    // not a stream-K case after all (batched, device-side row count, no scratch): the data-parallel twin
    if (batch != 1 || a.m_dev || !g || !sk_acquire(st, g, a)) { launch_variant(st, variant - 8, a, M, N, batch); return; }
    a.sk_full = sk_full_rounds((long long)a.tiles_m * a.tiles_n, g);
    const dim3 sgrid(g);
    if (variant == 39) UVX_GEMM_LAUNCH((gemm_nt_bf16_ph8_kernel<256, 2, 0, false, 2, true>), sgrid, dim3(512), st, a);
    else if (variant == 40) UVX_GEMM_LAUNCH((gemm_nt_bf16_ph8_kernel<192, 2, 0, false, 2, true>), sgrid, dim3(512), st, a);
    else if (variant == 41) UVX_GEMM_LAUNCH((gemm_nt_bf16_ph8_kernel<160, 2, 0, false, 2, true>), sgrid, dim3(512), st, a);
    else UVX_GEMM_LAUNCH((gemm_nt_bf16_ph8_kernel<128, 2, 0, false, 2, true>), sgrid, dim3(512), st, a);
    return;
  }
#endif
  switch (variant) {
    case 0:
      if (a.act >= 2) UVX_GEMM_LAUNCH(gemm_nt_bf16_kernel<true>, grid, dim3(256), st, a);
      else UVX_GEMM_LAUNCH(gemm_nt_bf16_kernel<false>, grid, dim3(256), st, a);
      break;
    case 11: UVX_GEMM_LAUNCH(gemm_nt_bf16_ph8_kernel<256>, grid, dim3(512), st, a); break;
    case 15: UVX_GEMM_LAUNCH(gemm_nt_bf16_ph8_kernel<160>, grid, dim3(512), st, a); break;
    case 16: UVX_GEMM_LAUNCH(gemm_nt_bf16_ph8_kernel<192>, grid, dim3(512), st, a); break;
    case 17: UVX_GEMM_LAUNCH(gemm_nt_bf16_ph8_kernel<128>, grid, dim3(512), st, a); break;
    case 18: UVX_GEMM_LAUNCH((gemm_nt_bf16_ph8_kernel<160, 3>), grid, dim3(512), st, a); break;
    case 19: UVX_GEMM_LAUNCH((gemm_nt_bf16_ph8_kernel<128, 3>), grid, dim3(512), st, a); break;
#define UVX_PH2(BMV)                                                                                                          \
  do {                                                                                                                        \
    if (a.b_kn) UVX_GEMM_LAUNCH((gemm_nt_bf16_ph8_kernel<BMV, 2, 0, false, 2, false, false, true>), grid, dim3(512), st, a);  \
    else if (a.act >= 2) UVX_GEMM_LAUNCH((gemm_nt_bf16_ph8_kernel<BMV, 2, 0, false, 2, false, false, false, true>), grid, dim3(512), st, a); \
    else UVX_GEMM_LAUNCH((gemm_nt_bf16_ph8_kernel<BMV, 2, 0, false, 2>), grid, dim3(512), st, a);                             \
  } while (0)
    case 31:      // (no act 2 / 3 build of the 256-row tile: it would spill - launch_variant sends those launches to the 192-row tile)
      if (a.b_kn) UVX_GEMM_LAUNCH((gemm_nt_bf16_ph8_kernel<256, 2, 0, false, 2, false, false, true>), grid, dim3(512), st, a);
      else UVX_GEMM_LAUNCH((gemm_nt_bf16_ph8_kernel<256, 2, 0, false, 2>), grid, dim3(512), st, a);
      break;
    case 32: UVX_PH2(192); break;
    case 33: UVX_PH2(160); break;
    case 34: UVX_PH2(128); break;
#undef UVX_PH2
    case 59: UVX_GEMM_LAUNCH((gemm_nt_bf16_ph8_kernel<160, 3, 0, false, 2>), grid, dim3(512), st, a); break;
    case 60: UVX_GEMM_LAUNCH((gemm_nt_bf16_ph8_kernel<128, 3, 0, false, 2>), grid, dim3(512), st, a); break;
    case 61: case 62: {
      // the 32 x 32 x 16 kernels carry the plain whole-line epilogue only
      const auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
      const bool plain = !a.out_f32 && a.swiglu == 0 && a.act < 2 && a.wide_io == 2 && a.ksplit <= 1 && !a.m_dev && (a.N & 7) == 0 && (a.ldc & 7) == 0 &&
                         (a.sC & 7) == 0 && al16(a.C) && (!a.bias || ((uintptr_t)a.bias & 7) == 0) &&
                         (!a.residual || ((a.ldr & 7) == 0 && (a.sR & 7) == 0 && al16(a.residual)));
      if (!plain) { launch_variant(st, variant == 61 ? 31 : 34, a, M, N, batch); return; }
      if (variant == 61) UVX_GEMM_LAUNCH((gemm_nt_bf16_ph8_kernel<256, 2, 0, false, 2, false, true>), grid, dim3(512), st, a);
      else UVX_GEMM_LAUNCH((gemm_nt_bf16_ph8_kernel<128, 2, 0, false, 2, false, true>), grid, dim3(512), st, a);
      break;
    }
#ifdef UVX_PROBES
    case 43: UVX_GEMM_LAUNCH((gemm_nt_bf16_a4_kernel<2, 8, 0>), grid, dim3(256), st, a); break;
    case 49: UVX_GEMM_LAUNCH((gemm_nt_bf16_a4_kernel<4, 8, 0>), grid, dim3(512), st, a); break;
    case 55: UVX_GEMM_LAUNCH((gemm_nt_bf16_a8pp_kernel<0>), grid, dim3(512), st, a); break;
    case 56: UVX_GEMM_LAUNCH((gemm_nt_bf16_a8pp_kernel<1>), grid, dim3(512), st, a); break;
    case 57: UVX_GEMM_LAUNCH((gemm_nt_bf16_a8pp_kernel<2>), grid, dim3(512), st, a); break;
    case 58: UVX_GEMM_LAUNCH((gemm_nt_bf16_a8pp_kernel<3>), grid, dim3(512), st, a); break;
    case 44: UVX_GEMM_LAUNCH((gemm_nt_bf16_a4_kernel<2, 8, 1>), grid, dim3(256), st, a); break;
    case 45: UVX_GEMM_LAUNCH((gemm_nt_bf16_a4_kernel<2, 8, 2>), grid, dim3(256), st, a); break;
    case 46: UVX_GEMM_LAUNCH((gemm_nt_bf16_a4_kernel<2, 8, 3>), grid, dim3(256), st, a); break;
    case 47: UVX_GEMM_LAUNCH((gemm_nt_bf16_a4_kernel<2, 8, 4>), grid, dim3(256), st, a); break;
    case 48: UVX_GEMM_LAUNCH((gemm_nt_bf16_a4_kernel<2, 8, 5>), grid, dim3(256), st, a); break;
    case 50: UVX_GEMM_LAUNCH((gemm_nt_bf16_a4_kernel<4, 8, 1>), grid, dim3(512), st, a); break;
    case 51: UVX_GEMM_LAUNCH((gemm_nt_bf16_a4_kernel<4, 8, 2>), grid, dim3(512), st, a); break;
    case 52: UVX_GEMM_LAUNCH((gemm_nt_bf16_a4_kernel<4, 8, 3>), grid, dim3(512), st, a); break;
    case 53: UVX_GEMM_LAUNCH((gemm_nt_bf16_a4_kernel<4, 8, 4>), grid, dim3(512), st, a); break;
    case 54: UVX_GEMM_LAUNCH((gemm_nt_bf16_a4_kernel<4, 8, 5>), grid, dim3(512), st, a); break;
    case 1: UVX_GEMM_LAUNCH(gemm_nt_bf16_wide_kernel<128>, grid, dim3(512), st, a); break;
    case 2: UVX_GEMM_LAUNCH(gemm_nt_bf16_wide_kernel<160>, grid, dim3(512), st, a); break;
    case 3: UVX_GEMM_LAUNCH(gemm_nt_bf16_wide_kernel<192>, grid, dim3(512), st, a); break;
    case 4: UVX_GEMM_LAUNCH(gemm_nt_bf16_wide_kernel<256>, grid, dim3(512), st, a); break;
    case 5: UVX_GEMM_LAUNCH(gemm_nt_bf16_pp_kernel<128>, grid, dim3(512), st, a); break;
    case 6: UVX_GEMM_LAUNCH(gemm_nt_bf16_pp_kernel<160>, grid, dim3(512), st, a); break;
    case 7: UVX_GEMM_LAUNCH(gemm_nt_bf16_pp_kernel<192>, grid, dim3(512), st, a); break;
    case 8: UVX_GEMM_LAUNCH(gemm_nt_bf16_pp_kernel<256>, grid, dim3(512), st, a); break;
    case 9: UVX_GEMM_LAUNCH(gemm_nt_bf16_wide3_kernel<128>, grid, dim3(512), st, a); break;
    case 10: UVX_GEMM_LAUNCH(gemm_nt_bf16_wide3_kernel<160>, grid, dim3(512), st, a); break;
    case 20: UVX_GEMM_LAUNCH((gemm_nt_bf16_ph8_kernel<256, 2, 1>), grid, dim3(512), st, a); break;   // operand delivery only
    case 21: UVX_GEMM_LAUNCH((gemm_nt_bf16_ph8_kernel<256, 2, 2>), grid, dim3(512), st, a); break;   // arithmetic only
    case 22: UVX_GEMM_LAUNCH((gemm_nt_bf16_ph8_kernel<256, 2, 3>), grid, dim3(512), st, a); break;   // no epilogue
    case 27: UVX_GEMM_LAUNCH((gemm_nt_bf16_ph8_kernel<256, 2, 4>), grid, dim3(512), st, a); break;   // s_memtime timeline
    case 28: UVX_GEMM_LAUNCH((gemm_nt_bf16_ph8_kernel<256, 2, 5>), grid, dim3(512), st, a); break;
    case 29: UVX_GEMM_LAUNCH((gemm_nt_bf16_ph8_kernel<256, 2, 6>), grid, dim3(512), st, a); break;
    case 30: UVX_GEMM_LAUNCH((gemm_nt_bf16_ph8_kernel<256, 2, 7>), grid, dim3(512), st, a); break;
    case 35: case 36: case 37: case 38: {   // persistent merged-phase: next tile's pipeline fill under the epilogue
      if (batch != 1) { launch_variant(st, variant - 4, a, M, N, batch); return; }
      const dim3 pgrid(a.tiles_m * a.tiles_n < 256 ? a.tiles_m * a.tiles_n : 256);
      if (variant == 35) UVX_GEMM_LAUNCH((gemm_nt_bf16_ph8_kernel<256, 2, 0, true, 2>), pgrid, dim3(512), st, a);
      else if (variant == 36) UVX_GEMM_LAUNCH((gemm_nt_bf16_ph8_kernel<192, 2, 0, true, 2>), pgrid, dim3(512), st, a);
      else if (variant == 37) UVX_GEMM_LAUNCH((gemm_nt_bf16_ph8_kernel<160, 2, 0, true, 2>), pgrid, dim3(512), st, a);
      else UVX_GEMM_LAUNCH((gemm_nt_bf16_ph8_kernel<128, 2, 0, true, 2>), pgrid, dim3(512), st, a);
      break;
    }
    case 23: case 24: case 25: case 26: {
      // persistent: one block per CU walks the tiles (batched problems use the plain kernels: grid.y would oversubscribe)
      if (batch != 1) { launch_variant(st, variant == 23 ? 11 : variant == 24 ? 16 : variant == 25 ? 15 : 17, a, M, N, batch); return; }
      const dim3 pgrid(a.tiles_m * a.tiles_n < 256 ? a.tiles_m * a.tiles_n : 256);
      if (variant == 23) UVX_GEMM_LAUNCH((gemm_nt_bf16_ph8_kernel<256, 2, 0, true>), pgrid, dim3(512), st, a);
      else if (variant == 24) UVX_GEMM_LAUNCH((gemm_nt_bf16_ph8_kernel<192, 2, 0, true>), pgrid, dim3(512), st, a);
      else if (variant == 25) UVX_GEMM_LAUNCH((gemm_nt_bf16_ph8_kernel<160, 2, 0, true>), pgrid, dim3(512), st, a);
      else UVX_GEMM_LAUNCH((gemm_nt_bf16_ph8_kernel<128, 2, 0, true>), pgrid, dim3(512), st, a);
      break;
    }
    case 12: UVX_GEMM_LAUNCH(gemm_nt_bf16_q4_kernel, grid, dim3(256), st, a); break;
    case 13: UVX_GEMM_LAUNCH((gemm_nt_bf16_pp_kernel<256, 1>), grid, dim3(512), st, a); break;   // probe: delivery only
    case 14: UVX_GEMM_LAUNCH((gemm_nt_bf16_pp_kernel<256, 2>), grid, dim3(512), st, a); break;   // probe: arithmetic only
#endif
    default: break;   // unavailable variants are rejected in gemm_nt before any launch
  }
}

// ------------------------------------------------------------------------------------------------
// Split-K (round 5).  The prefill of generate() runs the LLM's linears on M = 64 ... ~700 rows (one or two prompts of 30 s of audio +
// text: 316 positions each).  With 160 / 256-row tiles that is 2-4 row tiles x N / 256 weight panels - 64 tiles at N = 8192 on a chip of
// 256 CUs - and the 128 x 128 fallback re-reads the weights per row tile at a third of the big tiles' rate: Llama-3.3-70B's B = 1 prefill
// sat at 0.17 of EITHER roofline (round 4: 102 ms for 44.6 TFLOP / 139 GB).  A tile's K loop is therefore cut over `s` blocks
// (grid.y = s; block z runs K-tiles [z nk / s, (z + 1) nk / s) of its tile through the unchanged main loop and writes an f32 partial tile
// to slab z of the caller's scratch), so tiles x s ~ one full round of the 256 CUs, every weight byte still leaves HBM once (the row
// tiles that share a weight panel and K range sit in the same XCD's L2), and splitk_reduce_k sums the slabs in slab order - fixed, hence
// bit-reproducible - and applies the epilogue with store_tile's arithmetic and rounding points.
// Why not a block that owns all M rows and a narrow N slice with the K split over its waves: its activation traffic.  Per K-tile a
// 320 x 32 block pulls 40 KB of activations + 4 KB of weights from L2 for 1.3 MFLOP, a 160 x 256 tile 53 KB for 5.2 MFLOP - and
// the wide tile's main loop is already bounded by that L2 -> LDS stream (DESIGN 3.1, round 4 probes).  Splitting K over CUs keeps the
// wide tile's bytes per flop and pays s x M x N x 4 bytes of partials through the L2 / Infinity Cache instead.
struct ReduceArgs {
  const float* P; long long slab; int s, ldp;
  bf16_t* C; int ldc; const bf16_t* bias; const bf16_t* residual; int ldr, res_mod;
  int M, N, act; float alpha; bf16_t* C2; int ldc2, swiglu;
};
// one thread = 8 consecutive output columns of a row (SwiGLU epilogue: 8 gate columns of a 16-column block and the matching 8 up columns)
__global__ __launch_bounds__(256) void splitk_reduce_k(ReduceArgs p) {
  const int per_row = p.swiglu ? p.N / 16 : p.N / 8;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)p.M * per_row) return;
  const int m = (int)(idx / per_row), c = (int)(idx % per_row);
  const int n0 = p.swiglu ? (c >> 1) * 32 + (c & 1) * 8 : c * 8;       // first (gate) column
  const float* src = p.P + (long long)m * p.ldp + n0;
  float g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, u[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int z = 0; z < p.s; ++z) {                                       // slab order: the sum is the same on every run
    const float4 a = *reinterpret_cast<const float4*>(src + z * p.slab), b = *reinterpret_cast<const float4*>(src + z * p.slab + 4);
    g[0] += a.x; g[1] += a.y; g[2] += a.z; g[3] += a.w; g[4] += b.x; g[5] += b.y; g[6] += b.z; g[7] += b.w;
    if (p.swiglu) {
      const float4 e = *reinterpret_cast<const float4*>(src + z * p.slab + 16), f = *reinterpret_cast<const float4*>(src + z * p.slab + 20);
      u[0] += e.x; u[1] += e.y; u[2] += e.z; u[3] += e.w; u[4] += f.x; u[5] += f.y; u[6] += f.z; u[7] += f.w;
    }
  }
  uint32_t o[4];
  if (p.swiglu) {   // C = gate|up pre-activations (interleaved 16-column blocks), C2 = round(silu(round(gate))) * round(up)
    uint32_t ou[4], oa[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a2[2];
      o[e] = pack2(g[2 * e] * p.alpha, g[2 * e + 1] * p.alpha);
      ou[e] = pack2(u[2 * e] * p.alpha, u[2 * e + 1] * p.alpha);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float gg = h ? unpack_hi(o[e]) : unpack_lo(o[e]), uu = h ? unpack_hi(ou[e]) : unpack_lo(ou[e]);
        a2[h] = bf2f(f2bf(gg / (1.0f + __expf(-gg)))) * uu;
      }
      oa[e] = pack2(a2[0], a2[1]);
    }
    bf16_t* crow = p.C + (long long)m * p.ldc + n0;
    *reinterpret_cast<uint4*>(crow) = make_uint4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<uint4*>(crow + 16) = make_uint4(ou[0], ou[1], ou[2], ou[3]);
    *reinterpret_cast<uint4*>(p.C2 + (long long)m * p.ldc2 + (c >> 1) * 16 + (c & 1) * 8) = make_uint4(oa[0], oa[1], oa[2], oa[3]);
    return;
  }
  float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (p.bias) {
    const uint4 b4 = *reinterpret_cast<const uint4*>(p.bias + n0);
    bv[0] = unpack_lo(b4.x); bv[1] = unpack_hi(b4.x); bv[2] = unpack_lo(b4.y); bv[3] = unpack_hi(b4.y);
    bv[4] = unpack_lo(b4.z); bv[5] = unpack_hi(b4.z); bv[6] = unpack_lo(b4.w); bv[7] = unpack_hi(b4.w);
  }
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float t = bf2f(f2bf(g[e] * p.alpha + bv[e]));
    if (p.act == 1) t = bf2f(f2bf(gelu_fast(t)));
    v[e] = t;
  }
  if (p.residual) {
    const int rm = p.res_mod > 0 ? (m % p.res_mod) : m;
    const uint4 r = *reinterpret_cast<const uint4*>(p.residual + (long long)rm * p.ldr + n0);
    v[0] += unpack_lo(r.x); v[1] += unpack_hi(r.x); v[2] += unpack_lo(r.y); v[3] += unpack_hi(r.y);
    v[4] += unpack_lo(r.z); v[5] += unpack_hi(r.z); v[6] += unpack_lo(r.w); v[7] += unpack_hi(r.w);
  }
  *reinterpret_cast<uint4*>(p.C + (long long)m * p.ldc + n0) = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
}

// The reduce with the RMSNorm that FOLLOWS the linear fused in (round 5): in the decoder stack every o_proj / down_proj (+ residual) is
// followed by post_attention_layernorm / the next layer's input_layernorm over the row it has just completed - a separate launch of 8-14 us on
// a few hundred rows (160 of them per 70B prefill, 160 per decode step of a batch beyond 16 rows).  One block per output row, the
// thread -> column mapping and the summation order of rmsnorm_fwd_k (norms.hip: norm_threads(N) threads, 8-column chunks strided by the block,
// block_sum), so C is what splitk_reduce_k writes and Y is bit for bit what rmsnorm_fwd_k would compute from it: y = w * round(x * rstd)
// (flavor 0, LlamaRMSNorm) or (x * rstd) * (1 + w) (flavor 1, GemmaRMSNorm), x = the bf16-rounded C row.  No SwiGLU form (never followed by a norm).
struct ReduceNormArgs { ReduceArgs r; const bf16_t* norm_w; bf16_t* Y; int ldy; float eps; int flavor; };
constexpr int kReduceNormMaxChunks = 8;      // 8-column chunks per thread: N <= 256 * 8 * 8 = 16384
__global__ __launch_bounds__(256) void splitk_reduce_norm_k(ReduceNormArgs q) {
  __shared__ float red[16];
  const ReduceArgs& p = q.r;
  const int m = blockIdx.x;
  const float* src = p.P + (long long)m * p.ldp;
  const int rm = p.res_mod > 0 ? (m % p.res_mod) : m;
  uint4 xo[kReduceNormMaxChunks];
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < kReduceNormMaxChunks; ++k) {
    const int n0 = (threadIdx.x + k * blockDim.x) * 8;
    if (n0 >= p.N) break;
    float g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int z = 0; z < p.s; ++z) {                                     // slab order: the sum is the same on every run
      const float4 a = *reinterpret_cast<const float4*>(src + z * p.slab + n0), b = *reinterpret_cast<const float4*>(src + z * p.slab + n0 + 4);
      g[0] += a.x; g[1] += a.y; g[2] += a.z; g[3] += a.w; g[4] += b.x; g[5] += b.y; g[6] += b.z; g[7] += b.w;
    }
    float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
      const uint4 b4 = *reinterpret_cast<const uint4*>(p.bias + n0);
      bv[0] = unpack_lo(b4.x); bv[1] = unpack_hi(b4.x); bv[2] = unpack_lo(b4.y); bv[3] = unpack_hi(b4.y);
      bv[4] = unpack_lo(b4.z); bv[5] = unpack_hi(b4.z); bv[6] = unpack_lo(b4.w); bv[7] = unpack_hi(b4.w);
    }
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = bf2f(f2bf(g[e] * p.alpha + bv[e]));
      if (p.act == 1) t = bf2f(f2bf(gelu_fast(t)));
      v[e] = t;
    }
    if (p.residual) {
      const uint4 r = *reinterpret_cast<const uint4*>(p.residual + (long long)rm * p.ldr + n0);
      v[0] += unpack_lo(r.x); v[1] += unpack_hi(r.x); v[2] += unpack_lo(r.y); v[3] += unpack_hi(r.y);
      v[4] += unpack_lo(r.z); v[5] += unpack_hi(r.z); v[6] += unpack_lo(r.w); v[7] += unpack_hi(r.w);
    }
    xo[k] = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
    *reinterpret_cast<uint4*>(p.C + (long long)m * p.ldc + n0) = xo[k];
    const uint32_t xw[4] = {xo[k].x, xo[k].y, xo[k].z, xo[k].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float lo = unpack_lo(xw[e]), hi = unpack_hi(xw[e]); ss += lo * lo; ss += hi * hi; }
  }
  const float rstd = rsqrtf(block_sum(ss, red) / p.N + q.eps);
#pragma unroll
  for (int k = 0; k < kReduceNormMaxChunks; ++k) {
    const int n0 = (threadIdx.x + k * blockDim.x) * 8;
    if (n0 >= p.N) break;
    const uint4 w4 = *reinterpret_cast<const uint4*>(q.norm_w + n0);
    const uint32_t xw[4] = {xo[k].x, xo[k].y, xo[k].z, xo[k].w}, ww[4] = {w4.x, w4.y, w4.z, w4.w};
    uint32_t o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xl = unpack_lo(xw[e]), xh = unpack_hi(xw[e]), wl = unpack_lo(ww[e]), wh = unpack_hi(ww[e]);
      o[e] = q.flavor ? pack2((xl * rstd) * (1.0f + wl), (xh * rstd) * (1.0f + wh))
                      : pack2(wl * bf2f(f2bf(xl * rstd)), wh * bf2f(f2bf(xh * rstd)));
    }
    *reinterpret_cast<uint4*>(q.Y + (long long)m * q.ldy + n0) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}
int reduce_norm_threads(int cols) {      // == norm_threads(cols) of norms.hip (the summation order depends on it)
  const int t = ((cols / 8) + 63) / 64 * 64;
  return t < 64 ? 64 : (t > 256 ? 256 : t);
}

// Which (tile variant, split factor).  Two regimes, both in microseconds:
//  * at most one block per CU (blocks = tiles x s <= 256; the few-hundred-row problems this path exists for): every block runs its K loop
//    alone on a CU at the tile's own pace, kt1_us per K-tile, + kSparseFix K-tiles of fill and epilogue - the time is that of ONE block,
//    however few there are.  (The training-shape model above prices a quarter-filled round at 0.55 of a full one and has the 128 x 128
//    tile at 880 TF/s; on cold weights a lone 128 x 128 block needs 1.3 us per K-tile - no prefetch across its two barriers - and 64
//    blocks of 160 x 256 take exactly as long as 256: profiles/r05_gemm_splitk_probe.txt.)
//  * more blocks than CUs: the tile model above (cost units x kCostUs) on tiles x s blocks of K / s each.
// A split adds what the partials cost: the f32 tiles written by the GEMM and read back by the reduce kernel (through L2 / Infinity
// Cache: kSplitBytesPerUs, fitted: 2.4 TB/s for the pair) and the second launch (kSplitFixUs).  The model reproduces the probe's
// 5 tiles x 8 factors x 16 shapes to ~10 %; its picks are within 5 % of the best measured pair on every shape.
constexpr double kCostUs = 128.0 * 256.0 / 1e6;       // variant_cost units -> microseconds (2 x 64 flop per tile element and K-tile, 256 CUs, speed in TF/s)
constexpr double kSplitFixUs = 4.0, kSplitBytesPerUs = 2.4e6, kSparseFix = 2.0;
constexpr int kSplitMinKTiles = 4, kSplitMax = 16;
double kt1_us(int v) { return v == 0 ? 1.30 : v == 34 ? 0.85 : v == 33 ? 0.975 : v == 32 ? 1.00 : v == 31 ? 1.09 : kVariants[v].bm * kVariants[v].bn * kCostUs / 1300.; }
double sparse_cost_us(int v, int M, int N, int nk_s, int s, bool gelu) {
  const long long blocks = (long long)cdiv(M, kVariants[v].bm) * cdiv(N, kVariants[v].bn) * s;
  // between half and all of the CUs busy the K-tile time climbs from the lone block's to the full chip's (the tile model's asymptotic
  // rate): 1.07 -> 1.44 us for the 256 x 256 tile between 128 and 256 blocks, 0.98 -> 1.04 for 160 x 256 (same probe)
  const double full = kVariants[v].bm * kVariants[v].bn * kCostUs / (kVariants[v].speed > 0. ? kVariants[v].speed : 1300.), x = blocks / 256.0;
  const double kt = v == 0 ? kt1_us(0) : kt1_us(v) + fmax(0., full - kt1_us(v)) * fmin(1., fmax(0., (x - 0.5) / 0.5));
  const double alone = (nk_s + kSparseFix + (gelu ? 1.0 : 0.0)) * kt;
  if (v != 0 && blocks <= 256) return alone;
  const double model = variant_cost(v, M, N, nk_s * 64, s, gelu) * kCostUs;
  return v == 0 ? fmax(model, alone) : model;          // (several 128 x 128 blocks share a CU, none runs its K loop faster than alone)
}
struct SplitPick { int variant, s; double us; };
SplitPick pick_split(int M, int N, int K, size_t ws_bytes, bool gelu, int force_s) {
  const int nk = K / 64, fv = forced_variant(M, N, K);
  SplitPick best{fv >= 0 ? fv : 0, 1, 1e30};
  for (int v : {0, 34, 33, 32, 31, 59, 60, 18, 19, 11, 15, 16, 17}) {
    if (fv >= 0 ? v != fv : (v > 34 || v < 31) && v != 0) continue;      // (outside the two-set merged-phase kernels - incl. the three-set 59 / 60: only when forced)
    if (fv < 0 && !uvx::g_options[6] && v != 0) continue;                // (option 6 = 0, the round-1 four-phase set: A/B builds, never split)
    const long long tiles = (long long)cdiv(M, kVariants[v].bm) * cdiv(N, kVariants[v].bn);
    for (int s = 1; s <= kSplitMax; ++s) {
      if (force_s > 0 && s != force_s) continue;
      if (s > 1 && (nk / s < 1 || (size_t)s * M * N * 4 > ws_bytes)) break;
      if (s > 1 && force_s <= 1 && (nk / s < kSplitMinKTiles || tiles * s > 640)) break;   // (> ~2.5 rounds of blocks: nothing left to fill)
      const double us = sparse_cost_us(v, M, N, cdiv(nk, s), s, gelu) + (s > 1 ? kSplitFixUs + (double)s * M * N * 4.0 / kSplitBytesPerUs : 0.);
      if (us < best.us) best = SplitPick{v, s, us};
    }
  }
  if (best.us >= 1e30) {       // a forced tile outside the split set (probe builds): unsplit
    double c0 = 0.;
    const int v0 = pick_variant(M, N, K, 1, &c0, gelu);
    best = SplitPick{v0, 1, c0 * kCostUs};
  }
  return best;
}

}  // namespace

size_t uvx::gemm_splitk_ws_bytes(int M, int N) {
  // enough for one full round of the largest tiles' f32 partials and never more than 16 slabs of the problem
  const size_t cap = (size_t)256 * 256 * 256 * 4 * 2, want = (size_t)kSplitMax * (size_t)M * N * 4;
  return want < cap ? want : cap;
}

// Stream-K spin waits that gave up (see the kernel), summed over every stream's scratch; synchronises the device.  0 on a
// healthy run - anything else means wrong output tiles in some stream-K launch since the last call (the counters are reset).
int uvx::gemm_streamk_timeouts() {
  std::lock_guard<std::mutex> lock(g_sk_mu);
  const int g = sk_grid();
  long long total = 0;
  if (!g || g_sk_scratch.empty()) return 0;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  for (auto& kv : g_sk_scratch) {
    unsigned n = 0;
    if (!kv.second.flags) continue;
    if (hipMemcpy(&n, kv.second.flags + g, sizeof(n), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (n && hipMemset(kv.second.flags + g, 0, sizeof(n)) != hipSuccess) return -1;
    total += n;
  }
  return (int)(total > 0x7fffffff ? 0x7fffffff : total);
}

// launches the four-wave kernel's epilogue can serve (16-byte accesses on every operand it touches; no SwiGLU-backward fusion)
static bool a4_applicable(const uvx::GemmDesc& d) {
  if (d.swiglu == 2) return false;
  if (d.out_f32) return true;
  const auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  if ((d.N & 7) || (d.ldc & 7) || (d.sC & 7) || !al16(d.C)) return false;
  if (d.residual && ((d.ldr & 7) || (d.sR & 7) || !al16(d.residual))) return false;
  if (d.swiglu == 1 && ((d.ldc2 & 7) || !al16(d.C2))) return false;
  return true;
}

int uvx::gemm_pick_variant(int M, int N, int K, int batch) { return pick_variant(M, N, K, batch > 0 ? batch : 1); }
int uvx::gemm_pick_split(int M, int N, int K, size_t ws_bytes, int* variant) {
  const SplitPick sp = pick_split(M, N, K, ws_bytes, false, 0);
  if (variant) *variant = sp.variant;
  return sp.s;
}

int uvx::gemm_nt(hipStream_t st, const GemmDesc& d) {
  UVX_CHECK(d.M > 0 && d.N > 0 && d.K > 0, UVX_ERR_SHAPE, "gemm: empty problem %dx%dx%d", d.M, d.N, d.K);
  UVX_CHECK(d.K % BK == 0, UVX_ERR_SHAPE, "gemm: K=%d must be a multiple of %d", d.K, BK);
  UVX_CHECK(d.N % 4 == 0 && d.ldc % 4 == 0, UVX_ERR_SHAPE, "gemm: N=%d / ldc=%d must be multiples of 4", d.N, d.ldc);
  UVX_CHECK(d.lda % 8 == 0 && d.ldb % 8 == 0, UVX_ERR_SHAPE, "gemm: lda=%d / ldb=%d must be multiples of 8", d.lda, d.ldb);
  UVX_CHECK(!d.residual || d.ldr % 4 == 0, UVX_ERR_SHAPE, "gemm: ldr=%d must be a multiple of 4", d.ldr);
  UVX_CHECK(!d.accumulate || d.out_f32, UVX_ERR_INVALID, "gemm: accumulate needs f32 output");
  // few rows: stream the weights.  Beyond 16 rows the weight-streaming kernels serve 16-row tiles one after the other (MT = 2, 4) and lose to a
  // 128 x 256 tile whose K loop is cut over 6-8 blocks wherever the caller lent split-K scratch (Llama-3.3-70B's four linears, us per
  // launch, staged kernel / tiled split-K: 32 rows 447 / 345, 64 rows 871 / 358 - profiles/r05_gemm_splitk_decode_rows.txt)
  const bool split_ok = d.splitk_ws && d.batch <= 1 && !d.out_f32 && !d.m_dev && d.swiglu != 2 && d.act < 2 && !d.b_kn && d.splitk_force != 1 && d.N % 8 == 0 && d.ldc % 8 == 0 &&
      ((uintptr_t)d.C & 15) == 0 && ((uintptr_t)d.splitk_ws & 15) == 0 && (!d.bias || ((uintptr_t)d.bias & 15) == 0) &&
      (!d.residual || (d.ldr % 8 == 0 && ((uintptr_t)d.residual & 15) == 0)) &&
      (!d.swiglu || (d.ldc2 % 8 == 0 && ((uintptr_t)d.C2 & 15) == 0));      // (the reduce kernel's 16-byte accesses apply)
  if (uvx::g_gemm_variant < 0 && uvx::gemm_skinny_applicable(d)) {
    // (M = 17..64 with scratch: the tiled split-K path only where the picker really splits - K/64 < 8, too little scratch or option 6 = 0
    //  leave s = 1, and an UNSPLIT 128 x 128 / 128 x 256 tile on a handful of CUs is slower than the staged skinny kernel: ADVICE r5)
    const bool tiled = split_ok && d.M > 16 && pick_split(d.M, d.N, d.K, d.splitk_ws_bytes, d.act == 1, d.splitk_force).s > 1;
    if (!tiled) return uvx::gemm_skinny_bf16(st, d);
  }
  GemmArgs a;
  a.A = (const bf16_t*)d.A; a.B = (const bf16_t*)d.B; a.C = d.C;
  a.bias = (const bf16_t*)d.bias; a.residual = (const bf16_t*)d.residual;
  a.M = d.M; a.N = d.N; a.K = d.K;
  a.lda = d.lda; a.ldb = d.ldb; a.ldc = d.ldc; a.ldr = d.ldr;
  a.res_mod = d.res_mod;
  a.sA = d.sA; a.sB = d.sB; a.sC = d.sC; a.sR = d.sR;
  a.act = d.act; a.out_f32 = d.out_f32; a.accumulate = d.accumulate; a.alpha = d.alpha;
  a.C2 = (bf16_t*)d.C2; a.ldc2 = d.ldc2; a.swiglu = d.swiglu;
  a.wide_io = uvx::g_options[1];
  a.sw_stage = uvx::g_options[2] != 1;     // SwiGLU-backward epilogue through the LDS stage (option 2 = 1: the round-2 fragment-layout form)
  a.m_major = uvx::g_options[7] && d.M > d.N && (d.batch <= 1);
  a.m_dev = d.m_dev; a.m_dev_off = d.m_dev_off;
  a.sk_full = 0; a.sk_ws = nullptr; a.sk_flags = nullptr; a.sk_epoch = 0; a.ksplit = 1;
  a.b_kn = d.b_kn;
  UVX_CHECK(!d.b_kn || (d.N % 8 == 0 && d.N >= 8 && d.batch <= 1 && ((uintptr_t)d.B & 15) == 0), UVX_ERR_INVALID,
            "gemm: the NN form (B stored [K, N]) needs N %% 8 == 0, a 16-byte-aligned B and no batch");
  UVX_CHECK(!d.swiglu || (d.C2 && !d.out_f32 && !d.bias && !d.residual && d.act == 0 && d.N % 32 == 0 && d.ldc2 % 4 == 0 && (d.batch <= 1)),
            UVX_ERR_INVALID, "gemm: swiglu epilogue needs C2, bf16 output, N %% 32 == 0 and no bias/act/residual/batch");
  UVX_CHECK(d.swiglu != 2 || (d.N % 16 == 0 && d.ldc >= 2 * d.N && d.ldc2 >= 2 * d.N), UVX_ERR_INVALID,
            "gemm: swiglu-backward epilogue writes [M, 2N]: ldc=%d / ldc2=%d too small for N=%d", d.ldc, d.ldc2, d.N);
  UVX_CHECK(d.act < 2 || (d.act <= 3 && d.C2 && !d.out_f32 && !d.residual && !d.swiglu && d.ldc2 % 4 == 0 && d.batch <= 1 && (d.act == 2 || !d.bias)),
            UVX_ERR_INVALID, "gemm: act %d (GELU keeping / consuming the pre-activation) needs C2 [M, N], bf16 output and no residual / swiglu / batch", d.act);
  const int batch = d.batch > 0 ? d.batch : 1;
  double cost_whole = 0.;
  const bool gelu = d.act == 1 || d.act == 2 || d.act == 3;
  int sparse_variant = -1;       // the split picker's unsplit choice for a launch of at most one block per CU (its model, not the tile model's)
  // split-K (see splitk_reduce_k): only where the caller lent scratch for the partial tiles and the reduce kernel's 16-byte accesses apply
  if (split_ok) {
    const SplitPick sp = pick_split(d.M, d.N, d.K, d.splitk_ws_bytes, gelu, d.splitk_force);
    if (sp.s == 1 && (long long)cdiv(d.M, kVariants[sp.variant].bm) * cdiv(d.N, kVariants[sp.variant].bn) <= 256) sparse_variant = sp.variant;
    if (sp.s > 1) {
      UVX_CHECK(variant_available(sp.variant), UVX_ERR_INVALID, "gemm: tile variant %d is not in this build", sp.variant);
      hipEvent_t ev_a = nullptr, ev_b = nullptr;
      const bool timed = uvx::g_prof_on &&
                         uvx::prof_take(uvx::PROF_GEMM, 2.0 * d.M * d.N * (double)d.K,
                                        ((double)d.M * d.K + (double)d.N * d.K) * 2.0 + (double)d.M * d.N * 2.0, &ev_a, &ev_b);
      if (timed) uvx::prof_tag(d.M, d.N, d.K, sp.s, 300 + sp.variant);   // (records: batch = split factor, variant = 300 + tile)
      GemmArgs g = a;
      g.C = d.splitk_ws; g.ldc = d.N; g.sA = 0; g.sB = 0; g.sC = (long long)d.M * d.N; g.out_f32 = 1; g.accumulate = 0;
      g.bias = nullptr; g.residual = nullptr; g.act = 0; g.alpha = 1.0f; g.swiglu = 0; g.C2 = nullptr; g.ksplit = sp.s;
      g.m_major = 0;
      g_launch_ev = LaunchEvents{ev_a, nullptr};
      launch_variant(st, sp.variant, g, d.M, d.N, sp.s);
      g_launch_ev = LaunchEvents{};
      ReduceArgs r;
      r.P = (const float*)d.splitk_ws; r.slab = (long long)d.M * d.N; r.s = sp.s; r.ldp = d.N;
      r.C = (bf16_t*)d.C; r.ldc = d.ldc; r.bias = a.bias; r.residual = a.residual; r.ldr = d.ldr; r.res_mod = d.res_mod;
      r.M = d.M; r.N = d.N; r.act = d.act; r.alpha = d.alpha; r.C2 = a.C2; r.ldc2 = d.ldc2; r.swiglu = d.swiglu;
      const long long items = (long long)d.M * (d.swiglu ? d.N / 16 : d.N / 8);
      const dim3 rgrid((unsigned)((items + 255) / 256));
      // the RMSNorm that follows this linear, in the same launch (one block per row), where the caller asked for it
      // (N = 512 / 1024 / 2048: rmsnorm_fwd runs its one-wave-per-row kernel there - another summation order, so those widths keep the separate launch)
      const bool with_norm = d.norm_w && d.norm_out && !d.swiglu && uvx::g_options[17] && d.N <= 256 * 8 * kReduceNormMaxChunks &&
                             d.N != 512 && d.N != 1024 && d.N != 2048 &&
                             d.norm_ld % 8 == 0 && ((uintptr_t)d.norm_w & 15) == 0 && ((uintptr_t)d.norm_out & 15) == 0;
      if (with_norm) {
        ReduceNormArgs q{r, (const bf16_t*)d.norm_w, (bf16_t*)d.norm_out, d.norm_ld, d.norm_eps, d.norm_flavor};
        const dim3 ngrid((unsigned)d.M), nblk((unsigned)reduce_norm_threads(d.N));
        if (ev_b) hipExtLaunchKernelGGL(splitk_reduce_norm_k, ngrid, nblk, 0, st, nullptr, ev_b, 0, q);
        else hipLaunchKernelGGL(splitk_reduce_norm_k, ngrid, nblk, 0, st, q);
        if (d.norm_done) *d.norm_done = true;
      } else if (ev_b) hipExtLaunchKernelGGL(splitk_reduce_k, rgrid, dim3(256), 0, st, nullptr, ev_b, 0, r);
      else hipLaunchKernelGGL(splitk_reduce_k, rgrid, dim3(256), 0, st, r);
      if (timed) uvx::prof_commit();
      UVX_LAUNCH_CHECK();
      return UVX_OK;
    }
  }
  int variant = pick_variant(d.M, d.N, d.K, batch, &cost_whole, gelu);
  if (sparse_variant >= 0 && sparse_variant != variant) { variant = sparse_variant; cost_whole = variant_cost(variant, d.M, d.N, d.K, batch, gelu); }
  if (is_a4(variant) && !a4_applicable(d)) variant = 31;     // (the eight-wave 256 x 256 kernel takes any alignment)
  if (d.b_kn && (variant < 31 || variant > 34)) {             // the NN form exists for the merged-phase tiles: the cheapest of them
    double best = 1e300;
    for (int v = 31; v <= 34; ++v) {
      const double cst = variant_cost(v, d.M, d.N, d.K, batch, gelu);
      if (cst < best) { best = cst; variant = v; }
    }
    cost_whole = best;
  }
  UVX_CHECK(variant_available(variant), UVX_ERR_INVALID,
            "gemm: tile variant %d is not in this build (probe variants live in libuvx_probes.so, built with -DUVX_PROBES)", variant);
  // (device-side row count: the true work is unknown here, such launches are left out of the timing)
  hipEvent_t ev_a = nullptr, ev_b = nullptr;
  const bool timed = uvx::g_prof_on && d.m_dev == nullptr &&
                     uvx::prof_take(uvx::PROF_GEMM, 2.0 * d.M * d.N * (double)d.K * batch,
                                    ((double)d.M * d.K + (double)d.N * d.K) * 2.0 * batch + (double)d.M * d.N * batch * (d.out_f32 ? 4.0 : 2.0),
                                    &ev_a, &ev_b);
  // Tail split (tile-quantisation fix): when the last round of big tiles would leave most CUs idle, the
  // trailing weight panels (a column range of C) are computed by a second launch with its own best variant.
  const Variant& V = kVariants[variant];
  const int tm = cdiv(d.M, V.bm), tn = cdiv(d.N, V.bn);
  const long long tiles = (long long)tm * tn;
  int n_main = d.N, tail_variant = -1;
  if (batch == 1 && uvx::g_gemm_split && !is_streamk(variant) && tiles > 256 && tiles % 256 != 0) {
    const int full_rounds = (int)(tiles / 256);
    const int main_panels = (int)((full_rounds * 256LL) / tm);           // whole weight panels in the full rounds
    const int tail_n = d.N - main_panels * V.bn;
    if (main_panels > 0 && tail_n > 0) {
      double cost_tail = 0.;
      const int tv = pick_variant(d.M, tail_n, d.K, 1, &cost_tail, gelu);
      const double cost_split = variant_cost(variant, d.M, main_panels * V.bn, d.K, 1, gelu) + cost_tail;
      // (the second launch costs a round of its own plus a launch gap: only worth it for a clear modelled win)
      // (probe option 10: the threshold in per cent, 0 = the default 88)
      const double thr = uvx::g_options[10] > 0 ? uvx::g_options[10] / 100.0 : 0.88;
      if (cost_split < thr * cost_whole) { n_main = main_panels * V.bn; tail_variant = tv; }
    }
  }
  if (timed) uvx::prof_tag(d.M, d.N, d.K, batch, tail_variant >= 0 ? 100 + variant : variant);
  g_launch_ev = LaunchEvents{ev_a, tail_variant >= 0 ? nullptr : ev_b};     // start on the (first) launch, stop on the last one
  launch_variant(st, variant, a, d.M, n_main, batch);
  g_launch_ev = LaunchEvents{};
  if (tail_variant >= 0) {
    GemmArgs t = a;
    t.B = a.b_kn ? a.B + n_main : a.B + (long long)n_main * a.ldb;
    if (a.bias) t.bias = a.bias + n_main;
    if (a.residual) t.residual = a.residual + n_main;
    if (a.swiglu == 1) t.C2 = a.C2 + n_main / 2;
    if (a.act >= 2) t.C2 = a.C2 + n_main;
    t.C = a.out_f32 ? (void*)((float*)a.C + n_main) : (void*)((bf16_t*)a.C + n_main);
    if (a.swiglu == 2) { t.C2 = a.C2 + 2 * n_main; t.C = (void*)((bf16_t*)a.C + 2 * n_main); }   // [M, 2N] operands
    g_launch_ev = LaunchEvents{nullptr, ev_b};
    launch_variant(st, tail_variant, t, d.M, d.N - n_main, 1);
    g_launch_ev = LaunchEvents{};
  }
  if (timed) uvx::prof_commit();
  UVX_LAUNCH_CHECK();
  return UVX_OK;
}
