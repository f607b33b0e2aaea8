"""GPU probe (round 4): what a SPLIT-K TAIL would buy on the two C2 shapes whose 256 x 256 tiles leave a mostly empty last round
(2528 x 14336 x 4096: 560 tiles = 2.19 rounds of 256 CUs; 2528 x 28672 x 4096: 1120 tiles = 4.375 rounds).
Arms, cold weights (a pool of weight matrices, as tools/gpu_gemm_cold_probe.py):
  whole  - one launch, the library's own choice
  split  - the whole weight panels of the full rounds as one launch + the trailing panels as ONE batched launch that splits K
           s ways into f32 partials (s x tail tiles ~ one round) + a reduction (here torch's sum + cast as a stand-in for a fused kernel)
  main / tail / reduce - the three parts alone."""
import ctypes as C

import torch
from ultravox_amd import _lib
from ultravox_amd._lib import check
from ultravox_amd.ops import stream_ptr

torch.manual_seed(0)
dev = "cuda"
L = _lib.lib()


def desc(a, w, out, M, N, K, batch=1, sa=0, sb=0, sc=0, out_f32=0):
    d = _lib.GemmDesc()
    d.A, d.B, d.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = a.stride(0), w.stride(0), out.stride(0)
    d.batch, d.stride_a, d.stride_b, d.stride_c = batch, sa, sb, sc
    d.out_f32, d.alpha = out_f32, 1.0
    return d


shapes = [(2528, 14336, 4096), (2528, 28672, 4096)]
BF = _lib.BF16
for (M, N, K) in shapes:
    npool = min(64, max(2, -(-(1200 << 20) // (N * K * 2))))
    ws = [torch.randn(N, K, device=dev).bfloat16() for _ in range(npool)]
    a = torch.randn(M, K, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    tm, tn = -(-M // 256), N // 256
    tiles = tm * tn
    main_panels = (tiles // 256) * 256 // tm
    n_main, tail_n = main_panels * 256, N - main_panels * 256
    tail_tiles = (tn - main_panels) * tm

    def run(fn, reps=2):
        for i in range(npool): fn(ws[i])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(reps):
            for i in range(npool): fn(ws[i])
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (reps * npool) * 1e3      # us

    def whole(w):
        d = desc(a, w, out, M, N, K)
        check(L.uvx_gemm(stream_ptr(), BF, C.byref(d)), "uvx_gemm")

    print(f"{M} x {N} x {K}: {tiles} tiles, main {main_panels} panels ({main_panels * tm} tiles), tail {tail_n} columns ({tail_tiles} tiles)", flush=True)
    t_whole = run(whole)
    print(f"  whole                      {t_whole:8.1f} us  {2.0 * M * N * K / t_whole / 1e6:7.1f} TF/s", flush=True)
    # the trailing panels as a second plain launch with a smaller tile (what gemm_nt's tail split does when its cost model agrees)
    L.uvx_gemm_force_variant(-2)          # automatic variant, the library's own tail split off: the arms below split by hand
    for tv in (0, 34, 33, 31):
        def main2(w):
            d = desc(a, w, out, M, n_main, K)
            check(L.uvx_gemm(stream_ptr(), BF, C.byref(d)), "uvx_gemm")

        def tail2(w):
            L.uvx_gemm_force_variant(tv)
            d = desc(a, w[n_main:], out[:, n_main:], M, tail_n, K)
            check(L.uvx_gemm(stream_ptr(), BF, C.byref(d)), "uvx_gemm")
            L.uvx_gemm_force_variant(-2)

        def both(w):
            main2(w); tail2(w)

        t_both, t_tail = run(both), run(tail2)
        print(f"  main + tail via variant {tv:2d}   {t_both:8.1f} us  {2.0 * M * N * K / t_both / 1e6:7.1f} TF/s   tail alone {t_tail:6.1f}", flush=True)
    for s in (2, 4, 8):
        if (K // 64) % s or tail_tiles * s > 320:
            continue
        Kc = K // s
        part = torch.empty(s, M, tail_n, device=dev, dtype=torch.float32)

        def main(w):
            d = desc(a, w, out, M, n_main, K)
            check(L.uvx_gemm(stream_ptr(), BF, C.byref(d)), "uvx_gemm")

        def tail(w):
            d = desc(a, w[n_main:], part, M, tail_n, Kc, batch=s, sa=Kc, sb=Kc, sc=M * tail_n, out_f32=1)
            d.ldc = tail_n
            check(L.uvx_gemm(stream_ptr(), BF, C.byref(d)), "uvx_gemm")

        def reduce(w):
            out[:, n_main:].copy_(part.sum(0))

        def split(w):
            main(w); tail(w); reduce(w)

        # correctness of the decomposition (against the whole launch, bf16 rounding of a different f32 summation order)
        whole(ws[0]); ref = out.clone(); split(ws[0]); torch.cuda.synchronize()
        err = ((out.float() - ref.float()).norm() / ref.float().norm()).item()
        t_split, t_main, t_tail, t_red = run(split), run(main), run(tail), run(reduce)
        print(f"  split-K {s} tail ({tail_tiles * s:3d} blocks) {t_split:8.1f} us  {2.0 * M * N * K / t_split / 1e6:7.1f} TF/s   "
              f"main {t_main:7.1f}  tail {t_tail:6.1f}  reduce(torch) {t_red:5.1f}   rel-L2 vs whole {err:.1e}", flush=True)
    L.uvx_gemm_force_variant(-1)
    del ws
