"""GPU probe (round 5): split-K of the prefill's GEMMs with COLD weights (every launch reads the next weight matrix of a > 1 GB pool, as
the prefill does: 139 GB of weights per pass).  For each (M, N, K): every production tile x split factor through uvx_gemm_splitk
(force_split), the cost model's own pick, and the unsplit pick; prints microseconds per launch (GEMM + reduce) and TF/s.
usage: gpu_gemm_splitk_probe.py [70b|8b|all] [M,M,...] [variant,variant,...]"""
import ctypes as C
import sys
import torch
from ultravox_amd import ops, _lib

torch.manual_seed(0)
dev = "cuda"
L = _lib.lib()
which = sys.argv[1] if len(sys.argv) > 1 else "all"
Ms = [int(m) for m in sys.argv[2].split(",")] if len(sys.argv) > 2 else [316, 632]
NK = {"70b": [(10240, 8192), (8192, 8192), (57344, 8192), (8192, 28672)], "8b": [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336)]}
shapes = [(m, n, k) for key in (["70b", "8b"] if which == "all" else [which]) for m in Ms for (n, k) in NK[key]]
VARIANTS = [int(v) for v in sys.argv[3].split(',')] if len(sys.argv) > 3 else [0, 34, 33, 32, 31]
SPLITS = [1, 2, 3, 4, 6, 8, 12, 16]
for (M, N, K) in shapes:
    npool = min(48, max(2, -(-(1200 << 20) // (N * K * 2))))
    ws = [torch.randn(N, K, device=dev).bfloat16() for _ in range(npool)]
    a = torch.randn(M, K, device=dev).bfloat16()
    resid = torch.randn(M, N, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    wsp = torch.empty(int(L.uvx_gemm_splitk_ws_bytes(M, N)), device=dev, dtype=torch.uint8)

    def run(fn):
        for i in range(npool): fn(ws[i])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(2):
            for i in range(npool): fn(ws[i])
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (2 * npool) * 1e3       # us per launch

    v = C.c_int32()
    s_auto = L.uvx_gemm_pick_split(M, N, K, C.c_size_t(wsp.numel()), C.byref(v))
    t_auto = run(lambda w: ops.gemm_splitk(a, w, residual=resid, out=out, workspace=wsp))
    t_plain = run(lambda w: ops.gemm(a, w, residual=resid, out=out))
    fl = 2.0 * M * N * K
    print(f"{M:5d} {N:6d} {K:6d} pool={npool:2d} | auto: v{v.value} s{s_auto} {t_auto:7.1f} us {fl / t_auto / 1e6:6.0f} TF/s | unsplit v{L.uvx_gemm_pick_variant(M, N, K, 1)} "
          f"{t_plain:7.1f} us {fl / t_plain / 1e6:6.0f} TF/s | weights at {N * K * 2 / t_auto / 1e6:5.2f} TB/s", flush=True)
    for var in VARIANTS:
        L.uvx_gemm_force_variant(var)
        row = []
        for s in SPLITS:
            if s > 1 and (K // 64 // s < 2 or s * M * N * 4 > wsp.numel()):
                continue
            t = run(lambda w: ops.gemm_splitk(a, w, residual=resid, out=out, workspace=wsp, force_split=s))
            row.append(f"s{s}={t:6.1f}")
        L.uvx_gemm_force_variant(-1)
        print(f"      v{var:<2d} " + " ".join(row), flush=True)
    del ws
