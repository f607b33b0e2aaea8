"""GPU probe (round 5): the prefill's few-hundred-row GEMMs on cold weights - this library (split-K picked by its cost model, reduce included)
against hipBLASLt as torch.matmul reaches it (a second opinion on what 'good' is at M = 316 / 632; tools only: the product never calls torch.matmul).
Same method as gpu_gemm_splitk_probe.py: every launch takes the next weight matrix of a > 1 GB pool.
usage: gpu_prefill_vs_blaslt_probe.py [70b|8b|all] [M,M,...]"""
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
tot = {}
print("#     M      N      K | libuvx (auto split-K + reduce)        | hipBLASLt (torch.matmul)   | libuvx / hipBLASLt time")
for key in (["70b", "8b"] if which == "all" else [which]):
    for M in Ms:
        for (N, K) in NK[key]:
            npool = min(48, max(2, -(-(1200 << 20) // (N * K * 2))))
            ws = [torch.randn(N, K, device=dev).bfloat16() for _ in range(npool)]
            a = torch.randn(M, K, device=dev).bfloat16()
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
            best = {"uvx": 1e30, "blaslt": 1e30}
            for rnd in range(2):
                best["uvx"] = min(best["uvx"], run(lambda w: ops.gemm_splitk(a, w, out=out, workspace=wsp)))
                best["blaslt"] = min(best["blaslt"], run(lambda w: torch.matmul(a, w.t(), out=out)))
            fl = 2.0 * M * N * K
            print(f"  {M:5d} {N:6d} {K:6d} | v{v.value} s{s_auto:<2d} {best['uvx']:7.1f} us {fl / best['uvx'] / 1e6:6.0f} TF/s | "
                  f"{best['blaslt']:7.1f} us {fl / best['blaslt'] / 1e6:6.0f} TF/s | {best['uvx'] / best['blaslt']:5.2f}", flush=True)
            t = tot.setdefault((key, M), [0.0, 0.0])
            t[0] += best["uvx"]; t[1] += best["blaslt"]
            del ws
for (key, M), (u, b) in tot.items():
    print(f"# {key} layer (q|k|v + o + gate|up + down) at M = {M}: libuvx {u:7.1f} us, hipBLASLt {b:7.1f} us, ratio {u / b:.2f}")
