"""GPU probe: stream-K variants (39..42) next to their data-parallel twins (31..34) and torch (hipBLASLt), cold weights
(same method as gpu_gemm_cold_probe.py: every launch reads the next weight matrix of a > 1 GB pool).  TF/s.
usage: gpu_gemm_streamk_probe.py [llm|enc]"""
import os
os.environ.setdefault("UVX_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ultravox_amd", "libuvx_probes.so"))  # probe tile variants live in the probes build
import sys
import torch
from ultravox_amd import ops, _lib

torch.manual_seed(0)
dev = "cuda"
L = _lib.lib()
shapes = [(2528, 4096, 14336), (2528, 4096, 4096), (2528, 4096, 6144), (2528, 4096, 28672), (2528, 28672, 4096),
          (2528, 14336, 4096), (2528, 6144, 4096)]
if len(sys.argv) > 1 and sys.argv[1] == "enc":
    shapes = [(12000, 4096, 1024), (12000, 1024, 4096), (12000, 3072, 1024), (12000, 1024, 1024), (1504, 4096, 8192),
              (1504, 8192, 4096), (4096, 8192, 1536)]
for (M, N, K) in shapes:
    npool = min(64, max(2, -(-(1200 << 20) // (N * K * 2))))
    ws = [torch.randn(N, K, device=dev).bfloat16() for _ in range(npool)]
    a = torch.randn(M, K, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    rec = {}
    def run(fn):
        for i in range(npool): fn(ws[i])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(2):
            for i in range(npool): fn(ws[i])
        e1.record(); torch.cuda.synchronize()
        return 2.0 * M * N * K / (e0.elapsed_time(e1) / (2 * npool)) / 1e9
    for rnd in range(2):
        for v, o9 in [(31, 1), (39, 1), (39, 0), (32, 1), (40, 1), (40, 0), (33, 1), (41, 1), (41, 0), (-1, 1)]:
            L.uvx_gemm_force_variant(v)
            L.uvx_set_option(9, o9)
            L.uvx_set_option(8, 0)
            key = (f"v{v}" + ("" if o9 else "p")) if v >= 0 else "auto"
            rec[key] = max(rec.get(key, 0.0), run(lambda w: ops.gemm(a, w, out=out)))
    L.uvx_gemm_force_variant(-1)
    L.uvx_set_option(8, 1)
    rec["auto_sk"] = run(lambda w: ops.gemm(a, w, out=out))
    rec["picked"] = L.uvx_gemm_pick_variant(M, N, K, 1)
    L.uvx_set_option(8, 0)
    rec["torch"] = run(lambda w: torch.matmul(a, w.t(), out=out))
    print(f"{M:6d} {N:7d} {K:7d} pool={npool:3d} | " + " ".join(f"{k}={v:7.1f}" for k, v in rec.items()) + f" give-ups={L.uvx_gemm_streamk_timeouts()}", flush=True)
    del ws
L.uvx_gemm_force_variant(-1)
