"""GPU probe: bf16 GEMM tile variants with COLD weights.  Inside the training step every weight matrix is read once
per launch from HBM (16 GB of frozen weights per pass), while a back-to-back probe on one weight buffer keeps it in
the 256 MB Infinity Cache and overstates every variant differently.  Here each launch takes the next weight matrix
from a pool of > 1 GB, activations stay warm (they were just written by the previous kernel in the real step)."""
import os
os.environ.setdefault("UVX_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ultravox_amd", "libuvx_probes.so"))  # probe tile variants live in the probes build
import sys
import torch
from ultravox_amd import ops, _lib

torch.manual_seed(0)
dev = "cuda"
L = _lib.lib()
VARIANTS = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2, 4, 6, 8, 10]
shapes = [(2528, 4096, 14336), (2528, 4096, 4096), (2528, 4096, 6144), (2528, 4096, 28672), (2528, 28672, 4096),
          (2528, 14336, 4096), (2528, 6144, 4096)]
if len(sys.argv) > 2 and sys.argv[2] == "enc":   # encoder / projector shapes (small weights: the pool still rotates them)
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
        for v in VARIANTS + [-1]:
            L.uvx_gemm_force_variant(v)
            key = f"v{v}" if v >= 0 else "auto"
            rec[key] = max(rec.get(key, 0.0), run(lambda w: ops.gemm(a, w, out=out)))
    rec["torch"] = run(lambda w: torch.matmul(a, w.t(), out=out))
    # the same with ONE weight buffer (Infinity-Cache warm) for the picked variant, to show the gap
    L.uvx_gemm_force_variant(-1)
    w0 = ws[0]
    rec["auto_warm"] = run(lambda w: ops.gemm(a, w0, out=out))
    print(f"{M:6d} {N:7d} {K:7d} pool={npool:3d} | " + " ".join(f"{k}={v:7.1f}" for k, v in rec.items()), flush=True)
    del ws
L.uvx_gemm_force_variant(-1)
