"""GPU probe: fixed per-tile cost of a bf16 GEMM tile variant.  time(K) for K = 64 .. 4096 at fixed M, N is a straight line
a + b * (K / 64); a / rounds is the pipeline fill + epilogue cost of one round of tiles, b the steady-state K-tile time."""
import os
os.environ.setdefault("UVX_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ultravox_amd", "libuvx_probes.so"))  # probe tile variants live in the probes build
import sys
import torch
from ultravox_amd import ops, _lib

dev = "cuda"
L = _lib.lib()
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [11]
shapes = [(2528, 28672), (2528, 6144), (12000, 4096)]
if len(sys.argv) > 2: shapes = shapes[: int(sys.argv[2])]
for (M, N) in shapes:
    for v in variants:
        L.uvx_gemm_force_variant(v)
        pts = []
        for K in (64, 128, 256, 512, 1024, 2048, 4096):
            a = torch.randn(M, K, device=dev).bfloat16()
            ws = [torch.randn(N, K, device=dev).bfloat16() for _ in range(max(2, min(16, (600 << 20) // (N * K * 2))))]
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            for w in ws: ops.gemm(a, w, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 4
            e0.record()
            for _ in range(reps):
                for w in ws: ops.gemm(a, w, out=out)
            e1.record(); torch.cuda.synchronize()
            pts.append((K // 64, e0.elapsed_time(e1) * 1e3 / (reps * len(ws))))
        # least squares on the last 4 points (steady state), intercept from them
        xs = [p[0] for p in pts[-4:]]; ys = [p[1] for p in pts[-4:]]
        n = len(xs); sx, sy = sum(xs), sum(ys); sxx = sum(x * x for x in xs); sxy = sum(x * y for x, y in zip(xs, ys))
        b = (n * sxy - sx * sy) / (n * sxx - sx * sx); a0 = (sy - b * sx) / n
        print(f"v{v} M={M} N={N}: " + " ".join(f"nk={k}:{t:.1f}us" for k, t in pts) + f" | fit: {a0:.1f} us + {b:.2f} us/k-tile  (fixed cost = {a0 / b:.1f} k-tiles)", flush=True)
L.uvx_gemm_force_variant(-1)
