"""Hypothesis test: are the GEMM operand loads limited by L2-channel conflicts of power-of-two row strides?
Same GEMM, weight (and/or activation) rows padded by a few hundred bytes."""
import os
os.environ.setdefault("UVX_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ultravox_amd", "libuvx_probes.so"))  # probe tile variants live in the probes build
import torch
from ultravox_amd import ops, _lib
L = _lib.lib(); dev = "cuda"
def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (M, N, K) in [(2528, 28672, 4096), (2528, 4096, 28672), (2528, 4096, 4096), (2528, 14336, 4096), (12000, 4096, 1024)]:
    for v in (4, 2, 8):
        L.uvx_gemm_force_variant(v)
        row = {}
        for padw, pada in [(0, 0), (64, 0), (128, 0), (64, 64), (256, 256), (8, 8)]:
            A = torch.randn(M, K + pada, device=dev).bfloat16()[:, :K]
            B = torch.randn(N, K + padw, device=dev).bfloat16()[:, :K]
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            ms = timeit(lambda: ops.gemm(A, B, out=out))
            row[f"w+{padw},a+{pada}"] = round(2.0 * M * N * K / ms / 1e9, 1)
        print(M, N, K, "v", v, row, flush=True)
