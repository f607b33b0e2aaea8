"""Per-K-tile cost of GEMM tile variants: time(K) is a line for a launch of exactly one round of 256 x 256 tiles (4096 x 4096 output);
slope = ns per K-tile of 64 (x clock = cycles), intercept = pipeline fill + epilogue.  Warm operands (one buffer): this isolates the
kernel's own issue / LDS / latency structure from HBM effects.  usage: gpu_gemm_ktile_probe.py v1,v2,... [M]"""
import os
os.environ.setdefault("UVX_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ultravox_amd", "libuvx_probes.so"))
import sys
import torch
from ultravox_amd import ops, _lib

L = _lib.lib()
dev = "cuda"
VARIANTS = [int(v) for v in sys.argv[1].split(",")]
M = int(sys.argv[2]) if len(sys.argv) > 2 else 4096      # 2560: 160 tiles of 256 x 256 - 160 of the 256 CUs, not power-limited
N = 4096
KS = (2048, 4096, 8192)
bufs = {K: ((torch.randn(M, K, device=dev) * 0.5).bfloat16(), (torch.randn(N, K, device=dev) * 0.5).bfloat16()) for K in KS}
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)


def t_us(v, K, iters=20):
    a, b = bufs[K]
    L.uvx_gemm_force_variant(v)
    for _ in range(3): ops.gemm(a, b, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.gemm(a, b, out=out)
    e1.record(); torch.cuda.synchronize()
    L.uvx_gemm_force_variant(-1)
    return e0.elapsed_time(e1) / iters * 1e3


for rnd in range(2):
    for v in VARIANTS:
        t = {K: t_us(v, K) for K in KS}
        slope = (t[8192] - t[2048]) / 96.0            # us per K-tile
        icpt = t[2048] - 32 * slope
        print(f"round {rnd} v{v}: " + " ".join(f"K={K}: {t[K]:7.1f} us ({2.0 * M * N * K / t[K] / 1e6:6.0f} TF/s)" for K in KS) +
              f" | {slope * 1e3:6.1f} ns per K-tile (= {slope * 1e3 * 2.0:5.0f} cycles at 2.0 GHz), fixed {icpt:5.1f} us", flush=True)
