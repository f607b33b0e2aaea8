"""What the fused epilogues of the encoder's short-K GEMMs cost: the same launch with no epilogue / bias / bias + GELU / bias +
residual, cold weights (a pool of weight matrices larger than the Infinity Cache, one per launch, like the training step).
Usage: python tools/gpu_gemm_epilogue_probe.py  (on the GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ultravox_amd import ops  # noqa: E402

DEV = "cuda"


def bench(M, N, K, mode, pool=48, reps=3, cold_io=False):
    """cold_io: the activation operand and the output rotate through pools larger than the Infinity Cache as well (in the step
    a layer's activations were last touched ~0.5 GB of traffic ago)."""
    n_io = max(2, int(600e6 // (M * max(N, K) * 2))) if cold_io else 1
    acts = [torch.randn(M, K, device=DEV).bfloat16() for _ in range(n_io)]
    outs = [torch.empty(M, N, device=DEV, dtype=torch.bfloat16) for _ in range(n_io)]
    ws = [torch.randn(N, K, device=DEV).bfloat16() for _ in range(pool)]
    bias = torch.randn(N, device=DEV).bfloat16()
    res = [torch.randn(M, N, device=DEV).bfloat16() for _ in range(n_io)]
    kws = [{"none": {}, "bias": {"bias": bias}, "bias+gelu": {"bias": bias, "act": "gelu"},
            "bias+residual": {"bias": bias, "residual": res[i]}}[mode] for i in range(n_io)]
    for i, w in enumerate(ws[:4]):
        ops.gemm(acts[i % n_io], w, out=outs[i % n_io], **kws[i % n_io])
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i, w in enumerate(ws):
            ops.gemm(acts[i % n_io], w, out=outs[i % n_io], **kws[i % n_io])
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / pool)
    return best * 1e3


if __name__ == "__main__":
    print("# M N K | mode: us (TF/s)")
    for (M, N, K) in [(12000, 4096, 1024), (12000, 1024, 4096), (12000, 3072, 1024), (12000, 1024, 1024), (2528, 6144, 4096), (2528, 4096, 4096),
                      (2528, 14336, 4096), (1264, 6144, 4096), (1264, 4096, 4096)]:
        line = f"{M:6d} {N:6d} {K:6d} |"
        for mode in ("none", "bias", "bias+gelu", "bias+residual"):
            us = bench(M, N, K, mode)
            line += f" {mode}: {us:7.1f} us ({2.0 * M * N * K / us / 1e6:7.1f})"
        for mode in ("none", "bias+gelu", "bias+residual"):
            us = bench(M, N, K, mode, cold_io=True)
            line += f" | cold-io {mode}: {us:7.1f} us ({2.0 * M * N * K / us / 1e6:7.1f})"
        print(line, flush=True)
