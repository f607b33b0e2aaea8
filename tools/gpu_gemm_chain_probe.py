"""Why is a K = 4096 forward GEMM 10-25 % slower inside the step than in the cold-weight probe loop?  The same launch (2528 x 6144 x 4096,
cold weights) timed with events around it (a) back to back, (b) right after the RMSNorm kernel that PRODUCES its activation operand,
(c) after an unrelated HBM-bound kernel of the same size, (d) after an idle gap.  Usage: python tools/gpu_gemm_chain_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ultravox_amd import ops  # noqa: E402

DEV = "cuda"


def run(M, N, K, mode, pool=40, reps=3):
    x = torch.randn(M, K, device=DEV).bfloat16()
    w_ln = torch.ones(K, device=DEV).bfloat16()
    other = torch.randn(M, K, device=DEV).bfloat16()
    ws = [torch.randn(N, K, device=DEV).bfloat16() for _ in range(pool)]
    outs = [torch.empty(M, N, device=DEV, dtype=torch.bfloat16) for _ in range(8)]
    n = ops.rmsnorm(x, w_ln)
    best = 1e9
    for _ in range(reps):
        evs = []
        for i, w in enumerate(ws):
            if mode == "after_producer":
                n = ops.rmsnorm(x, w_ln)
            elif mode == "after_unrelated":
                _ = ops.rmsnorm(other, w_ln)
            elif mode == "after_gap":
                torch.cuda.synchronize(); time.sleep(0.0005)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.gemm(n, w, out=outs[i % 8])
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) for a, b in evs[4:])
        best = min(best, t[len(t) // 2])
    return best * 1e3


if __name__ == "__main__":
    for (M, N, K) in [(2528, 6144, 4096), (2528, 14336, 4096), (2528, 4096, 14336)]:
        line = f"{M} x {N} x {K}:"
        for mode in ("back_to_back", "after_producer", "after_unrelated", "after_gap"):
            us = run(M, N, K, mode)
            line += f"  {mode} {us:6.1f} us ({2.0 * M * N * K / us / 1e6:6.1f} TF/s)"
        print(line, flush=True)
