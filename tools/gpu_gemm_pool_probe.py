"""Does the SIZE of the weight pool matter?  The training step streams 32 GB of frozen weights per pass, each matrix once every ~85 ms;
the cold-weight probes cycle through ~2 GB.  The same launch with pools of 2 / 8 / 24 GB (events around every launch, median).
Usage: python tools/gpu_gemm_pool_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ultravox_amd import ops  # noqa: E402

DEV = "cuda"


def run(M, N, K, pool_gb, one_alloc):
    n_w = max(4, int(pool_gb * 2 ** 30 // (N * K * 2)))
    a = torch.randn(M, K, device=DEV).bfloat16()
    if one_alloc:       # one big allocation carved into matrices (what a packed checkpoint buffer would be)
        big = torch.empty(n_w, N, K, device=DEV, dtype=torch.bfloat16)
        big.normal_()
        ws = [big[i] for i in range(n_w)]
    else:               # one allocation per matrix (what the per-tensor state dict gives)
        ws = [torch.randn(N, K, device=DEV).bfloat16() for _ in range(n_w)]
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    for w in ws[:3]:
        ops.gemm(a, w, out=out)
    torch.cuda.synchronize()
    evs = []
    for rep in range(2):
        for w in ws:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.gemm(a, w, out=out); e1.record()
            evs.append((e0, e1))
    torch.cuda.synchronize()
    t = sorted(x.elapsed_time(y) for x, y in evs[len(ws):])          # second sweep: every matrix last touched a whole sweep ago
    del ws
    torch.cuda.empty_cache()
    return t[len(t) // 2] * 1e3, n_w


if __name__ == "__main__":
    for (M, N, K) in [(2528, 6144, 4096), (2528, 14336, 4096)]:
        for one_alloc in (False, True):
            line = f"{M} x {N} x {K} {'one allocation ' if one_alloc else 'per-matrix alloc'}:"
            for gb in (2, 8, 24):
                us, n_w = run(M, N, K, gb, one_alloc)
                line += f"  pool {gb:2d} GB ({n_w:3d} matrices) {us:6.1f} us ({2.0 * M * N * K / us / 1e6:6.1f} TF/s)"
            print(line, flush=True)
