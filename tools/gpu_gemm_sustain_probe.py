"""Is the gap between the probe loops (1250-1340 TF/s) and the same GEMM inside the training step (~1030 TF/s) a matter of DURATION?
One launch shape, cold weights, run back to back for several seconds; the rate per half-second window.  Usage: python tools/gpu_gemm_sustain_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ultravox_amd import ops  # noqa: E402

DEV = "cuda"
M, N, K = 2528, 6144, 4096
a = torch.randn(M, K, device=DEV).bfloat16()
ws = [torch.randn(N, K, device=DEV).bfloat16() for _ in range(160)]
out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
for w in ws[:4]:
    ops.gemm(a, w, out=out)
torch.cuda.synchronize()
t_start = time.perf_counter()
win = 0
while time.perf_counter() - t_start < 8.0:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 0
    for rep in range(30):
        for w in ws:
            ops.gemm(a, w, out=out)
            n += 1
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(f"t = {time.perf_counter() - t_start:5.2f} s  {n} launches  {ms / n * 1e3:6.1f} us each  {2.0 * M * N * K * n / ms / 1e9:7.1f} TF/s", flush=True)
