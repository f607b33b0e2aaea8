"""GPU probe: the picker on the whisper-large-v3 tower GEMM shapes (BASELINE config 3: d = 1280, ffn 5120, 12000 rows), cold weights, bias epilogue: every merged-phase tile vs the automatic choice."""
import os, sys, torch
from ultravox_amd import ops, _lib
L = _lib.lib()
dev = "cuda"
shapes = [(12000, 3840, 1280), (12000, 1280, 1280), (12000, 5120, 1280), (12000, 1280, 5120)]
for (M, N, K) in shapes:
    npool = 64
    ws = [torch.randn(N, K, device=dev).bfloat16() for _ in range(npool)]
    a = torch.randn(M, K, device=dev).bfloat16()
    bias = torch.randn(N, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    def run(fn):
        for i in range(npool): fn(ws[i])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(2):
            for i in range(npool): fn(ws[i])
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (2 * npool) * 1e3
    rec = {}
    for rnd in range(2):
        for v in (31, 32, 33, 34, -1):
            L.uvx_gemm_force_variant(v)
            key = f"v{v}" if v >= 0 else f"auto(v{L.uvx_gemm_pick_variant(M, N, K, 1)})"
            rec[key] = min(rec.get(key, 1e30), run(lambda w: ops.gemm(a, w, out=out, bias=bias)))
    L.uvx_gemm_force_variant(-1)
    print(f"{M} {N} {K} | " + " ".join(f"{k}={v:6.1f}us" for k, v in rec.items()), flush=True)
