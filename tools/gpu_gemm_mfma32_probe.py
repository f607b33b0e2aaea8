"""GPU probe (round 6): the merged-phase bf16 GEMM on 32 x 32 x 16 MFMAs (variants 61 / 62) against its 16 x 16 x 32 twins (31 / 34) - same tile, same
DMA schedule - on the seven LLM shapes of the C2 step and the four encoder shapes, COLD weights (every launch reads the next weight matrix of a
> 1 GB pool, as the training step does).  Wall time per launch by HIP events over the pool; a second pass under rocprofv3 --pmc (tools/calls) gives
GRBM_GUI_ACTIVE / duration = the clock the chip sustained under each body.  usage: gpu_gemm_mfma32_probe.py [variants] [rounds]"""
import sys
import torch
from ultravox_amd import ops, _lib

torch.manual_seed(0)
dev = "cuda"
L = _lib.lib()
VARIANTS = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [31, 61, 34, 62, 33]
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
shapes = [(2528, 4096, 28672), (2528, 28672, 4096), (2528, 14336, 4096), (2528, 4096, 14336), (2528, 4096, 4096), (2528, 4096, 6144), (2528, 6144, 4096),
          (12000, 4096, 1024), (12000, 1024, 4096), (12000, 3072, 1024), (12000, 1024, 1024)]
for (M, N, K) in shapes:
    npool = min(64, max(2, -(-(1200 << 20) // (N * K * 2))))
    ws = [torch.randn(N, K, device=dev).bfloat16() for _ in range(npool)]
    a = torch.randn(M, K, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)

    def run(fn):
        for i in range(npool):
            fn(ws[i])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(2):
            for i in range(npool):
                fn(ws[i])
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (2 * npool) * 1e3      # us per launch

    rec = {}
    for rnd in range(ROUNDS):
        for v in VARIANTS + [-1]:
            L.uvx_gemm_force_variant(v)
            key = f"v{v}" if v >= 0 else "auto"
            rec[key] = min(rec.get(key, 1e30), run(lambda w: ops.gemm(a, w, out=out)))
    L.uvx_gemm_force_variant(-1)
    fl = 2.0 * M * N * K
    print(f"{M:6d} {N:7d} {K:7d} pool={npool:3d} auto=v{L.uvx_gemm_pick_variant(M, N, K, 1)} | " +
          " ".join(f"{k}={v:7.1f}us ({fl / v / 1e6:6.0f} TF/s)" for k, v in rec.items()), flush=True)
    del ws
