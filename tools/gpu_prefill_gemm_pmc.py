"""GPU probe (round 5), meant to run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (its own pass): the prefill's GEMMs at Llama-3.3-70B
shapes with cold weights (two alternating weight matrices, together far beyond the 256 MB Infinity Cache where they fit at all), the cost
model's own (tile, split) pick.  tools/calls/r5_call2.sh turns the counter file into memory-side read bytes per launch against the
algorithmic bytes (weights once + activations once): whether the two 160-row tiles that share a weight panel also share its fetch.
usage: gpu_prefill_gemm_pmc.py [M]"""
import sys
import torch
from ultravox_amd import ops, _lib

M = int(sys.argv[1]) if len(sys.argv) > 1 else 316
dev = "cuda"
torch.manual_seed(0)
L = _lib.lib()
for (N, K) in [(57344, 8192), (8192, 28672), (8192, 8192), (10240, 8192)]:
    ws = [torch.randn(N, K, device=dev).bfloat16() for _ in range(2)]
    a = torch.randn(M, K, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    wsp = torch.empty(int(L.uvx_gemm_splitk_ws_bytes(M, N)), device=dev, dtype=torch.uint8)
    for r in range(3):
        for w in ws:
            ops.gemm_splitk(a, w, out=out, workspace=wsp)
    torch.cuda.synchronize()
    print(f"{M} {N} {K}: algorithmic read bytes {(N * K + M * K) * 2 / 1e6:.1f} MB", flush=True)
    del ws
