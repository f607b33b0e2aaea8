"""GPU probe (round 5): does the weight matrix's ROW STRIDE matter?  With ldb = K (a power of two times 2 bytes: 8 / 16 / 56 KB) every row of
a 256-row weight panel starts at the same offset modulo the stride, and every CU reads the same K offset of its own rows at the same
time: if the memory system picks channels from low address bits, all of a K-tile's fetches land on few channels.  Here the same GEMMs
run on weights stored with a padded row stride (ldb = K + pad elements), cold (a pool of matrices), the cost model's own pick.
usage: gpu_gemm_ldb_probe.py"""
import torch
from ultravox_amd import ops, _lib

torch.manual_seed(0)
dev = "cuda"
L = _lib.lib()
for (M, N, K) in [(316, 57344, 8192), (316, 8192, 28672), (316, 28672, 4096), (316, 8192, 8192), (2528, 28672, 4096), (2528, 4096, 14336), (2528, 4096, 4096)]:
    row = []
    for pad in (0, 64, 128, 256, 320):
        npool = min(24, max(2, -(-(1200 << 20) // (N * (K + pad) * 2))))
        ws = [torch.randn(N, K + pad, device=dev).bfloat16()[:, :K] for _ in range(npool)]
        a = torch.randn(M, K, device=dev).bfloat16()
        resid = torch.randn(M, N, device=dev).bfloat16()
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        wsp = torch.empty(int(L.uvx_gemm_splitk_ws_bytes(M, N)), device=dev, dtype=torch.uint8)
        fn = (lambda w: ops.gemm_splitk(a, w, residual=resid, out=out, workspace=wsp)) if M < 1000 else (lambda w: ops.gemm(a, w, residual=resid, out=out))
        for w in ws: fn(w)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(3):
            for w in ws: fn(w)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / (3 * npool) * 1e3
        row.append(f"pad{pad}={t:7.1f} us ({2.0 * M * N * K / t / 1e6:5.0f} TF/s)")
        del ws
    print(f"{M:5d} {N:6d} {K:6d} | " + "  ".join(row), flush=True)
