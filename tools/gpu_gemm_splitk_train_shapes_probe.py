"""GPU probe (round 5): would split-K help the training step's deep-K, single-round GEMMs?  2528 x 4096 x 28672 (the gate|up dgrad: 256 tiles of
160 x 256 = one round, 448 K-tiles each; the step's largest item, where hipBLASLt is 8-10 % ahead on cold weights) and 2528 x 4096 x 14336, every
merged-phase tile x split factor through uvx_gemm_splitk (reduce included), cold weights.  usage: gpu_gemm_splitk_train_shapes_probe.py"""
import torch
from ultravox_amd import ops, _lib

dev = "cuda"
L = _lib.lib()
torch.manual_seed(0)
for (M, N, K) in [(2528, 4096, 28672), (2528, 4096, 14336), (2528, 4096, 4096)]:
    npool = min(48, max(2, -(-(1200 << 20) // (N * K * 2))))
    ws = [torch.randn(N, K, device=dev).bfloat16() for _ in range(npool)]
    a = torch.randn(M, K, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    wsp = torch.empty(int(L.uvx_gemm_splitk_ws_bytes(M, N)), device=dev, dtype=torch.uint8)

    def run(fn):
        for i in range(npool): fn(ws[i])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(2):
            for i in range(npool): fn(ws[i])
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (2 * npool) * 1e3
    fl = 2.0 * M * N * K
    t_plain = run(lambda w: ops.gemm(a, w, out=out))
    t_torch = run(lambda w: torch.matmul(a, w.t(), out=out))
    print(f"{M} x {N} x {K}: production pick {t_plain:7.1f} us ({fl / t_plain / 1e6:5.0f} TF/s), hipBLASLt {t_torch:7.1f} us ({fl / t_torch / 1e6:5.0f} TF/s)", flush=True)
    for var in (31, 32, 33, 34):
        L.uvx_gemm_force_variant(var)
        row = []
        for s in (1, 2, 3, 4, 5, 8):
            if s > 1 and s * M * N * 4 > wsp.numel():
                continue
            t = run(lambda w: ops.gemm_splitk(a, w, out=out, workspace=wsp, force_split=s))
            row.append(f"s{s}={t:6.1f} ({fl / t / 1e6:4.0f})")
        L.uvx_gemm_force_variant(-1)
        print(f"   v{var}: " + "  ".join(row), flush=True)
    del ws
