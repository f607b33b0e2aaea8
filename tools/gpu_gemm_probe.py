"""GPU probe: every bf16 GEMM tile variant — correctness (asymmetric data, ragged M/N) and TF/s on the
hot-path shapes, interleaved in one process (within-probe A/B)."""
import os
os.environ.setdefault("UVX_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ultravox_amd", "libuvx_probes.so"))  # probe tile variants live in the probes build
import json
import torch
from ultravox_amd import ops, _lib

torch.manual_seed(0)
dev = "cuda"
L = _lib.lib()
res = {"checks": [], "perf": []}
VARIANTS = [2, 4, 8, 10]

def check(v, M, N, K):
    L.uvx_gemm_force_variant(v)
    a = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
    b = (torch.randn(N, K, device=dev) * 0.5).bfloat16()
    bias = torch.randn(N, device=dev).bfloat16()
    resid = torch.randn(M, N, device=dev).bfloat16()
    out = ops.gemm(a, b, bias=bias, residual=resid)
    ref = (a.float() @ b.float().t() + bias.float()).bfloat16().float() + resid.float()
    err = (out.float() - ref).abs().max().item()
    res["checks"].append({"v": v, "M": M, "N": N, "K": K, "err": err, "ok": bool(err <= 2e-2 * ref.abs().max().item() + 1e-3)})

for v in VARIANTS:
    for shp in [(256, 256, 64), (300, 520, 192), (2528, 4096, 256), (77, 132, 128), (1000, 1028, 640)]:
        check(v, *shp)

def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

shapes = [(2528, 6144, 4096), (2528, 4096, 4096), (2528, 28672, 4096), (2528, 4096, 14336), (2528, 14336, 4096),
          (2528, 4096, 28672), (2528, 4096, 6144), (2528, 128256, 4096), (2528, 4096, 128256),
          (12000, 3072, 1024), (12000, 1024, 1024), (12000, 4096, 1024), (12000, 1024, 4096), (12000, 1024, 3072),
          (1504, 4096, 8192), (1504, 4096, 2048), (4096, 4096, 4096), (8192, 8192, 8192)]
for (M, N, K) in shapes:
    a = torch.randn(M, K, device=dev).bfloat16()
    b = torch.randn(N, K, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    rec = {"M": M, "N": N, "K": K}
    for rnd in range(2):
        for v in VARIANTS + [-1, -2]:
            L.uvx_gemm_force_variant(v)
            ms = timeit(lambda: ops.gemm(a, b, out=out))
            key = f"v{v}" if v >= 0 else ("auto" if v == -1 else "nosplit")
            rec[key] = max(rec.get(key, 0.0), 2.0 * M * N * K / ms / 1e9)
    ms = timeit(lambda: torch.matmul(a, b.t()))
    rec["torch"] = 2.0 * M * N * K / ms / 1e9
    res["perf"].append(rec)
L.uvx_gemm_force_variant(-1)
print(json.dumps(res))
print("ALL_OK" if all(c["ok"] for c in res["checks"]) else "SOME_FAILED")
for r in res["perf"]:
    print(f"{r['M']:6d} {r['N']:7d} {r['K']:7d} | " + " ".join(f"{k}={r[k]:7.1f}" for k in ["v2","v4","v8","v10","nosplit","auto","torch"]))
