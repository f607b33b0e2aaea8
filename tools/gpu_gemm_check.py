"""GPU probe: GEMM correctness (transpose-detecting, asymmetric data) + throughput on hot-path shapes."""
import os
os.environ.setdefault("UVX_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ultravox_amd", "libuvx_probes.so"))  # probe tile variants live in the probes build
import json
import torch
from ultravox_amd import ops

torch.manual_seed(0)
dev = "cuda"
res = {"checks": [], "perf": []}

def check(M, N, K, **kw):
    a = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
    b = (torch.randn(N, K, device=dev) * 0.5).bfloat16()
    bias = torch.randn(N, device=dev).bfloat16() if kw.get("bias") else None
    resid = torch.randn(M, N, device=dev).bfloat16() if kw.get("residual") else None
    out = ops.gemm(a, b, bias=bias, residual=resid, act=kw.get("act", "none"), out_f32=kw.get("out_f32", False))
    ref = a.float() @ b.float().t()
    if bias is not None: ref = ref + bias.float()
    if not kw.get("out_f32"): ref = ref.bfloat16().float()
    if kw.get("act") == "gelu": ref = torch.nn.functional.gelu(ref).bfloat16().float()
    if resid is not None: ref = ref + resid.float()
    err = (out.float() - ref).abs().max().item()
    scale = ref.abs().max().item()
    res["checks"].append({"M": M, "N": N, "K": K, **kw, "max_abs_err": err, "ref_max": scale,
                          "ok": bool(err <= 2e-2 * scale + 1e-3)})

for (M, N, K) in [(128, 128, 64), (256, 384, 128), (100, 132, 192), (2528, 4096, 4096), (1504, 2048, 8192), (37, 8, 64)]:
    check(M, N, K)
check(300, 256, 256, bias=True)
check(300, 256, 256, bias=True, act="gelu")
check(300, 256, 256, bias=True, residual=True)
check(300, 256, 1536, out_f32=True)

def perf(M, N, K, iters=20):
    a = torch.randn(M, K, device=dev).bfloat16()
    b = torch.randn(N, K, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3): ops.gemm(a, b, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.gemm(a, b, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    tf = 2.0 * M * N * K / ms / 1e9
    # torch (hipBLASLt) as a second opinion on what the chip does for this shape
    for _ in range(3): torch.matmul(a, b.t())
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): torch.matmul(a, b.t())
    e1.record(); torch.cuda.synchronize()
    ms2 = e0.elapsed_time(e1) / iters
    res["perf"].append({"M": M, "N": N, "K": K, "ms": ms, "TF": tf, "torch_ms": ms2,
                        "torch_TF": 2.0 * M * N * K / ms2 / 1e9})

for shp in [(4096, 4096, 4096), (8192, 8192, 8192), (2528, 6144, 4096), (2528, 4096, 4096), (2528, 28672, 4096),
            (2528, 4096, 14336), (2528, 14336, 4096), (2528, 4096, 28672), (2528, 128256, 4096), (2528, 4096, 128256),
            (12000, 3072, 1024), (12000, 1024, 1024), (12000, 4096, 1024), (12000, 1024, 4096), (1504, 4096, 8192)]:
    perf(*shp)
print(json.dumps(res, indent=1))
ok = all(c["ok"] for c in res["checks"])
print("ALL_OK" if ok else "SOME_FAILED")
