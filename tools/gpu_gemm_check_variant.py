"""GPU check of ONE bf16 GEMM tile variant: correctness on ragged / tiny-K / deep-K shapes with every epilogue, and a
race screen (the same launch repeated 30x must be bit-identical: a mis-placed wait or barrier shows up as rare
different tiles).  usage: gpu_gemm_check_variant.py <variant>"""
import os
os.environ.setdefault("UVX_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ultravox_amd", "libuvx_probes.so"))  # probe tile variants live in the probes build
import sys
import torch
from ultravox_amd import ops, _lib

v = int(sys.argv[1])
torch.manual_seed(0)
dev = "cuda"
L = _lib.lib()
ok = True
for (M, N, K) in [(256, 256, 64), (256, 256, 128), (300, 520, 192), (2528, 4096, 256), (77, 132, 128), (1000, 1028, 640),
                  (2528, 6144, 4096), (4096, 4096, 4096), (12000, 1024, 1024)]:
    a = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
    b = (torch.randn(N, K, device=dev) * 0.5).bfloat16()
    bias = torch.randn(N, device=dev).bfloat16()
    resid = torch.randn(M, N, device=dev).bfloat16()
    L.uvx_gemm_force_variant(v)
    out = ops.gemm(a, b, bias=bias, residual=resid)
    L.uvx_gemm_force_variant(0)
    base = ops.gemm(a, b, bias=bias, residual=resid)      # 128x128 kernel: same arithmetic per element up to k order
    ref = (a.float() @ b.float().t() + bias.float()).bfloat16().float() + resid.float()
    err = (out.float() - ref).abs().max().item()
    tol = 2e-2 * ref.abs().max().item() + 1e-3
    same = (out.float() - base.float()).abs().max().item()
    L.uvx_gemm_force_variant(v)
    first = ops.gemm(a, b)
    races = 0
    for _ in range(30):
        races += int(not torch.equal(ops.gemm(a, b), first))
    good = err <= tol and races == 0
    ok &= good
    print(f"v{v} {M}x{N}x{K}: max err {err:.4f} (tol {tol:.4f}), vs v0 {same:.4f}, nondeterministic repeats {races}/30 {'OK' if good else 'FAIL'}", flush=True)
L.uvx_gemm_force_variant(-1)
print("ALL_OK" if ok else "SOME_FAILED")
