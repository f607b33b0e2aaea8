"""GPU check of the four-wave hand-scheduled GEMM (tile variants 43..45): every epilogue it serves must be BIT-IDENTICAL to
the eight-wave 256 x 256 kernel (variant 31: same MFMA, operand roles and k order), on ragged / tiny-K / deep-K shapes, and
30 repeats of one launch must be bit-identical (race screen: a mis-placed wait or barrier shows up as rare different tiles).
usage: gpu_gemm_a4_check.py [variants, default 43]"""
import os
os.environ.setdefault("UVX_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ultravox_amd", "libuvx_probes.so"))
import sys
import torch
from ultravox_amd import ops, _lib

VARIANTS = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [43]
REF = 31
torch.manual_seed(0)
dev = "cuda"
L = _lib.lib()
SHAPES = [(256, 256, 64), (256, 256, 128), (256, 256, 192), (256, 512, 256), (300, 520, 192), (77, 136, 128), (1000, 1032, 640),
          (2528, 4096, 256), (2528, 6144, 4096), (4096, 4096, 4096), (12000, 1024, 1024), (2528, 4096, 14336)]


def run(v, fn):
    L.uvx_gemm_force_variant(v)
    try:
        return fn()
    finally:
        L.uvx_gemm_force_variant(-1)


def modes(M, N, K):
    a = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
    b = (torch.randn(N, K, device=dev) * 0.5).bfloat16()
    bias = torch.randn(N, device=dev).bfloat16()
    resid = torch.randn(M, N, device=dev).bfloat16()
    pos = torch.randn(64, N, device=dev).bfloat16()
    acc0 = torch.randn(M, N, device=dev)
    yield "plain", lambda: (ops.gemm(a, b),)
    yield "bias+res", lambda: (ops.gemm(a, b, bias=bias, residual=resid),)
    yield "bias+gelu", lambda: (ops.gemm(a, b, bias=bias, act="gelu"),)
    yield "res_mod+alpha", lambda: (ops.gemm(a, b, residual=pos, res_mod=64, alpha=0.5),)
    yield "f32", lambda: (ops.gemm(a, b, out_f32=True),)
    yield "f32+acc", lambda: (ops.gemm(a, b, out_f32=True, accumulate=True, out=acc0.clone()),)
    if N % 32 == 0:
        def sw():
            c2 = torch.empty(M, N // 2, device=dev, dtype=torch.bfloat16)
            out = ops.gemm(a, b, epilogue=1, c2=c2)
            return out, c2
        yield "swiglu", sw


ok = True
for v in VARIANTS:
    for (M, N, K) in SHAPES:
        bad = []
        for name, fn in modes(M, N, K):
            want = run(REF, fn)
            got = run(v, fn)
            if not all(torch.equal(g, w) for g, w in zip(got, want)):
                d = max((g.float() - w.float()).abs().max().item() for g, w in zip(got, want))
                nbad = sum(int((g != w).sum().item()) for g, w in zip(got, want))
                bad.append(f"{name}: {nbad} elements differ (max {d:.4g})")
        # sanity against torch as well (the reference variant could be wrong too)
        a = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
        b = (torch.randn(N, K, device=dev) * 0.5).bfloat16()
        got = run(v, lambda: ops.gemm(a, b))
        ref = a.float() @ b.float().t()
        err = (got.float() - ref).abs().max().item()
        tol = 2e-2 * ref.abs().max().item() + 1e-3
        races = 0
        for _ in range(30):
            races += int(not torch.equal(run(v, lambda: ops.gemm(a, b)), got))
        good = not bad and err <= tol and races == 0
        ok &= good
        print(f"v{v} {M}x{N}x{K}: vs torch {err:.4f} (tol {tol:.4f}), nondeterministic repeats {races}/30, "
              f"{'bit-identical to v%d in every mode' % REF if not bad else '; '.join(bad)} {'OK' if good else 'FAIL'}", flush=True)
print("ALL_OK" if ok else "SOME_FAILED")
