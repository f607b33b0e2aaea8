"""GPU check of the stream-K GEMM variants (39..42): every epilogue on shapes whose tiles split across blocks in each way
(one K-tile, fewer K-tiles than blocks, tiles < CUs, several data-parallel rounds + a partial one, ragged edges), both
flavours (option 9), bit-identical repeats (fixed ranges and a fixed order of partial sums), bit-identical or 1-ulp
agreement with the data-parallel twin, and the give-up counter at zero.
usage: gpu_gemm_streamk_check.py [variants, default 39,40,41,42]"""
import os
os.environ.setdefault("UVX_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ultravox_amd", "libuvx_probes.so"))  # probe tile variants live in the probes build
import sys
import torch
from ultravox_amd import ops, _lib

variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [39, 40, 41, 42]
torch.manual_seed(0)
dev = "cuda"
L = _lib.lib()
ok = True
shapes = [(256, 256, 64), (256, 256, 128), (300, 520, 192), (77, 132, 128), (1000, 1028, 640), (2528, 4096, 256),
          (2528, 6144, 4096), (4096, 4096, 4096), (12000, 1024, 1024), (2528, 4096, 14336), (2528, 4096, 28672),
          (2528, 28672, 4096), (2528, 14336, 4096), (12000, 4096, 1024), (1504, 4096, 8192)]
for v in variants:
    for flavour in (1, 0):
        L.uvx_set_option(9, flavour)
        for (M, N, K) in shapes:
            a = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
            b = (torch.randn(N, K, device=dev) * 0.5).bfloat16()
            bias = torch.randn(N, device=dev).bfloat16()
            resid = torch.randn(M, N, device=dev).bfloat16()
            L.uvx_gemm_force_variant(v)
            out = ops.gemm(a, b, bias=bias, residual=resid)
            L.uvx_gemm_force_variant(v - 8)                      # the data-parallel twin: same tile, whole-K sums
            twin = ops.gemm(a, b, bias=bias, residual=resid)
            ref = (a.float() @ b.float().t() + bias.float()).bfloat16().float() + resid.float()
            err = (out.float() - ref).abs().max().item()
            tol = 2e-2 * ref.abs().max().item() + 1e-3
            diff = (out != twin).float().mean().item()           # split tiles sum their K halves in a different order
            dmax = (out.float() - twin.float()).abs().max().item()
            L.uvx_gemm_force_variant(v)
            first = ops.gemm(a, b)
            races = 0
            for _ in range(20):
                races += int(not torch.equal(ops.gemm(a, b), first))
            to = L.uvx_gemm_streamk_timeouts()
            good = err <= tol and races == 0 and to == 0 and diff < 0.05 and dmax <= 0.07 * ref.abs().max().item()
            ok &= good
            print(f"v{v} opt9={flavour} {M}x{N}x{K}: max err {err:.4f} (tol {tol:.4f}), differs from twin in {diff:.2e} of elements "
                  f"(max {dmax:.4f}), nondeterministic repeats {races}/20, give-ups {to} {'OK' if good else 'FAIL'}", flush=True)
L.uvx_set_option(9, 1)
L.uvx_gemm_force_variant(-1)
print("ALL_OK" if ok else "SOME_FAILED")
