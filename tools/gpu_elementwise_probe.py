"""GPU probe: the HBM-bound kernels between the LLM's GEMMs at the C2 shape (2528 rows), timed over a POOL of buffers larger
than the Infinity Cache so that every launch streams from HBM as in the step.  UVX_LIB selects the library (A/B of two builds).
usage: [UVX_LIB=...] PYTHONPATH=. python tools/gpu_elementwise_probe.py"""
import torch
from ultravox_amd import ops

dev = "cuda"
torch.manual_seed(0)
M, H, F = 2528, 4096, 14336
NP = 24                                            # pool entries: 24 x (3 x 20.7 MB) = 1.5 GB for the norm, more for SwiGLU


def timed(fn, reps=96):
    for i in range(4):
        fn(i)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    for i in range(reps):
        fn(i)
    e[1].record(); torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]) / reps * 1e3


xs = [(torch.randn(M, H, device=dev)).bfloat16() for _ in range(NP)]
dys = [(torch.randn(M, H, device=dev) * 0.1).bfloat16() for _ in range(NP)]
adds = [(torch.randn(M, H, device=dev) * 0.1).bfloat16() for _ in range(NP)]
w = torch.ones(H, device=dev).bfloat16()
t = timed(lambda i: ops.rmsnorm_bwd(dys[i % NP], xs[i % NP], w, dx_add=adds[i % NP]))
print(f"rmsnorm_bwd  [{M} x {H}] + dx_add: {t:6.1f} us  ({4 * M * H * 2 / t / 1e6:5.2f} TB/s over 4 streams; incl. the wrapper's allocation)")
t = timed(lambda i: ops.rmsnorm(xs[i % NP], w))
print(f"rmsnorm_fwd  [{M} x {H}]:          {t:6.1f} us  ({2 * M * H * 2 / t / 1e6:5.2f} TB/s over 2 streams)")
