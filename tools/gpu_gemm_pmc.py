"""Run ONE GEMM shape with ONE forced tile variant a few times (for rocprofv3 --pmc passes)."""
import os
os.environ.setdefault("UVX_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ultravox_amd", "libuvx_probes.so"))  # probe tile variants live in the probes build
import sys, torch
from ultravox_amd import ops, _lib
v, M, N, K = (int(x) for x in sys.argv[1:5])
_lib.lib().uvx_gemm_force_variant(v)
torch.manual_seed(0)
a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(4): ops.gemm(a, b, out=out)
torch.cuda.synchronize()
