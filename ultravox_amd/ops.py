"""Tensor-level wrappers over the single-op C-ABI entry points (PyTorch-ROCm tensors in/out).

These are plumbing for tests, benches and the host-side model code; all arithmetic happens in
libuvx.so.  Every wrapper raises if the library is missing — there is no fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import check, dtype_code, ptr, stream_ptr


def gemm(a: torch.Tensor, b: torch.Tensor, *, bias: Optional[torch.Tensor] = None,
         residual: Optional[torch.Tensor] = None, act: str = "none", out: Optional[torch.Tensor] = None,
         out_f32: bool = False, accumulate: bool = False, alpha: float = 1.0, res_mod: int = 0
         ) -> torch.Tensor:
    """out[M,N] = act(alpha * a[M,K] @ b[N,K]^T + bias) + residual   (nn.Linear layout)."""
    assert a.dim() == 2 and b.dim() == 2 and a.shape[1] == b.shape[1]
    assert a.stride(1) == 1 and b.stride(1) == 1
    M, K = a.shape
    N = b.shape[0]
    dt = dtype_code(a.dtype)
    if out is None:
        out = torch.empty((M, N), device=a.device,
                          dtype=torch.float32 if (out_f32 or dt == _lib.F32) else a.dtype)
    d = _lib.GemmDesc()
    d.A, d.B, d.C = a.data_ptr(), b.data_ptr(), out.data_ptr()
    d.bias = 0 if bias is None else bias.data_ptr()
    d.residual = 0 if residual is None else residual.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = a.stride(0), b.stride(0), out.stride(0)
    d.ldr = 0 if residual is None else residual.stride(0)
    d.res_mod = res_mod
    d.batch = 1
    d.act = {"none": 0, "gelu": 1}[act]
    d.out_f32 = int(out_f32)
    d.accumulate = int(accumulate)
    d.alpha = alpha
    check(_lib.lib().uvx_gemm(stream_ptr(), dt, C.byref(d)), "uvx_gemm")
    return out
