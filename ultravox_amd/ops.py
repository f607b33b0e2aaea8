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
         out_f32: bool = False, accumulate: bool = False, alpha: float = 1.0, res_mod: int = 0,
         epilogue: int = 0, c2: Optional[torch.Tensor] = None, b_kn: bool = False) -> torch.Tensor:
    """out[M,N] = act(alpha * a[M,K] @ b[N,K]^T + bias) + residual   (nn.Linear layout); b_kn: b is [K, N] and out = a @ b (the NN form)."""
    assert a.dim() == 2 and b.dim() == 2 and a.shape[1] == b.shape[0 if b_kn else 1]
    assert a.stride(1) == 1 and b.stride(1) == 1
    M, K = a.shape
    N = b.shape[1 if b_kn else 0]
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
    d.act = {"none": 0, "gelu": 1, "gelu_keep": 2, "gelu_bwd": 3}[act]      # 2 / 3: include/uvx.h uvx_gemm_desc_t.act (c2 = activation out / pre-activation in)
    d.out_f32 = int(out_f32)
    d.accumulate = int(accumulate)
    d.alpha = alpha
    d.epilogue = epilogue
    d.C2, d.ldc2 = (0, 0) if c2 is None else (c2.data_ptr(), c2.stride(0))
    d.b_kn = int(b_kn)
    check(_lib.lib().uvx_gemm(stream_ptr(), dt, C.byref(d)), "uvx_gemm")
    return out


def gemm_splitk(a: torch.Tensor, b: torch.Tensor, *, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
                act: str = "none", epilogue: int = 0, c2: Optional[torch.Tensor] = None, alpha: float = 1.0, res_mod: int = 0,
                force_split: int = 0, out: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """uvx_gemm_splitk: gemm() with split-K scratch (the prefill's few-hundred-row problems); force_split 0 = the cost model decides."""
    M, K = a.shape
    N = b.shape[0]
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=a.dtype)
    if workspace is None:
        workspace = torch.empty(int(_lib.lib().uvx_gemm_splitk_ws_bytes(M, N)), device=a.device, dtype=torch.uint8)
    d = _lib.GemmDesc()
    d.A, d.B, d.C = a.data_ptr(), b.data_ptr(), out.data_ptr()
    d.bias = 0 if bias is None else bias.data_ptr()
    d.residual = 0 if residual is None else residual.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = a.stride(0), b.stride(0), out.stride(0)
    d.ldr = 0 if residual is None else residual.stride(0)
    d.res_mod, d.batch, d.alpha = res_mod, 1, alpha
    d.act = {"none": 0, "gelu": 1}[act]
    d.epilogue = epilogue
    d.C2, d.ldc2 = (0, 0) if c2 is None else (c2.data_ptr(), c2.stride(0))
    check(_lib.lib().uvx_gemm_splitk(stream_ptr(), dtype_code(a.dtype), C.byref(d), ptr(workspace), C.c_size_t(workspace.numel()),
                                     int(force_split)), "uvx_gemm_splitk")
    return out


def gemm_rmsnorm(a: torch.Tensor, norm_w: torch.Tensor, b: torch.Tensor, *, eps: float = 1e-5, flavor: int = 0,
                 bias: Optional[torch.Tensor] = None, epilogue: int = 0, c2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[M,N] = epilogue(RMSNorm(a; norm_w) @ b[N,K]^T + bias) through uvx_gemm_rmsnorm (the decode step's fused norm + GEMV)."""
    M, K = a.shape
    N = b.shape[0]
    out = torch.empty((M, N), device=a.device, dtype=a.dtype)
    scratch = torch.empty_like(a)
    d = _lib.GemmDesc()
    d.A, d.B, d.C = a.data_ptr(), b.data_ptr(), out.data_ptr()
    d.bias = 0 if bias is None else bias.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = a.stride(0), b.stride(0), out.stride(0)
    d.batch, d.alpha, d.epilogue = 1, 1.0, epilogue
    d.C2, d.ldc2 = (0, 0) if c2 is None else (c2.data_ptr(), c2.stride(0))
    check(_lib.lib().uvx_gemm_rmsnorm(stream_ptr(), dtype_code(a.dtype), C.byref(d), ptr(norm_w), C.c_float(eps), int(flavor), ptr(scratch)),
          "uvx_gemm_rmsnorm")
    return out


def _code(t: torch.Tensor) -> int:
    return dtype_code(t.dtype)


def layernorm(x, w, b, eps=1e-5):
    y = torch.empty_like(x)
    rows = x.numel() // x.shape[-1]
    check(_lib.lib().uvx_layernorm(stream_ptr(), _code(x), ptr(x), ptr(w), ptr(b), ptr(y), rows, x.shape[-1],
                                   C.c_float(eps)), "uvx_layernorm")
    return y


def rmsnorm(x, w, eps=1e-6):
    y = torch.empty_like(x)
    rows = x.numel() // x.shape[-1]
    check(_lib.lib().uvx_rmsnorm(stream_ptr(), _code(x), ptr(x), ptr(w), ptr(y), rows, x.shape[-1], C.c_float(eps)),
          "uvx_rmsnorm")
    return y


def rmsnorm_bwd(dy, x, w, eps=1e-6, dx_add=None, want_dx=True, want_dw=False):
    rows, cols = x.numel() // x.shape[-1], x.shape[-1]
    dx = torch.empty_like(x) if want_dx else None
    dw = torch.zeros(cols, device=x.device, dtype=torch.float32) if want_dw else None
    check(_lib.lib().uvx_rmsnorm_bwd(stream_ptr(), _code(x), ptr(dy), ptr(x), ptr(w), ptr(dx_add), ptr(dx), ptr(dw),
                                     rows, cols, C.c_float(eps)), "uvx_rmsnorm_bwd")
    return dx, dw


def qk_norm_rope_(qkv, wq, wk, cos_sin, T, Hq, Hkv, head_dim, eps=1e-6, keep_raw=False, flavor=0):
    """Qwen3 q_norm / k_norm + RoPE in place on the q | k columns of qkv [rows, ld]; -> the raw q | k rows when keep_raw."""
    rows = qkv.numel() // qkv.shape[-1]
    raw = torch.empty(rows, (Hq + Hkv) * head_dim, device=qkv.device, dtype=qkv.dtype) if keep_raw else None
    check(_lib.lib().uvx_qk_norm_rope(stream_ptr(), _code(qkv), ptr(qkv), ptr(wq), ptr(wk), ptr(raw), ptr(cos_sin), rows, T, Hq, Hkv,
                                      head_dim, qkv.shape[-1], C.c_float(eps), flavor), "uvx_qk_norm_rope")
    return raw


def qk_norm_bwd_(d_qkv, raw, wq, wk, Hq, Hkv, head_dim, eps=1e-6, flavor=0):
    """In place on the q | k columns of d_qkv [rows, ld]: gradient of the normalised rows -> gradient of the raw rows."""
    rows = d_qkv.numel() // d_qkv.shape[-1]
    check(_lib.lib().uvx_qk_norm_bwd(stream_ptr(), _code(d_qkv), ptr(d_qkv), ptr(raw), ptr(wq), ptr(wk), rows, Hq, Hkv, head_dim,
                                     d_qkv.shape[-1], C.c_float(eps), flavor), "uvx_qk_norm_bwd")
    return d_qkv


def swiglu(x, gate_first=False):
    half = x.shape[-1] // 2
    out = torch.empty(*x.shape[:-1], half, device=x.device, dtype=x.dtype)
    check(_lib.lib().uvx_swiglu(stream_ptr(), _code(x), ptr(x), ptr(out), x.numel() // x.shape[-1], half,
                                int(gate_first)), "uvx_swiglu")
    return out


def swiglu_bwd(dout, x, gate_first=False):
    half = x.shape[-1] // 2
    din = torch.empty_like(x)
    check(_lib.lib().uvx_swiglu_bwd(stream_ptr(), _code(x), ptr(dout), ptr(x), ptr(din), x.numel() // x.shape[-1],
                                    half, int(gate_first)), "uvx_swiglu_bwd")
    return din


def rope_(x, cos_sin, T, n_heads, head_dim, inverse=False):
    """In place on x [rows, ld]: heads 0..n_heads-1 (contiguous from column 0) are rotated."""
    rows = x.numel() // x.shape[-1]
    check(_lib.lib().uvx_rope(stream_ptr(), _code(x), ptr(x), ptr(cos_sin), rows, T, n_heads, head_dim, x.shape[-1],
                              int(inverse)), "uvx_rope")
    return x


def _attn_desc(q, k, v, o, lse, causal, block, scale, kv_start, kv_len, window=0):
    B, T, Hq, D = q.shape
    d = _lib.AttnDesc()
    d.q, d.k, d.v, d.o = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr()
    d.lse = 0 if lse is None else lse.data_ptr()
    d.kv_start = 0 if kv_start is None else kv_start.data_ptr()
    d.kv_len = 0 if kv_len is None else kv_len.data_ptr()
    d.B, d.T, d.Hq, d.Hkv, d.D = B, T, Hq, k.shape[2], D
    d.ldq, d.ldk, d.ldv, d.ldo = q.stride(1), k.stride(1), v.stride(1), o.stride(1)
    d.causal, d.block, d.scale, d.window = int(causal), int(block), scale, int(window)
    return d


def attention(q, k, v, causal=False, block=0, scale=None, kv_start=None, kv_len=None, need_lse=True, window=0):
    """q [B,T,Hq,D], k/v [B,T,Hkv,D] (last two dims contiguous, token stride free) -> o [B,T,Hq*D], lse."""
    B, T, Hq, D = q.shape
    scale = D ** -0.5 if scale is None else scale
    o = torch.empty(B, T, Hq * D, device=q.device, dtype=q.dtype)
    lse = torch.empty(B, Hq, T, device=q.device, dtype=torch.float32) if need_lse else None
    d = _attn_desc(q, k, v, o, lse, causal, block, scale, kv_start, kv_len, window)
    nb = _lib.lib().uvx_attention_ws_bytes(_code(q), C.byref(d), 0)
    ws = torch.empty(nb, device=q.device, dtype=torch.uint8)
    check(_lib.lib().uvx_attention_fwd(stream_ptr(), _code(q), C.byref(d), ptr(ws), C.c_size_t(nb)), "uvx_attention_fwd")
    return o, lse


def attention_bwd(q, k, v, o, lse, dout, causal=False, block=0, scale=None, kv_start=None, kv_len=None, window=0):
    B, T, Hq, D = q.shape
    scale = D ** -0.5 if scale is None else scale
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    d = _attn_desc(q, k, v, o, lse, causal, block, scale, kv_start, kv_len, window)
    d.dout, d.dq, d.dk, d.dv = dout.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
    d.lddq, d.lddk, d.lddv = dq.stride(1), dk.stride(1), dv.stride(1)
    nb = _lib.lib().uvx_attention_ws_bytes(_code(q), C.byref(d), 1)
    ws = torch.empty(nb, device=q.device, dtype=torch.uint8)
    check(_lib.lib().uvx_attention_bwd(stream_ptr(), _code(q), C.byref(d), ptr(ws), C.c_size_t(nb)), "uvx_attention_bwd")
    return dq, dk, dv


def ce_loss(logits, labels, want_grad=True, grad_scale=1.0, in_place=False):
    """logits [B,T,V], labels [B,T] int64 -> (loss f32 scalar tensor, dlogits or None)."""
    B, T, V = logits.shape
    loss = torch.zeros(1, device=logits.device, dtype=torch.float32)
    dl = (logits if in_place else torch.empty_like(logits)) if want_grad else None
    scratch = torch.empty(2 + B * T, device=logits.device, dtype=torch.float32)
    check(_lib.lib().uvx_ce_loss(stream_ptr(), _code(logits), ptr(logits), ptr(labels.contiguous()), ptr(loss), ptr(dl),
                                 B, T, V, logits.stride(1), C.c_float(grad_scale), ptr(scratch)), "uvx_ce_loss")
    return loss[0], dl


def kl_loss(student, teacher, pair_row, pair_w, temperature, want_grad=True, grad_scale=1.0):
    """student [R, V], teacher [Rt, V]; pair_row int32 [2, R], pair_w f32 [2, R] -> (loss, dlogits or None)."""
    R, V = student.shape
    loss = torch.zeros(1, device=student.device, dtype=torch.float32)
    dl = torch.empty_like(student) if want_grad else None
    scratch = torch.empty(R, device=student.device, dtype=torch.float32)
    check(_lib.lib().uvx_kl_loss(stream_ptr(), _code(student), ptr(student), ptr(teacher), ptr(pair_row.contiguous()),
                                 ptr(pair_w.contiguous()), ptr(loss), ptr(dl), C.c_int64(R), V, student.stride(0),
                                 teacher.stride(0), C.c_float(temperature), C.c_float(grad_scale), ptr(scratch)), "uvx_kl_loss")
    return loss[0], dl


def layernorm_bwd(dy, x, w, eps=1e-5, dx_add=None):
    rows, cols = x.numel() // x.shape[-1], x.shape[-1]
    dx = torch.empty_like(x)
    check(_lib.lib().uvx_layernorm_bwd(stream_ptr(), _code(x), ptr(dy), ptr(x), ptr(w), ptr(dx_add), ptr(dx), rows, cols,
                                       C.c_float(eps)), "uvx_layernorm_bwd")
    return dx


def gelu(pre):
    out = torch.empty_like(pre)
    check(_lib.lib().uvx_gelu(stream_ptr(), _code(pre), ptr(pre), ptr(out), C.c_int64(pre.numel())), "uvx_gelu")
    return out


def gelu_bwd(dout, pre):
    din = torch.empty_like(pre)
    check(_lib.lib().uvx_gelu_bwd(stream_ptr(), _code(pre), ptr(dout), ptr(pre), ptr(din), C.c_int64(pre.numel())), "uvx_gelu_bwd")
    return din
