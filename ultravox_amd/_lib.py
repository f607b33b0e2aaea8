"""ctypes loader for libuvx.so — the thin C-ABI layer (north_star asks for cffi; cffi is not installed
in this image and cannot be, ctypes is the same idea with zero dependencies).

The product path FAILS LOUDLY when the HIP library is missing: there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

_LIB_PATH = Path(__file__).resolve().parent / "libuvx.so"
_lib = None

BF16, F32 = 0, 1


class UvxError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p),
        ("residual", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32), ("ldr", C.c_int32),
        ("res_mod", C.c_int32), ("batch", C.c_int32),
        ("stride_a", C.c_int64), ("stride_b", C.c_int64), ("stride_c", C.c_int64),
        ("stride_r", C.c_int64),
        ("act", C.c_int32), ("out_f32", C.c_int32), ("accumulate", C.c_int32), ("alpha", C.c_float),
    ]


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            raise UvxError(
                f"{_LIB_PATH} not found: build it with `python -m ultravox_amd.build` "
                "(hipcc, gfx950). There is no CPU fallback for the device path."
            )
        _lib = C.CDLL(str(_LIB_PATH))
        _lib.uvx_last_error.restype = C.c_char_p
        _lib.uvx_abi_version.restype = C.c_int32
    return _lib


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = lib().uvx_last_error().decode("utf-8", "replace")
        kind = ValueError if status in (-1, -2) else UvxError
        raise kind(f"libuvx {what} failed (status {status}): {msg}")


def stream_ptr() -> C.c_void_p:
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def dtype_code(t) -> int:
    import torch

    if t == torch.bfloat16:
        return BF16
    if t == torch.float32:
        return F32
    raise ValueError(f"unsupported dtype {t}")
