"""ctypes loader for libuvx.so — the thin C-ABI layer (north_star asks for cffi; cffi is not installed
in this image and cannot be, ctypes is the same idea with zero dependencies).

The product path FAILS LOUDLY when the HIP library is missing: there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import os

# UVX_LIB: the probe tools (tools/gpu_gemm_*.py) load libuvx_probes.so, the same library plus the GEMM probe variants
_LIB_PATH = Path(os.environ["UVX_LIB"]).resolve() if os.environ.get("UVX_LIB") else Path(__file__).resolve().parent / "libuvx.so"
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
        ("C2", C.c_void_p), ("ldc2", C.c_int32), ("epilogue", C.c_int32),
        ("b_kn", C.c_int32), ("reserved_", C.c_int32),
    ]


ABI_VERSION = 19  # UVX_ABI_VERSION of include/uvx.h that the struct mirrors below follow


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
        got = _lib.uvx_abi_version()
        if got != ABI_VERSION:      # a stale .so next to newer struct mirrors would corrupt memory silently
            _lib = None
            raise UvxError(f"{_LIB_PATH} has ABI version {got}, this package expects {ABI_VERSION}: rebuild with "
                           "`python -m ultravox_amd.build --force`")
        _declare(_lib)
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


# ------------------------------------------------------------------ struct mirrors of include/uvx.h
class Config(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32),
        ("enc_layers", C.c_int32), ("enc_d", C.c_int32), ("enc_heads", C.c_int32), ("enc_ffn", C.c_int32),
        ("n_mels", C.c_int32), ("enc_max_pos", C.c_int32), ("enc_block", C.c_int32), ("ln_eps", C.c_float),
        ("stack_factor", C.c_int32), ("proj_hidden", C.c_int32), ("proj_ln_mid", C.c_int32),
        ("proj_eps", C.c_float),
        ("llm_layers", C.c_int32), ("llm_d", C.c_int32), ("llm_heads", C.c_int32), ("llm_kv_heads", C.c_int32),
        ("llm_head_dim", C.c_int32), ("llm_inter", C.c_int32), ("vocab", C.c_int32), ("rms_eps", C.c_float),
        ("llm_flavor", C.c_int32), ("llm_act", C.c_int32), ("llm_qk_norm", C.c_int32), ("llm_wt_stream", C.c_int32),
        ("llm_attn_scale", C.c_float), ("llm_window", C.c_int32), ("proj_act", C.c_int32),
    ]


_ENC_LAYER_FIELDS = ["ln1_w", "ln1_b", "wqkv", "bqkv", "wo", "bo", "ln2_w", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b",
                     "wqkv_t", "wo_t", "fc1_t", "fc2_t"]


class EncLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _ENC_LAYER_FIELDS]


class EncoderWeights(C.Structure):
    _fields_ = [("conv1_w", C.c_void_p), ("conv1_b", C.c_void_p), ("conv2_w", C.c_void_p), ("conv2_b", C.c_void_p),
                ("pos", C.c_void_p), ("layers", C.POINTER(EncLayer)), ("lnf_w", C.c_void_p), ("lnf_b", C.c_void_p)]


class W2vConfig(C.Structure):      # uvx_w2v_config_t
    _fields_ = [("dtype", C.c_int32), ("n_conv", C.c_int32), ("conv_dim", C.c_int32),
                ("conv_kernel", C.c_int32 * 8), ("conv_stride", C.c_int32 * 8),
                ("d", C.c_int32), ("heads", C.c_int32), ("ffn", C.c_int32), ("layers", C.c_int32),
                ("pos_k", C.c_int32), ("pos_groups", C.c_int32), ("ln_eps", C.c_float),
                ("feat_norm_layer", C.c_int32), ("conv_bias", C.c_int32), ("stable_ln", C.c_int32)]


class W2vWeights(C.Structure):     # uvx_w2v_weights_t
    _fields_ = [("conv0_w", C.c_void_p), ("gn_w", C.c_void_p), ("gn_b", C.c_void_p), ("conv_w", C.c_void_p * 8),
                ("fp_ln_w", C.c_void_p), ("fp_ln_b", C.c_void_p), ("fp_w", C.c_void_p), ("fp_b", C.c_void_p),
                ("pos_w", C.c_void_p), ("pos_b", C.c_void_p), ("ln_w", C.c_void_p), ("ln_b", C.c_void_p),
                ("layers", C.POINTER(EncLayer)),
                ("conv_b", C.c_void_p * 8), ("conv_ln_w", C.c_void_p * 8), ("conv_ln_b", C.c_void_p * 8)]


class LoraProj(C.Structure):
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p)]


class EncLoraLayer(C.Structure):
    _fields_ = [("q", LoraProj), ("k", LoraProj), ("v", LoraProj), ("o", LoraProj), ("g", LoraProj), ("u", LoraProj), ("d", LoraProj)]      # a NULL `a` = not adapted (ABI 17 / 18)


class EncoderLora(C.Structure):
    _fields_ = [("r", C.c_int32), ("scaling", C.c_float), ("layers", C.POINTER(EncLoraLayer))]


class EncoderLoraGrads(C.Structure):   # uvx_lora_proj_grad_t has the layout of uvx_lora_proj_t (two pointers)
    _fields_ = [("layers", C.POINTER(EncLoraLayer))]


class ProjectorWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ln_pre", "w1", "ln_mid", "w2", "ln_post")]


class ProjectorGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ln_pre", "w1", "ln_mid", "w2", "ln_post")]


_LLM_LAYER_FIELDS = ["ln1", "wqkv", "wo", "ln2", "wgu", "wd", "wqkv_t", "wo_t", "wgu_t", "wd_t",
                     "bqkv", "q_norm", "k_norm",     # family extras (Qwen2 biases, Qwen3 / Gemma-3 per-head norms): null when absent
                     "ln1_post", "ln2_post"]         # Gemma-3's post norms


class LlmLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _LLM_LAYER_FIELDS]


class LlmWeights(C.Structure):
    _fields_ = [("embed", C.c_void_p), ("layers", C.POINTER(LlmLayer)), ("norm", C.c_void_p),
                ("lm_head", C.c_void_p), ("lm_head_t", C.c_void_p), ("rope_cos_sin", C.c_void_p),
                ("rope_len", C.c_int32), ("rope_cos_sin_local", C.c_void_p), ("layer_local", C.POINTER(C.c_int32))]


class AttnDesc(C.Structure):
    _fields_ = [("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("o", C.c_void_p), ("lse", C.c_void_p),
                ("kv_start", C.c_void_p), ("kv_len", C.c_void_p),
                ("B", C.c_int32), ("T", C.c_int32), ("Hq", C.c_int32), ("Hkv", C.c_int32), ("D", C.c_int32),
                ("ldq", C.c_int32), ("ldk", C.c_int32), ("ldv", C.c_int32), ("ldo", C.c_int32),
                ("causal", C.c_int32), ("block", C.c_int32), ("scale", C.c_float),
                ("dout", C.c_void_p), ("dq", C.c_void_p), ("dk", C.c_void_p), ("dv", C.c_void_p),
                ("lddq", C.c_int32), ("lddk", C.c_int32), ("lddv", C.c_int32), ("window", C.c_int32)]


EXPORTS = [
    "uvx_last_error", "uvx_abi_version", "uvx_logmel", "uvx_encoder_ws_bytes", "uvx_encoder_fwd",
    "uvx_projector_ws_bytes", "uvx_projector_fwd", "uvx_projector_bwd", "uvx_embed_merge", "uvx_merge_embeds_bwd",
    "uvx_llm_ws_bytes", "uvx_llm_fwd", "uvx_llm_bwd", "uvx_adamw_clip_step", "uvx_gemm", "uvx_layernorm",
    "uvx_rmsnorm", "uvx_gemm_rmsnorm", "uvx_rmsnorm_bwd", "uvx_swiglu", "uvx_swiglu_bwd", "uvx_rope", "uvx_qk_norm_rope", "uvx_qk_norm_bwd", "uvx_attention_ws_bytes",
    "uvx_attention_fwd", "uvx_attention_bwd", "uvx_ce_loss", "uvx_prof_begin", "uvx_prof_enable", "uvx_prof_end", "uvx_prof_records", "uvx_prof_union_ms", "uvx_probe_lds_tr", "uvx_gemm_force_variant", "uvx_attention_force_qt", "uvx_kv_cache_bytes", "uvx_llm_infer_ws_bytes",
    "uvx_llm_prefill", "uvx_llm_prefill_chunk", "uvx_llm_prefill_chunk_logits", "uvx_llm_prefill_chunk_ws_bytes", "uvx_llm_decode", "uvx_argmax", "uvx_greedy_select", "uvx_llm_kl_loss", "uvx_kl_loss", "uvx_gemm_override_variant", "uvx_gemm_pick_variant", "uvx_set_option", "uvx_get_option",
    "uvx_encoder_train_ws_bytes", "uvx_encoder_fwd_train", "uvx_encoder_bwd", "uvx_layernorm_bwd", "uvx_gelu", "uvx_gelu_bwd",
    "uvx_llm_fwd_rows", "uvx_llm_kl_loss_rows", "uvx_llm_bwd_rows", "uvx_llm_bwd_rows_from", "uvx_llm_fwd_lora", "uvx_llm_bwd_lora",
    "uvx_llm_fwd_train", "uvx_llm_bwd_train", "uvx_llm_bwd_train_from",
    "uvx_wav2vec2_frames", "uvx_wav2vec2_ws_bytes", "uvx_wav2vec2_fwd", "uvx_wav2vec2_train_ws_bytes", "uvx_wav2vec2_fwd_train", "uvx_wav2vec2_bwd",
    "uvx_gemm_splitk_ws_bytes", "uvx_gemm_splitk", "uvx_gemm_pick_split",
    "uvx_comm_unique_id", "uvx_comm_init", "uvx_comm_world_size", "uvx_comm_version", "uvx_comm_allreduce_f32", "uvx_comm_destroy",
]


def _declare(l: C.CDLL) -> None:
    for name in ("uvx_encoder_ws_bytes", "uvx_projector_ws_bytes", "uvx_llm_ws_bytes", "uvx_attention_ws_bytes",
                 "uvx_kv_cache_bytes", "uvx_llm_infer_ws_bytes", "uvx_llm_prefill_chunk_ws_bytes", "uvx_encoder_train_ws_bytes",
                 "uvx_wav2vec2_ws_bytes", "uvx_wav2vec2_train_ws_bytes", "uvx_gemm_splitk_ws_bytes"):
        getattr(l, name).restype = C.c_size_t
    for name in EXPORTS:
        f = getattr(l, name)
        if name.endswith("_bytes") or name in ("uvx_last_error", "uvx_abi_version"):
            continue
        if name == "uvx_prof_union_ms":
            f.restype, f.argtypes = C.c_double, [C.c_int32]
            continue
        f.restype = C.c_int32
