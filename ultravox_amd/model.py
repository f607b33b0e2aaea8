"""Host-side mirror of ultravox/model/ultravox_model.py — UltravoxModel.forward (:277-352),
_prepare_audio_embeds (:354-396), _audio_iter (:259-275), UltravoxProjector (:745-800),
ModifiedWhisperEncoder (:803-994) — over the libuvx C ABI.  Tensors in, tensors out; every FLOP of
the path runs in hand-written HIP (gfx950).  There is NO PyTorch / CPU fallback: without libuvx.so or
without a GPU the constructors raise.

Training scope = the reference's default recipe: audio tower and LLM frozen (apply_lora with r = 0,
ultravox_model.py:697-703), projector trained; `UltravoxTrainer` reproduces one HF-Trainer optimizer
step (loss -> backward -> DP mean of projector grads -> clip 1.0 -> AdamW), SURVEY.md Appendix B.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import dataclasses
import json
from typing import Dict, Optional

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr
from .config import PROJECTOR_ACTS, LossConfig, LossFunction, UltravoxConfig
from .weights import (LORA_FIELD, LORA_TARGETS, audio_lora_key, init_lora_state_dict, lora_targets, llm_lora_key, lora_key, pack_encoder, pack_llm, pack_wav2vec2,
                      unpack_encoder, unpack_llm, unpack_wav2vec2, wav2vec2_param_names, check_encoder_exportable, encoder_param_names, llm_param_names,
                      random_state_dict)


@dataclasses.dataclass
class KVState:
    """What `generate()` hands back as `past_key_values`: the caller-owned KV cache ([L][2][B][Tmax][kv_heads * head_dim],
    include/uvx.h) with its bookkeeping.  `tokens` are the ids whose keys / values fill rows [0, cur_len) — a later
    `generate(input_ids, past_key_values=state)` runs only `input_ids[:, cur_len:]` when its prefix still matches them.
    Like HF's in-place caches, a state is CONSUMED by the call it is handed to: the call may append to the same buffer
    (rows below `cur_len` are never rewritten, so the old state stays readable, but two continuations of one state would
    share - and overwrite - the rows above it).  Branch a dialogue from `copy.deepcopy(state)`.
    With `partial_ok` a prompt that departs from `tokens` part-way (a reply that re-tokenised differently) still reuses the
    rows of the longest common prefix - under causal attention they do not depend on what follows - instead of dropping the
    cache; `LocalInference` turns it on because in a conversation only the cache remembers earlier AUDIO turns."""
    cache: torch.Tensor
    Tmax: int
    cur_len: int
    pos_next: torch.Tensor        # [B] int32: RoPE position of the next token of each sequence
    kv_start: torch.Tensor        # [B] int32: first real cache row (left padding)
    tokens: torch.Tensor          # [B, cur_len] int64
    partial_ok: bool = False      # reuse the longest common prefix when the new prompt departs from `tokens` part-way

    def get_seq_length(self) -> int:
        return self.cur_len


@dataclasses.dataclass
class GenerateOutput:
    sequences: torch.Tensor
    past_key_values: Optional[KVState] = None
    logits: Optional[tuple] = None      # output_logits=True: one [B, vocab] tensor of raw next-token logits per generated token (HF's field)
    sequences_scores: Optional[torch.Tensor] = None      # beam search: final (length-normalised) score of every returned sequence (HF's field)


@dataclasses.dataclass
class CausalLMOutputWithPast:
    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    past_key_values: Optional[object] = None


_PROJ_ORDER = ("ln_pre", "linear_1", "ln_norm", "linear_2")  # flat-bucket order; ln_norm = ln_mid | ln_post


def _torch_dtype(cfg: UltravoxConfig, dtype=None) -> torch.dtype:
    if dtype is not None:
        return dtype
    d = cfg.torch_dtype
    return getattr(torch, d) if isinstance(d, str) else d


def kl_row_pairs(labels: torch.Tensor, alt_labels: torch.Tensor, eot_loss_weight: float):
    """Host mirror of UltravoxModel._get_prediction_mask (ultravox_model.py:157-198) for both label tensors, turned
    into the row pairs uvx_llm_kl_loss takes: the i-th prediction position of the student is scored against the
    i-th prediction position of the teacher (boolean-mask indexing order, :226-237), likewise the end-of-turn
    positions (:240-254).  Returns (pair_row int32 [2, B*T], pair_w f32 [2, B*T], n_pred)."""
    def masks(lab):
        lab = lab.detach().to("cpu")
        lm = lab != -100
        pred = torch.zeros_like(lm)
        pred[:, :-1] = lm[:, 1:]
        has = pred.any(dim=1)
        last = pred.shape[1] - 1 - torch.flip(pred, dims=[1]).to(torch.int8).argmax(dim=1)
        eot = torch.zeros_like(pred)
        eot[has, last[has]] = True
        return pred.reshape(-1), eot.reshape(-1)
    ps, es = masks(labels)
    pt, et = masks(alt_labels)
    R = ps.numel()
    pair_row = torch.full((2, R), -1, dtype=torch.int32)
    pair_w = torch.zeros((2, R), dtype=torch.float32)
    for slot, (ms, mt, w) in enumerate(((ps, pt, 1.0), (es, et, float(eot_loss_weight)))):
        if slot == 1 and not eot_loss_weight > 0:
            break                                                   # :240: the eot term only exists for a positive weight
        i_s, i_t = torch.nonzero(ms)[:, 0], torch.nonzero(mt)[:, 0]
        if len(i_s) != len(i_t):
            raise ValueError(f"KL loss: {len(i_s)} student positions vs {len(i_t)} teacher positions "
                             f"({'prediction' if slot == 0 else 'end-of-turn'} mask); the reference's F.kl_div cannot pair them")
        pair_row[slot, i_s] = i_t.to(torch.int32)
        if len(i_s):
            pair_w[slot, i_s] = w / len(i_s)
    return pair_row, pair_w, int(ps.sum())


def kl_compact_pairs(pair_row: torch.Tensor, pair_w: torch.Tensor):
    """kl_row_pairs' [2, B*T] tables reduced to the rows that take part: student rows (ascending), teacher rows
    (ascending, unique) and, per student row, the INDEX of its partner(s) in the teacher list - the operands of
    uvx_llm_fwd_rows / uvx_llm_kl_loss_rows."""
    has = (pair_row >= 0).any(dim=0)
    rows_s = torch.nonzero(has)[:, 0].to(torch.int32)
    pr = pair_row[:, has]
    rows_t = torch.unique(pr[pr >= 0]).to(torch.int32)              # sorted ascending
    pos = torch.full((int(rows_t.max()) + 1 if rows_t.numel() else 1,), -1, dtype=torch.int32)
    pos[rows_t.long()] = torch.arange(rows_t.numel(), dtype=torch.int32)
    pair_c = torch.where(pr >= 0, pos[pr.clamp_min(0).long()], torch.full_like(pr, -1))
    return rows_s, rows_t, pair_c.contiguous(), pair_w[:, has].contiguous()


class UltravoxModel:
    """Same call surface as the reference's UltravoxModel for the hot path (forward / train step)."""

    config_class = UltravoxConfig
    accepts_loss_kwargs = False  # ultravox_model.py:50-53: loss = mean over this rank's tokens

    def __init__(self, config: UltravoxConfig, state_dict: Optional[Dict[str, torch.Tensor]] = None,
                 device: str = "cuda", dtype: Optional[torch.dtype] = None, seed: int = 0,
                 with_backward: bool = True, rope_len: Optional[int] = None, consume_state_dict: bool = False,
                 stream_weight_transposes: Optional[bool] = None, dgrad_nn: bool = False):
        """consume_state_dict: pop the LLM's q/k/v/gate/up tensors from `state_dict` as they are packed (the dict is left
        without them) so that loading peaks at one copy of the model plus a layer - for the 70B-parameter LLM (C4).
        stream_weight_transposes: the backward pass needs the frozen LLM's weights transposed; True = made on the fly, one layer
        ahead, on a side stream (uvx_config_t.llm_wt_stream: half the resident weight bytes for one extra read + write of the
        weights per step), False = resident copies, None = resident unless the LLM's weights exceed a third of the GPU's memory
        (a 70B-parameter LLM: 141 GB of 288)."""
        _lib.lib()  # fail loudly if the HIP library is missing
        if not torch.cuda.is_available():
            raise _lib.UvxError("UltravoxModel needs a GPU (MI355X / gfx950); there is no CPU path")
        # llm_only_training (ultravox_config.py:120; ultravox_model.py:62-67: no audio tower, no projector - the reference's text-only LoRA
        # pre-stage, training/model_types.py:139-163): only the language model and its adapters exist; audio inputs are refused
        self.llm_only = bool(config.llm_only_training)
        self.config = config
        self.device = torch.device(device)
        self.dtype = _torch_dtype(config, dtype)
        self.code = _lib.dtype_code(self.dtype)
        self.training = False
        # training step: last layer's o_proj / MLP on the supervised rows only (uvx_llm_fwd_train); A/B switch for the probes
        self.top_layer_supervised_rows = os.environ.get("UVX_TOP_LAYER_ROWS", "1") != "0"
        self._llm_train_pair = False             # which uvx_llm_bwd* entry point pairs with the last language_model_forward
        self._llm_top_rows = False
        self._kl_grad_scale = 1.0
        self._before_projector = None
        self.keep_params = set()                 # ultravox_model.py:59
        self.loss_config = LossConfig()
        self.vocab_size = config.vocab_size
        a, t = config.audio_config, config.text_config
        self.audio_tower_context_length = a.max_source_positions * 2  # ultravox_model.py:826-832
        if config.audio_latency_block_size is not None:
            assert self.audio_tower_context_length % config.audio_latency_block_size == 0, (
                f"audio_latency_block_size {config.audio_latency_block_size} must divide "
                f"{self.audio_tower_context_length} evenly.")  # ultravox_model.py:846-848
        if state_dict is None:
            gen_dev = "cuda" if t.num_hidden_layers * t.hidden_size > 64 * 1024 else "cpu"
            state_dict = random_state_dict(config, seed=seed, dtype=self.dtype, device=gen_dev)
            consume_state_dict = True              # nobody else holds this dict: let the packer free its sources as it goes
        self.with_backward = with_backward
        self._consume_sd = bool(consume_state_dict)
        if stream_weight_transposes is None:
            per_layer = ((t.num_attention_heads + 2 * t.num_key_value_heads) * t.head_dim * t.hidden_size
                         + t.num_attention_heads * t.head_dim * t.hidden_size + 3 * t.intermediate_size * t.hidden_size)
            llm_bytes = (per_layer * t.num_hidden_layers + 2 * t.vocab_size * t.hidden_size) * (2 if self.dtype == torch.bfloat16 else 4)
            total = torch.cuda.get_device_properties(self.device).total_memory
            stream_weight_transposes = llm_bytes > total / 3
        # dgrad_nn (bf16, round 6): neither resident nor streamed W^T - the frozen LLM's dgrads read the forward weights through the GEMM's NN
        # form (uvx_gemm_desc_t.b_kn; bit-identical results): -14 GB at Llama-3-8B, no transposing side stream at 70B.  lm_head^T stays.
        self.dgrad_nn = bool(dgrad_nn) and with_backward and self.dtype == torch.bfloat16
        if self.dgrad_nn:
            stream_weight_transposes = False
        self.stream_weight_transposes = bool(stream_weight_transposes) and with_backward
        self._load(state_dict, rope_len)
        self._ws: Dict[str, torch.Tensor] = {}
        self._proj_ctx = None
        self._llm_ctx = None

    # ------------------------------------------------------------------ weights
    def _load(self, sd, rope_len):
        cfg, dev, dt = self.config, self.device, self.dtype
        a, t = cfg.audio_config, cfg.text_config
        self.lora_r = int((getattr(cfg, "audio_model_lora_config", None) or {}).get("r", 0) or 0)      # encoder LoRA rank (0: frozen tower)
        self.is_wav2vec2 = bool(getattr(a, "is_wav2vec2", False))
        if self.llm_only:
            self.lora_r, self.is_wav2vec2, self._enc = 0, False, None      # (LLMOnlyModelPack passes audio_model_lora_config = None)
        elif self.is_wav2vec2:      # (apply_lora wraps whatever AutoModel tower was loaded, ultravox_model.py:460-467: uvx_wav2vec2_fwd_train / _bwd)
            self._enc = pack_wav2vec2(sd, cfg, dt, dev, with_transposes=self.lora_r > 0 and self.with_backward)
        else:
            self._enc = pack_encoder(sd, cfg, dt, dev, with_transposes=self.lora_r > 0 and self.with_backward)
        self._llm = pack_llm(sd, cfg, dt, dev, with_transposes=("head" if self.dgrad_nn else self.with_backward and not self.stream_weight_transposes), rope_len=rope_len,
                             consume=getattr(self, "_consume_sd", False))
        # projector: one flat trainable bucket with views (ln_pre | linear_1 | ln_mid/ln_post | linear_2)
        P = "multi_modal_projector."
        norm_key = "ln_mid" if cfg.projector_ln_mid else "ln_post"
        parts = [] if self.llm_only else [sd[P + "ln_pre.weight"], sd[P + "linear_1.weight"], sd[P + norm_key + ".weight"],
                                          sd[P + "linear_2.weight"]]
        names = [] if self.llm_only else list(_PROJ_ORDER)
        # encoder LoRA: lora_A / lora_B of q_proj and k_proj of every layer join the SAME flat trainable bucket (one
        # all-reduce, one AdamW launch); missing keys get peft's initialisation (A kaiming-uniform, B zero)
        self._lora_names = []
        self.text_lora_r = int((getattr(cfg, "text_model_lora_config", None) or {}).get("r", 0) or 0)   # LLM LoRA rank (0: frozen LLM)
        if self.lora_r > 0 or self.text_lora_r > 0:
            init = init_lora_state_dict(cfg, seed=0, dtype=dt)
            self._audio_lora_key = audio_lora_key(cfg)      # Whisper: layers.N.self_attn..., wav2vec2: encoder.layers.N.attention...
            todo = ([(self._audio_lora_key, a.encoder_layers, lora_targets(cfg, "audio"))] * (self.lora_r > 0)
                    + [(llm_lora_key, t.num_hidden_layers, lora_targets(cfg, "text"))] * (self.text_lora_r > 0))
            for keyfn, nl, targets in todo:
                for i in range(nl):
                    for pj in targets:      # the adapted attention projections (target_modules; default q_proj + k_proj)
                        for which in "AB":
                            k = keyfn(i, pj, which)
                            parts.append(sd[k] if k in sd else init[k])
                            names.append(k)
                            self._lora_names.append(k)
        sizes = [p.numel() for p in parts]
        pad = [(-s) % 64 for s in sizes]  # keep every view 128-byte aligned
        total = sum(s + p for s, p in zip(sizes, pad))
        self.proj_flat = torch.zeros(total, device=dev, dtype=dt)
        self.proj_grad = torch.zeros(total, device=dev, dtype=torch.float32)
        self._proj_views, self._grad_views, off = {}, {}, 0
        for name, p, s, pd in zip(names, parts, sizes, pad):
            self.proj_flat[off:off + s].copy_(p.reshape(-1).to(device=dev, dtype=dt))
            self._proj_views[name] = self.proj_flat[off:off + s].view(p.shape)
            self._grad_views[name] = self.proj_grad[off:off + s].view(p.shape)
            off += s + pd
        self._norm_key = norm_key
        if self.lora_r > 0:
            nl = a.encoder_layers
            self._lora_layers = (_lib.EncLoraLayer * nl)()
            self._lora_grad_layers = (_lib.EncLoraLayer * nl)()
            self._lora_targets = lora_targets(cfg, "audio")
            for i in range(nl):
                for pj, fld in ((pj, LORA_FIELD[pj]) for pj in self._lora_targets):      # projections not named keep NULL pointers: not adapted
                    ak = self._audio_lora_key
                    getattr(self._lora_layers[i], fld).a = self._proj_views[ak(i, pj, "A")].data_ptr()
                    getattr(self._lora_layers[i], fld).b = self._proj_views[ak(i, pj, "B")].data_ptr()
                    getattr(self._lora_grad_layers[i], fld).a = self._grad_views[ak(i, pj, "A")].data_ptr()
                    getattr(self._lora_grad_layers[i], fld).b = self._grad_views[ak(i, pj, "B")].data_ptr()
            self._lora = _lib.EncoderLora()
            self._lora.r = self.lora_r
            self._lora.scaling = float(cfg.audio_model_lora_config.get("lora_alpha", 8)) / self.lora_r
            self._lora.layers = self._lora_layers
            self._lora_grads = _lib.EncoderLoraGrads()
            self._lora_grads.layers = self._lora_grad_layers
        if self.text_lora_r > 0:
            nl = t.num_hidden_layers
            self._tlora_layers = (_lib.EncLoraLayer * nl)()
            self._tlora_grad_layers = (_lib.EncLoraLayer * nl)()
            self._tlora_targets = lora_targets(cfg, "text")
            for i in range(nl):
                for pj, fld in ((pj, LORA_FIELD[pj]) for pj in self._tlora_targets):
                    getattr(self._tlora_layers[i], fld).a = self._proj_views[llm_lora_key(i, pj, "A")].data_ptr()
                    getattr(self._tlora_layers[i], fld).b = self._proj_views[llm_lora_key(i, pj, "B")].data_ptr()
                    getattr(self._tlora_grad_layers[i], fld).a = self._grad_views[llm_lora_key(i, pj, "A")].data_ptr()
                    getattr(self._tlora_grad_layers[i], fld).b = self._grad_views[llm_lora_key(i, pj, "B")].data_ptr()
            self._tlora = _lib.EncoderLora()
            self._tlora.r = self.text_lora_r
            self._tlora.scaling = float(cfg.text_model_lora_config.get("lora_alpha", 8)) / self.text_lora_r
            self._tlora.layers = self._tlora_layers
            self._tlora_grads = _lib.EncoderLoraGrads()
            self._tlora_grads.layers = self._tlora_grad_layers

        c = _lib.Config()
        c.dtype = self.code
        c.enc_layers, c.enc_d, c.enc_heads, c.enc_ffn = a.encoder_layers, a.d_model, a.encoder_attention_heads, a.encoder_ffn_dim
        c.n_mels, c.enc_max_pos = a.num_mel_bins, a.max_source_positions
        c.enc_block = cfg.audio_latency_block_size or 0
        c.ln_eps = a.layer_norm_eps
        c.stack_factor, c.proj_hidden, c.proj_ln_mid, c.proj_eps = cfg.stack_factor, cfg.hidden_size, int(cfg.projector_ln_mid), 1e-6
        c.proj_act = PROJECTOR_ACTS[cfg.projector_act]      # UVX_PROJ_* (include/uvx.h): SwiGLU or a plain ACT2FN activation
        c.llm_layers, c.llm_d, c.llm_heads, c.llm_kv_heads = t.num_hidden_layers, t.hidden_size, t.num_attention_heads, t.num_key_value_heads
        c.llm_head_dim, c.llm_inter, c.vocab, c.rms_eps = t.head_dim, t.intermediate_size, t.vocab_size, t.rms_norm_eps
        c.llm_flavor = 2 if t.is_gemma3 else (1 if t.is_gemma else 0)      # UVX_LLM_GEMMA3 / UVX_LLM_GEMMA / UVX_LLM_LLAMA (include/uvx.h)
        if t.is_gemma3:      # [3P] Gemma3Attention: scaling = query_pre_attn_scalar ** -0.5; sliding-window layers (window checked per call)
            c.llm_attn_scale = float(t.query_pre_attn_scalar) ** -0.5
            c.llm_window = int(t.sliding_window or 0)
        elif t.window_layers:      # Mistral with a live sliding_window: every layer windowed, head_dim ** -0.5 scale, one rotary table
            c.llm_window = int(t.sliding_window)
        c.llm_act = {"silu": 0, "gelu_pytorch_tanh": 1, "gelu": 2}[t.hidden_act]      # UVX_ACT_* : [3P] ACT2FN[hidden_act]
        c.llm_qk_norm = int(t.has_qk_norm)         # Qwen3: per-head q_norm / k_norm before RoPE
        c.llm_wt_stream = int(self.stream_weight_transposes)
        self._c = c

        e = self._enc if not self.llm_only else {"layers": []}
        self._enc_layers = (_lib.EncLayer * a.encoder_layers)()
        for i, L in enumerate(e["layers"]):
            for n in _lib._ENC_LAYER_FIELDS:
                setattr(self._enc_layers[i], n, 0 if L.get(n) is None else L[n].data_ptr())
        if self.is_wav2vec2:
            wc = _lib.W2vConfig()
            wc.dtype, wc.n_conv, wc.conv_dim = self.code, len(a.conv_kernel), a.conv_dim[0]
            for i, (k, st) in enumerate(zip(a.conv_kernel, a.conv_stride)):
                wc.conv_kernel[i], wc.conv_stride[i] = k, st
            wc.d, wc.heads, wc.ffn, wc.layers = a.d_model, a.encoder_attention_heads, a.encoder_ffn_dim, a.encoder_layers
            wc.pos_k, wc.pos_groups, wc.ln_eps = a.num_conv_pos_embeddings, a.num_conv_pos_embedding_groups, a.layer_norm_eps
            wc.feat_norm_layer, wc.conv_bias, wc.stable_ln = int(a.feat_extract_norm == "layer"), int(a.conv_bias), int(a.do_stable_layer_norm)
            ww = _lib.W2vWeights()
            for n in ("conv0_w", "gn_w", "gn_b", "fp_ln_w", "fp_ln_b", "fp_w", "fp_b", "pos_w", "pos_b", "ln_w", "ln_b"):
                setattr(ww, n, 0 if e[n] is None else e[n].data_ptr())
            for name in ("conv_w", "conv_b", "conv_ln_w", "conv_ln_b"):
                for i, t_ in enumerate(e[name]):
                    getattr(ww, name)[i] = 0 if t_ is None else t_.data_ptr()
            ww.layers = self._enc_layers
            self._w2v_cfg, self._w2v_w, self._ew = wc, ww, None
        elif self.llm_only:
            self._ew = None
        else:
            ew = _lib.EncoderWeights()
            for n in ("conv1_w", "conv1_b", "conv2_w", "conv2_b", "pos", "lnf_w", "lnf_b"):
                setattr(ew, n, e[n].data_ptr())
            ew.layers = self._enc_layers
            self._ew = ew

        pw, pg = _lib.ProjectorWeights(), _lib.ProjectorGrads()
        if not self.llm_only:
            pw.ln_pre = self._proj_views["ln_pre"].data_ptr()
            pw.w1 = self._proj_views["linear_1"].data_ptr()
            pw.w2 = self._proj_views["linear_2"].data_ptr()
            pg.ln_pre = self._grad_views["ln_pre"].data_ptr()
            pg.w1 = self._grad_views["linear_1"].data_ptr()
            pg.w2 = self._grad_views["linear_2"].data_ptr()
            setattr(pw, norm_key, self._proj_views["ln_norm"].data_ptr())
            setattr(pg, norm_key, self._grad_views["ln_norm"].data_ptr())
        self._pw, self._pg = pw, pg

        m = self._llm
        self._llm_layers = (_lib.LlmLayer * t.num_hidden_layers)()
        for i, L in enumerate(m["layers"]):
            for n in _lib._LLM_LAYER_FIELDS:
                setattr(self._llm_layers[i], n, 0 if L.get(n) is None else L[n].data_ptr())
        lw = _lib.LlmWeights()
        lw.embed, lw.norm, lw.lm_head = m["embed"].data_ptr(), m["norm"].data_ptr(), m["lm_head"].data_ptr()
        lw.lm_head_t = 0 if m["lm_head_t"] is None else m["lm_head_t"].data_ptr()
        lw.layers = self._llm_layers
        lw.rope_cos_sin, lw.rope_len = m["rope"].data_ptr(), m["rope_len"]
        if m.get("layer_local") is not None:       # the per-layer sliding-window flags (host array): Gemma-3's local layers, every Mistral layer
            self._layer_local = (C.c_int32 * t.num_hidden_layers)(*m["layer_local"])
            lw.layer_local = self._layer_local
        if m.get("rope_local") is not None:        # Gemma-3: the sliding-window layers' own rotary table
            lw.rope_cos_sin_local = m["rope_local"].data_ptr()
        self._lw = lw

    def projector_state_dict(self) -> Dict[str, torch.Tensor]:
        """Trainable keys under the reference's checkpoint names (ultravox_model.py:565-594 saves these): the projector
        and, with audio_model_lora_config.r > 0, the encoder's LoRA matrices under peft's names."""
        P = "multi_modal_projector."
        out = {} if self.llm_only else {P + "ln_pre.weight": self._proj_views["ln_pre"], P + "linear_1.weight": self._proj_views["linear_1"],
                                        P + self._norm_key + ".weight": self._proj_views["ln_norm"],
                                        P + "linear_2.weight": self._proj_views["linear_2"]}
        out.update({k: self._proj_views[k] for k in self._lora_names})
        return out

    # ------------------------------------------------------------------ checkpoint I/O (ultravox_model.py:565-594)
    def trainable_parameter_names(self):
        return list(self.projector_state_dict().keys())

    def _full_state_dict(self, strict: bool = False) -> Dict[str, torch.Tensor]:
        """What can be saved from this model: the trainable tensors plus the frozen-tower tensors a loaded checkpoint carried
        (`keep_params`, retained on the host by from_pretrained).  Deviation from the reference, stated: there `keep_params`
        may name ANY key of the module's state dict (ultravox_model.py:59, :565-584: the whole model lives in one nn.Module);
        here the frozen towers exist as packed device weights: a keep_param nobody retained on the host is read back from them
        under its plain HF name (weights.unpack_encoder / unpack_wav2vec2 / unpack_llm).  What still cannot be re-saved:
        tower keys while that tower's adapters are un-merged (peft then nests the base weights under other names).
        strict=True (what save_pretrained / save_checkpoint use by default) raises for those; otherwise they are reported with a
        warning and left out (the reload then takes that tower from its base model id)."""
        sd = {**getattr(self, "_kept_tensors", {}), **self.projector_state_dict()}
        # frozen-tower keys nobody retained on the host are read back from the packed device weights (exact inverses of the
        # packing: weights.unpack_*) - the merged towers after merge_and_unload, or keep_params a caller added by name
        for prefix, ok, unpack, packed in (("audio_tower.", not self.llm_only and self.lora_r == 0, unpack_wav2vec2 if self.is_wav2vec2 else unpack_encoder, self._enc),
                                           ("language_model.", self.text_lora_r == 0, unpack_llm, self._llm)):
            if ok and any(k.startswith(prefix) and k not in sd for k in self.keep_params):
                sd = {**unpack(packed, self.config, prefix), **sd}
        lost = sorted(k for k in self.keep_params if k not in sd)
        if lost:
            msg = (f"keep_params names {len(lost)} tensor(s) this model cannot re-save (e.g. {lost[:3]}): not a trainable key, not retained "
                   "from a loaded checkpoint and not a parameter of a plain (adapter-free) Whisper tower / LLM under its checkpoint name")
            if strict:
                raise KeyError(msg)
            import warnings
            warnings.warn(msg + "; they are left out of the saved state dict")
        return sd

    def diff_state_dict(self, state_dict: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
        """Trainable parameters + keys carried by a previously loaded checkpoint (`keep_params`)."""
        from . import checkpoint
        sd = self._full_state_dict() if state_dict is None else state_dict
        return checkpoint.diff_state_dict(sd, self.trainable_parameter_names(), self.keep_params)

    def save_pretrained(self, save_directory: str, state_dict: Optional[Dict[str, torch.Tensor]] = None, strict: bool = True):
        """strict (default): a keep_param this model cannot re-save ABORTS the save - a checkpoint that silently lost the frozen-tower
        tensors the reference would have kept reloads with the base model's instead.  strict=False: warn and leave them out."""
        from . import checkpoint
        sd = self._full_state_dict(strict=strict) if state_dict is None else state_dict
        return checkpoint.save_pretrained(save_directory, self.config, sd, self.trainable_parameter_names(), self.keep_params)

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = False):
        """In-place load of projector keys (the trainable set); frozen-tower keys are only accepted at construction
        time (`from_pretrained`), where they are packed into the device layouts - here they are skipped (and reported as
        unexpected).  Every key that was loaded joins keep_params."""
        mine = self.projector_state_dict()
        unexpected = [k for k in state_dict if k not in mine]
        missing = [k for k in mine if k not in state_dict]
        if unexpected and (strict or any(not k.startswith(("audio_tower.", "language_model.")) for k in unexpected)):
            raise KeyError(f"unexpected key(s): {unexpected[:5]}")
        if strict and missing:
            raise KeyError(f"missing key(s): {missing}")
        for k, v in state_dict.items():
            if k in mine:
                if tuple(v.shape) != tuple(mine[k].shape):
                    raise ValueError(f"size mismatch for {k}: {tuple(v.shape)} vs {tuple(mine[k].shape)}")
                mine[k].copy_(v.to(device=self.device, dtype=self.dtype))
        # only what was actually loaded joins keep_params: tower keys are ignored here (strict=False), and recording them would
        # promise a re-save of tensors this model does not hold
        self.keep_params.update(k for k in state_dict if k in mine)
        return missing, unexpected

    @classmethod
    def from_pretrained(cls, directory: str, base_state_dict: Optional[Dict[str, torch.Tensor]] = None,
                        audio_model_dir: Optional[str] = None, text_model_dir: Optional[str] = None, **kwargs):
        """config.json + model.safetensors written by save_pretrained (here or by the reference).  The towers come from
        `base_state_dict`, or from local HF checkpoint directories of the models `audio_model_id` / `text_model_id` name
        (`audio_model_dir`, `text_model_dir`: safetensors, single file or sharded - there is no hub access here), or, absent
        both, from the seeded random initialisation; checkpoint keys override either."""
        from . import checkpoint
        config, ckpt = checkpoint.load_pretrained(directory)
        dtype = _torch_dtype(config, kwargs.get("dtype"))
        if base_state_dict is None and (audio_model_dir or text_model_dir):
            if not (audio_model_dir and text_model_dir):
                raise ValueError("give both audio_model_dir and text_model_dir (or a complete base_state_dict)")
            base_state_dict = {**checkpoint.audio_tower_state_dict(audio_model_dir),
                               **checkpoint.language_model_state_dict(text_model_dir)}
            for k, v in random_state_dict(config, seed=kwargs.get("seed", 0), dtype=dtype).items():
                if k.startswith("multi_modal_projector."):
                    base_state_dict.setdefault(k, v)       # a projector the checkpoint does not carry starts from its init
        base = dict(base_state_dict) if base_state_dict is not None else random_state_dict(
            config, seed=kwargs.get("seed", 0), dtype=dtype)
        # adapters (apply_lora -> get_peft_model, ultravox_model.py:690-709) exist in the model before the checkpoint is loaded
        # over it: peft's initialisation unless the base already carries them
        for k, v in init_lora_state_dict(config, seed=kwargs.get("seed", 0), dtype=dtype).items():
            base.setdefault(k, v)
        merged, keep = checkpoint.merge_state_dict(base, ckpt)
        model = cls(config, state_dict=merged, **kwargs)
        model.keep_params.update(keep)
        # frozen-tower tensors a checkpoint carried are re-saved with it (ultravox_model.py:565-591 keeps every keep_param);
        # they live packed on the device, so the originals are retained on the host for save_pretrained
        trainable = set(model.trainable_parameter_names())
        model._kept_tensors = {k: merged[k].detach().to("cpu") for k in keep if k not in trainable}
        return model

    @torch.no_grad()
    def merge_and_unload(self) -> None:
        """UltravoxModel.merge_and_unload (ultravox_model.py:528-559; peft's merge): fold every LoRA adapter into the packed
        base weights, W += scaling * B @ A, and drop the adapters - afterwards the towers run their plain (frozen) kernels.
        As in the reference, a tower that carried adapters can no longer be re-created from its base model id: the id is
        cleared, every parameter of that tower joins keep_params (the next save_pretrained writes the merged tower whole, read
        back from the packed device weights - weights.unpack_encoder / unpack_wav2vec2 / unpack_llm) and the two LoRA configs leave the config."""
        def fold(w_rows: torch.Tensor, A: torch.Tensor, B: torch.Tensor, scale: float) -> None:
            w_rows.copy_((w_rows.float() + scale * (B.float() @ A.float())).to(w_rows.dtype))
        kept = getattr(self, "_kept_tensors", {})
        if self.lora_r > 0:
            check_encoder_exportable(self.config)      # BEFORE any weight changes: a tower that cannot be re-exported is not half-merged
        def fold_layer(L, rows, targets, keyfn, i, sc, q_scale=1.0) -> None:
            for pj in targets:      # q / k / v: row blocks of the packed wqkv; out_proj / o_proj: the whole wo; the MLP's linears (ABI 18)
                A, B = self._proj_views[keyfn(i, pj, "A")], self._proj_views[keyfn(i, pj, "B")]
                if pj in rows:
                    lo, hi = rows[pj]
                    fold(L["wqkv"][lo:hi], A, B, sc * (q_scale if pj == "q_proj" else 1.0))
                elif pj in ("gate_proj", "up_proj"):      # rows of the packed gate|up matrix: alternating 16-row gate / up blocks (weights.pack_llm)
                    idx = _gu_rows(L["wgu"].shape[0] // 2, pj == "up_proj", L["wgu"].device)
                    L["wgu"][idx] = (L["wgu"][idx].float() + sc * (B.float() @ A.float())).to(L["wgu"].dtype)
                else:
                    fold(L[{"fc1": "fc1_w", "fc2": "fc2_w", "down_proj": "wd"}.get(pj, "wo")], A, B, sc)
            for n, nt in (("wqkv", "wqkv_t"), ("wo", "wo_t"), ("wgu", "wgu_t"), ("wd", "wd_t"), ("fc1_w", "fc1_t"), ("fc2_w", "fc2_t")):
                if L.get(nt) is not None:
                    L[nt].copy_(L[n].t())
        if self.lora_r > 0:
            d = self.config.audio_config.d_model
            qs = (d // self.config.audio_config.encoder_attention_heads) ** -0.5      # folded into the packed q rows
            rows = {"q_proj": (0, d), "k_proj": (d, 2 * d), "v_proj": (2 * d, 3 * d)}
            for i, L in enumerate(self._enc["layers"]):
                fold_layer(L, rows, self._lora_targets, self._audio_lora_key, i, float(self._lora.scaling), qs)
            self.lora_r = 0
            self._merged_tower("audio_tower.", "audio_model_id", kept)
        if self.text_lora_r > 0:
            t = self.config.text_config
            qc, kc = t.num_attention_heads * t.head_dim, t.num_key_value_heads * t.head_dim
            rows = {"q_proj": (0, qc), "k_proj": (qc, qc + kc), "v_proj": (qc + kc, qc + 2 * kc)}
            for i, L in enumerate(self._llm["layers"]):
                fold_layer(L, rows, self._tlora_targets, llm_lora_key, i, float(self._tlora.scaling))
            self.text_lora_r = 0
            self._merged_tower("language_model.", "text_model_id", kept)
        self._lora_names = []        # the adapters are gone: only the projector remains trainable
        for param in ("text_model_lora_config", "audio_model_lora_config"):      # ultravox_model.py:555-557
            if hasattr(self.config, param):
                delattr(self.config, param)

    def _merged_tower(self, prefix: str, id_attr: str, kept: Dict[str, torch.Tensor]) -> None:
        """Book-keeping of a merged tower (ultravox_model.py:529-553): the base id no longer describes the weights, every tower
        parameter is kept; adapter keys and host copies of pre-merge tensors a loaded checkpoint carried are stale and go."""
        setattr(self.config, id_attr, None)
        self.keep_params = {k for k in self.keep_params if not k.startswith(prefix)}
        for k in [k for k in kept if k.startswith(prefix)]:
            del kept[k]
        self.keep_params.update(self._tower_param_names(prefix))

    def _tower_param_names(self, prefix: str):
        """named_parameters() of a (plain, un-wrapped) tower under the reference's key names."""
        if prefix == "audio_tower.":
            if self.is_wav2vec2:
                return wav2vec2_param_names(self._enc, self.config, prefix)
            return encoder_param_names(self.config, prefix)
        return llm_param_names(self._llm, self.config, prefix)

    def projector_grads(self) -> Dict[str, torch.Tensor]:
        P = "multi_modal_projector."
        if self.llm_only:
            return {k: self._grad_views[k] for k in self._lora_names}
        out = {P + "ln_pre.weight": self._grad_views["ln_pre"], P + "linear_1.weight": self._grad_views["linear_1"],
               P + self._norm_key + ".weight": self._grad_views["ln_norm"],
               P + "linear_2.weight": self._grad_views["linear_2"]}
        out.update({k: self._grad_views[k] for k in self._lora_names})
        return out

    def train(self, mode: bool = True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def __call__(self, *args, **kwargs) -> "CausalLMOutputWithPast":
        """`model(**batch)` as the reference's callers write it (validate.py:35, train loop): the forward pass."""
        return self.forward(*args, **kwargs)

    def set_loss_config(self, loss_config: LossConfig):
        self.loss_config = loss_config

    def get_input_embeddings(self):
        return self._llm["embed"]

    # ------------------------------------------------------------------ workspaces
    def _workspace(self, key: str, nbytes: int) -> torch.Tensor:
        w = self._ws.get(key)
        if w is None or w.numel() < nbytes:
            self._ws[key] = w = torch.empty(nbytes, device=self.device, dtype=torch.uint8)
        return w

    # ------------------------------------------------------------------ device stages
    def audio_tower_forward(self, audio_values: torch.Tensor, audio_len: Optional[torch.Tensor]) -> torch.Tensor:
        """ModifiedWhisperEncoder.forward(input_features, audio_len) -> last_hidden_state [A, Te, d]."""
        l = _lib.lib()
        if self.llm_only:      # the reference's module simply has no such attribute (ultravox_model.py:62-65)
            raise AttributeError("'UltravoxModel' object has no attribute 'audio_tower' (config.llm_only_training: text-only model)")
        if self.is_wav2vec2:
            return self._wav2vec2_forward(audio_values)
        A, n_mels, F = audio_values.shape
        if F > self.audio_tower_context_length:
            raise ValueError(
                f"Whisper expects the mel input features to be of length {self.audio_tower_context_length} or less, "
                f"but found {F}. Make sure to pad the input mel features to {self.audio_tower_context_length}.")
        is_f32 = audio_values.dtype == torch.float32
        if not is_f32 and audio_values.dtype != self.dtype:
            audio_values = audio_values.to(self.dtype)
        audio_values = audio_values.contiguous()
        Te = (F - 1) // 2 + 1
        out = torch.empty((A, Te, self.config.audio_config.d_model), device=self.device, dtype=self.dtype)
        lens = None if audio_len is None else audio_len.to(device=self.device, dtype=torch.int64).contiguous()
        if self.lora_r > 0:
            # LoRA-adapted tower (peft merges nothing at run time: base + lora_B(lora_A(x)) * scaling, also in eval mode);
            # the training workspace keeps the per-layer activations for uvx_encoder_bwd
            nb = l.uvx_encoder_train_ws_bytes(C.byref(self._c), A, F)
            ws = self._workspace("enc_train", nb)
            check(l.uvx_encoder_fwd_train(stream_ptr(), C.byref(self._c), C.byref(self._ew), C.byref(self._lora),
                                          ptr(audio_values), int(is_f32), ptr(lens), A, F, ptr(out), ptr(ws), C.c_size_t(nb)),
                  "uvx_encoder_fwd_train")
            self._enc_ctx = (A, F, nb, lens)
            return out
        nb = l.uvx_encoder_ws_bytes(C.byref(self._c), A, F)
        ws = self._workspace("enc", nb)
        check(l.uvx_encoder_fwd(stream_ptr(), C.byref(self._c), C.byref(self._ew), ptr(audio_values), int(is_f32),
                                ptr(lens), A, F, ptr(out), ptr(ws), C.c_size_t(nb)), "uvx_encoder_fwd")
        return out

    def _wav2vec2_forward(self, input_values: torch.Tensor) -> torch.Tensor:
        """AutoModel branch of the audio tower (ultravox_model.py:460-467, :476-485): Wav2Vec2Model(input_values).last_hidden_state,
        input_values [A, L] = the normalised waveform (the `input_values` fallback, ultravox_processing.py:308); no mask."""
        l = _lib.lib()
        if input_values.dim() != 2:
            raise ValueError(f"the wav2vec2 tower takes input_values [n_audio, n_samples], got shape {tuple(input_values.shape)}")
        A, L = input_values.shape
        is_f32 = input_values.dtype == torch.float32
        if not is_f32 and input_values.dtype != self.dtype:
            input_values = input_values.to(self.dtype)
        input_values = input_values.contiguous()
        Tn = l.uvx_wav2vec2_frames(C.byref(self._w2v_cfg), L)
        if Tn <= 0:
            raise ValueError(f"{L} samples are shorter than the wav2vec2 feature encoder's receptive field")
        out = torch.empty((A, Tn, self.config.audio_config.d_model), device=self.device, dtype=self.dtype)
        if self.lora_r > 0:      # the LoRA-adapted tower (adapters active in eval mode too, as peft's wrapped modules are); stash for uvx_wav2vec2_bwd
            nb = l.uvx_wav2vec2_train_ws_bytes(C.byref(self._w2v_cfg), A, L)
            ws = self._workspace("enc_train", nb)
            check(l.uvx_wav2vec2_fwd_train(stream_ptr(), C.byref(self._w2v_cfg), C.byref(self._w2v_w), C.byref(self._lora), ptr(input_values),
                                           int(is_f32), A, L, ptr(out), ptr(ws), C.c_size_t(nb)), "uvx_wav2vec2_fwd_train")
            self._enc_ctx = (A, L, nb, None)
            return out
        nb = l.uvx_wav2vec2_ws_bytes(C.byref(self._w2v_cfg), A, L)
        ws = self._workspace("enc", nb)
        check(l.uvx_wav2vec2_fwd(stream_ptr(), C.byref(self._w2v_cfg), C.byref(self._w2v_w), ptr(input_values), int(is_f32), A, L,
                                 ptr(out), ptr(ws), C.c_size_t(nb)), "uvx_wav2vec2_fwd")
        return out

    def multi_modal_projector_forward(self, audio_features: torch.Tensor) -> torch.Tensor:
        """UltravoxProjector.forward: [A, Te, C] -> [A, ceil(Te/S), D]."""
        l = _lib.lib()
        A, Te, Cc = audio_features.shape
        Na = (Te + self.config.stack_factor - 1) // self.config.stack_factor
        out = torch.empty((A, Na, self.config.text_config.hidden_size), device=self.device, dtype=self.dtype)
        nb = l.uvx_projector_ws_bytes(C.byref(self._c), A, Te)
        ws = self._workspace("proj", nb)
        check(l.uvx_projector_fwd(stream_ptr(), C.byref(self._c), C.byref(self._pw), ptr(audio_features.contiguous()),
                                  A, Te, ptr(out), ptr(ws), C.c_size_t(nb)), "uvx_projector_fwd")
        self._proj_ctx = (A, Te, nb)
        return out

    def _projector_backward(self, d_audio_embeds: torch.Tensor) -> None:
        l = _lib.lib()
        A, Te, nb = self._proj_ctx
        d_enc = None
        if self.lora_r > 0:
            d_enc = torch.empty((A, Te, self.config.audio_config.d_model), device=self.device, dtype=self.dtype)
        check(l.uvx_projector_bwd(stream_ptr(), C.byref(self._c), C.byref(self._pw), ptr(d_audio_embeds), A, Te,
                                  C.byref(self._pg), ptr(d_enc), ptr(self._ws["proj"]), C.c_size_t(nb)), "uvx_projector_bwd")
        if self.lora_r > 0 and self.is_wav2vec2:
            Ae, L, nbe, _ = self._enc_ctx
            check(l.uvx_wav2vec2_bwd(stream_ptr(), C.byref(self._w2v_cfg), C.byref(self._w2v_w), C.byref(self._lora), ptr(d_enc), Ae, L,
                                     C.byref(self._lora_grads), ptr(self._ws["enc_train"]), C.c_size_t(nbe)), "uvx_wav2vec2_bwd")
        elif self.lora_r > 0:
            Ae, F, nbe, lens = self._enc_ctx
            check(l.uvx_encoder_bwd(stream_ptr(), C.byref(self._c), C.byref(self._ew), C.byref(self._lora), ptr(d_enc),
                                    ptr(lens), Ae, F, C.byref(self._lora_grads), ptr(self._ws["enc_train"]), C.c_size_t(nbe)),
                  "uvx_encoder_bwd")

    def _prepare_audio_embeds(self, inputs_embeds, input_ids, audio_values, audio_token_start_idx, audio_lens,
                              audio_token_len, audio_batch_size):
        # same argument checks (and messages) as ultravox_model.py:363-379
        assert (audio_values is not None and audio_token_start_idx is not None and audio_token_len is not None
                and audio_lens is not None and audio_batch_size is not None), \
            "inputs_embeds/audio_values/audio_token_start_idx/audio_token_len/audio_lens/audio_batch_size must be provided."
        assert len(audio_token_start_idx) == len(audio_token_len) == len(audio_lens) == len(audio_values), \
            "audio_token_start_idx/audio_token_len/audio_lens/audio_values must have the same batch size."
        B, T = (inputs_embeds.shape[:2] if inputs_embeds is not None else input_ids.shape)
        assert len(audio_batch_size) == B, "audio_batch_size and inputs_embeds must have the same batch size."
        if self._before_projector is not None and self.lora_r > 0:
            self._before_projector()             # a LoRA-adapted encoder reads trainable weights: nothing to hide behind
        tower = self.audio_tower_forward(audio_values, audio_lens)
        if self._before_projector is not None:   # the trainer's deferred all-reduce + optimizer step (overlapped with the
            self._before_projector()             # frozen encoder above, which does not read the trainable weights)
        audio_embeds = self.multi_modal_projector_forward(tower)
        if self.is_wav2vec2 and not audio_token_len.is_cuda:
            # raw-waveform tower: audio_token_len comes from the PROCESSOR's frame formula (its audio_frames_fn defaults to the
            # standard 7-layer conv stack); a tower with other conv_kernel / conv_stride values would silently disagree with the
            # rows the projector really produced - checked here when the lengths live on the host (no synchronisation)
            assert int(audio_token_len.max()) <= audio_embeds.shape[1], (
                f"audio_token_len {int(audio_token_len.max())} exceeds the {audio_embeds.shape[1]} rows the projector produced: the "
                "processor's audio_frames_fn does not match audio_config.conv_kernel / conv_stride")
        return self._embed_merge(inputs_embeds, input_ids, audio_embeds, audio_token_start_idx, audio_token_len,
                                 audio_batch_size, B, T)

    def _embed_merge(self, inputs_embeds, input_ids, audio_embeds, start, tok_len, batch_size, B, T):
        l = _lib.lib()
        dev = self.device
        D = self.config.text_config.hidden_size
        if inputs_embeds is None:
            inputs_embeds = torch.empty((B, T, D), device=dev, dtype=self.dtype)
            ids = input_ids.to(device=dev, dtype=torch.int64).contiguous()
        else:
            ids = None  # caller-provided embeddings are overwritten in place like the reference (:394)
            assert inputs_embeds.is_contiguous() and inputs_embeds.dtype == self.dtype
        n_items = 0 if audio_embeds is None else audio_embeds.shape[0]
        Na = 0 if audio_embeds is None else audio_embeds.shape[1]
        scratch = torch.empty(B * T + max(n_items, 1), device=dev, dtype=torch.int32)
        st = None if start is None else start.to(device=dev, dtype=torch.int64).contiguous()
        tl = None if tok_len is None else tok_len.to(device=dev, dtype=torch.int32).contiguous()
        bs = None if batch_size is None else batch_size.reshape(-1).to(device=dev, dtype=torch.int64).contiguous()
        check(l.uvx_embed_merge(stream_ptr(), C.byref(self._c), ptr(self._llm["embed"]), ptr(ids), ptr(audio_embeds),
                                ptr(bs), ptr(st), ptr(tl), B, T, n_items, Na, ptr(inputs_embeds), ptr(scratch)),
              "uvx_embed_merge")
        self._merge_ctx = (st, tl, B, T, n_items, Na, scratch)
        return inputs_embeds

    def language_model_forward(self, inputs_embeds, labels=None, attention_mask=None, want_logits=True,
                               save_for_bwd=False) -> CausalLMOutputWithPast:
        l = _lib.lib()
        B, T, D = inputs_embeds.shape
        dev = self.device
        V = self.config.vocab_size
        nb = l.uvx_llm_ws_bytes(C.byref(self._c), B, T, int(save_for_bwd))
        ws = self._workspace("llm", nb)
        logits = torch.empty((B, T, V), device=dev, dtype=self.dtype) if want_logits else None
        loss = torch.zeros(1, device=dev, dtype=torch.float32) if labels is not None else None
        lab = None if labels is None else labels.to(device=dev, dtype=torch.int64).contiguous()
        am = None if attention_mask is None else attention_mask.to(device=dev, dtype=torch.int64).contiguous()
        # the training step's pair (uvx_llm_fwd_train / uvx_llm_bwd_train): last layer's row-wise half on the supervised rows
        train_pair = (save_for_bwd and not want_logits and lab is not None and self.dtype == torch.bfloat16
                      and self.text_lora_r == 0 and self.top_layer_supervised_rows)
        if train_pair:
            check(l.uvx_llm_fwd_train(stream_ptr(), C.byref(self._c), C.byref(self._lw), ptr(inputs_embeds.contiguous()), ptr(am),
                                      ptr(lab), B, T, ptr(loss), ptr(ws), C.c_size_t(nb)), "uvx_llm_fwd_train")
        elif self.text_lora_r > 0:
            check(l.uvx_llm_fwd_lora(stream_ptr(), C.byref(self._c), C.byref(self._lw), C.byref(self._tlora),
                                     ptr(inputs_embeds.contiguous()), ptr(am), ptr(lab), B, T, ptr(logits), ptr(loss),
                                     int(save_for_bwd), ptr(ws), C.c_size_t(nb)), "uvx_llm_fwd_lora")
        else:
            check(l.uvx_llm_fwd(stream_ptr(), C.byref(self._c), C.byref(self._lw), ptr(inputs_embeds.contiguous()), ptr(am),
                                ptr(lab), B, T, ptr(logits), ptr(loss), int(save_for_bwd), ptr(ws), C.c_size_t(nb)),
                  "uvx_llm_fwd")
        self._llm_ctx = (B, T, nb, lab)
        self._llm_train_pair = bool(train_pair)
        # (what the pair really skipped: Gemma-3's post norms keep its last layer on all rows - bench.py's FLOP count reads this)
        self._llm_top_rows = bool(train_pair) and not self.config.text_config.is_gemma3
        return CausalLMOutputWithPast(loss=None if loss is None else loss[0], logits=logits)

    def language_model_backward(self, grad_scale: float = 1.0, first_pos: int = 0) -> torch.Tensor:
        """d loss / d inputs_embeds of the last ``language_model_forward(save_for_bwd=True)`` (or KL forward): the uvx_llm_bwd*
        entry point that pairs with the forward that ran.  LoRA gradients (text_model_lora_config) land in their buffers.
        first_pos > 0 (the training pair only): the caller needs no gradient below that position - uvx_llm_bwd_train_from."""
        l = _lib.lib()
        B, T, nb, lab = self._llm_ctx
        D = self.config.text_config.hidden_size
        d_embeds = torch.empty((B, T, D), device=self.device, dtype=self.dtype)
        if isinstance(lab, str) and first_pos > 0:
            check(l.uvx_llm_bwd_rows_from(stream_ptr(), C.byref(self._c), C.byref(self._lw), B, T, int(first_pos), ptr(d_embeds),
                                          ptr(self._ws["llm"]), C.c_size_t(nb)), "uvx_llm_bwd_rows_from")
        elif isinstance(lab, str):      # "rows": the compact KL path left d logits for its row list in the workspace
            check(l.uvx_llm_bwd_rows(stream_ptr(), C.byref(self._c), C.byref(self._lw), B, T, ptr(d_embeds),
                                     ptr(self._ws["llm"]), C.c_size_t(nb)), "uvx_llm_bwd_rows")
        elif self.text_lora_r > 0:
            check(l.uvx_llm_bwd_lora(stream_ptr(), C.byref(self._c), C.byref(self._lw), C.byref(self._tlora), ptr(lab), B, T,
                                     C.c_float(grad_scale), ptr(d_embeds), C.byref(self._tlora_grads), ptr(self._ws["llm"]),
                                     C.c_size_t(nb)), "uvx_llm_bwd_lora")
        elif self._llm_train_pair and first_pos > 0:
            check(l.uvx_llm_bwd_train_from(stream_ptr(), C.byref(self._c), C.byref(self._lw), ptr(lab), B, T, int(first_pos), C.c_float(grad_scale),
                                           ptr(d_embeds), ptr(self._ws["llm"]), C.c_size_t(nb)), "uvx_llm_bwd_train_from")
        elif self._llm_train_pair:
            check(l.uvx_llm_bwd_train(stream_ptr(), C.byref(self._c), C.byref(self._lw), ptr(lab), B, T, C.c_float(grad_scale),
                                      ptr(d_embeds), ptr(self._ws["llm"]), C.c_size_t(nb)), "uvx_llm_bwd_train")
        else:
            check(l.uvx_llm_bwd(stream_ptr(), C.byref(self._c), C.byref(self._lw), ptr(lab), B, T, C.c_float(grad_scale),
                                ptr(d_embeds), ptr(self._ws["llm"]), C.c_size_t(nb)), "uvx_llm_bwd")
        return d_embeds

    # ------------------------------------------------------------------ reference API
    def forward(self, input_ids: Optional[torch.Tensor] = None, audio_values: Optional[torch.Tensor] = None,
                inputs_embeds: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None,
                attention_mask: Optional[torch.Tensor] = None, audio_token_start_idx: Optional[torch.Tensor] = None,
                audio_lens: Optional[torch.Tensor] = None, audio_token_len: Optional[torch.Tensor] = None,
                audio_batch_size: Optional[torch.Tensor] = None, past_key_values=None, alt_input_ids=None,
                alt_attention_mask=None, alt_labels=None, return_logits: bool = True, _save_for_bwd: bool = False,
                **kwargs) -> CausalLMOutputWithPast:
        """UltravoxModel.forward (ultravox_model.py:277-352).  `attention_mask` rows must keep ONE contiguous run of positions
        (right or left padding, what DataCollatorForSeq2SeqWithAudio produces): the device path turns each row into a
        [start, end) key range, so a mask with holes would be honoured only at its outer edges.  A CPU mask is checked here;
        a device mask is checked synchronously the first time its shape is seen and asynchronously afterwards (`_check_mask`)."""
        if past_key_values is not None:
            with self._llm_adapters_folded():      # (an un-merged LLM LoRA adapter is folded per call: see generate())
                return self._forward_with_cache(past_key_values, input_ids, inputs_embeds, audio_values, audio_token_start_idx,
                                                audio_lens, audio_token_len, audio_batch_size, labels, attention_mask,
                                                kwargs.get("logits_to_keep", kwargs.get("num_logits_to_keep", 0)))
        if attention_mask is not None:
            self._check_mask(attention_mask)
        use_kl = False
        if self.training and self.loss_config.loss_function != LossFunction.CrossEntropy:
            if self.loss_config.loss_function != LossFunction.KL_Divergence:
                raise ValueError(f"Unsupported loss function: {self.loss_config.loss_function}")
            use_kl = True
        kl_ctx = None
        if (use_kl and getattr(self, "kl_teacher_side_stream", True) and self.device.type == "cuda" and self.dtype == torch.bfloat16
                and not return_logits and self.text_lora_r == 0 and labels is not None and alt_input_ids is not None and alt_labels is not None):
            # the teacher pass depends on the text-only view alone: it starts NOW, on a side stream, next to the encoder, the projector
            # and the student forward (see _kl_teacher_launch)
            kl_ctx = self._kl_teacher_launch(labels, alt_input_ids, alt_attention_mask, alt_labels)
        if audio_values is not None and len(audio_values) > 0:
            inputs_embeds = self._prepare_audio_embeds(inputs_embeds, input_ids, audio_values, audio_token_start_idx,
                                                       audio_lens, audio_token_len, audio_batch_size)
        elif inputs_embeds is None:
            B, T = input_ids.shape
            inputs_embeds = self._embed_merge(None, input_ids, None, None, None, None, B, T)
        if kl_ctx is not None:
            return self._kl_forward_rows(inputs_embeds, attention_mask, alt_input_ids, alt_attention_mask, kl_ctx["pair_row"],
                                         kl_ctx["pair_w"], teacher=kl_ctx)
        if not use_kl:
            return self.language_model_forward(inputs_embeds, labels=labels, attention_mask=attention_mask,
                                               want_logits=return_logits, save_for_bwd=_save_for_bwd)
        return self._kl_forward(inputs_embeds, labels, attention_mask, alt_input_ids, alt_attention_mask, alt_labels,
                                return_logits)

    __call__ = forward

    _MASK_MSG = ("attention_mask rows must keep one contiguous run of positions (padding on one side or both), "
                 "masks with holes are not supported")

    @staticmethod
    def _mask_has_holes(mask: torch.Tensor) -> torch.Tensor:
        m = mask != 0
        flips = (m[:, 1:] != m[:, :-1]).sum(-1)
        return ((flips > 2) | ((flips == 2) & m[:, 0])).any()

    def _check_mask(self, mask: torch.Tensor) -> None:
        """Host masks: checked on the spot.  Device masks: a host synchronisation per step would serialise the pipeline, so the
        verdict is read synchronously only the FIRST time a mask shape is seen; later masks of that shape fold theirs into a
        device flag that is read at the next natural synchronisation point (`raise_pending_errors`, called by
        `UltravoxTrainer.flush / grad_norm / save_checkpoint` and by the next first-time shape)."""
        bad = self._mask_has_holes(mask)
        if not mask.is_cuda:
            if bool(bad):
                raise ValueError(self._MASK_MSG)
            return
        seen = self.__dict__.setdefault("_mask_shapes_seen", set())
        if tuple(mask.shape) not in seen:
            seen.add(tuple(mask.shape))
            self.raise_pending_errors()
            if bool(bad):
                raise ValueError(self._MASK_MSG)
            return
        flag = self.__dict__.get("_mask_err")
        self._mask_err = bad if flag is None else torch.logical_or(flag, bad)

    def raise_pending_errors(self) -> None:
        """Reads (synchronously) the device-side verdict of the attention masks checked asynchronously since the last call."""
        flag = self.__dict__.pop("_mask_err", None)
        if flag is not None and bool(flag):
            raise ValueError(self._MASK_MSG + " (detected on a device mask of an earlier call)")

    def _forward_with_cache(self, past, input_ids, inputs_embeds, audio_values, audio_token_start_idx, audio_lens, audio_token_len,
                            audio_batch_size, labels, attention_mask, logits_to_keep) -> CausalLMOutputWithPast:
        """forward(..., past_key_values=KVState) - what the reference forwards to the language model (ultravox_model.py:328-334)
        and HF's generation loop calls every step: `input_ids` / `inputs_embeds` hold only the NEW positions (HF's contract for
        a cache handed to forward), their keys / values are appended to the cache (uvx_llm_prefill_chunk*) and the logits of
        EVERY new position come back, [B, Tn, V], as from the HF language model; with `logits_to_keep=1` (what HF's generate()
        asks for) only the last position's, [B, 1, V].  The returned `past_key_values` is the extended state (the one handed in is consumed,
        as HF's in-place caches are)."""
        if not isinstance(past, KVState):
            raise TypeError("past_key_values must be a KVState (generate(return_dict_in_generate=True).past_key_values)")
        if labels is not None:
            raise ValueError("forward() with a KV cache is an inference call: labels are not supported")
        l = _lib.lib()
        dev = self.device
        if audio_values is not None and len(audio_values) > 0:
            inputs_embeds = self._prepare_audio_embeds(inputs_embeds, input_ids, audio_values, audio_token_start_idx,
                                                       audio_lens, audio_token_len, audio_batch_size)
        elif inputs_embeds is None:
            inputs_embeds = self._embed_merge(None, input_ids, None, None, None, None, *input_ids.shape)
        B, Tn, D = inputs_embeds.shape
        if B != past.tokens.shape[0]:
            raise ValueError(f"batch size {B} does not match the cache ({past.tokens.shape[0]})")
        keep = int(logits_to_keep or 0)      # HF: 0 = every new position, k = the last k (its generate() asks for 1)
        if keep < 0:
            raise ValueError(f"logits_to_keep={keep} must be >= 0")
        all_rows = Tn > 1 and keep != 1
        if attention_mask is not None and not bool(torch.as_tensor(attention_mask)[:, -Tn:].to("cpu").bool().all()):
            raise ValueError("padding inside the new positions of a cached sequence is not supported")
        P, V = past.cur_len, self.config.vocab_size
        if P + Tn > self._llm["rope_len"]:
            raise ValueError(f"cache + new tokens = {P + Tn} exceeds the RoPE table ({self._llm['rope_len']})")
        cache, Tmax = past.cache, past.Tmax
        if Tmax < P + Tn:                      # grow: rows [0, P) of every (layer, k|v, sequence) plane move over
            new_T = min(max(P + Tn, 2 * Tmax), self._llm["rope_len"])     # (doubling, but never past the RoPE table the kernels check)
            nbytes = l.uvx_kv_cache_bytes(C.byref(self._c), B, new_T)
            grown = torch.empty(nbytes, device=dev, dtype=torch.uint8)
            planes = self._c.llm_layers * 2 * B
            row = nbytes // (planes * new_T)
            grown[:nbytes].view(planes, new_T, row)[:, :P].copy_(cache[:planes * Tmax * row].view(planes, Tmax, row)[:, :P])
            cache, Tmax = grown, new_T
        nb = l.uvx_llm_prefill_chunk_ws_bytes(C.byref(self._c), B, Tn, P)
        ws = self._workspace("infer", nb)
        logits = torch.empty((B, Tn, V) if all_rows else (B, 1, V), device=dev, dtype=self.dtype)
        pos0 = past.pos_next.to(torch.int32).contiguous()
        entry = l.uvx_llm_prefill_chunk_logits if all_rows else l.uvx_llm_prefill_chunk      # [B, Tn, V] / the last position's [B, V]
        check(entry(stream_ptr(), C.byref(self._c), C.byref(self._lw), ptr(inputs_embeds.contiguous()), B, Tn,
                    ptr(cache), Tmax, P, ptr(pos0), ptr(past.kv_start), ptr(logits), ptr(ws), C.c_size_t(nb)),
              "uvx_llm_prefill_chunk_logits" if all_rows else "uvx_llm_prefill_chunk")
        new_ids = (input_ids.to(dev) if input_ids is not None else torch.full((B, Tn), -1, device=dev, dtype=torch.int64))
        state = KVState(cache=cache, Tmax=Tmax, cur_len=P + Tn, pos_next=(pos0 + Tn).contiguous(), kv_start=past.kv_start,
                        tokens=torch.cat([past.tokens, new_ids.to(torch.int64)], dim=1), partial_ok=past.partial_ok)
        if all_rows and 1 < keep < Tn:         # the last `keep` positions (computed with the rest: one LM-head GEMM over the chunk either way)
            logits = logits[:, Tn - keep:].contiguous()
        return CausalLMOutputWithPast(loss=None, logits=logits, past_key_values=state)

    def _kl_forward(self, inputs_embeds, labels, attention_mask, alt_input_ids, alt_attention_mask, alt_labels,
                    return_logits) -> CausalLMOutputWithPast:
        """LossFunction.KL_Divergence (ultravox_model.py:335-345 -> _compute_kl_loss :200-256): a text-only teacher
        pass of the same frozen LLM over alt_input_ids (no_grad), then KL(teacher || student) at kl_temperature over
        the prediction positions plus eot_loss_weight x the end-of-turn positions."""
        if labels is None:
            raise ValueError("labels must be provided")          # _get_prediction_mask, :178-179
        if alt_input_ids is None or alt_labels is None:
            raise ValueError("alt_input_ids / alt_labels are required for the KL loss (include_alt_fields)")
        l = _lib.lib()
        dev = self.device
        V = self.config.vocab_size
        pair_row, pair_w, n_pred = kl_row_pairs(labels, alt_labels, self.loss_config.eot_loss_weight)
        # (under text_model_lora_config the teacher is the SAME adapted model, adapters active, no_grad - `self.language_model.forward`,
        #  ultravox_model.py:212-222 - so both passes go through uvx_llm_fwd_lora; the compact-rows entry points are frozen-LLM only)
        if self.dtype == torch.bfloat16 and not return_logits and n_pred > 0 and self.text_lora_r == 0:
            return self._kl_forward_rows(inputs_embeds, attention_mask, alt_input_ids, alt_attention_mask, pair_row, pair_w)
        # teacher (its own workspace: the student's holds the activations for the backward pass)
        Bt, Tt = alt_input_ids.shape
        student_merge = self._merge_ctx          # the teacher's plain embedding lookup must not replace it
        alt_embeds = self._embed_merge(None, alt_input_ids, None, None, None, None, Bt, Tt)
        self._merge_ctx = student_merge
        nbt = l.uvx_llm_ws_bytes(C.byref(self._c), Bt, Tt, 0)
        wst = self._workspace("llm_teacher", nbt)
        t_logits = self._workspace("teacher_logits", Bt * Tt * V * self.proj_flat.element_size()).view(self.dtype)[: Bt * Tt * V]
        am = None if alt_attention_mask is None else alt_attention_mask.to(device=dev, dtype=torch.int64).contiguous()
        if self.text_lora_r > 0:
            check(l.uvx_llm_fwd_lora(stream_ptr(), C.byref(self._c), C.byref(self._lw), C.byref(self._tlora), ptr(alt_embeds.contiguous()),
                                     ptr(am), None, Bt, Tt, ptr(t_logits), None, 0, ptr(wst), C.c_size_t(nbt)), "uvx_llm_fwd_lora")
        else:
            check(l.uvx_llm_fwd(stream_ptr(), C.byref(self._c), C.byref(self._lw), ptr(alt_embeds.contiguous()), ptr(am), None,
                                Bt, Tt, ptr(t_logits), None, 0, ptr(wst), C.c_size_t(nbt)), "uvx_llm_fwd")
        # student: activations kept for the backward pass; logits stay in the workspace
        out = self.language_model_forward(inputs_embeds, labels=None, attention_mask=attention_mask,
                                          want_logits=return_logits, save_for_bwd=True)
        B, T, nb, _ = self._llm_ctx
        loss = torch.zeros(1, device=dev, dtype=torch.float32)
        pr, pw = pair_row.to(dev), pair_w.to(dev)
        check(l.uvx_llm_kl_loss(stream_ptr(), C.byref(self._c), ptr(t_logits), C.c_int64(Bt * Tt), ptr(pr), ptr(pw), B, T,
                                C.c_float(self.loss_config.kl_temperature), C.c_float(self._kl_grad_scale), ptr(loss),
                                ptr(self._ws["llm"]), C.c_size_t(nb)), "uvx_llm_kl_loss")
        if n_pred == 0:
            loss = loss + float("nan")       # F.kl_div(reduction="batchmean") over zero rows: 0 / 0
        self._llm_ctx = (B, T, nb, None)     # uvx_llm_bwd(labels = NULL): gradient already in place of the logits
        return CausalLMOutputWithPast(loss=loss[0], logits=out.logits)

    def _kl_teacher_launch(self, labels, alt_input_ids, alt_attention_mask, alt_labels, pair=None) -> Optional[dict]:
        """The teacher half of the compact-rows KL step: host row pairs, the text-only embedding lookup and uvx_llm_fwd_rows(teacher).
        The teacher pass is independent of the student's (frozen LLM, no gradient) and of the audio path: with kl_teacher_side_stream
        (default) it is issued on a side stream - as early as forward() can, before the encoder - and its GEMMs (1408 rows at C2: 96-176
        tiles, less than one round of the 256 CUs) fill the CUs the other stream's partly filled rounds leave idle: -6.5 ms per C2 step
        (profiles/r04_kl_side_stream_ab.txt).  Same kernels in the same order per stream: identical results.  None: nothing to pair."""
        l = _lib.lib()
        dev = self.device
        V = self.config.vocab_size
        pair_row, pair_w, n_pred = pair if pair is not None else kl_row_pairs(labels, alt_labels, self.loss_config.eot_loss_weight)
        if n_pred == 0:
            return None
        rows_s, rows_t, pair_c, pw = kl_compact_pairs(pair_row, pair_w)
        ns, nt = int(rows_s.numel()), int(rows_t.numel())
        rows_s, rows_t, pair_c, pw = rows_s.to(dev), rows_t.to(dev), pair_c.to(dev), pw.to(dev)
        Bt, Tt = alt_input_ids.shape
        student_merge = self.__dict__.get("_merge_ctx")
        alt_embeds = self._embed_merge(None, alt_input_ids, None, None, None, None, Bt, Tt).contiguous()
        self._merge_ctx = student_merge          # the teacher's plain embedding lookup must not replace the student's merge context
        nbt = l.uvx_llm_ws_bytes(C.byref(self._c), Bt, Tt, 0)
        wst = self._workspace("llm_teacher", nbt)
        t_logits = self._workspace("teacher_logits", nt * V * 2).view(self.dtype)[: nt * V]
        am = None if alt_attention_mask is None else alt_attention_mask.to(device=dev, dtype=torch.int64).contiguous()
        side = None
        # With several LLM layer chains (uvx_set_option(11, n >= 2)) every uvx_llm_fwd* call forks the device's ONE set of side streams
        # and events; two calls issued from two streams would share it (ordering held only because one host thread issues
        # everything, and the chains of the two calls serialised on the same side streams) - the teacher then runs on the caller's stream.
        if getattr(self, "kl_teacher_side_stream", True) and dev.type == "cuda" and l.uvx_get_option(11) < 2:
            if "_kl_side" not in self.__dict__:
                self._kl_side = torch.cuda.Stream(device=dev)
            side = self._kl_side
            side.wait_stream(torch.cuda.current_stream(dev))
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            check(l.uvx_llm_fwd_rows(stream_ptr(), C.byref(self._c), C.byref(self._lw), ptr(alt_embeds), ptr(am), Bt, Tt,
                                     ptr(rows_t), nt, ptr(t_logits), 0, ptr(wst), C.c_size_t(nbt)), "uvx_llm_fwd_rows")
        return dict(pair_row=pair_row, pair_w=pair_w, rows_s=rows_s, ns=ns, pair_c=pair_c, pw=pw, t_logits=t_logits, side=side,
                    keep=(alt_embeds, am, rows_t))       # (tensors the side stream reads stay referenced until the join)

    def _kl_forward_rows(self, inputs_embeds, attention_mask, alt_input_ids, alt_attention_mask, pair_row, pair_w, teacher=None):
        """The KL step with both LM heads restricted to the rows that enter the loss (prediction / end-of-turn positions):
        identical loss and gradients, ~10x less head and KL-kernel work than full [B, T, V] logits for teacher and student."""
        l = _lib.lib()
        dev = self.device
        if teacher is None:
            teacher = self._kl_teacher_launch(None, alt_input_ids, alt_attention_mask, None, pair=(pair_row, pair_w, 1))
        rows_s, ns, pair_c, pw, t_logits, side = (teacher[k] for k in ("rows_s", "ns", "pair_c", "pw", "t_logits", "side"))
        B, T, D = inputs_embeds.shape
        nb = l.uvx_llm_ws_bytes(C.byref(self._c), B, T, 1)
        ws = self._workspace("llm", nb)
        ams = None if attention_mask is None else attention_mask.to(device=dev, dtype=torch.int64).contiguous()
        check(l.uvx_llm_fwd_rows(stream_ptr(), C.byref(self._c), C.byref(self._lw), ptr(inputs_embeds.contiguous()), ptr(ams), B, T,
                                 ptr(rows_s), ns, None, 1, ptr(ws), C.c_size_t(nb)), "uvx_llm_fwd_rows")
        if side is not None:
            torch.cuda.current_stream(dev).wait_stream(side)
        loss = torch.zeros(1, device=dev, dtype=torch.float32)
        check(l.uvx_llm_kl_loss_rows(stream_ptr(), C.byref(self._c), ptr(t_logits), ptr(pair_c), ptr(pw), B, T, ns,
                                     C.c_float(self.loss_config.kl_temperature), C.c_float(self._kl_grad_scale), ptr(loss),
                                     ptr(ws), C.c_size_t(nb)), "uvx_llm_kl_loss_rows")
        self._llm_ctx = (B, T, nb, "rows")        # forward_backward: uvx_llm_bwd_rows
        self._llm_train_pair = False
        # (round 6: uvx_llm_fwd_rows / uvx_llm_bwd_rows run the last layer's row-wise half on the listed rows only, like the CE pair)
        self._llm_top_rows = not self.config.text_config.is_gemma3 and l.uvx_get_option(3) != 0
        return CausalLMOutputWithPast(loss=loss[0], logits=None)

    @torch.no_grad()
    def generate(self, input_ids: torch.Tensor, audio_values: Optional[torch.Tensor] = None,
                 inputs_embeds: Optional[torch.Tensor] = None, audio_token_start_idx: Optional[torch.Tensor] = None,
                 audio_lens: Optional[torch.Tensor] = None, audio_token_len: Optional[torch.Tensor] = None,
                 audio_batch_size: Optional[torch.Tensor] = None, attention_mask: Optional[torch.Tensor] = None,
                 max_new_tokens: int = 20, eos_token_id=None, pad_token_id: Optional[int] = None,
                 do_sample: bool = False, temperature: float = 1.0, top_k: Optional[int] = None,
                 top_p: Optional[float] = None, generator: Optional[torch.Generator] = None, streamer=None,
                 **kwargs) -> torch.Tensor:
        """UltravoxModel.generate (ultravox_model.py:398-426).  With an un-merged LLM LoRA adapter (text_model_lora_config.r > 0)
        the reference's peft-wrapped LLM decodes with the adapters active; here the decode kernels know plain weights only, so the
        adapters are folded into copies-on-the-side of the q / k rows for the duration of the call and the original rows put back
        afterwards (_llm_adapters_folded): W + scaling * B A applied as one matrix instead of base(x) + scaling * B(A(x)) - the
        same function to bf16 rounding."""
        with self._llm_adapters_folded():
            return self._generate(input_ids, audio_values=audio_values, inputs_embeds=inputs_embeds,
                                  audio_token_start_idx=audio_token_start_idx, audio_lens=audio_lens, audio_token_len=audio_token_len,
                                  audio_batch_size=audio_batch_size, attention_mask=attention_mask, max_new_tokens=max_new_tokens,
                                  eos_token_id=eos_token_id, pad_token_id=pad_token_id, do_sample=do_sample, temperature=temperature,
                                  top_k=top_k, top_p=top_p, generator=generator, streamer=streamer, **kwargs)

    @contextlib.contextmanager
    def _llm_adapters_folded(self):
        """Inference under an un-merged LLM LoRA adapter: the adapted rows of every layer's packed wqkv (and wo under an o_proj adapter)
        <- W + scaling * B A for the body of the `with`, the saved matrices restored on exit (training continues on the un-merged
        pair; one copy of wqkv per layer of scratch: 1.6 GB for Llama-3-8B).  No-op without adapters."""
        r = self.text_lora_r
        if r == 0:
            yield
            return
        t = self.config.text_config
        qc, kc = t.num_attention_heads * t.head_dim, t.num_key_value_heads * t.head_dim
        spans = {"q_proj": (0, qc), "k_proj": (qc, qc + kc), "v_proj": (qc + kc, qc + 2 * kc)}
        whole = {"o_proj": "wo", "down_proj": "wd"}
        touched = sorted({"wqkv" if pj in spans else "wgu" if pj in ("gate_proj", "up_proj") else whole[pj] for pj in self._tlora_targets})
        sc = float(self._tlora.scaling)
        saved = []
        try:
            with torch.no_grad():
                for i, L in enumerate(self._llm["layers"]):
                    saved.append({n: L[n].clone() for n in touched})
                    for pj in self._tlora_targets:
                        A, B = self._proj_views[llm_lora_key(i, pj, "A")], self._proj_views[llm_lora_key(i, pj, "B")]
                        delta = sc * (B.float() @ A.float())
                        if pj in ("gate_proj", "up_proj"):
                            idx = _gu_rows(L["wgu"].shape[0] // 2, pj == "up_proj", L["wgu"].device)
                            L["wgu"][idx] = (L["wgu"][idx].float() + delta).to(L["wgu"].dtype)
                        else:
                            rows = L[whole[pj]] if pj in whole else L["wqkv"][spans[pj][0]:spans[pj][1]]
                            rows.copy_((rows.float() + delta).to(rows.dtype))
            self.text_lora_r = 0
            yield
        finally:
            self.text_lora_r = r
            with torch.no_grad():
                for L, keep in zip(self._llm["layers"], saved):
                    for n, w in keep.items():
                        L[n].copy_(w)

    def _generate(self, input_ids: torch.Tensor, audio_values: Optional[torch.Tensor] = None,
                 inputs_embeds: Optional[torch.Tensor] = None, audio_token_start_idx: Optional[torch.Tensor] = None,
                 audio_lens: Optional[torch.Tensor] = None, audio_token_len: Optional[torch.Tensor] = None,
                 audio_batch_size: Optional[torch.Tensor] = None, attention_mask: Optional[torch.Tensor] = None,
                 max_new_tokens: int = 20, eos_token_id=None, pad_token_id: Optional[int] = None,
                 do_sample: bool = False, temperature: float = 1.0, top_k: Optional[int] = None,
                 top_p: Optional[float] = None, generator: Optional[torch.Generator] = None, streamer=None,
                 **kwargs) -> torch.Tensor:
        """UltravoxModel.generate (ultravox_model.py:398-426): merged embeddings built ONCE, then the LLM's
        prefill + KV-cache decode loop (greedy or sampling).  Returns prompt + generated ids, [B, T + n_new], finished
        sequences padded with pad_token_id like HF's GenerationMixin.  `eos_token_id` may be one id or a list of
        terminators (infer.py:326-328); `streamer` follows HF's protocol: put(prompt ids), put(new ids) per step, end()."""
        past: Optional[KVState] = kwargs.get("past_key_values")
        return_dict = bool(kwargs.get("return_dict_in_generate", False))
        rep = kwargs.get("repetition_penalty")
        rep = None if rep is None or float(rep) == 1.0 else float(rep)
        if rep is not None and not rep > 0:
            raise ValueError(f"`repetition_penalty` has to be a strictly positive float, but is {rep}")
        want_logits = bool(kwargs.get("output_logits", False)) and return_dict      # HF: only reported in the dict form
        step_logits = []
        # the rest of HF's generation keywords (ultravox_model.py:422-426 forwards them all): score processors / stopping criteria are applied in the
        # loop below (generation.py); fields HF would act on and this loop does not build RAISE - ignoring them would return other tokens
        from . import generation
        ignored = generation.check_kwargs(kwargs, ("past_key_values", "return_dict_in_generate", "repetition_penalty", "num_beams", "output_logits",
                                                   "length_penalty", "early_stopping", "num_return_sequences", "output_scores"))
        if ignored:
            import warnings
            warnings.warn(f"generate(): these arguments have no effect here: {ignored}")
        crit = list(kwargs.get("stopping_criteria") or [])
        min_p = kwargs.get("min_p")
        if past is not None and not isinstance(past, KVState):
            raise TypeError("past_key_values must be the KVState a previous generate(return_dict_in_generate=True) returned")
        num_beams = int(kwargs.get("num_beams", 1) or 1)
        nrs = int(kwargs.get("num_return_sequences", 1) or 1)
        if num_beams > 1:
            if streamer is not None:
                raise ValueError("`streamer` cannot be used with beam search (yet!). Make sure that `num_beams` is set to 1.")
            if past is not None:
                raise NotImplementedError("beam search starts from the prompt: past_key_values is not supported with num_beams > 1")
            if nrs > num_beams:
                raise ValueError(f"`num_return_sequences` ({nrs}) has to be smaller or equal to `num_beams` ({num_beams}).")
            if min_p is not None and not do_sample:
                raise NotImplementedError("min_p is a sampling warper: it needs do_sample=True")
        elif nrs != 1:
            raise ValueError(f"Greedy methods without beam search do not support `num_return_sequences` different than 1 (got {nrs}).")
        if do_sample and not temperature > 0:
            raise ValueError("`temperature` has to be a strictly positive float for sampling")
        l = _lib.lib()
        dev = self.device
        if audio_values is not None and len(audio_values) > 0:
            inputs_embeds = self._prepare_audio_embeds(inputs_embeds, input_ids, audio_values, audio_token_start_idx,
                                                       audio_lens, audio_token_len, audio_batch_size)
        elif inputs_embeds is None:
            inputs_embeds = self._embed_merge(None, input_ids, None, None, None, None, *input_ids.shape)
        B, T, D = inputs_embeds.shape
        V = self.config.vocab_size
        eos = self.config.text_config.eos_token_id if eos_token_id is None else eos_token_id
        eos_list = [int(e) for e in eos] if isinstance(eos, (list, tuple)) else [int(eos)]
        eos_ids = torch.tensor(eos_list, device=dev, dtype=torch.int64)
        pad = eos_list[0] if pad_token_id is None else pad_token_id
        Tmax = T + max_new_tokens
        if Tmax > self._llm["rope_len"]:
            raise ValueError(f"prompt + max_new_tokens = {Tmax} exceeds the RoPE table ({self._llm['rope_len']})")
        ids_dev = input_ids.to(dev)
        am = None if attention_mask is None else attention_mask.to(device=dev, dtype=torch.int64).contiguous()
        procs = generation.ScoreProcessors(kwargs, T, max_new_tokens, eos_list, rep)
        if num_beams > 1:
            lp = kwargs.get("length_penalty")
            # do_sample: HF's beam sampling - the warpers join the processor list with min_tokens_to_keep = 1 + the number of terminators
            sample = dict(temperature=float(temperature), top_k=top_k, top_p=top_p, min_p=min_p, keep=len(eos_list) + 1, generator=generator) if do_sample else None
            return self._beam_search(inputs_embeds, ids_dev, am, max_new_tokens, eos_list, pad_token_id, num_beams,
                                     1.0 if lp is None else float(lp), kwargs.get("early_stopping", False), nrs, procs, return_dict, sample, crit)
        # A cache handed in is reused only while it still describes this prompt's prefix (HF trusts the caller here; a
        # re-tokenised reply that no longer matches would silently corrupt the dialogue, so it is checked and dropped).
        P = 0
        if past is not None and past.tokens.shape[0] == B and past.cur_len > 0 and T > 1:
            n = min(past.cur_len, T - 1)                 # at least one token must run to produce logits
            eq = (ids_dev[:, :n] == past.tokens[:, :n]).all(dim=0)
            n_match = n if bool(eq.all()) else int(torch.nonzero(~eq)[0, 0])
            whole = n_match == past.cur_len
            if (whole or (past.partial_ok and n_match > int(past.kv_start.max()))) \
                    and (am is None or bool(am[:, n_match:].all())):
                P = n_match
        self.last_prefill_reused = P
        own_cache = return_dict or P > 0      # a cache that outlives this call cannot live in the shared workspace
        cache_bytes = l.uvx_kv_cache_bytes(C.byref(self._c), B, Tmax)
        # RoPE position of the first token that runs: positions are consecutive past the left padding
        pos0 = None if P == 0 else (past.pos_next - (past.cur_len - P)).to(torch.int32).contiguous()
        if P > 0 and P == past.cur_len and past.Tmax >= Tmax:
            cache, Tmax = past.cache, past.Tmax          # append in place (rows below cur_len stay as they are)
        else:
            cache = (torch.empty(cache_bytes, device=dev, dtype=torch.uint8) if own_cache else self._workspace("kv", cache_bytes))
            if P > 0:                          # grow: rows [0, P) of every (layer, k|v, sequence) plane move over
                planes = self._c.llm_layers * 2 * B
                row = cache_bytes // (planes * Tmax)                      # bytes of one position: kv_heads * head_dim elements
                cache[:cache_bytes].view(planes, Tmax, row)[:, :P].copy_(
                    past.cache[:planes * past.Tmax * row].view(planes, past.Tmax, row)[:, :P])
        nb = max(l.uvx_llm_infer_ws_bytes(C.byref(self._c), B, T), l.uvx_llm_infer_ws_bytes(C.byref(self._c), B, 1),
                 l.uvx_llm_prefill_chunk_ws_bytes(C.byref(self._c), B, T - P, P) if P > 0 else 0)
        ws = self._workspace("infer", nb)
        next_pos = torch.empty(B, device=dev, dtype=torch.int32)
        kv_start = torch.empty(B, device=dev, dtype=torch.int32)
        logits = torch.empty(B, V, device=dev, dtype=self.dtype)
        if P > 0:
            kv_start = past.kv_start
            chunk = inputs_embeds[:, P:].contiguous()
            check(l.uvx_llm_prefill_chunk(stream_ptr(), C.byref(self._c), C.byref(self._lw), ptr(chunk), B, T - P, ptr(cache),
                                          Tmax, P, ptr(pos0), ptr(kv_start), ptr(logits), ptr(ws), C.c_size_t(nb)),
                  "uvx_llm_prefill_chunk")
            next_pos = pos0 + (T - P)
        else:
            check(l.uvx_llm_prefill(stream_ptr(), C.byref(self._c), C.byref(self._lw), ptr(inputs_embeds.contiguous()), ptr(am),
                                    B, T, ptr(cache), Tmax, ptr(next_pos), ptr(kv_start), ptr(logits), ptr(ws),
                                    C.c_size_t(nb)), "uvx_llm_prefill")
        out = [ids_dev]
        if streamer is not None:
            streamer.put(input_ids.cpu())
        nxt = torch.empty(B, device=dev, dtype=torch.int64)
        emb = torch.empty(B, D, device=dev, dtype=self.dtype)
        offs = torch.empty(B + 1, device=dev, dtype=torch.int32)
        unfinished = torch.ones(B, device=dev, dtype=torch.bool)
        n_decoded = 0
        # Greedy decoding without score processors (the BASELINE inference configuration): the loop's per-token bookkeeping - argmax, pad once a
        # sequence has finished, the EOS test, the next RoPE position - is ONE launch (uvx_greedy_select) and one 4-byte read-back instead of
        # argmax + eight small torch kernels (round 6; profiles/r05_decode70_b8_kernel_stats.txt lists them as at::native rows).  Same tokens.
        if not procs.active and not crit and not do_sample and not want_logits and os.environ.get("UVX_GREEDY_SELECT", "1") != "0":
            seq = torch.empty(B, T + max_new_tokens, device=dev, dtype=torch.int64)
            seq[:, :T] = ids_dev
            live = torch.ones(B, device=dev, dtype=torch.int32)
            counter = torch.zeros(2, device=dev, dtype=torch.int32)
            pos = torch.empty(B, device=dev, dtype=torch.int32)
            n_out = 0
            for step in range(max_new_tokens):
                check(l.uvx_greedy_select(stream_ptr(), self.code, ptr(logits), B, V, ptr(eos_ids), len(eos_list), C.c_int64(int(pad)), ptr(live),
                                          ptr(nxt), ptr(seq), C.c_int64(seq.stride(0)), C.c_int64(T + step), ptr(next_pos), ptr(pos), step,
                                          ptr(counter)), "uvx_greedy_select")
                n_out = step + 1
                if streamer is not None:
                    streamer.put(nxt.cpu())
                if step + 1 == max_new_tokens or int(counter[step & 1]) == 0:
                    break
                check(l.uvx_embed_merge(stream_ptr(), C.byref(self._c), ptr(self._llm["embed"]), ptr(nxt), None, None, None,
                                        None, B, 1, 0, 0, ptr(emb), ptr(offs)), "uvx_embed_merge")
                check(l.uvx_llm_decode(stream_ptr(), C.byref(self._c), C.byref(self._lw), ptr(emb), ptr(pos), ptr(kv_start),
                                       ptr(cache), Tmax, T + step, B, ptr(logits), ptr(ws), C.c_size_t(nb)), "uvx_llm_decode")
                n_decoded += 1
            out = [seq[:, :T + n_out]]
            max_new_tokens = 0          # (the generic loop below is skipped)
        for step in range(max_new_tokens):
            if want_logits:
                step_logits.append(logits.clone())
            if procs.active:      # HF order: logits processors on f32 scores first (generation.py), then the warpers / argmax
                scores = procs(torch.cat(out, dim=1), logits.float())
                nxt = self._sample(scores, temperature, top_k, top_p, generator, min_p) if do_sample else torch.argmax(scores, dim=-1)
            elif do_sample:
                nxt = self._sample(logits, temperature, top_k, top_p, generator, min_p)
            else:
                check(l.uvx_argmax(stream_ptr(), self.code, ptr(logits), B, V, ptr(nxt)), "uvx_argmax")
            tok = torch.where(unfinished, nxt, torch.full_like(nxt, pad))
            out.append(tok[:, None])
            if streamer is not None:
                streamer.put(tok.cpu())
            unfinished = unfinished & ~torch.isin(tok, eos_ids)
            if crit:              # [3P] StoppingCriteriaList on the sequence INCLUDING the new token, OR-ed with the EOS test
                unfinished = unfinished & ~generation.stopped(crit, torch.cat(out, dim=1), logits.float())
            if step + 1 == max_new_tokens or not bool(unfinished.any()):
                break
            tok = tok.contiguous()
            check(l.uvx_embed_merge(stream_ptr(), C.byref(self._c), ptr(self._llm["embed"]), ptr(tok), None, None, None,
                                    None, B, 1, 0, 0, ptr(emb), ptr(offs)), "uvx_embed_merge")
            pos = (next_pos + step).contiguous()
            check(l.uvx_llm_decode(stream_ptr(), C.byref(self._c), C.byref(self._lw), ptr(emb), ptr(pos), ptr(kv_start),
                                   ptr(cache), Tmax, T + step, B, ptr(logits), ptr(ws), C.c_size_t(nb)), "uvx_llm_decode")
            n_decoded += 1
        if streamer is not None:
            streamer.end()
        sequences = torch.cat(out, dim=1)
        if not return_dict:
            return sequences
        # like HF's cache after generate: every token but the last one produced has its keys / values stored
        state = KVState(cache=cache, Tmax=Tmax, cur_len=T + n_decoded, pos_next=(next_pos + n_decoded).to(torch.int32).contiguous(),
                        kv_start=kv_start, tokens=sequences[:, :T + n_decoded].contiguous(),
                        partial_ok=bool(past is not None and past.partial_ok))
        return GenerateOutput(sequences=sequences, past_key_values=state, logits=tuple(step_logits) if want_logits else None)

    def _beam_search(self, inputs_embeds: torch.Tensor, ids_dev: torch.Tensor, am: Optional[torch.Tensor], max_new_tokens: int, eos_list,
                     pad_token_id: Optional[int], nb: int, length_penalty: float, early_stopping, nrs: int, procs, return_dict: bool, sample=None, crit=()):
        """generate(num_beams > 1): the reference forwards the keyword to [3P] HF `generate` (ultravox_model.py:422-426), i.e. HF's beam
        search.  Here: ONE prefill of the B prompts, the KV cache rows then replicated per beam ([L][2][B * beams][Tmax][..]); every step is
        a decode batch of B * beams rows through uvx_llm_decode, the search policy runs on the [B, beams * V] f32 log-probabilities on the
        device (top K = max(2, 1 + n_eos) * beams continuations; the `beams` best that do not end run on, ending ones among the first
        `beams` enter the finished set scored sum_logprobs / generated_length ** length_penalty; stop when no prompt can still improve,
        or - early_stopping=True - all have `beams` finished hypotheses, or the length limit ends everything), and the cache rows are
        re-gathered to follow the surviving beams.  One host read-back per step (continue? / did the beams move?).  Returns
        [B * num_return_sequences, T + longest returned hypothesis], shorter ones padded (pad_token_id, else the first terminator) -
        token for token what HF returns (tests/test_generate_gpu.py, oracle: OracleModel.generate_beam pinned against HF).
        sample (do_sample=True: HF's beam sampling, GenerationMixin._get_top_k_continuations): the K continuations are DRAWN without replacement
        from softmax(running score + warped log-probabilities) - one torch.multinomial on the [B, beams * V] matrix per step, the draws unsorted -
        instead of taken by top-k; temperature / top-k / top-p / min-p act on the log-probabilities behind the score processors, each keeping at
        least 1 + n_terminators tokens.  (Token-exact against HF on the CPU generator: tests/test_generate_host_cpu.py; a CUDA generator draws other
        numbers.)"""
        l = _lib.lib()
        dev = self.device
        B, T, D = inputs_embeds.shape
        V = self.config.vocab_size
        BB, Lg = B * nb, max_new_tokens
        Tmax = T + max_new_tokens
        pad = int(pad_token_id) if pad_token_id else int(eos_list[0])
        eos_ids = torch.tensor(eos_list, device=dev, dtype=torch.int64)
        K = max(2, 1 + len(eos_list)) * nb
        if K > nb * V:
            raise ValueError(f"beam search: {K} continuations kept per step exceed beams * vocab = {nb * V}")
        NEG = -1.0e9
        # ---- prefill on the B prompts, then one cache plane per beam ----
        nbytes = max(l.uvx_llm_infer_ws_bytes(C.byref(self._c), B, T), l.uvx_llm_infer_ws_bytes(C.byref(self._c), BB, 1))
        ws = self._workspace("infer", nbytes)
        bytes1 = l.uvx_kv_cache_bytes(C.byref(self._c), B, Tmax)
        cache1 = torch.empty(bytes1, device=dev, dtype=torch.uint8)
        next_pos = torch.empty(B, device=dev, dtype=torch.int32)
        kv_start = torch.empty(B, device=dev, dtype=torch.int32)
        logits1 = torch.empty(B, V, device=dev, dtype=self.dtype)
        check(l.uvx_llm_prefill(stream_ptr(), C.byref(self._c), C.byref(self._lw), ptr(inputs_embeds.contiguous()), ptr(am), B, T, ptr(cache1),
                                Tmax, ptr(next_pos), ptr(kv_start), ptr(logits1), ptr(ws), C.c_size_t(nbytes)), "uvx_llm_prefill")
        planes = self._c.llm_layers * 2
        row = bytes1 // (planes * B * Tmax)                       # bytes of one cached position
        assert l.uvx_kv_cache_bytes(C.byref(self._c), BB, Tmax) == planes * BB * Tmax * row
        cache = cache1.view(planes, B, Tmax * row).repeat_interleave(nb, dim=1).contiguous()
        del cache1
        cv = cache.view(planes, BB, Tmax, row)
        next_pos = next_pos.repeat_interleave(nb).contiguous()
        kv_start = kv_start.repeat_interleave(nb).contiguous()
        logits = logits1.repeat_interleave(nb, dim=0).contiguous()
        prompt_flat = ids_dev.repeat_interleave(nb, dim=0)
        # ---- search state (generated tokens only; the prompt is put back at the end) ----
        run_seq = torch.full((B, nb, Lg), pad, device=dev, dtype=torch.int64)
        run_sc = torch.zeros(B, nb, device=dev, dtype=torch.float32)
        run_sc[:, 1:] = NEG                                       # step 0: all beams hold the same prompt, only the first may branch
        fin_seq = run_seq.clone()
        fin_sc = torch.full((B, nb), NEG, device=dev, dtype=torch.float32)
        fin_done = torch.zeros(B, nb, device=dev, dtype=torch.bool)
        fin_len = torch.zeros(B, nb, device=dev, dtype=torch.int64)
        can_improve = torch.ones(B, 1, device=dev, dtype=torch.bool)
        first_nb = (torch.arange(K, device=dev) < nb)[None, :]
        item0 = (torch.arange(B, device=dev) * nb)[:, None]
        identity = torch.arange(BB, device=dev)
        emb = torch.empty(BB, D, device=dev, dtype=self.dtype)
        offs = torch.empty(BB + 1, device=dev, dtype=torch.int32)
        take = torch.take_along_dim
        for s in range(Lg):
            logp = torch.log_softmax(logits.float(), dim=-1)
            if procs.active:                                      # HF: the processors (generation.py) see the log-probabilities of the flat running sequences
                logp = procs(torch.cat([prompt_flat, run_seq.view(BB, Lg)[:, :s]], dim=1), logp)
            if sample is not None:
                logp = self._warp(logp, sample["temperature"], sample["top_k"], sample["top_p"], sample["min_p"], keep=sample["keep"])
            acc = (logp.view(B, nb, V) + run_sc[:, :, None]).view(B, nb * V)
            if sample is not None:
                idx = torch.multinomial(torch.softmax(acc, dim=-1), K, generator=sample["generator"])
                vals = take(acc, idx, 1)
            else:
                vals, idx = torch.topk(acc, K, dim=1)
            src, tok = idx // V, idx % V
            cand = take(run_seq, src[:, :, None], 1)
            cand[:, :, s] = tok
            hit = torch.isin(tok, eos_ids)
            if crit:      # the caller's stopping criteria see prompt + candidate (HF: the flattened top-K running sequences; scores None), OR-ed with the terminators
                from .generation import stopped
                flat_c = torch.cat([ids_dev.repeat_interleave(K, dim=0), cand[:, :, :s + 1].reshape(B * K, s + 1)], dim=1)
                hit = hit | stopped(crit, flat_c, None).view(B, K)
            if s + 1 >= Lg:
                hit = torch.ones_like(hit)
            masked = vals + hit.to(torch.float32) * NEG
            keep = torch.topk(masked, nb, dim=1)[1]
            run_seq, run_sc, moved = take(cand, keep[:, :, None], 1), take(masked, keep, 1), take(src, keep, 1)
            # finished set: candidates that just ended among the first `nb`, length-normalised; merged with what is there, best nb kept
            just = hit & first_nb
            sc = vals / (float(s + 1) ** length_penalty)
            sc = sc + (fin_done.all(dim=-1, keepdim=True) & (early_stopping is True)).to(torch.float32) * NEG
            sc = sc + (~can_improve).to(torch.float32) * NEG
            sc = sc + (~just).to(torch.float32) * NEG
            m_sc = torch.cat([fin_sc, sc], dim=1)
            order = torch.topk(m_sc, nb, dim=1)[1]
            fin_seq = take(torch.cat([fin_seq, cand], dim=1), order[:, :, None], 1)
            fin_sc = take(m_sc, order, 1)
            fin_done = take(torch.cat([fin_done, just], dim=1), order, 1)
            fin_len = take(torch.cat([fin_len, torch.full((B, K), s + 1, device=dev, dtype=torch.int64)], dim=1), order, 1)
            best_len = Lg if (early_stopping == "never" and length_penalty > 0.0) else s + 1
            best = run_sc[:, :1] / (float(best_len) ** length_penalty)
            worst = torch.where(fin_done, fin_sc.min(dim=1, keepdim=True)[0], torch.full_like(fin_sc, NEG))
            can_improve = can_improve & (best > worst).any(dim=-1, keepdim=True)
            go = can_improve.any() & ~(fin_done.all() & (early_stopping is True)) & ~hit.all()
            flat = (item0 + moved).view(BB)
            go_h, still = torch.stack([go, (flat == identity).all()]).tolist()
            if not go_h:
                break
            cur = T + s                                           # cache rows written so far
            if not still:
                cv[:, :, :cur] = cv[:, flat, :cur]
            tok_next = run_seq[:, :, s].reshape(BB).contiguous()
            check(l.uvx_embed_merge(stream_ptr(), C.byref(self._c), ptr(self._llm["embed"]), ptr(tok_next), None, None, None,
                                    None, BB, 1, 0, 0, ptr(emb), ptr(offs)), "uvx_embed_merge")
            pos = (next_pos + s).contiguous()
            check(l.uvx_llm_decode(stream_ptr(), C.byref(self._c), C.byref(self._lw), ptr(emb), ptr(pos), ptr(kv_start), ptr(cache), Tmax,
                                   cur, BB, ptr(logits), ptr(ws), C.c_size_t(nbytes)), "uvx_llm_decode")
        n = int(fin_len[:, :nrs].max())
        sequences = torch.cat([ids_dev.repeat_interleave(nrs, dim=0), fin_seq[:, :nrs, :n].reshape(B * nrs, n)], dim=1)
        if not return_dict:
            return sequences
        return GenerateOutput(sequences=sequences, past_key_values=None, logits=None, sequences_scores=fin_sc[:, :nrs].reshape(B * nrs))

    @staticmethod
    def _repetition_penalty(scores: torch.Tensor, seen_ids: torch.Tensor, penalty: float) -> torch.Tensor:
        """[3P] HF RepetitionPenaltyLogitsProcessor (the reference's pipeline passes 1.1, ultravox_pipeline.py:100-119): every
        id already in the sequence — prompt, padding and generated tokens alike — has its score divided by the penalty if
        positive, multiplied if negative.  scores: f32 [B, V] (modified in place), seen_ids: int64 [B, n]."""
        s = torch.gather(scores, 1, seen_ids)
        s = torch.where(s < 0, s * penalty, s / penalty)
        return scores.scatter_(1, seen_ids, s)

    @staticmethod
    def _sample(logits: torch.Tensor, temperature: float, top_k, top_p, generator, min_p=None) -> torch.Tensor:
        """HF's sampling policy on the last-position logits (the reference's inference default, infer.py:317-324):
        TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper (-> MinPLogitsWarper) -> multinomial.  Host-side policy on [B, V]."""
        return torch.multinomial(torch.softmax(UltravoxModel._warp(logits, temperature, top_k, top_p, min_p), dim=-1), 1, generator=generator)[:, 0]

    @staticmethod
    def _warp(scores: torch.Tensor, temperature: float, top_k, top_p, min_p=None, keep: int = 1) -> torch.Tensor:
        """[3P] HF's sampling warpers on a [rows, V] score matrix, in HF's order: TemperatureLogitsWarper (only when != 1), TopKLogitsWarper,
        TopPLogitsWarper, MinPLogitsWarper, each with min_tokens_to_keep = `keep` (1; beam sampling: 1 + the number of terminators)."""
        x = scores.float()
        if temperature != 1.0:
            x = x / temperature
        if top_k is not None and top_k > 0:
            kth = torch.topk(x, min(max(int(top_k), keep), x.shape[-1]), dim=-1).values[:, -1:]
            x = x.masked_fill(x < kth, float("-inf"))
        if top_p is not None and top_p < 1.0:
            sv, si = torch.sort(x, dim=-1, descending=False)
            cum = torch.softmax(sv, dim=-1).cumsum(dim=-1)
            remove = cum <= (1.0 - top_p)
            remove[:, -keep:] = False                              # keep at least the `keep` most likely tokens
            x = x.masked_fill(remove.scatter(1, si, remove), float("-inf"))
        if min_p is not None:
            from .generation import min_p_
            x = min_p_(x, float(min_p), keep)
        return x

    # The text before the first audio token needs no backward (include/uvx.h uvx_llm_bwd_train_from): the smallest audio_token_start_idx of the
    # batch as a HOST integer.  Collator output on the host: free.  A device tensor: its minimum is copied to pinned memory right away (queued
    # before the forward) and waited for only when the backward is about to be launched - the GPU has the whole forward queued by then.
    skip_prefix_backward = True

    def _first_audio_pos_begin(self, start):
        if not self.skip_prefix_backward or start is None or start.numel() == 0 or self.text_lora_r > 0:
            return 0
        if not start.is_cuda:
            return int(start.min())
        if self.__dict__.get("_first_pos_host") is None:
            self._first_pos_host = torch.empty(1, dtype=torch.int64, pin_memory=True)
            self._first_pos_event = torch.cuda.Event()
        self._first_pos_host.copy_(start.min().reshape(1), non_blocking=True)
        self._first_pos_event.record()
        return None

    def _first_audio_pos_end(self, first) -> int:
        if first is not None:
            return first
        self._first_pos_event.synchronize()
        return int(self._first_pos_host[0])

    def forward_backward(self, grad_scale: float = 1.0, **batch) -> torch.Tensor:
        """loss = model(**batch).loss; (loss * grad_scale).backward() for the trainable (projector)
        parameters.  Gradients land in `self.proj_grad` (flat f32 bucket, overwritten)."""
        if not self.with_backward:
            raise _lib.UvxError("model was built with with_backward=False")
        assert batch.get("labels") is not None, "labels are required for a training step"
        self._kl_grad_scale = grad_scale
        first = self._first_audio_pos_begin(batch.get("audio_token_start_idx"))
        out = self.forward(return_logits=False, _save_for_bwd=True, **batch)
        self._kl_grad_scale = 1.0
        d_embeds = self.language_model_backward(grad_scale, self._first_audio_pos_end(first))
        l = _lib.lib()
        D = self.config.text_config.hidden_size
        st, tl, B, T, n_items, Na, scratch = self._merge_ctx
        if n_items == 0:     # text-only batch: no gradient reaches the projector / the encoder adapters
            if self.text_lora_r > 0:
                for k, g in self._grad_views.items():
                    if not k.startswith("language_model."):
                        g.zero_()
            else:
                self.proj_grad.zero_()
            return out.loss
        d_audio = torch.empty((n_items, Na, D), device=self.device, dtype=self.dtype)
        check(l.uvx_merge_embeds_bwd(stream_ptr(), C.byref(self._c), ptr(d_embeds), ptr(st), ptr(tl), B, T, n_items, Na,
                                     ptr(d_audio), ptr(scratch)), "uvx_merge_embeds_bwd")
        self._projector_backward(d_audio)
        self._last_d_embeds, self._last_d_audio = d_embeds, d_audio
        return out.loss


def _gu_rows(inter: int, up: bool, device) -> torch.Tensor:
    """Rows of the packed gate|up matrix [2 I, D] that hold gate_proj (up = False) / up_proj (True) row 0 .. I - 1: alternating 16-row blocks."""
    c = torch.arange(inter, device=device)
    return (c // 16) * 32 + (16 if up else 0) + (c % 16)


class UltravoxTrainer:
    """One optimizer step with the reference's semantics: per-rank mean CE loss,
    DP gradient MEAN over ranks (torch DDP), clip_grad_norm_(1.0), AdamW(beta 0.9/0.999, eps 1e-8, wd 0)."""

    def __init__(self, model: UltravoxModel, lr: float = 2e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, max_grad_norm: float = 1.0, master_weights: bool = False,
                 gradient_accumulation_steps: int = 1, overlap_comm: bool = False, lr_scheduler: str = "constant",
                 lr_warmup_steps: float = 0, max_steps: int = 0, lr_scheduler_kwargs: Optional[dict] = None, comm=None):
        """comm: a parallel.UvxComm - the gradient exchange then runs through libuvx.so's own RCCL communicator
        (uvx_comm_allreduce_f32) instead of torch.distributed.all_reduce; same arithmetic (sum over ranks, x 1/world)."""
        from .schedule import LRSchedule
        self.model = model
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.schedule = LRSchedule(lr, lr_scheduler, lr_warmup_steps, max_steps, lr_scheduler_kwargs)
        self.last_lr = self.schedule(0)
        self.max_grad_norm = max_grad_norm
        self.step_count = 0
        self.grad_accum = max(1, int(gradient_accumulation_steps))
        self._micro = 0                       # micro-batches since the last optimizer step
        self._accum = None                    # f32 sum of the (1 / grad_accum)-scaled micro-batch gradients
        self.overlap_comm = overlap_comm
        n = model.proj_flat.numel()
        if n == 0:      # llm_only_training without text_model_lora_config.r > 0: torch.optim raises the same way in the reference's Trainer
            raise ValueError("optimizer got an empty parameter list (llm_only_training trains the language model's LoRA adapters: "
                             "set text_model_lora_config.r > 0)")
        dev = model.device
        self.master = model.proj_flat.float().clone() if master_weights else None
        st_dtype = torch.float32 if (master_weights or model.dtype == torch.float32) else model.dtype
        self.exp_avg = torch.zeros(n, device=dev, dtype=st_dtype)
        self.exp_avg_sq = torch.zeros(n, device=dev, dtype=st_dtype)
        self.scratch = torch.zeros(1025, device=dev, dtype=torch.float32)
        self.comm = comm
        self.world = comm.world if comm is not None else (torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1)
        self._comm_stream = torch.cuda.Stream(device=dev) if comm is not None else None
        self._pending = None
        # diagnostics for the scaling runs (bench.py --gpus N): time the compute stream spent WAITING for the deferred
        # all-reduce in flush() (the exposed part of the collective), from event pairs around the wait
        self.measure_comm = False
        self._comm_waits = []
        # Schedule auto-tuning (autotune_schedule()): the number of LLM layer chains (uvx_set_option 11) that is faster depends
        # on the box (one chain won by 3 ms per step on one MI355X, two chains by 0.7-3.6 ms on others - profiles/r03_*):
        # the first steps try the candidates in turn, timed with events, and the trainer keeps the faster one.  Results are
        # bit-identical under every candidate, so tuning never changes what is trained.
        self._tune = None
        self.llm_schedule = None              # the tuner's verdict (None: not tuned - the library defaults apply)
        self.schedule_chains = None
        self.schedule_timings = {}

    def save_checkpoint(self, directory: str) -> None:
        """checkpoint-N/ of the HF Trainer: the model's diff state dict + optimizer moments + step."""
        from . import checkpoint
        self.flush()
        self.model.raise_pending_errors()
        self.model.save_pretrained(directory)
        checkpoint.save_trainer_state(directory, self.step_count,
                                      {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "master": self.master},
                                      {"lr": self.lr, "betas": list(self.betas), "eps": self.eps, "weight_decay": self.wd,
                                       "max_grad_norm": self.max_grad_norm,
                                       # the tuner's verdict (uvx_set_option keys): a resumed trainer re-applies it, so that the
                                       # attention backward keeps the summation order the run has been using
                                       "llm_schedule": None if not self.llm_schedule else {str(k): int(v) for k, v in self.llm_schedule.items()}})

    def load_checkpoint(self, directory: str) -> None:
        """resume_from_checkpoint: restores projector weights, AdamW moments, master weights and the step counter,
        so that training continues bit-identically."""
        from . import checkpoint
        _, ckpt = checkpoint.load_pretrained(directory)
        self.model.load_state_dict(ckpt)
        step, t, extra = checkpoint.load_trainer_state(directory)
        self.step_count = step
        if (extra or {}).get("llm_schedule"):
            self.llm_schedule = {int(k): int(v) for k, v in extra["llm_schedule"].items()}
            self._apply_schedule(self.llm_schedule)
        self.exp_avg.copy_(t["exp_avg"].to(self.exp_avg.device))
        self.exp_avg_sq.copy_(t["exp_avg_sq"].to(self.exp_avg_sq.device))
        if self.master is not None:
            if "master" not in t:
                raise KeyError("checkpoint has no f32 master weights but the trainer was built with master_weights=True")
            self.master.copy_(t["master"].to(self.master.device))

    def all_reduce_grads(self) -> None:
        """torch DDP semantics: sum over ranks then divide by world size (RCCL over xGMI)."""
        if self.comm is not None:
            self.comm.all_reduce_mean_(self.model.proj_grad)
            return
        from .parallel import dp_mean_
        dp_mean_(self.model.proj_grad)

    def optimizer_step(self) -> None:
        """clip + AdamW with the schedule's lr for this step (HF steps its LambdaLR after the optimizer: step k, counted
        from 0, runs with lr * factor(k))."""
        self.last_lr = self.schedule(self.step_count)
        self.step_count += 1
        self._adamw(self.last_lr)

    def _adamw(self, lr: float) -> None:
        m = self.model
        check(_lib.lib().uvx_adamw_clip_step(
            stream_ptr(), _lib.dtype_code(m.dtype), ptr(m.proj_flat), ptr(self.master), ptr(m.proj_grad),
            ptr(self.exp_avg), ptr(self.exp_avg_sq), C.c_int64(m.proj_flat.numel()), C.c_float(self.max_grad_norm),
            C.c_float(lr), C.c_float(self.betas[0]), C.c_float(self.betas[1]), C.c_float(self.eps),
            C.c_float(self.wd), self.step_count, ptr(self.scratch)), "uvx_adamw_clip_step")

    def grad_norm(self) -> torch.Tensor:
        self.flush()
        self.model.raise_pending_errors()
        return self.scratch[0].sqrt()

    # the two schedules worth trying (profiles/r03_*): ONE layer chain with the fused attention backward (fewer, longer kernels;
    # wins by 1-4 ms per step on the faster MI355X boxes) and TWO chains with the dQ + dK/dV kernel pair (the chains fill each
    # other's GEMM tail rounds; wins by up to 3 ms on the slower ones).  Keys are uvx_set_option keys.
    SCHEDULES = ({11: 1, 13: 1}, {11: 2, 13: 0})

    def autotune_schedule(self, candidates=None, rounds: int = 2) -> None:
        """Arms the tuner: the next 1 + rounds * len(candidates) calls of train_step (one throw-away step first) alternate
        between the candidate schedules (dicts of uvx_set_option key -> value); afterwards the fastest (by its best step) stays
        set.  `self.llm_schedule` holds the verdict (None while tuning), `self.schedule_timings` the best step time of each.
        NOT bit-neutral across candidates: the default pair also switches the attention backward (option 13), whose fused form sums
        in another order than the dQ + dK/dV pair - results agree to rounding, and a run that tunes may round differently from box to
        box.  For reproducible runs / bit-exact resume keep the library default (do not call this) or pass candidates that differ in
        option 11 only (bit-identical); the verdict is saved by `save_checkpoint` and re-applied by `load_checkpoint`.  The options
        are process-global: tuning one trainer sets the schedule of every model in the process."""
        cands = [dict(c) for c in (candidates or self.SCHEDULES)]
        self._tune = {"cands": cands, "plan": [None] + [i for _ in range(rounds) for i in range(len(cands))], "i": 0, "t": {},
                      "pending": None}
        self.llm_schedule = None

    def _apply_schedule(self, sched: dict) -> None:
        for k, v in sched.items():
            _lib.lib().uvx_set_option(int(k), int(v))

    def _tune_begin(self) -> None:
        tn = self._tune
        if tn["pending"] is not None:            # the previous tuned step: read its time (it has finished long ago, or we wait)
            cand, e0, e1 = tn["pending"]
            e1.synchronize()
            if cand is not None:
                tn["t"].setdefault(cand, []).append(e0.elapsed_time(e1))
            tn["pending"] = None
        if tn["i"] >= len(tn["plan"]):
            best = min(tn["t"], key=lambda c: min(tn["t"][c]))
            self._apply_schedule(tn["cands"][best])
            self.llm_schedule = dict(tn["cands"][best])
            self.schedule_chains = int(self.llm_schedule.get(11, 0)) or None
            self.schedule_timings = {json.dumps(tn["cands"][c], sort_keys=True): min(v) for c, v in tn["t"].items()}
            self._tune = None
            return
        cand = tn["plan"][tn["i"]]
        tn["i"] += 1
        if cand is not None:
            self._apply_schedule(tn["cands"][cand])
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        tn["pending"] = (cand, e0, None)

    def _tune_end(self) -> None:
        tn = self._tune
        if tn is not None and tn["pending"] is not None and tn["pending"][2] is None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            tn["pending"] = (tn["pending"][0], tn["pending"][1], e1)

    def train_step(self, **batch) -> torch.Tensor:
        if self._tune is not None:
            self.flush()
            self._tune_begin()
        try:
            return self._train_step(**batch)
        finally:
            if self._tune is not None:
                self.flush()                     # a tuned step includes its own (otherwise deferred) exchange + update
                self._tune_end()

    def _train_step(self, **batch) -> torch.Tensor:
        """One optimizer step.  With `overlap_comm` (and more than one rank) the gradient all-reduce is launched
        asynchronously after the backward pass and waited for - together with clip + AdamW - only when the NEXT step
        reaches the projector: the next step's log-mel and frozen-encoder forward, which do not read the trainable
        weights, hide the collective.  Same arithmetic in the same order as the sequential schedule; call flush() before
        reading the parameters or the gradient norm."""
        self.model.train()
        if self._pending is not None:
            av = batch.get("audio_values")
            if av is not None and len(av) > 0:
                self.model._before_projector = self.flush
            else:
                self.flush()    # text-only batch: no encoder work to hide behind, and its backward rewrites the bucket
        try:
            loss = self.model.forward_backward(grad_scale=1.0 / self.grad_accum, **batch)
        finally:
            self.model._before_projector = None
        assert self._pending is None
        if self.grad_accum > 1:
            # HF / accelerate: micro-batch losses are divided by gradient_accumulation_steps (train.py:285), gradients add up
            # locally (DDP no_sync) and ranks synchronise once, at the boundary.  forward_backward overwrites the bucket, so
            # the running sum lives next to it.
            self._micro += 1
            if self._micro == 1:
                if self._accum is None:
                    self._accum = torch.empty_like(self.model.proj_grad)
                self._accum.copy_(self.model.proj_grad)
            else:
                self._accum.add_(self.model.proj_grad)
            if self._micro < self.grad_accum:
                return loss
            self.model.proj_grad.copy_(self._accum)
            self._micro = 0
        if self.overlap_comm and self.comm is not None:
            # the exchange runs on a side stream behind the backward pass; flush() makes the compute stream wait for it
            ready = torch.cuda.Event()
            ready.record()
            self._comm_stream.wait_event(ready)
            self.comm.all_reduce_mean_(self.model.proj_grad, stream=self._comm_stream)
            done = torch.cuda.Event()
            done.record(self._comm_stream)
            self._pending = done
        elif self.overlap_comm and torch.distributed.is_available() and torch.distributed.is_initialized():
            self._pending = torch.distributed.all_reduce(self.model.proj_grad, op=torch.distributed.ReduceOp.SUM, async_op=True)
        else:
            self.all_reduce_grads()
            self.optimizer_step()
        return loss

    def flush(self) -> None:
        """Complete a deferred all-reduce + optimizer step (no-op when nothing is pending)."""
        if self._pending is None:
            return
        work, self._pending = self._pending, None
        if self.measure_comm:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        if self.comm is not None:
            torch.cuda.current_stream().wait_event(work)   # uvx_comm_allreduce_f32 already applied the 1 / world
        else:
            work.wait()                                    # the compute stream waits for the collective
        if self.measure_comm:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self._comm_waits.append((e0, e1))
        if self.comm is None:
            self.model.proj_grad.mul_(1.0 / self.world)    # DDP: sum, then divide by the world size
        self.optimizer_step()

    def comm_exposed_ms(self, reset: bool = True) -> float:
        """Sum over the flushes since the last call of the time the compute stream waited for the deferred all-reduce
        (measure_comm = True).  Synchronises."""
        torch.cuda.synchronize()
        total = sum(a.elapsed_time(b) for a, b in self._comm_waits)
        if reset:
            self._comm_waits = []
        return float(total)
