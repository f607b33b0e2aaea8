"""ORACLE — test infrastructure only.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this module; the product path (ultravox_amd/) never does.

A CPU restatement, in plain PyTorch, of the reference's audio->LLM hot path.  Each function cites the
reference lines (relative to /root/reference) or the third-party (transformers 4.51.3, pinned by the
reference's poetry.lock:7739-7740) routine it follows.  The reference's own model module cannot be
imported as-is in this image (`peft` missing, transformers 5.x API drift — SURVEY.md §8c), hence a
restatement; it is pinned against
  * the reference's UltravoxProjector / StackAudioFrames / RMSNorm / SwiGLU classes and the reference's
    UltravoxProcessor, imported from /root/reference in the build container (tests/golden/make_golden.py
    writes the fixtures, tests/test_oracle_pinning.py replays them anywhere);
  * the reference's `_get_prediction_mask` / `_compute_kl_loss`, `init_latency_mask` and `diff_state_dict`, called from the
    imported reference module on stub objects (fixtures kl_loss.npz, latency_mask.npz, diff_state_dict.json);
  * the installed HF WhisperFeatureExtractor, WhisperEncoderLayer and LlamaForCausalLM blocks
    (tests/test_oracle_pinning.py), which are the third-party arithmetic the reference calls;
  * the reference's `apply_lora` (ultravox_model.py:690-709) run on installed-HF towers through tests/peft_stub.py (fixture
    lora_reference.npz / .json): adapted modules, trainable and checkpoint key names, forward and adapter gradients of the
    LoRA sub-path.  peft itself (pinned ~0.11.1) cannot be installed here; its Linear.forward is restated by that stub, which is
    the one part of this pin that is not the reference's or HF's own code.
Floating-point parity of mel / encoder / logits / loss / grads is NOT pinned by any reference test
("parity unpinned" at the fp level); integer/index behaviour is pinned.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

N_FFT, HOP = 400, 160

# bf16 runs only.  False: eager attention as HF's `eager` class computes it in the model dtype (scores rounded to bf16).
# True: attention as a FUSED flash kernel computes it (torch SDPA's flash back end, which the reference's default `sdpa`
# attention class dispatches to on a GPU, and this repo's kernels): q, k, v in the model dtype, scores and softmax statistics
# in f32, keys walked in tiles of FLASH_TILE with a running row maximum, the probabilities of a tile rounded to the model
# dtype RELATIVE TO THE RUNNING MAXIMUM at that tile (exp(s - m_run), un-normalised) as the P.V operand, f32 accumulation
# with the usual rescaling, one division by the f32 row sum and one rounding of the output.  The backward recomputes
# P = exp(s - lse) from the saved log-sum-exp and rounds P and dS = P * (dP - delta) to the model dtype as matmul operands.
# Why the tile walk is restated at all: the rounding of P is the one rounding point of the path whose REALISATION depends on
# the implementation (normalised vs running-max-relative), and one differently rounded attention output decorrelates every
# later bf16 rounding - two bf16 implementations that differ only there end up as far apart as bf16 is from f32 (measured:
# 4.6e-3 rel-L2 after two encoder layers either way).  Every other rounding point (linear outputs, norms, RoPE, activations,
# residual adds) is torch's own bf16 module-by-module rounding in both settings.
FUSED_ATTENTION = False
FLASH_TILE = 64


class fused_attention:
    """with fused_attention(): ... - run the bf16 oracle with flash-kernel attention rounding (see FUSED_ATTENTION)."""

    def __enter__(self):
        global FUSED_ATTENTION
        self._old, FUSED_ATTENTION = FUSED_ATTENTION, True

    def __exit__(self, *a):
        global FUSED_ATTENTION
        FUSED_ATTENTION = self._old


class _FlashAttend(torch.autograd.Function):
    """Flash attention with the rounding points stated at FUSED_ATTENTION, over [B, H, T, dh] operands."""

    @staticmethod
    def forward(ctx, q, k, v, mask, scale):
        dt = q.dtype
        s = (q.float() @ k.float().transpose(-1, -2)) * scale
        if mask is not None:
            s = s.masked_fill(mask.float().expand_as(s) < -1e30, float("-inf"))
        Tk = s.shape[-1]
        m = torch.full(s.shape[:-1], float("-inf"))
        l = torch.zeros(s.shape[:-1])
        o = torch.zeros(*s.shape[:-1], v.shape[-1])
        vf = v.float()
        for k0 in range(0, Tk, FLASH_TILE):
            st = s[..., k0:k0 + FLASH_TILE]
            m_new = torch.maximum(m, st.max(-1).values)
            m_use = torch.where(torch.isinf(m_new), torch.zeros_like(m_new), m_new)      # a row with no valid key so far
            alpha = torch.exp(m - m_use)                                                  # exp(-inf) = 0
            pt = torch.exp(st - m_use[..., None])
            l = l * alpha + pt.sum(-1)
            o = o * alpha[..., None] + pt.to(dt).float() @ vf[..., k0:k0 + FLASH_TILE, :]
            m = m_new
        out = (o / l[..., None]).to(dt)
        ctx.save_for_backward(q, k, v, out, m + torch.log(l))
        ctx.mask, ctx.scale = mask, scale
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse = ctx.saved_tensors
        dt = q.dtype
        s = (q.float() @ k.float().transpose(-1, -2)) * ctx.scale
        if ctx.mask is not None:
            s = s.masked_fill(ctx.mask.float().expand_as(s) < -1e30, float("-inf"))
        pr = torch.exp(s - lse[..., None])
        do = dout.float()
        delta = (do * out.float()).sum(-1, keepdim=True)
        dp = do @ v.float().transpose(-1, -2)
        ds = (pr * (dp - delta)).to(dt).float()
        dv = (pr.to(dt).float().transpose(-1, -2) @ do).to(dt)
        dq = ((ds @ k.float()) * ctx.scale).to(dt)
        dk = ((ds.transpose(-1, -2) @ q.float()) * ctx.scale).to(dt)
        return dq, dk, dv, None, None


def _attend(q, k, v, mask, scale: float):
    """softmax(q k^T * scale + mask) v over [B, H, T, dh] operands; mask additive (finfo.min) or None."""
    dt = q.dtype
    if FUSED_ATTENTION and dt != torch.float32:
        return _FlashAttend.apply(q, k, v, mask, scale)
    s = q @ k.transpose(-1, -2)
    if scale != 1.0:
        s = s * scale
    if mask is not None:
        s = s + mask
    p = torch.softmax(s.float(), dim=-1).to(dt)
    return p @ v


# ----------------------------------------------------------------------------------------------
# K1 — log-mel.  [3P] WhisperFeatureExtractor._torch_extract_fbank_features, called from
# ultravox_processing.py:295-303.
# ----------------------------------------------------------------------------------------------
def mel_filters_ref(n_mels: int) -> np.ndarray:
    """[3P] audio_utils.mel_filter_bank(201, n_mels, 0, 8000, 16000, norm='slaney', mel_scale='slaney')."""

    def hz2mel(f):
        f = np.asarray(f, np.float64)
        m = 3.0 * f / 200.0
        big = f >= 1000.0
        m = np.where(big, 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) * (27.0 / np.log(6.4)), m)
        return m

    def mel2hz(m):
        m = np.asarray(m, np.float64)
        f = 200.0 * m / 3.0
        big = m >= 15.0
        return np.where(big, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), f)

    edges = mel2hz(np.linspace(hz2mel(0.0), hz2mel(8000.0), n_mels + 2))
    bins = np.linspace(0, 8000, 201)
    out = np.zeros((201, n_mels))
    for j in range(n_mels):
        lo, ce, hi = edges[j], edges[j + 1], edges[j + 2]
        rise = (bins - lo) / (ce - lo)
        fall = (hi - bins) / (hi - ce)
        out[:, j] = np.maximum(0.0, np.minimum(rise, fall)) * (2.0 / (hi - lo))
    return out


def logmel_ref(waveform: torch.Tensor, n_mels: int = 80) -> torch.Tensor:
    """waveform [B, L] f32 -> [B, n_mels, L // 160] f32 (STFT, power, drop last frame, mel, log10,
    per-clip floor at max - 8, (x + 4) / 4)."""
    waveform = waveform.to(torch.float32)
    window = torch.hann_window(N_FFT)
    stft = torch.stft(waveform, N_FFT, HOP, window=window, return_complex=True)
    power = stft[..., :-1].abs() ** 2
    fb = torch.from_numpy(mel_filters_ref(n_mels)).to(torch.float32)
    mel = fb.T @ power
    log_spec = torch.clamp(mel, min=1e-10).log10()
    mx = log_spec.max(dim=2, keepdim=True)[0].max(dim=1, keepdim=True)[0]
    log_spec = torch.maximum(log_spec, mx - 8.0)
    return (log_spec + 4.0) / 4.0


class FeatureExtractorRef:
    """The subset of the HF feature-extractor call contract that ultravox_processing.py:295-303 uses."""

    hop_length = HOP
    sampling_rate = 16000
    model_input_names = ["input_features"]

    def __init__(self, feature_size: int = 80):
        self.feature_size = feature_size

    @property
    def feature_extractor(self):
        return self

    def __call__(self, raw_speech, sampling_rate=None, padding="longest", pad_to_multiple_of=None, truncation=False,
                 return_attention_mask=True, **kw):
        raw = [np.asarray(x, np.float32) for x in raw_speech]
        L = max(len(x) for x in raw)
        if pad_to_multiple_of:
            L = -(-L // pad_to_multiple_of) * pad_to_multiple_of
        batch = np.zeros((len(raw), L), np.float32)
        mask = np.zeros((len(raw), L), np.int32)
        for i, x in enumerate(raw):
            batch[i, : len(x)] = x
            mask[i, : len(x)] = 1
        feats = logmel_ref(torch.from_numpy(batch), self.feature_size)
        fmask = mask[:, ::HOP]
        if L % HOP != 0:
            fmask = fmask[:, :-1]
        return {"input_features": feats, "attention_mask": torch.from_numpy(np.ascontiguousarray(fmask))}


# ----------------------------------------------------------------------------------------------
# Encoder — ModifiedWhisperEncoder.forward, ultravox_model.py:865-994 (+ [3P] WhisperEncoderLayer /
# WhisperAttention / get_extended_attention_mask).
# ----------------------------------------------------------------------------------------------
def latency_mask_ref(max_context: int, block: int, dtype: torch.dtype) -> torch.Tensor:
    """init_latency_mask, ultravox_model.py:834-863."""
    assert max_context % block == 0, f"audio_latency_block_size {block} must divide {max_context} evenly."
    nb = max_context // block
    m = torch.tril(torch.ones(nb, nb)).repeat_interleave(block, 0).repeat_interleave(block, 1)
    return ((1.0 - m) * torch.finfo(dtype).min)[None, None]


def whisper_encoder_ref(sd: Dict[str, torch.Tensor], cfg, input_features: torch.Tensor,
                        audio_len: Optional[torch.Tensor], prefix: str = "audio_tower.",
                        return_layer: Optional[int] = None, lora: Optional[dict] = None) -> torch.Tensor:
    a = cfg.audio_config
    dt = input_features.dtype
    H, d = a.encoder_attention_heads, a.d_model
    dh = d // H
    W = lambda k: sd[prefix + k].to(dt)
    max_ctx = a.max_source_positions * 2                                           # :826-832
    if input_features.shape[-1] > max_ctx:                                         # :874-878
        raise ValueError(f"Whisper expects the mel input features to be of length {max_ctx} or less, but found "
                         f"{input_features.shape[-1]}. Make sure to pad the input mel features to {max_ctx}.")
    x = F.gelu(F.conv1d(input_features, W("conv1.weight"), W("conv1.bias"), padding=1))         # :893
    x = F.gelu(F.conv1d(x, W("conv2.weight"), W("conv2.bias"), stride=2, padding=1))            # :894
    x = x.permute(0, 2, 1)                                                                       # :896
    x = x + W("embed_positions.weight")[: x.size(-2)]                                            # :897-899
    B, S, _ = x.shape
    mask = None
    if audio_len is not None:                                                                    # :915-926
        feat_len = (audio_len - 1) // 2 + 1
        keep = torch.arange(S)[None, :].lt(feat_len.view(-1, 1))
        mask = (1.0 - keep[:, None, None, :].to(dt)) * torch.finfo(dt).min
    if cfg.audio_latency_block_size is not None:                                                 # :928-936
        sm = latency_mask_ref(max_ctx, cfg.audio_latency_block_size, dt)[:, :, :S, :S]
        mask = torch.minimum(sm, mask) if mask is not None else sm
        mask = mask.to(dt)
    scaling = dh ** -0.5
    for i in range(a.encoder_layers):                                                            # :944-975
        L = f"layers.{i}."
        res = x
        h = F.layer_norm(x, (d,), W(L + "self_attn_layer_norm.weight"), W(L + "self_attn_layer_norm.bias"), a.layer_norm_eps)
        q = F.linear(h, W(L + "self_attn.q_proj.weight"), W(L + "self_attn.q_proj.bias"))
        k = F.linear(h, W(L + "self_attn.k_proj.weight"))
        if lora is not None:
            # peft LoRA (apply_lora -> get_peft_model, ultravox_model.py:690-709) on q_proj / k_proj, dropout 0:
            # result = base(x) + lora_B(lora_A(x)) * (lora_alpha / r); the adapter matrices live in sd under peft's names.
            # Pinned by tests/golden/lora_reference.npz: the reference's apply_lora on an HF WhisperEncoder through
            # tests/peft_stub.py (peft ~0.11.1, pyproject.toml:15, is not installable here; the stub restates its Linear)
            def lo(pj):
                A = sd[f"{prefix}base_model.model.layers.{i}.self_attn.{pj}.lora_A.default.weight"].to(dt)
                Bm = sd[f"{prefix}base_model.model.layers.{i}.self_attn.{pj}.lora_B.default.weight"].to(dt)
                return F.linear(F.linear(h, A), Bm) * lora["scaling"]
            q = q + lo("q_proj")
            k = k + lo("k_proj")
        q = q * scaling                                      # WhisperAttention: q_proj(x) * head_dim^-0.5
        v = F.linear(h, W(L + "self_attn.v_proj.weight"), W(L + "self_attn.v_proj.bias"))
        q, k, v = (t.view(B, S, H, dh).transpose(1, 2) for t in (q, k, v))
        o = _attend(q, k, v, mask, 1.0).transpose(1, 2).reshape(B, S, d)
        x = res + F.linear(o, W(L + "self_attn.out_proj.weight"), W(L + "self_attn.out_proj.bias"))
        res = x
        h = F.layer_norm(x, (d,), W(L + "final_layer_norm.weight"), W(L + "final_layer_norm.bias"), a.layer_norm_eps)
        h = F.gelu(F.linear(h, W(L + "fc1.weight"), W(L + "fc1.bias")))
        x = res + F.linear(h, W(L + "fc2.weight"), W(L + "fc2.bias"))
        if return_layer is not None and i == return_layer:
            return x
    return F.layer_norm(x, (d,), W("layer_norm.weight"), W("layer_norm.bias"), a.layer_norm_eps)  # :980


# ----------------------------------------------------------------------------------------------
# Alt audio tower (BASELINE.json config 5): [3P] transformers Wav2Vec2Model.forward, the AutoModel branch of
# UltravoxModel._create_audio_tower (ultravox_model.py:460-467, :476-485).  In this reference snapshot that branch cannot
# run end to end (max_context_length / audio_len / hop_length exist only for the Whisper tower - SURVEY.md §8f-4), so the
# contract is the third-party module's: input_values [B, L] (zero-mean / unit-variance waveform, the `input_values` fallback
# of ultravox_processing.py:308) -> last_hidden_state [B, frames, hidden], no attention mask (wav2vec2-large-960h is a
# feat_extract_norm="group" model: trained and used without one).  Pinned against the installed HF Wav2Vec2Model.
# ----------------------------------------------------------------------------------------------
def wav2vec2_normalize_ref(pcm: torch.Tensor) -> torch.Tensor:
    """[3P] Wav2Vec2FeatureExtractor.zero_mean_unit_var_norm for equal-length clips [B, L]: (x - mean) / sqrt(var + 1e-7)."""
    return (pcm - pcm.mean(-1, keepdim=True)) / torch.sqrt(pcm.var(-1, unbiased=False, keepdim=True) + 1e-7)


def pos_conv_weight_ref(sd: Dict[str, torch.Tensor], prefix: str) -> torch.Tensor:
    """weight_norm(conv, dim=2): w = g * v / ||v|| with the norm over (out, in) per kernel position; torch's parametrization
    names (original0 = g, original1 = v) or the older weight_g / weight_v."""
    P = prefix + "encoder.pos_conv_embed.conv."
    if P + "parametrizations.weight.original0" in sd:
        g, v = sd[P + "parametrizations.weight.original0"], sd[P + "parametrizations.weight.original1"]
    else:
        g, v = sd[P + "weight_g"], sd[P + "weight_v"]
    return v * (g / v.float().norm(dim=(0, 1), keepdim=True).to(v.dtype))


def wav2vec2_encoder_ref(sd: Dict[str, torch.Tensor], cfg, input_values: torch.Tensor, prefix: str = "audio_tower.") -> torch.Tensor:
    a = cfg.audio_config
    dt = input_values.dtype
    W = lambda k: sd[prefix + k].to(dt)
    H, d = a.encoder_attention_heads, a.d_model
    dh = d // H
    x = input_values[:, None]                                                     # Wav2Vec2FeatureEncoder
    for i, (k, st) in enumerate(zip(a.conv_kernel, a.conv_stride)):
        x = F.conv1d(x, W(f"feature_extractor.conv_layers.{i}.conv.weight"), stride=st)
        if i == 0:      # Wav2Vec2GroupNormConvLayer: GroupNorm(num_groups = channels) = per-channel statistics over time
            C = x.shape[1]
            x = F.group_norm(x, C, W("feature_extractor.conv_layers.0.layer_norm.weight"),
                             W("feature_extractor.conv_layers.0.layer_norm.bias"), 1e-5)
        x = F.gelu(x)
    x = x.transpose(1, 2)                                                         # [B, T, conv_dim]
    C = x.shape[-1]
    x = F.layer_norm(x, (C,), W("feature_projection.layer_norm.weight"), W("feature_projection.layer_norm.bias"), a.layer_norm_eps)
    x = F.linear(x, W("feature_projection.projection.weight"), W("feature_projection.projection.bias"))
    K, G = a.num_conv_pos_embeddings, a.num_conv_pos_embedding_groups             # Wav2Vec2PositionalConvEmbedding
    pos = F.conv1d(x.transpose(1, 2), pos_conv_weight_ref(sd, prefix).to(dt), W("encoder.pos_conv_embed.conv.bias"), padding=K // 2, groups=G)
    if K % 2 == 0:
        pos = pos[:, :, :-1]                                                      # Wav2Vec2SamePadLayer
    x = x + F.gelu(pos).transpose(1, 2)
    x = F.layer_norm(x, (d,), W("encoder.layer_norm.weight"), W("encoder.layer_norm.bias"), a.layer_norm_eps)
    B, S, _ = x.shape
    for i in range(a.encoder_layers):                                             # Wav2Vec2EncoderLayer (post-LN)
        L = f"encoder.layers.{i}."
        q = F.linear(x, W(L + "attention.q_proj.weight"), W(L + "attention.q_proj.bias")) * dh ** -0.5
        k = F.linear(x, W(L + "attention.k_proj.weight"), W(L + "attention.k_proj.bias"))
        v = F.linear(x, W(L + "attention.v_proj.weight"), W(L + "attention.v_proj.bias"))
        q, k, v = (t.view(B, S, H, dh).transpose(1, 2) for t in (q, k, v))
        o = _attend(q, k, v, None, 1.0).transpose(1, 2).reshape(B, S, d)
        x = x + F.linear(o, W(L + "attention.out_proj.weight"), W(L + "attention.out_proj.bias"))
        x = F.layer_norm(x, (d,), W(L + "layer_norm.weight"), W(L + "layer_norm.bias"), a.layer_norm_eps)
        h = F.gelu(F.linear(x, W(L + "feed_forward.intermediate_dense.weight"), W(L + "feed_forward.intermediate_dense.bias")))
        x = x + F.linear(h, W(L + "feed_forward.output_dense.weight"), W(L + "feed_forward.output_dense.bias"))
        x = F.layer_norm(x, (d,), W(L + "final_layer_norm.weight"), W(L + "final_layer_norm.bias"), a.layer_norm_eps)
    return x


# ----------------------------------------------------------------------------------------------
# Projector — StackAudioFrames / RMSNorm / SwiGLU / UltravoxProjector, ultravox_model.py:712-800.
# ----------------------------------------------------------------------------------------------
def rmsnorm_ref(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """[3P] LlamaRMSNorm.forward (subclassed at ultravox_model.py:733-736)."""
    dt = x.dtype
    h = x.to(torch.float32)
    h = h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + eps)
    return w * h.to(dt)


def projector_ref(p: Dict[str, torch.Tensor], cfg, audio_features: torch.Tensor) -> torch.Tensor:
    S = cfg.stack_factor
    B, T, Cc = audio_features.shape
    Tp = (T + S - 1) // S * S                                              # :724-729
    x = F.pad(audio_features, (0, 0, 0, Tp - T)).view(B, Tp // S, Cc * S)
    x = rmsnorm_ref(x, p["ln_pre.weight"], 1e-6)                           # :791
    x = F.linear(x, p["linear_1.weight"])                                  # :793
    val, gate = x.chunk(2, dim=-1)                                         # :739-742 (first half = value)
    x = F.silu(gate) * val
    if cfg.projector_ln_mid:                                               # :761-766
        x = rmsnorm_ref(x, p["ln_mid.weight"], 1e-6)
        return F.linear(x, p["linear_2.weight"])
    return rmsnorm_ref(F.linear(x, p["linear_2.weight"]), p["ln_post.weight"], 1e-6)


# ----------------------------------------------------------------------------------------------
# Merge — _audio_iter + the in-place loop, ultravox_model.py:259-275, :390-394.
# ----------------------------------------------------------------------------------------------
def merge_ref(inputs_embeds, audio_embeds, audio_token_start_idx, audio_token_len, audio_batch_size):
    out = inputs_embeds.clone()
    i_a = 0
    for i_b, cnt in enumerate(audio_batch_size.reshape(-1).tolist()):
        for _ in range(int(cnt)):
            s, n = int(audio_token_start_idx[i_a]), int(audio_token_len[i_a])
            out[i_b][s:s + n] = audio_embeds[i_a][:n]
            i_a += 1
    return out


# ----------------------------------------------------------------------------------------------
# LLM — [3P] LlamaForCausalLM.forward + ForCausalLMLoss, reached at ultravox_model.py:328-334.
# ----------------------------------------------------------------------------------------------
def rope_cos_sin_ref(tc, T: int, dtype) -> tuple:
    dh = tc.head_dim
    inv = 1.0 / (tc.rope_theta ** (torch.arange(0, dh, 2, dtype=torch.int64).float() / dh))
    rs = tc.rope_scaling
    if rs and rs.get("rope_type", rs.get("type")) == "llama3":
        factor, lo, hi, old = rs["factor"], rs["low_freq_factor"], rs["high_freq_factor"], rs["original_max_position_embeddings"]
        wl = 2 * math.pi / inv
        inv_l = torch.where(wl > old / lo, inv / factor, inv)
        smooth = (old / wl - lo) / (hi - lo)
        mid = ~(wl < old / hi) * ~(wl > old / lo)
        inv = torch.where(mid, (1 - smooth) * inv_l / factor + smooth * inv_l, inv_l)
    freqs = torch.arange(T, dtype=torch.float32)[:, None] * inv[None, :]
    emb = torch.cat([freqs, freqs], dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def _rotate_half(x):
    a, b = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-b, a), dim=-1)


def gemma_rmsnorm_ref(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """[3P] GemmaRMSNorm.forward: the whole of x_hat * (1 + w) in f32, one cast at the end
    ("Llama does x.to(float16) * w whilst Gemma is (x * w).to(float16)")."""
    h = x.float()
    h = h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + eps)
    return (h * (1.0 + w.float())).type_as(x)


def llama_ref(sd: Dict[str, torch.Tensor], cfg, inputs_embeds: torch.Tensor,
              attention_mask: Optional[torch.Tensor] = None, prefix: str = "language_model.",
              n_layers: Optional[int] = None, position_ids: Optional[torch.Tensor] = None,
              lora: Optional[dict] = None) -> torch.Tensor:
    """-> logits [B, T, V] in the dtype of inputs_embeds.  text_config.model_type "gemma" (BASELINE config 5; [3P]
    transformers 4.51.3 modeling_gemma.py): inputs_embeds * tensor(sqrt(hidden_size), dtype) INSIDE the model - the pinned
    version scales whatever it is given, text and merged audio rows alike (the installed 5.x moved the
    scale into the embedding module, so the pin in tests feeds pre-scaled embeddings to the installed decoder stack) -,
    GemmaRMSNorm, gelu_pytorch_tanh(gate) * up, head_dim from the config, lm_head = embed_tokens.
    lora = {"scaling": ..}: peft adapters on q_proj / k_proj
    (text_model_lora_config; keys under `language_model.base_model.model.model.layers.N.self_attn.*`), pinned like the
    encoder's by tests/golden/lora_reference.npz (the reference's apply_lora on an HF LlamaForCausalLM via tests/peft_stub.py)."""
    tc = cfg.text_config
    dt = inputs_embeds.dtype
    B, T, D = inputs_embeds.shape
    Hq, Hkv, dh = tc.num_attention_heads, tc.num_key_value_heads, tc.head_dim
    W = lambda k: sd[prefix + k].to(dt) if not sd[prefix + k].requires_grad else sd[prefix + k]
    gemma = getattr(tc, "model_type", "llama") == "gemma"
    norm = gemma_rmsnorm_ref if gemma else rmsnorm_ref
    act = (lambda g: F.gelu(g, approximate="tanh")) if gemma else F.silu
    if position_ids is None:
        cos, sin = rope_cos_sin_ref(tc, T, dt)
    else:  # [B, T] position ids (HF generate derives them from the attention mask) -> [B, 1, T, dh] tables
        cos_t, sin_t = rope_cos_sin_ref(tc, int(position_ids.max()) + 1, dt)
        cos, sin = cos_t[position_ids][:, None], sin_t[position_ids][:, None]
    neg = torch.finfo(dt).min
    causal = torch.full((T, T), neg, dtype=dt).triu(1)[None, None]
    if attention_mask is not None:
        pad = (1.0 - attention_mask[:, None, None, :].to(dt)) * neg
        causal = torch.clamp(causal + pad, min=neg)
    x = inputs_embeds
    if gemma:
        x = x * torch.tensor(tc.hidden_size ** 0.5, dtype=dt)
    L_ = tc.num_hidden_layers if n_layers is None else n_layers
    for i in range(L_):
        P = f"model.layers.{i}."
        h = norm(x, W(P + "input_layernorm.weight"), tc.rms_norm_eps)
        q = F.linear(h, W(P + "self_attn.q_proj.weight"))
        k = F.linear(h, W(P + "self_attn.k_proj.weight"))
        if lora is not None:
            def lo(pj):
                A = sd[f"{prefix}base_model.model.model.layers.{i}.self_attn.{pj}.lora_A.default.weight"].to(dt)
                Bm = sd[f"{prefix}base_model.model.model.layers.{i}.self_attn.{pj}.lora_B.default.weight"].to(dt)
                return F.linear(F.linear(h, A), Bm) * lora["scaling"]
            q = q + lo("q_proj")
            k = k + lo("k_proj")
        q = q.view(B, T, Hq, dh).transpose(1, 2)
        k = k.view(B, T, Hkv, dh).transpose(1, 2)
        v = F.linear(h, W(P + "self_attn.v_proj.weight")).view(B, T, Hkv, dh).transpose(1, 2)
        q = q * cos + _rotate_half(q) * sin
        k = k * cos + _rotate_half(k) * sin
        k = k.repeat_interleave(Hq // Hkv, dim=1)
        v = v.repeat_interleave(Hq // Hkv, dim=1)
        o = _attend(q, k, v, causal, dh ** -0.5).transpose(1, 2).reshape(B, T, Hq * dh)
        x = x + F.linear(o, W(P + "self_attn.o_proj.weight"))
        h = norm(x, W(P + "post_attention_layernorm.weight"), tc.rms_norm_eps)
        h = act(F.linear(h, W(P + "mlp.gate_proj.weight"))) * F.linear(h, W(P + "mlp.up_proj.weight"))
        x = x + F.linear(h, W(P + "mlp.down_proj.weight"))
    x = norm(x, W("model.norm.weight"), tc.rms_norm_eps)
    return F.linear(x, W("lm_head.weight") if prefix + "lm_head.weight" in sd else W("model.embed_tokens.weight"))


def causal_lm_loss_ref(logits: torch.Tensor, labels: torch.Tensor, ignore_index: int = -100) -> torch.Tensor:
    """[3P] loss_utils.ForCausalLMLoss with num_items_in_batch=None (accepts_loss_kwargs = False,
    ultravox_model.py:50-53): upcast, pad one ignore label on the right, shift, mean CE."""
    logits = logits.float()
    shifted = F.pad(labels, (0, 1), value=ignore_index)[..., 1:].contiguous()
    return F.cross_entropy(logits.view(-1, logits.shape[-1]), shifted.view(-1), ignore_index=ignore_index,
                           reduction="mean")


def prediction_mask_ref(labels: torch.Tensor):
    """UltravoxModel._get_prediction_mask (ultravox_model.py:157-198): positions whose NEXT token carries a label,
    and, per sequence, the last such position (the one that predicts the end-of-turn token)."""
    label_mask = labels != -100
    pred = torch.zeros_like(label_mask)
    pred[:, :-1] = label_mask[:, 1:]
    eot = torch.zeros_like(pred)
    for b in range(labels.shape[0]):
        pos = torch.where(pred[b])[0]
        if len(pos) > 0:
            eot[b, pos[-1]] = True
    return pred, eot


def kl_loss_ref(student_logits, labels, teacher_logits, alt_labels, temperature: float = 2.0, eot_loss_weight: float = 1.0):
    """UltravoxModel._compute_kl_loss (ultravox_model.py:223-256): KL(teacher || student) at temperature over the
    prediction positions ("batchmean": divided by the number of rows), plus eot_loss_weight x the same over the
    end-of-turn positions.  The teacher is detached (:212-222 run it under no_grad)."""
    pm, em = prediction_mask_ref(labels)
    apm, aem = prediction_mask_ref(alt_labels)
    t = teacher_logits.detach()
    kl = F.kl_div(F.log_softmax(student_logits[pm] / temperature, dim=-1), F.softmax(t[apm] / temperature, dim=-1),
                  reduction="batchmean")
    if eot_loss_weight > 0:
        kl = kl + eot_loss_weight * F.kl_div(F.log_softmax(student_logits[em] / temperature, dim=-1),
                                             F.softmax(t[aem] / temperature, dim=-1), reduction="batchmean")
    return kl


# ----------------------------------------------------------------------------------------------
# Whole forward / train step — UltravoxModel.forward (:277-352), _prepare_audio_embeds (:354-396).
# ----------------------------------------------------------------------------------------------
class OracleModel:
    def __init__(self, cfg, state_dict: Dict[str, torch.Tensor], dtype=torch.float32):
        self.cfg, self.dtype = cfg, dtype
        self.sd = {k: v.detach().to("cpu", dtype).clone() for k, v in state_dict.items()}
        # apply_lora r = 0 (ultravox_model.py:697-703): towers frozen, projector trainable
        self.trainable = [k for k in self.sd if k.startswith("multi_modal_projector.") or ".lora_" in k]
        for k in self.trainable:
            self.sd[k].requires_grad_(True)
        r = int((getattr(cfg, "audio_model_lora_config", None) or {}).get("r", 0) or 0)
        self.lora = None if r == 0 else {"scaling": float(cfg.audio_model_lora_config.get("lora_alpha", 8)) / r}
        rt = int((getattr(cfg, "text_model_lora_config", None) or {}).get("r", 0) or 0)
        self.text_lora = None if rt == 0 else {"scaling": float(cfg.text_model_lora_config.get("lora_alpha", 8)) / rt}

    def projector_params(self):
        P = "multi_modal_projector."
        return {k[len(P):]: self.sd[k] for k in self.trainable if k.startswith(P)}

    def audio_embeds(self, audio_values, audio_lens):
        if getattr(self.cfg.audio_config, "is_wav2vec2", False):
            with torch.no_grad():                                                          # frozen tower (apply_lora r = 0)
                tower = wav2vec2_encoder_ref(self.sd, self.cfg, audio_values.to(self.dtype))
            return tower, projector_ref(self.projector_params(), self.cfg, tower.to(self.dtype))
        with torch.set_grad_enabled(self.lora is not None and torch.is_grad_enabled()):   # frozen tower unless LoRA-adapted
            tower = whisper_encoder_ref(self.sd, self.cfg, audio_values.to(self.dtype), audio_lens, lora=self.lora)   # :382-385
        return tower, projector_ref(self.projector_params(), self.cfg, tower.to(self.dtype))        # :386-387

    def forward(self, input_ids, audio_values=None, labels=None, attention_mask=None, audio_token_start_idx=None,
                audio_lens=None, audio_token_len=None, audio_batch_size=None, inputs_embeds=None, alt_input_ids=None,
                alt_attention_mask=None, alt_labels=None, kl: Optional[dict] = None):
        """kl = {"temperature": .., "eot_loss_weight": ..} selects LossFunction.KL_Divergence in training mode
        (ultravox_model.py:335-351): the text-only teacher pass over alt_* (:212-222) and _compute_kl_loss."""
        if inputs_embeds is None:
            inputs_embeds = F.embedding(input_ids, self.sd["language_model.model.embed_tokens.weight"])  # :314-316
        audio_embeds = None
        if audio_values is not None and len(audio_values) > 0:
            _, audio_embeds = self.audio_embeds(audio_values, audio_lens)
            inputs_embeds = merge_ref(inputs_embeds, audio_embeds, audio_token_start_idx, audio_token_len, audio_batch_size)
        logits = llama_ref(self.sd, self.cfg, inputs_embeds, attention_mask, lora=self.text_lora)
        loss = causal_lm_loss_ref(logits, labels) if labels is not None else None
        out = {"loss": loss, "logits": logits, "inputs_embeds": inputs_embeds, "audio_embeds": audio_embeds}
        if kl is not None:
            with torch.no_grad():
                alt_embeds = F.embedding(alt_input_ids, self.sd["language_model.model.embed_tokens.weight"])
                out["alt_logits"] = llama_ref(self.sd, self.cfg, alt_embeds, alt_attention_mask)
            out["loss"] = kl_loss_ref(logits, labels, out["alt_logits"], alt_labels, kl["temperature"], kl["eot_loss_weight"])
        return out

    @torch.no_grad()
    def generate_greedy(self, max_new_tokens: int, eos_token_id: int, pad_token_id: Optional[int] = None, **batch):
        """UltravoxModel.generate (ultravox_model.py:398-426) + [3P] HF greedy search, restated WITHOUT a KV cache
        (every step re-runs the whole sequence): merged embeddings once, position ids = cumsum(mask) - 1 (1 where
        masked), next = argmax of the last position, finished sequences emit pad_token_id."""
        ids = batch["input_ids"]
        mask = batch.get("attention_mask")
        if mask is None:
            mask = torch.ones_like(ids)
        fwd = {k: v for k, v in batch.items() if k not in ("labels",)}
        embeds = self.forward(**{**fwd, "attention_mask": mask})["inputs_embeds"]
        table = self.sd["language_model.model.embed_tokens.weight"]
        pad = eos_token_id if pad_token_id is None else pad_token_id
        out, unfinished = [ids], torch.ones(ids.shape[0], dtype=torch.bool)
        for _ in range(max_new_tokens):
            pos = (mask.long().cumsum(-1) - 1).masked_fill(mask == 0, 1)
            logits = llama_ref(self.sd, self.cfg, embeds, mask, position_ids=pos)
            nxt = logits[:, -1].float().argmax(-1)
            tok = torch.where(unfinished, nxt, torch.full_like(nxt, pad))
            out.append(tok[:, None])
            unfinished = unfinished & (tok != eos_token_id)
            if not bool(unfinished.any()):
                break
            embeds = torch.cat([embeds, F.embedding(tok, table)[:, None].to(embeds.dtype)], 1)
            mask = torch.cat([mask, torch.ones_like(mask[:, :1])], 1)
        return torch.cat(out, 1)

    def train_step(self, batch, optimizer=None, max_grad_norm: float = 1.0):
        """loss.backward(); clip_grad_norm_(1.0); AdamW.step() — SURVEY.md Appendix B."""
        params = [self.sd[k] for k in self.trainable]
        for p in params:
            p.grad = None
        out = self.forward(**batch)
        out["loss"].backward()
        grads = {k: self.sd[k].grad.detach().clone() for k in self.trainable}
        gn = None
        if optimizer is not None:
            gn = torch.nn.utils.clip_grad_norm_(params, max_grad_norm)
            optimizer.step()
        return out, grads, gn


def synthetic_batch(cfg, B: int, seconds: float, n_text: int = 128, audio_start: int = 16, n_supervised: int = 32,
                    rank: int = 0, n_mels: Optional[int] = None):
    """The synthetic inputs of SURVEY.md §8d: PCM 0.1*N(0,1) clipped to [-1,1] (seed 1234 + rank), token ids
    uniform in [0, V-2] (seed 4321 + rank), audio inserted at `audio_start`, last `n_supervised` tokens
    supervised (LAST_ASSISTANT masking, ultravox_data_proc.py:106-110)."""
    g = torch.Generator().manual_seed(1234 + rank)
    L = int(round(seconds * 16000)) // HOP * HOP
    pcm = (0.1 * torch.randn(B, L, generator=g)).clamp_(-1, 1)
    Fm = L // HOP
    Na = -(-Fm // (2 * cfg.stack_factor))
    if getattr(cfg.audio_config, "is_wav2vec2", False):     # raw-waveform tower: audio_lens = encoder frames, no 2x factor
        Fm = cfg.audio_config.feat_extract_output_length(L)
        Na = -(-Fm // cfg.stack_factor)
    g2 = torch.Generator().manual_seed(4321 + rank)
    V = cfg.text_config.vocab_size
    text = torch.randint(0, V - 1, (B, n_text), generator=g2)
    eos = cfg.text_config.eos_token_id
    ids = torch.cat([text[:, :audio_start], torch.full((B, Na), eos, dtype=torch.long), text[:, audio_start:]], 1)
    T = ids.shape[1]
    labels = ids.clone()
    labels[:, : T - n_supervised] = -100
    return {
        "pcm": pcm,
        "input_ids": ids, "attention_mask": torch.ones(B, T, dtype=torch.long), "labels": labels,
        "audio_token_start_idx": torch.full((B,), audio_start, dtype=torch.long),
        "audio_lens": torch.full((B,), Fm, dtype=torch.long),
        "audio_token_len": torch.full((B,), Na, dtype=torch.int32),
        "audio_batch_size": torch.ones(B, dtype=torch.long),
    }
