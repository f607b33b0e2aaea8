"""State-dict plumbing: HF/Ultravox checkpoint key names -> packed device tensors.

`random_state_dict` builds a seeded random-init state dict at the true architecture shapes with the
reference's key names (no pretrained weights are reachable offline); `pack_*` fuse / reorder /
transpose them ONCE at load time into the layouts the HIP kernels want (include/uvx.h):
  - encoder  q/k/v -> one [3d, d] weight, q pre-scaled by head_dim^-0.5 (what WhisperAttention does
    to q_proj's output; exact for power-of-two scales), k bias = 0;
  - conv weights in im2col order (tap-major, channel-minor);
  - Llama q/k/v -> [(H+2Hkv)dh, D], gate/up -> [2I, D] (16-row gate / up blocks interleaved), plus a transposed copy of every frozen linear
    for the activation-gradient GEMMs (288 GB of HBM makes the second copy free).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, List

import torch

from .config import UltravoxConfig


def _sinusoids(length: int, channels: int) -> torch.Tensor:
    """Whisper's fixed positional table (what embed_positions.weight holds in released checkpoints)."""
    log_timescale_increment = math.log(10000) / (channels // 2 - 1)
    inv = torch.exp(-log_timescale_increment * torch.arange(channels // 2, dtype=torch.float32))
    t = torch.arange(length, dtype=torch.float32)[:, None] * inv[None, :]
    return torch.cat([t.sin(), t.cos()], dim=1)


def random_state_dict(cfg: UltravoxConfig, seed: int = 0, dtype=torch.float32, device="cpu",
                      std: float = 0.02) -> Dict[str, torch.Tensor]:
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    a, t = cfg.audio_config, cfg.text_config
    sd: Dict[str, torch.Tensor] = {}

    def rn(*shape, s=std):
        return (torch.randn(*shape, generator=g, device=device, dtype=torch.float32) * s).to(dtype)

    def near_one(n):
        return (1.0 + 0.1 * torch.randn(n, generator=g, device=device, dtype=torch.float32)).to(dtype)

    d = a.d_model
    P = "audio_tower."
    if getattr(a, "is_wav2vec2", False):
        _random_wav2vec2(sd, a, rn, near_one, P)
    else:
        _random_whisper(sd, a, rn, near_one, P, device, dtype)
    _random_rest(sd, cfg, rn, near_one, device, dtype)
    return sd


def _random_wav2vec2(sd, a, rn, near_one, P):
    """HF Wav2Vec2Model key names (wav2vec2-large-960h family: GroupNorm after the first conv, bias-free convs, post-LN)."""
    d, Cc = a.d_model, a.conv_dim[0]
    cin = 1
    layer_norm = a.feat_extract_norm == "layer"          # LayerNorm after every conv layer (else: GroupNorm after the first)
    for i, k in enumerate(a.conv_kernel):
        sd[P + f"feature_extractor.conv_layers.{i}.conv.weight"] = rn(Cc, cin, k, s=1.6 / math.sqrt(cin * k))
        if a.conv_bias:
            sd[P + f"feature_extractor.conv_layers.{i}.conv.bias"] = rn(Cc, s=0.1)
        if layer_norm or i == 0:
            sd[P + f"feature_extractor.conv_layers.{i}.layer_norm.weight"] = near_one(Cc)
            sd[P + f"feature_extractor.conv_layers.{i}.layer_norm.bias"] = rn(Cc)
        cin = Cc
    sd[P + "feature_projection.layer_norm.weight"] = near_one(Cc)
    sd[P + "feature_projection.layer_norm.bias"] = rn(Cc)
    sd[P + "feature_projection.projection.weight"] = rn(d, Cc, s=1.0 / math.sqrt(Cc))
    sd[P + "feature_projection.projection.bias"] = rn(d)
    K, G = a.num_conv_pos_embeddings, a.num_conv_pos_embedding_groups
    sd[P + "encoder.pos_conv_embed.conv.parametrizations.weight.original1"] = rn(d, d // G, K, s=1.0 / math.sqrt(K * d // G))
    sd[P + "encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = (
        sd[P + "encoder.pos_conv_embed.conv.parametrizations.weight.original1"].float().norm(dim=(0, 1), keepdim=True) * 1.1).to(
        sd[P + "encoder.pos_conv_embed.conv.parametrizations.weight.original1"].dtype)
    sd[P + "encoder.pos_conv_embed.conv.bias"] = rn(d)
    sd[P + "encoder.layer_norm.weight"] = near_one(d)
    sd[P + "encoder.layer_norm.bias"] = rn(d)
    for i in range(a.encoder_layers):
        L = f"{P}encoder.layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[L + f"attention.{nm}.weight"] = rn(d, d, s=1.0 / math.sqrt(d))
            sd[L + f"attention.{nm}.bias"] = rn(d)
        sd[L + "layer_norm.weight"] = near_one(d)
        sd[L + "layer_norm.bias"] = rn(d)
        sd[L + "feed_forward.intermediate_dense.weight"] = rn(a.encoder_ffn_dim, d, s=1.0 / math.sqrt(d))
        sd[L + "feed_forward.intermediate_dense.bias"] = rn(a.encoder_ffn_dim)
        sd[L + "feed_forward.output_dense.weight"] = rn(d, a.encoder_ffn_dim, s=1.0 / math.sqrt(a.encoder_ffn_dim))
        sd[L + "feed_forward.output_dense.bias"] = rn(d)
        sd[L + "final_layer_norm.weight"] = near_one(d)
        sd[L + "final_layer_norm.bias"] = rn(d)
    # Wav2Vec2Model.masked_spec_embed (SpecAugment's fill vector; a parameter of every released checkpoint, read by no kernel here) - from a generator of
    # its own so that the tensors above keep their values
    sd[P + "masked_spec_embed"] = torch.rand(d, generator=torch.Generator().manual_seed(977)).to(device=sd[P + "encoder.layer_norm.bias"].device,
                                                                                                  dtype=sd[P + "encoder.layer_norm.bias"].dtype)


def _random_whisper(sd, a, rn, near_one, P, device, dtype):
    d = a.d_model
    sd[P + "conv1.weight"] = rn(d, a.num_mel_bins, 3, s=1.0 / math.sqrt(3 * a.num_mel_bins))
    sd[P + "conv1.bias"] = rn(d)
    sd[P + "conv2.weight"] = rn(d, d, 3, s=1.0 / math.sqrt(3 * d))
    sd[P + "conv2.bias"] = rn(d)
    sd[P + "embed_positions.weight"] = _sinusoids(a.max_source_positions, d).to(device=device, dtype=dtype)
    for i in range(a.encoder_layers):
        L = f"{P}layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[L + f"self_attn.{nm}.weight"] = rn(d, d, s=1.0 / math.sqrt(d))
            if nm != "k_proj":
                sd[L + f"self_attn.{nm}.bias"] = rn(d)
        sd[L + "self_attn_layer_norm.weight"] = near_one(d)
        sd[L + "self_attn_layer_norm.bias"] = rn(d)
        sd[L + "fc1.weight"] = rn(a.encoder_ffn_dim, d, s=1.0 / math.sqrt(d))
        sd[L + "fc1.bias"] = rn(a.encoder_ffn_dim)
        sd[L + "fc2.weight"] = rn(d, a.encoder_ffn_dim, s=1.0 / math.sqrt(a.encoder_ffn_dim))
        sd[L + "fc2.bias"] = rn(d)
        sd[L + "final_layer_norm.weight"] = near_one(d)
        sd[L + "final_layer_norm.bias"] = rn(d)
    sd[P + "layer_norm.weight"] = near_one(d)
    sd[P + "layer_norm.bias"] = rn(d)


def _random_rest(sd, cfg, rn, near_one, device, dtype):
    a, t = cfg.audio_config, cfg.text_config
    d = a.d_model
    # projector (ultravox_model.py:745-766): RMSNorm weights start at norm_init (0.4)
    P = "multi_modal_projector."
    dim_in, H, D = d * cfg.stack_factor, cfg.hidden_size, t.hidden_size
    sd[P + "ln_pre.weight"] = torch.full((dim_in,), cfg.norm_init, device=device, dtype=dtype)
    sd[P + "linear_1.weight"] = rn(H, dim_in, s=1.0 / math.sqrt(dim_in))
    mid = cfg.projector_mid_dim                  # H // 2 behind SwiGLU, H behind a plain activation (ultravox_model.py:753-755)
    sd[P + "linear_2.weight"] = rn(D, mid, s=1.0 / math.sqrt(mid))
    if cfg.projector_ln_mid:
        sd[P + "ln_mid.weight"] = torch.full((mid,), cfg.norm_init, device=device, dtype=dtype)
    else:
        sd[P + "ln_post.weight"] = torch.full((D,), cfg.norm_init, device=device, dtype=dtype)

    P = "language_model.model."
    dh, Hq, Hkv, I = t.head_dim, t.num_attention_heads, t.num_key_value_heads, t.intermediate_size
    sd[P + "embed_tokens.weight"] = rn(t.vocab_size, D, s=1.0)
    for i in range(t.num_hidden_layers):
        L = f"{P}layers.{i}."
        sd[L + "input_layernorm.weight"] = near_one(D)
        sd[L + "post_attention_layernorm.weight"] = near_one(D)
        sd[L + "self_attn.q_proj.weight"] = rn(Hq * dh, D, s=1.0 / math.sqrt(D))
        sd[L + "self_attn.k_proj.weight"] = rn(Hkv * dh, D, s=1.0 / math.sqrt(D))
        sd[L + "self_attn.v_proj.weight"] = rn(Hkv * dh, D, s=1.0 / math.sqrt(D))
        sd[L + "self_attn.o_proj.weight"] = rn(D, Hq * dh, s=1.0 / math.sqrt(Hq * dh))
        sd[L + "mlp.gate_proj.weight"] = rn(I, D, s=1.0 / math.sqrt(D))
        sd[L + "mlp.up_proj.weight"] = rn(I, D, s=1.0 / math.sqrt(D))
        sd[L + "mlp.down_proj.weight"] = rn(D, I, s=1.0 / math.sqrt(I))
        if getattr(t, "has_qkv_bias", False):      # Qwen2Attention: biases on q / k / v only
            for n, rows in (("q", Hq * dh), ("k", Hkv * dh), ("v", Hkv * dh)):
                sd[L + f"self_attn.{n}_proj.bias"] = rn(rows, s=0.5)
        if getattr(t, "has_qk_norm", False):       # Qwen3Attention / Gemma3Attention q_norm / k_norm: RMSNorm weights over head_dim
            sd[L + "self_attn.q_norm.weight"] = near_one(dh)
            sd[L + "self_attn.k_norm.weight"] = near_one(dh)
        if getattr(t, "is_gemma3", False):         # Gemma3DecoderLayer: four norms (HF names; post_attention_layernorm is a POST norm here)
            sd[L + "pre_feedforward_layernorm.weight"] = near_one(D)
            sd[L + "post_feedforward_layernorm.weight"] = near_one(D)
    sd[P + "norm.weight"] = near_one(D)
    if getattr(t, "is_gemma", False) or getattr(t, "is_gemma3", False):
        # Gemma: norm weights are stored zero-centred (the norm multiplies by 1 + w), the embedding is divided by the
        # sqrt(hidden) the model multiplies back in, and the head is the embedding matrix (tied: no lm_head key)
        for k in [k for k in sd if k.startswith(P) and k.endswith(("layernorm.weight", "model.norm.weight", "q_norm.weight", "k_norm.weight"))]:
            sd[k] = (sd[k].float() - 1.0).to(dtype)
        sd[P + "embed_tokens.weight"] = (sd[P + "embed_tokens.weight"].float() / math.sqrt(D)).to(dtype)
    elif not getattr(t, "ties_head", False):
        sd["language_model.lm_head.weight"] = rn(t.vocab_size, D, s=1.0 / math.sqrt(D))
    return sd


def rope_inv_freq(tc, local: bool = False) -> torch.Tensor:
    """[3P] LlamaRotaryEmbedding: default, "llama3" (Llama-3.1/3.3) and "linear" (Gemma-3's global layers) rope_scaling.
    local: the table of Gemma-3's sliding-window layers (rope_local_base_freq, never scaled)."""
    dh = tc.head_dim
    inv = 1.0 / ((tc.rope_local_base_freq if local else tc.rope_theta) ** (torch.arange(0, dh, 2, dtype=torch.int64).float() / dh))
    rs = None if local else tc.rope_scaling
    if rs and rs.get("rope_type", rs.get("type")) == "linear":
        inv = inv / rs["factor"]
    elif rs and rs.get("rope_type", rs.get("type")) == "llama3":
        factor, lo, hi = rs["factor"], rs["low_freq_factor"], rs["high_freq_factor"]
        old = rs["original_max_position_embeddings"]
        wavelen = 2 * math.pi / inv
        inv_l = torch.where(wavelen > old / lo, inv / factor, inv)
        smooth = (old / wavelen - lo) / (hi - lo)
        smoothed = (1 - smooth) * inv_l / factor + smooth * inv_l
        mid = ~(wavelen < old / hi) * ~(wavelen > old / lo)
        inv = torch.where(mid, smoothed, inv_l)
    elif rs and rs.get("rope_type", rs.get("type", "default")) not in ("default", None):
        raise ValueError(f"rope_scaling {rs} is not supported")
    return inv


def rope_table(tc, length: int, device, local: bool = False) -> torch.Tensor:
    inv = rope_inv_freq(tc, local)
    freqs = torch.arange(length, dtype=torch.float32)[:, None] * inv[None, :]
    return torch.stack([freqs.cos(), freqs.sin()], dim=-1).contiguous().to(device)  # [len, dh/2, 2]


LORA_TARGETS = ("q_proj", "k_proj")     # of the encoder's self_attn (the default target_modules that exist in Whisper)
LORA_FIELD = {"q_proj": "q", "k_proj": "k", "v_proj": "v", "out_proj": "o", "o_proj": "o",      # module -> field of uvx_enc_lora_layer_t
              "fc1": "g", "fc2": "d", "gate_proj": "g", "up_proj": "u", "down_proj": "d"}


def lora_targets(cfg: UltravoxConfig, tower: str):
    """The adapted attention projections of `tower` ("audio" / "text") under cfg's LoRA config, () when its rank is 0
    (config.lora_target_modules: the reference's target_modules resolved against the tower's module names)."""
    from .config import lora_target_modules
    lc = getattr(cfg, f"{tower}_model_lora_config", None) or {}
    kind = "audio_w2v" if tower == "audio" and getattr(cfg.audio_config, "is_wav2vec2", False) else tower
    return lora_target_modules(lc, kind) if int(lc.get("r", 0) or 0) > 0 else ()


def lora_dims(cfg: UltravoxConfig, tower: str, proj: str):
    """(in_features, out_features) of an adapted projection."""
    a, t = cfg.audio_config, cfg.text_config
    if tower == "audio":
        return {"fc1": (a.d_model, a.encoder_ffn_dim), "fc2": (a.encoder_ffn_dim, a.d_model)}.get(proj, (a.d_model, a.d_model))
    qc, kc, D, I = t.num_attention_heads * t.head_dim, t.num_key_value_heads * t.head_dim, t.hidden_size, t.intermediate_size
    return {"q_proj": (D, qc), "k_proj": (D, kc), "v_proj": (D, kc), "o_proj": (qc, D), "gate_proj": (D, I), "up_proj": (D, I), "down_proj": (I, D)}[proj]


def lora_key(layer: int, proj: str, which: str, prefix="audio_tower.") -> str:
    """peft's parameter name for a LoRA matrix of the wrapped encoder (get_peft_model, ultravox_model.py:707): the base
    model sits under `base_model.model.` and the adapter is called `default`."""
    sub = "" if proj in ("fc1", "fc2") else "self_attn."      # WhisperEncoderLayer: fc1 / fc2 sit next to self_attn
    return f"{prefix}base_model.model.layers.{layer}.{sub}{proj}.lora_{which}.default.weight"


def w2v_lora_key(layer: int, proj: str, which: str, prefix="audio_tower.") -> str:
    """... of the wrapped HF Wav2Vec2Model (the AutoModel tower, ultravox_model.py:460-467): its attention lives at encoder.layers.N.attention."""
    return f"{prefix}base_model.model.encoder.layers.{layer}.attention.{proj}.lora_{which}.default.weight"


def audio_lora_key(cfg: UltravoxConfig):
    """The key function of cfg's audio tower: Whisper's `layers.N.self_attn` or wav2vec2's `encoder.layers.N.attention`."""
    return w2v_lora_key if getattr(cfg.audio_config, "is_wav2vec2", False) else lora_key


def llm_lora_key(layer: int, proj: str, which: str, prefix="language_model.") -> str:
    """peft's name for a LoRA matrix of the wrapped LlamaForCausalLM (whose own `model.` level follows peft's)."""
    sub = "mlp." if proj in ("gate_proj", "up_proj", "down_proj") else "self_attn."
    return f"{prefix}base_model.model.model.layers.{layer}.{sub}{proj}.lora_{which}.default.weight"


def init_lora_state_dict(cfg: UltravoxConfig, seed: int = 0, dtype=torch.float32, device="cpu", random_b: bool = False):
    """peft's initialisation: lora_A kaiming-uniform(a = sqrt(5)) = U(-1/sqrt(in), 1/sqrt(in)), lora_B zeros
    (random_b: small random B instead, so that tests see non-zero gradients for A too)."""
    a, t = cfg.audio_config, cfg.text_config
    g = torch.Generator(device=device).manual_seed(seed + 7919)
    out = {}

    def add(keyfn, i, pj, r, d_in, d_out):
        A = (torch.rand(r, d_in, generator=g, device=device) * 2 - 1) / math.sqrt(d_in)
        B = 0.05 * torch.randn(d_out, r, generator=g, device=device) if random_b else torch.zeros(d_out, r, device=device)
        out[keyfn(i, pj, "A")] = A.to(dtype)
        out[keyfn(i, pj, "B")] = B.to(dtype)

    # (a config that went through merge_and_unload no longer carries the LoRA configs - ultravox_model.py:555-557 - and means r = 0)
    for tower, keyfn, nl in (("audio", audio_lora_key(cfg), a.encoder_layers), ("text", llm_lora_key, t.num_hidden_layers)):
        r = int((getattr(cfg, f"{tower}_model_lora_config", None) or {}).get("r", 0) or 0)
        for i in range(nl if r else 0):
            for pj in lora_targets(cfg, tower):      # (q_proj, k_proj) unless target_modules says otherwise
                add(keyfn, i, pj, r, *lora_dims(cfg, tower, pj))
    return out


def pack_wav2vec2(sd, cfg: UltravoxConfig, dtype, device, prefix="audio_tower.", with_transposes: bool = False) -> Dict[str, object]:
    """HF Wav2Vec2Model weights -> the operands of csrc/wav2vec2.hip (layouts: include/uvx.h, uvx_w2v_weights_t).  with_transposes: the
    [K_in, N_out] copies of the encoder layers' linears that uvx_wav2vec2_bwd's dgrads read (LoRA training of the tower)."""
    a = cfg.audio_config
    d, H, Cc = a.d_model, a.encoder_attention_heads, a.conv_dim[0]
    scale = (d // H) ** -0.5
    cv = lambda x: x.to(device=device, dtype=dtype).contiguous()
    W = lambda k: sd[prefix + k]
    w0 = W("feature_extractor.conv_layers.0.conv.weight")                       # [C, 1, k0]
    conv0 = torch.zeros(Cc, 64, dtype=w0.dtype, device=w0.device)
    conv0[:, : w0.shape[-1]] = w0[:, 0, :]
    layer_norm = a.feat_extract_norm == "layer"
    n_conv = len(a.conv_kernel)
    out = {"conv0_w": cv(conv0), "conv_w": [None],
           # group-norm family: the first conv layer's GroupNorm; layer-norm family: a LayerNorm per conv layer; biases where conv_bias
           "gn_w": None if layer_norm else cv(W("feature_extractor.conv_layers.0.layer_norm.weight")),
           "gn_b": None if layer_norm else cv(W("feature_extractor.conv_layers.0.layer_norm.bias")),
           "conv_b": [cv(W(f"feature_extractor.conv_layers.{i}.conv.bias")) if a.conv_bias else None for i in range(n_conv)],
           "conv_ln_w": [cv(W(f"feature_extractor.conv_layers.{i}.layer_norm.weight")) if layer_norm else None for i in range(n_conv)],
           "conv_ln_b": [cv(W(f"feature_extractor.conv_layers.{i}.layer_norm.bias")) if layer_norm else None for i in range(n_conv)]}
    if not a.conv_bias and any(prefix + f"feature_extractor.conv_layers.{i}.conv.bias" in sd for i in range(n_conv)):
        raise ValueError("the checkpoint carries conv biases but audio_config.conv_bias is False")
    for i in range(1, n_conv):
        w = W(f"feature_extractor.conv_layers.{i}.conv.weight")                  # [C, C, k] -> [C, k*C], column k*C + c
        out["conv_w"].append(cv(w.permute(0, 2, 1).reshape(Cc, -1)))
    out["fp_ln_w"], out["fp_ln_b"] = cv(W("feature_projection.layer_norm.weight")), cv(W("feature_projection.layer_norm.bias"))
    out["fp_w"], out["fp_b"] = cv(W("feature_projection.projection.weight")), cv(W("feature_projection.projection.bias"))
    # positional conv: weight norm (dim = 2) folded in the SOURCE dtype, as torch's parametrization evaluates it
    P = prefix + "encoder.pos_conv_embed.conv."
    if P + "parametrizations.weight.original0" in sd:
        g, v = sd[P + "parametrizations.weight.original0"], sd[P + "parametrizations.weight.original1"]
    elif P + "weight_g" in sd:
        g, v = sd[P + "weight_g"], sd[P + "weight_v"]
    else:
        g, v = None, sd[P + "weight"]
    # (the parametrisation's own tensors - and masked_spec_embed, a Wav2Vec2Model parameter no kernel reads - are kept on the host as they came, for
    #  unpack_wav2vec2: the fold below has no exact inverse)
    out["host_keep"] = {k[len(prefix):]: sd[k].detach().to("cpu").clone() for k in sd
                        if k.startswith(P) and not k.endswith("conv.bias") or k == prefix + "masked_spec_embed"}
    wpos = v if g is None else v * (g / v.float().norm(dim=(0, 1), keepdim=True).to(v.dtype))      # [d, d/G, K]
    G, K = a.num_conv_pos_embedding_groups, a.num_conv_pos_embeddings
    dg = d // G
    out["pos_w"] = cv(wpos.to(dtype).view(G, dg, dg, K).permute(0, 1, 3, 2).reshape(G, dg, K * dg))   # [g][o][k*dg + c]
    out["pos_b"] = cv(sd[P + "bias"])
    out["ln_w"], out["ln_b"] = cv(W("encoder.layer_norm.weight")), cv(W("encoder.layer_norm.bias"))
    out["layers"] = []
    for i in range(a.encoder_layers):
        L = f"encoder.layers.{i}."
        wqkv = torch.cat([W(L + "attention.q_proj.weight") * scale, W(L + "attention.k_proj.weight"), W(L + "attention.v_proj.weight")], 0)
        bqkv = torch.cat([W(L + "attention.q_proj.bias") * scale, W(L + "attention.k_proj.bias"), W(L + "attention.v_proj.bias")], 0)
        out["layers"].append({
            "ln1_w": cv(W(L + "layer_norm.weight")), "ln1_b": cv(W(L + "layer_norm.bias")),
            "wqkv": cv(wqkv), "bqkv": cv(bqkv),
            "wo": cv(W(L + "attention.out_proj.weight")), "bo": cv(W(L + "attention.out_proj.bias")),
            "ln2_w": cv(W(L + "final_layer_norm.weight")), "ln2_b": cv(W(L + "final_layer_norm.bias")),
            "fc1_w": cv(W(L + "feed_forward.intermediate_dense.weight")), "fc1_b": cv(W(L + "feed_forward.intermediate_dense.bias")),
            "fc2_w": cv(W(L + "feed_forward.output_dense.weight")), "fc2_b": cv(W(L + "feed_forward.output_dense.bias")),
        })
        lay = out["layers"][-1]
        for n, src in (("wqkv_t", "wqkv"), ("wo_t", "wo"), ("fc1_t", "fc1_w"), ("fc2_t", "fc2_w")):
            lay[n] = lay[src].t().contiguous() if with_transposes else None
    return out


def pack_encoder(sd, cfg: UltravoxConfig, dtype, device, prefix="audio_tower.", with_transposes: bool = False) -> Dict[str, object]:
    if prefix + "conv1.weight" not in sd and prefix + "base_model.model.conv1.weight" in sd:
        prefix = prefix + "base_model.model."          # a peft-wrapped tower (LoRA checkpoints)
    a = cfg.audio_config
    d, H = a.d_model, a.encoder_attention_heads
    scale = (d // H) ** -0.5
    cv = lambda x: x.to(device=device, dtype=dtype).contiguous()
    kp1 = (3 * a.num_mel_bins + 63) // 64 * 64
    w1 = sd[prefix + "conv1.weight"].float().permute(0, 2, 1).reshape(d, 3 * a.num_mel_bins)
    w1p = torch.zeros(d, kp1)
    w1p[:, : 3 * a.num_mel_bins] = w1.cpu()
    out = {
        "conv1_w": cv(w1p), "conv1_b": cv(sd[prefix + "conv1.bias"]),
        "conv2_w": cv(sd[prefix + "conv2.weight"].permute(0, 2, 1).reshape(d, 3 * d)),
        "conv2_b": cv(sd[prefix + "conv2.bias"]),
        "pos": cv(sd[prefix + "embed_positions.weight"]),
        "lnf_w": cv(sd[prefix + "layer_norm.weight"]), "lnf_b": cv(sd[prefix + "layer_norm.bias"]),
        "layers": [],
    }
    for i in range(a.encoder_layers):
        L = f"{prefix}layers.{i}."
        q, k, v = (sd[L + f"self_attn.{n}_proj.weight"] for n in "qkv")
        bq, bv = sd[L + "self_attn.q_proj.bias"], sd[L + "self_attn.v_proj.bias"]
        # (x W_q^T + b_q) * scale, rounded like the reference: scale is a power of two for dh = 64
        wqkv = torch.cat([q.float() * scale, k.float(), v.float()], 0)
        bqkv = torch.cat([bq.float() * scale, torch.zeros_like(bq.float()), bv.float()], 0)
        out["layers"].append({
            "ln1_w": cv(sd[L + "self_attn_layer_norm.weight"]), "ln1_b": cv(sd[L + "self_attn_layer_norm.bias"]),
            "wqkv": cv(wqkv), "bqkv": cv(bqkv),
            "wo": cv(sd[L + "self_attn.out_proj.weight"]), "bo": cv(sd[L + "self_attn.out_proj.bias"]),
            "ln2_w": cv(sd[L + "final_layer_norm.weight"]), "ln2_b": cv(sd[L + "final_layer_norm.bias"]),
            "fc1_w": cv(sd[L + "fc1.weight"]), "fc1_b": cv(sd[L + "fc1.bias"]),
            "fc2_w": cv(sd[L + "fc2.weight"]), "fc2_b": cv(sd[L + "fc2.bias"]),
        })
        lay = out["layers"][-1]
        for n, src in (("wqkv_t", "wqkv"), ("wo_t", "wo"), ("fc1_t", "fc1_w"), ("fc2_t", "fc2_w")):
            lay[n] = lay[src].t().contiguous() if with_transposes else None   # [K_in, N_out]: dgrad operands (LoRA training)
    return out


def pack_llm(sd, cfg: UltravoxConfig, dtype, device, with_transposes: bool = True, rope_len: Optional[int] = None,
             prefix="language_model.", consume: bool = False) -> Dict[str, object]:
    """HF-named LLM weights -> the packed per-layer operands of csrc/model.hip.  q/k/v and gate/up are concatenated (new
    tensors); the other matrices are shared with `sd` when they already have the right device / dtype.  `consume=True` pops
    the q/k/v/gate/up entries from `sd` layer by layer as soon as they are packed, so the peak stays at one copy of the model
    plus one layer (what a 70B-parameter LLM needs to load into 288 GB next to its KV cache) — `sd` is left without them."""
    t = cfg.text_config
    cv = lambda x: x.to(device=device, dtype=dtype).contiguous()
    # with_transposes: True = a resident W^T per frozen linear + lm_head^T (the NT dgrads), False = none (inference, or the streamed copies),
    # "head" = lm_head^T only: the layers' dgrads then take the NN form on the forward weights (csrc/model.hip lin_dgrad; bf16, round 6)
    tr_head = lambda x: x.t().contiguous() if with_transposes else None
    tr = lambda x: x.t().contiguous() if with_transposes is True else None
    if prefix + "model.embed_tokens.weight" not in sd and prefix + "base_model.model.model.embed_tokens.weight" in sd:
        prefix = prefix + "base_model.model."          # a peft-wrapped LLM (LoRA checkpoints)
    P = prefix + "model."
    out = {"embed": cv(sd[P + "embed_tokens.weight"]), "norm": cv(sd[P + "norm.weight"]), "layers": []}
    # tied head (Gemma; HF tie_word_embeddings): lm_head IS the embedding matrix - one tensor, two uses
    head = sd.get(prefix + "lm_head.weight")
    # (checkpoint.language_model_state_dict hands a tied checkpoint's embedding over under both names: still one device tensor)
    out["lm_head"] = out["embed"] if head is None or head is sd[P + "embed_tokens.weight"] else cv(head)
    if prefix + "lm_head.weight" not in sd and not getattr(t, "ties_head", getattr(t, "is_gemma", False)):
        raise KeyError(f"{prefix}lm_head.weight is missing and the config does not tie the head to embed_tokens "
                       "(tie_word_embeddings; Gemma always ties)")
    out["lm_head_t"] = tr_head(out["lm_head"])
    for i in range(t.num_hidden_layers):
        L = f"{P}layers.{i}."
        wqkv = cv(torch.cat([sd[L + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0))
        # gate / up rows interleaved in 16-row blocks: a GEMM tile then holds matching gate and up columns and
        # the SwiGLU runs in its epilogue (csrc/gemm.hip store_tile)
        gate, up = sd[L + "mlp.gate_proj.weight"], sd[L + "mlp.up_proj.weight"]
        I, Dm = gate.shape
        assert I % 16 == 0, "intermediate_size must be a multiple of 16"
        wgu = cv(torch.stack([gate.reshape(I // 16, 16, Dm), up.reshape(I // 16, 16, Dm)], 1).reshape(2 * I, Dm))
        del gate, up
        if consume:
            for key in [L + f"self_attn.{n}_proj.weight" for n in "qkv"] + [L + "mlp.gate_proj.weight", L + "mlp.up_proj.weight"]:
                sd.pop(key)
        wo, wd = cv(sd[L + "self_attn.o_proj.weight"]), cv(sd[L + "mlp.down_proj.weight"])
        extras = {}      # family extras: present only where the family has them (uvx_llm_layer_t.bqkv / q_norm / k_norm, else NULL)
        if getattr(t, "has_qkv_bias", False):
            extras["bqkv"] = cv(torch.cat([sd[L + f"self_attn.{n}_proj.bias"] for n in "qkv"], 0))
        elif any(L + f"self_attn.{n}_proj.bias" in sd for n in "qkvo"):
            raise ValueError(f"{L}self_attn.*_proj.bias present: attention biases are built for the qwen2 family only")
        if getattr(t, "has_qk_norm", False):
            extras["q_norm"], extras["k_norm"] = cv(sd[L + "self_attn.q_norm.weight"]), cv(sd[L + "self_attn.k_norm.weight"])
        elif L + "self_attn.q_norm.weight" in sd:
            raise ValueError(f"{L}self_attn.q_norm.weight present: per-head q / k norms are built for the qwen3 / gemma3 families only")
        ln2_key = "post_attention_layernorm.weight"
        if getattr(t, "is_gemma3", False):      # four norms: ln1 | ln1_post (HF post_attention_layernorm) | ln2 (pre_feedforward) | ln2_post
            extras["ln1_post"] = cv(sd[L + "post_attention_layernorm.weight"])
            extras["ln2_post"] = cv(sd[L + "post_feedforward_layernorm.weight"])
            ln2_key = "pre_feedforward_layernorm.weight"
        out["layers"].append({
            **extras,
            "ln1": cv(sd[L + "input_layernorm.weight"]), "ln2": cv(sd[L + ln2_key]),
            "wqkv": wqkv, "wo": wo, "wgu": wgu, "wd": wd,
            "wqkv_t": tr(wqkv), "wo_t": tr(wo), "wgu_t": tr(wgu), "wd_t": tr(wd),
        })
    out["rope_len"] = rope_len or min(t.max_position_embeddings, 8192)
    out["rope"] = rope_table(t, out["rope_len"], device)
    if getattr(t, "is_gemma3", False):          # second rotary table for the sliding-window layers
        out["rope_local"] = rope_table(t, out["rope_len"], device, local=True)
    if getattr(t, "window_layers", None):       # per-layer flags: Gemma-3's local layers, every layer of a windowed Mistral
        out["layer_local"] = list(t.window_layers)
    return out


# ----------------------------------------------------------------------------------------------------------------------
# Packed device operands -> the reference's checkpoint names again.  What UltravoxModel.merge_and_unload needs to re-export a
# tower whose LoRA adapters were folded into the packed weights (ultravox_model.py:528-559: after the merge every tower
# parameter joins keep_params and the next save_pretrained writes the towers whole), and what lets ANY keep_param be re-saved
# from the tensors the kernels actually run on.  Exact inverses of pack_encoder / pack_llm: a split, a de-interleave, a
# power-of-two scale - no arithmetic that rounds.

def check_encoder_exportable(cfg: UltravoxConfig) -> None:
    """unpack_encoder divides the packed q_proj by head_dim^-0.5: exact only when that scale is a power of two (head_dim a power of 4;
    64 for every released Whisper).  Raises the ValueError unpack_encoder would - callers that are about to MUTATE the packed weights
    (merge_and_unload) ask first."""
    a = cfg.audio_config
    dh = a.d_model // a.encoder_attention_heads
    if dh & (dh - 1) or (dh.bit_length() - 1) % 2:
        raise ValueError(f"encoder head_dim {dh}: head_dim^-0.5 is not a power of two, the packed q_proj cannot be unscaled exactly")


def wav2vec2_param_names(enc: Dict[str, object], cfg: UltravoxConfig, prefix: str = "audio_tower.") -> List[str]:
    """The keys unpack_wav2vec2 produces (HF Wav2Vec2Model named_parameters(), checkpoint names), without touching a tensor."""
    a = cfg.audio_config
    layer_norm = a.feat_extract_norm == "layer"
    out = [prefix + "masked_spec_embed"] if "masked_spec_embed" in enc["host_keep"] else []
    for i in range(len(a.conv_kernel)):
        C = f"{prefix}feature_extractor.conv_layers.{i}."
        out.append(C + "conv.weight")
        if a.conv_bias:
            out.append(C + "conv.bias")
        if layer_norm or i == 0:
            out += [C + "layer_norm.weight", C + "layer_norm.bias"]
    out += [prefix + "feature_projection.layer_norm.weight", prefix + "feature_projection.layer_norm.bias",
            prefix + "feature_projection.projection.weight", prefix + "feature_projection.projection.bias"]
    out += [prefix + k for k in enc["host_keep"] if k != "masked_spec_embed"] + [prefix + "encoder.pos_conv_embed.conv.bias"]
    out += [prefix + "encoder.layer_norm.weight", prefix + "encoder.layer_norm.bias"]
    for i in range(a.encoder_layers):
        L = f"{prefix}encoder.layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            out += [L + f"attention.{nm}.weight", L + f"attention.{nm}.bias"]
        out += [L + "layer_norm.weight", L + "layer_norm.bias", L + "feed_forward.intermediate_dense.weight", L + "feed_forward.intermediate_dense.bias",
                L + "feed_forward.output_dense.weight", L + "feed_forward.output_dense.bias", L + "final_layer_norm.weight", L + "final_layer_norm.bias"]
    return out


def unpack_wav2vec2(enc: Dict[str, object], cfg: UltravoxConfig, prefix: str = "audio_tower.", device="cpu") -> Dict[str, torch.Tensor]:
    """pack_wav2vec2's operands -> HF Wav2Vec2Model names under `prefix` (what re-exports a LoRA-merged tower, ultravox_model.py:528-559): the im2col
    conv weights back to [C_out, C_in, k], q_proj un-scaled (head_dim^-0.5 folded into its packed rows: exact for a power-of-two scale only -
    check_encoder_exportable), the positional conv's weight-norm tensors as the checkpoint had them (kept on the host by pack_wav2vec2: the fold has no
    exact inverse and no adapter touches them)."""
    a = cfg.audio_config
    d, H, Cc = a.d_model, a.encoder_attention_heads, a.conv_dim[0]
    check_encoder_exportable(cfg)
    inv = float(d // H) ** 0.5
    cv = lambda x: x.detach().to(device).contiguous()
    layer_norm = a.feat_extract_norm == "layer"
    out: Dict[str, torch.Tensor] = {}
    if "masked_spec_embed" in enc["host_keep"]:
        out[prefix + "masked_spec_embed"] = cv(enc["host_keep"]["masked_spec_embed"])
    for i, k in enumerate(a.conv_kernel):
        C = f"{prefix}feature_extractor.conv_layers.{i}."
        out[C + "conv.weight"] = cv(enc["conv0_w"][:, :k].reshape(Cc, 1, k)) if i == 0 else cv(enc["conv_w"][i].reshape(Cc, k, Cc).permute(0, 2, 1))
        if a.conv_bias:
            out[C + "conv.bias"] = cv(enc["conv_b"][i])
        if layer_norm:
            out[C + "layer_norm.weight"], out[C + "layer_norm.bias"] = cv(enc["conv_ln_w"][i]), cv(enc["conv_ln_b"][i])
        elif i == 0:
            out[C + "layer_norm.weight"], out[C + "layer_norm.bias"] = cv(enc["gn_w"]), cv(enc["gn_b"])
    out[prefix + "feature_projection.layer_norm.weight"], out[prefix + "feature_projection.layer_norm.bias"] = cv(enc["fp_ln_w"]), cv(enc["fp_ln_b"])
    out[prefix + "feature_projection.projection.weight"], out[prefix + "feature_projection.projection.bias"] = cv(enc["fp_w"]), cv(enc["fp_b"])
    for k, v in enc["host_keep"].items():
        if k != "masked_spec_embed":
            out[prefix + k] = cv(v)
    out[prefix + "encoder.pos_conv_embed.conv.bias"] = cv(enc["pos_b"])
    out[prefix + "encoder.layer_norm.weight"], out[prefix + "encoder.layer_norm.bias"] = cv(enc["ln_w"]), cv(enc["ln_b"])
    for i, lay in enumerate(enc["layers"]):
        L = f"{prefix}encoder.layers.{i}."
        w, b = lay["wqkv"], lay["bqkv"]
        for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
            sc = inv if j == 0 else 1.0
            out[L + f"attention.{nm}.weight"], out[L + f"attention.{nm}.bias"] = cv(w[j * d:(j + 1) * d] * sc), cv(b[j * d:(j + 1) * d] * sc)
        out[L + "attention.out_proj.weight"], out[L + "attention.out_proj.bias"] = cv(lay["wo"]), cv(lay["bo"])
        out[L + "layer_norm.weight"], out[L + "layer_norm.bias"] = cv(lay["ln1_w"]), cv(lay["ln1_b"])
        out[L + "feed_forward.intermediate_dense.weight"], out[L + "feed_forward.intermediate_dense.bias"] = cv(lay["fc1_w"]), cv(lay["fc1_b"])
        out[L + "feed_forward.output_dense.weight"], out[L + "feed_forward.output_dense.bias"] = cv(lay["fc2_w"]), cv(lay["fc2_b"])
        out[L + "final_layer_norm.weight"], out[L + "final_layer_norm.bias"] = cv(lay["ln2_w"]), cv(lay["ln2_b"])
    return out


def encoder_param_names(cfg: UltravoxConfig, prefix: str = "audio_tower.") -> List[str]:
    """The keys unpack_encoder produces (HF WhisperEncoder named_parameters()), without touching a tensor."""
    out = [prefix + n for n in ("conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias", "embed_positions.weight",
                                "layer_norm.weight", "layer_norm.bias")]
    for i in range(cfg.audio_config.encoder_layers):
        L = f"{prefix}layers.{i}."
        out += [L + n for n in ("self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight", "self_attn.q_proj.bias",
                                "self_attn.v_proj.bias", "self_attn.out_proj.weight", "self_attn.out_proj.bias",
                                "self_attn_layer_norm.weight", "self_attn_layer_norm.bias", "final_layer_norm.weight", "final_layer_norm.bias",
                                "fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias")]
    return out


def llm_param_names(llm: Dict[str, object], cfg: UltravoxConfig, prefix: str = "language_model.") -> List[str]:
    """The keys unpack_llm produces, without touching a tensor (which family extras exist is read off the packed layer dicts)."""
    P = prefix + "model."
    out = [P + "embed_tokens.weight", P + "norm.weight"]
    if llm["lm_head"] is not llm["embed"]:
        out.append(prefix + "lm_head.weight")
    g3 = bool(getattr(cfg.text_config, "is_gemma3", False))
    for i, lay in enumerate(llm["layers"]):
        L = f"{P}layers.{i}."
        out += [L + n for n in ("self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight", "self_attn.o_proj.weight",
                                "mlp.gate_proj.weight", "mlp.up_proj.weight", "mlp.down_proj.weight", "input_layernorm.weight",
                                "post_attention_layernorm.weight")]
        if g3:
            out += [L + "pre_feedforward_layernorm.weight", L + "post_feedforward_layernorm.weight"]
        if lay.get("bqkv") is not None:
            out += [L + "self_attn.q_proj.bias", L + "self_attn.k_proj.bias", L + "self_attn.v_proj.bias"]
        if lay.get("q_norm") is not None:
            out += [L + "self_attn.q_norm.weight", L + "self_attn.k_norm.weight"]
    return out


def unpack_encoder(enc: Dict[str, object], cfg: UltravoxConfig, prefix: str = "audio_tower.", device="cpu") -> Dict[str, torch.Tensor]:
    """pack_encoder's operands -> HF WhisperEncoder names under `prefix` (no peft infix: a merged tower is a plain module).
    q_proj's weight and bias were stored pre-multiplied by head_dim^-0.5; Whisper's head_dim is 64 for every released size,
    so the division is exact (a non power-of-two scale raises rather than round a second time).  k_proj has no bias in
    Whisper: the zero block of bqkv is not exported."""
    a = cfg.audio_config
    d, H = a.d_model, a.encoder_attention_heads
    dh = d // H
    check_encoder_exportable(cfg)
    inv = float(dh) ** 0.5
    cv = lambda x: x.detach().to(device).contiguous()
    nm = a.num_mel_bins
    out = {
        prefix + "conv1.weight": cv(enc["conv1_w"][:, : 3 * nm].reshape(d, 3, nm).permute(0, 2, 1)),
        prefix + "conv1.bias": cv(enc["conv1_b"]),
        prefix + "conv2.weight": cv(enc["conv2_w"].reshape(d, 3, d).permute(0, 2, 1)),
        prefix + "conv2.bias": cv(enc["conv2_b"]),
        prefix + "embed_positions.weight": cv(enc["pos"]),
        prefix + "layer_norm.weight": cv(enc["lnf_w"]), prefix + "layer_norm.bias": cv(enc["lnf_b"]),
    }
    for i, lay in enumerate(enc["layers"]):
        L = f"{prefix}layers.{i}."
        w, b = lay["wqkv"], lay["bqkv"]
        out[L + "self_attn.q_proj.weight"] = cv(w[:d] * inv)
        out[L + "self_attn.k_proj.weight"] = cv(w[d:2 * d])
        out[L + "self_attn.v_proj.weight"] = cv(w[2 * d:])
        out[L + "self_attn.q_proj.bias"] = cv(b[:d] * inv)
        out[L + "self_attn.v_proj.bias"] = cv(b[2 * d:])
        out[L + "self_attn.out_proj.weight"], out[L + "self_attn.out_proj.bias"] = cv(lay["wo"]), cv(lay["bo"])
        out[L + "self_attn_layer_norm.weight"], out[L + "self_attn_layer_norm.bias"] = cv(lay["ln1_w"]), cv(lay["ln1_b"])
        out[L + "final_layer_norm.weight"], out[L + "final_layer_norm.bias"] = cv(lay["ln2_w"]), cv(lay["ln2_b"])
        out[L + "fc1.weight"], out[L + "fc1.bias"] = cv(lay["fc1_w"]), cv(lay["fc1_b"])
        out[L + "fc2.weight"], out[L + "fc2.bias"] = cv(lay["fc2_w"]), cv(lay["fc2_b"])
    return out


def unpack_llm(llm: Dict[str, object], cfg: UltravoxConfig, prefix: str = "language_model.", device="cpu") -> Dict[str, torch.Tensor]:
    """pack_llm's operands -> HF causal-LM names under `prefix`: q/k/v split out of wqkv, gate/up de-interleaved (16-row
    blocks), the family extras (Qwen2 biases, Qwen3 / Gemma-3 q_norm / k_norm, Gemma-3's four norms) under their HF names.  A tied
    head (lm_head IS embed_tokens, one tensor) is exported once, as embed_tokens - what named_parameters() lists."""
    t = cfg.text_config
    dh, Hq, Hkv, I = t.head_dim, t.num_attention_heads, t.num_key_value_heads, t.intermediate_size
    qc, kc = Hq * dh, Hkv * dh
    cv = lambda x: x.detach().to(device).contiguous()
    P = prefix + "model."
    out = {P + "embed_tokens.weight": cv(llm["embed"]), P + "norm.weight": cv(llm["norm"])}
    if llm["lm_head"] is not llm["embed"]:
        out[prefix + "lm_head.weight"] = cv(llm["lm_head"])
    g3 = bool(getattr(t, "is_gemma3", False))
    for i, lay in enumerate(llm["layers"]):
        L = f"{P}layers.{i}."
        w = lay["wqkv"]
        out[L + "self_attn.q_proj.weight"], out[L + "self_attn.k_proj.weight"] = cv(w[:qc]), cv(w[qc:qc + kc])
        out[L + "self_attn.v_proj.weight"] = cv(w[qc + kc:])
        out[L + "self_attn.o_proj.weight"] = cv(lay["wo"])
        gu = lay["wgu"].reshape(I // 16, 2, 16, -1)
        out[L + "mlp.gate_proj.weight"], out[L + "mlp.up_proj.weight"] = cv(gu[:, 0].reshape(I, -1)), cv(gu[:, 1].reshape(I, -1))
        out[L + "mlp.down_proj.weight"] = cv(lay["wd"])
        out[L + "input_layernorm.weight"] = cv(lay["ln1"])
        if g3:
            out[L + "post_attention_layernorm.weight"] = cv(lay["ln1_post"])
            out[L + "pre_feedforward_layernorm.weight"] = cv(lay["ln2"])
            out[L + "post_feedforward_layernorm.weight"] = cv(lay["ln2_post"])
        else:
            out[L + "post_attention_layernorm.weight"] = cv(lay["ln2"])
        if lay.get("bqkv") is not None:
            b = lay["bqkv"]
            out[L + "self_attn.q_proj.bias"], out[L + "self_attn.k_proj.bias"], out[L + "self_attn.v_proj.bias"] = (
                cv(b[:qc]), cv(b[qc:qc + kc]), cv(b[qc + kc:]))
        if lay.get("q_norm") is not None:
            out[L + "self_attn.q_norm.weight"], out[L + "self_attn.k_norm.weight"] = cv(lay["q_norm"]), cv(lay["k_norm"])
    return out
