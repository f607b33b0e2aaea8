"""Configuration mirror of ultravox/model/ultravox_config.py:56-203 (UltravoxConfig, LossConfig,
LossFunction, LossMaskType, LoraConfigSimplified) without the HuggingFace base class: plain Python
objects with the same field names, so a reference config dict loads unchanged.

Architecture constants for the model ids the reference recipes name are recorded here because the hub
is unreachable offline.
"""
from __future__ import annotations

import dataclasses
from enum import Enum
from typing import Any, Dict, List, Optional


@dataclasses.dataclass
class LoraConfigSimplified:  # ultravox_config.py:8-23
    r: int = 0
    lora_alpha: float = 8
    target_modules: Optional[List[str]] = dataclasses.field(
        default_factory=lambda: ["k_proj", "q_proj", "linear_k", "linear_q"])
    unfreeze_layers: Optional[List[str]] = None


class LossMaskType(str, Enum):  # ultravox_config.py:26-34
    LAST_ASSISTANT = "last_assistant"
    ALL = "all"
    AFTER_AUDIO = "after_audio"


class LossFunction(str, Enum):  # ultravox_config.py:37-39
    CrossEntropy = "ce"
    KL_Divergence = "kl"


@dataclasses.dataclass
class LossConfig:  # ultravox_config.py:42-53
    loss_function: LossFunction = LossFunction.CrossEntropy
    kl_temperature: float = 2.0
    initial_tokens_to_ignore: int = 0
    eot_loss_weight: float = 1.0

    @property
    def requires_alt_fields(self):
        return self.loss_function == LossFunction.KL_Divergence


@dataclasses.dataclass
class AudioConfig:
    """Field names of transformers.WhisperConfig that the encoder path reads."""
    model_type: str = "whisper"
    d_model: int = 384
    encoder_layers: int = 4
    encoder_attention_heads: int = 6
    encoder_ffn_dim: int = 1536
    num_mel_bins: int = 80
    max_source_positions: int = 1500
    layer_norm_eps: float = 1e-5
    # model_type "wav2vec2" (BASELINE.json config 5, the AutoModel branch of _create_audio_tower, ultravox_model.py:460-467,
    # :476-485): d_model / encoder_layers / encoder_attention_heads / encoder_ffn_dim carry Wav2Vec2Config's hidden_size /
    # num_hidden_layers / num_attention_heads / intermediate_size; the fields below are Wav2Vec2Config's own names.
    # Built: the wav2vec2-base / -large-960h family (GroupNorm on the first conv layer, bias-free convs, post-LN encoder) and - round 5 -
    # the layer-norm family (-large-lv60, -large-960h-lv60-self: feat_extract_norm "layer", conv_bias, do_stable_layer_norm); the
    # three switches are independent, as in Wav2Vec2Config.
    conv_dim: Optional[List[int]] = None
    conv_kernel: Optional[List[int]] = None
    conv_stride: Optional[List[int]] = None
    conv_bias: bool = False
    feat_extract_norm: str = "group"
    do_stable_layer_norm: bool = False
    num_conv_pos_embeddings: int = 128
    num_conv_pos_embedding_groups: int = 16

    def __post_init__(self):
        if self.model_type not in ("whisper", "wav2vec2"):
            raise ValueError(f"audio_config.model_type {self.model_type!r} is not built (whisper, wav2vec2)")
        if self.is_wav2vec2:
            self.conv_dim = list(self.conv_dim or (512,) * 7)
            self.conv_kernel = list(self.conv_kernel or (10, 3, 3, 3, 3, 2, 2))
            self.conv_stride = list(self.conv_stride or (5, 2, 2, 2, 2, 2, 2))
            if not (len(self.conv_dim) == len(self.conv_kernel) == len(self.conv_stride)):
                raise ValueError("audio_config: conv_dim / conv_kernel / conv_stride must have one entry per conv layer")
            if self.feat_extract_norm not in ("group", "layer"):
                raise ValueError(f"audio_config.feat_extract_norm {self.feat_extract_norm!r}: 'group' (wav2vec2-base / -large-960h) and "
                                 "'layer' (the -lv60 family) are what Wav2Vec2Config knows")
            if len(set(self.conv_dim)) != 1 or self.conv_dim[0] % 64 or self.d_model % self.num_conv_pos_embedding_groups:
                raise ValueError("audio_config: conv_dim must be one multiple of 64 for all layers; hidden size divisible by the "
                                 "positional-conv groups")

    @property
    def is_wav2vec2(self) -> bool:
        return self.model_type == "wav2vec2"

    @property
    def hidden_size(self) -> int:  # WhisperConfig.hidden_size aliases d_model (used at ultravox_model.py:750)
        return self.d_model

    def feat_extract_output_length(self, n_samples: int) -> int:
        """[3P] Wav2Vec2Model._get_feat_extract_output_lengths: frames the conv stack produces from n_samples."""
        n = int(n_samples)
        for k, st in zip(self.conv_kernel, self.conv_stride):
            n = (n - k) // st + 1
        return n


@dataclasses.dataclass
class TextConfig:
    """Field names of transformers.LlamaConfig / MistralConfig / GemmaConfig / Qwen2Config / Qwen3Config that the LLM path reads.  model_type
    "llama" (default); "gemma" (BASELINE.json config 5, the alt backbone behind AutoModelForCausalLM, ultravox_model.py:499-526):
    GemmaRMSNorm, GeGLU (hidden_act gelu_pytorch_tanh), sqrt(hidden_size) embedding scale, explicit head_dim, lm_head tied to
    embed_tokens; "qwen3" (the reference's v0.6 recipe, ultravox/training/configs/v0.6_config_qwen3_32b.yaml: text_model
    Qwen/Qwen3-32B): a Llama block with an RMSNorm over head_dim on every q / k head before RoPE, explicit head_dim; "qwen2": a
    Llama block whose q / k / v projections carry biases."""
    model_type: str = "llama"
    hidden_act: Optional[str] = None            # None = the family's own: silu (llama), gelu_pytorch_tanh (gemma)
    # None = the family's [3P] default (transformers LlamaConfig / GemmaConfig), so that a config.json that leaves a field
    # out means here what it means to the reference's AutoConfig (pinned against the imported reference config:
    # tests/golden/config.json)
    hidden_size: Optional[int] = None
    intermediate_size: Optional[int] = None
    num_hidden_layers: Optional[int] = None
    num_attention_heads: Optional[int] = None
    num_key_value_heads: Optional[int] = None
    head_dim: Optional[int] = None
    vocab_size: Optional[int] = None
    rms_norm_eps: Optional[float] = None
    rope_theta: Optional[float] = None          # None = the family's default (llama / gemma / qwen: 1e4, gemma3: 1e6)
    rope_scaling: Optional[Dict[str, Any]] = None
    max_position_embeddings: Optional[int] = None
    initializer_range: float = 0.02
    eos_token_id: Optional[int] = None
    tie_word_embeddings: Optional[bool] = None
    # gemma3 only ([3P] Gemma3TextConfig): attention scale = query_pre_attn_scalar ** -0.5; every layer whose type is
    # "sliding_attention" attends to the last `sliding_window` positions and rotates with rope_local_base_freq (no scaling), the
    # "full_attention" layers (every sliding_window_pattern-th) with rope_theta / rope_scaling (linear)
    query_pre_attn_scalar: Optional[float] = None
    sliding_window: Optional[int] = None
    sliding_window_pattern: Optional[int] = None
    layer_types: Optional[List[str]] = None
    rope_local_base_freq: Optional[float] = None

    _FAMILY_DEFAULTS = {
        "llama": dict(hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32,
                      vocab_size=32000, rms_norm_eps=1e-6, max_position_embeddings=2048, eos_token_id=2,
                      tie_word_embeddings=False, rope_theta=10000.0),
        "gemma": dict(hidden_size=3072, intermediate_size=24576, num_hidden_layers=28, num_attention_heads=16,
                      num_key_value_heads=16, head_dim=256, vocab_size=256000, rms_norm_eps=1e-6,
                      max_position_embeddings=8192, eos_token_id=1, tie_word_embeddings=True, rope_theta=10000.0),
        # [3P] transformers Qwen2Config / Qwen3Config defaults
        "qwen2": dict(hidden_size=4096, intermediate_size=22016, num_hidden_layers=32, num_attention_heads=32,
                      num_key_value_heads=32, vocab_size=151936, rms_norm_eps=1e-6, max_position_embeddings=32768,
                      tie_word_embeddings=False, rope_theta=10000.0),
        "qwen3": dict(hidden_size=4096, intermediate_size=22016, num_hidden_layers=32, num_attention_heads=32,
                      num_key_value_heads=32, head_dim=128, vocab_size=151936, rms_norm_eps=1e-6,
                      max_position_embeddings=32768, tie_word_embeddings=False, rope_theta=10000.0),
        # [3P] Gemma3TextConfig defaults (the 4B text stack); google/gemma-3-27b-it is a preset
        "gemma3": dict(hidden_size=2304, intermediate_size=9216, num_hidden_layers=26, num_attention_heads=8, num_key_value_heads=4,
                       head_dim=256, vocab_size=262208, rms_norm_eps=1e-6, max_position_embeddings=131072, eos_token_id=1,
                       tie_word_embeddings=True, query_pre_attn_scalar=256, sliding_window=4096, sliding_window_pattern=6,
                       rope_local_base_freq=10000.0, rope_theta=1000000.0),     # Gemma3TextConfig: 1e6 (the published
                       # google/gemma-3-27b-it config.json leaves the field out)
        # [3P] MistralConfig defaults (Mistral-7B-v0.1; ultravox_config.py:68 names MistralConfig next to LlamaConfig, README.md:27 "Llama 3,
        # Mistral, and Gemma"): a Llama block whose EVERY layer attends to the last sliding_window positions (MistralModel.forward:
        # create_sliding_window_causal_mask unless config.sliding_window is None - v0.2 / v0.3 / Nemo say null and are plain causal)
        "mistral": dict(hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32, num_key_value_heads=8,
                        vocab_size=32000, rms_norm_eps=1e-6, max_position_embeddings=131072, eos_token_id=2, tie_word_embeddings=False,
                        rope_theta=10000.0, sliding_window=4096),
    }
    FAMILIES = ("llama", "gemma", "qwen2", "qwen3", "gemma3", "mistral")

    def __post_init__(self):
        if self.model_type == "gemma3_text":        # the text stack of a Gemma3ForConditionalGeneration checkpoint
            self.model_type = "gemma3"
        if self.model_type not in self.FAMILIES:
            raise ValueError(f"text_config.model_type {self.model_type!r} is not built ({', '.join(self.FAMILIES)})")
        for k, v in self._FAMILY_DEFAULTS[self.model_type].items():
            if getattr(self, k) is None:
                setattr(self, k, v)
        if self.num_key_value_heads is None:     # LlamaConfig: defaults to num_attention_heads (no GQA)
            self.num_key_value_heads = self.num_attention_heads
        if self.head_dim is None:
            self.head_dim = self.hidden_size // self.num_attention_heads
        want = "gelu_pytorch_tanh" if self.model_type in ("gemma", "gemma3") else "silu"
        if self.hidden_act is None:
            self.hidden_act = want
        # [3P] GemmaMLP applies ACT2FN[config.hidden_act] (transformers 4.51.3 and the installed 5.x alike): "gelu" in a Gemma
        # checkpoint's config.json is the EXACT erf GELU, not the tanh approximation - built as its own GLU activation
        if self.hidden_act != want and not (self.model_type == "gemma" and self.hidden_act == "gelu"):
            raise ValueError(f"text_config.hidden_act {self.hidden_act!r} is not built for {self.model_type} ({want})")
        if self.model_type == "gemma3":
            if self.layer_types is None:      # [3P] Gemma3TextConfig: every sliding_window_pattern-th layer is global
                self.layer_types = ["full_attention" if (i + 1) % self.sliding_window_pattern == 0 else "sliding_attention"
                                    for i in range(self.num_hidden_layers)]
            if len(self.layer_types) != self.num_hidden_layers or set(self.layer_types) - {"full_attention", "sliding_attention"}:
                raise ValueError(f"text_config.layer_types {self.layer_types} does not fit {self.num_hidden_layers} gemma3 layers")
            rs = self.rope_scaling
            if rs and rs.get("rope_type", rs.get("type", "default")) not in ("default", "linear"):
                raise ValueError(f"gemma3 rope_scaling {rs}: default or linear (the global layers) are built")

    @property
    def window_layers(self) -> Optional[List[int]]:
        """Per-layer flags "attends to the last sliding_window positions only", or None when no layer does: Gemma-3's
        "sliding_attention" layers, every layer of a Mistral config whose sliding_window is set (0 / null = plain causal)."""
        if self.model_type == "gemma3":
            return [int(lt == "sliding_attention") for lt in self.layer_types]
        if self.model_type == "mistral" and self.sliding_window:
            return [1] * self.num_hidden_layers
        return None

    @property
    def is_gemma(self) -> bool:        # Gemma-1: GemmaRMSNorm, GeGLU, the embedding scale applied INSIDE the model (4.51.3)
        return self.model_type == "gemma"

    @property
    def is_gemma3(self) -> bool:       # Gemma-3 text stack: Gemma norms, GeGLU, four norms per layer, q / k norms, local / global layers
        return self.model_type == "gemma3"

    @property
    def has_qk_norm(self) -> bool:     # Qwen3Attention / Gemma3Attention q_norm / k_norm
        return self.model_type in ("qwen3", "gemma3")

    @property
    def has_qkv_bias(self) -> bool:    # Qwen2Attention: q_proj / k_proj / v_proj with bias, o_proj without
        return self.model_type == "qwen2"

    @property
    def ties_head(self) -> bool:       # lm_head IS embed_tokens (Gemma always; any checkpoint whose config says tie_word_embeddings)
        return self.model_type == "gemma" or bool(self.tie_word_embeddings)      # e.g. Llama-3.2-1B / 3B, the small Qwen models, Gemma-3


AUDIO_PRESETS: Dict[str, Dict[str, Any]] = {
    # conv feature encoder 512 x 7, kernels (10,3,3,3,3,2,2), strides (5,2,2,2,2,2,2) = 320x
    "facebook/wav2vec2-large-960h": dict(model_type="wav2vec2", d_model=1024, encoder_layers=24, encoder_attention_heads=16,
                                         encoder_ffn_dim=4096, layer_norm_eps=1e-5),
    "facebook/wav2vec2-large-960h-lv60-self": dict(model_type="wav2vec2", d_model=1024, encoder_layers=24, encoder_attention_heads=16,
                                                   encoder_ffn_dim=4096, layer_norm_eps=1e-5, feat_extract_norm="layer", conv_bias=True,
                                                   do_stable_layer_norm=True),
    "facebook/wav2vec2-large-lv60": dict(model_type="wav2vec2", d_model=1024, encoder_layers=24, encoder_attention_heads=16,
                                         encoder_ffn_dim=4096, layer_norm_eps=1e-5, feat_extract_norm="layer", conv_bias=True,
                                         do_stable_layer_norm=True),
    "facebook/wav2vec2-base-960h": dict(model_type="wav2vec2", d_model=768, encoder_layers=12, encoder_attention_heads=12,
                                        encoder_ffn_dim=3072, layer_norm_eps=1e-5),
    "openai/whisper-tiny": dict(d_model=384, encoder_layers=4, encoder_attention_heads=6, encoder_ffn_dim=1536, num_mel_bins=80),
    "openai/whisper-small": dict(d_model=768, encoder_layers=12, encoder_attention_heads=12, encoder_ffn_dim=3072, num_mel_bins=80),
    "openai/whisper-medium": dict(d_model=1024, encoder_layers=24, encoder_attention_heads=16, encoder_ffn_dim=4096, num_mel_bins=80),
    "openai/whisper-large-v3": dict(d_model=1280, encoder_layers=32, encoder_attention_heads=20, encoder_ffn_dim=5120, num_mel_bins=128),
    "openai/whisper-large-v3-turbo": dict(d_model=1280, encoder_layers=32, encoder_attention_heads=20, encoder_ffn_dim=5120, num_mel_bins=128),
}
_LLAMA3_SCALING = dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0,
                       original_max_position_embeddings=8192)
TEXT_PRESETS: Dict[str, Dict[str, Any]] = {
    "TinyLlama/TinyLlama-1.1B-Chat-v1.0": dict(hidden_size=2048, intermediate_size=5632, num_hidden_layers=22,
                                               num_attention_heads=32, num_key_value_heads=4, vocab_size=32000,
                                               rope_theta=10000.0, max_position_embeddings=2048, eos_token_id=2,
                                               rms_norm_eps=1e-5),
    "meta-llama/Meta-Llama-3-8B-Instruct": dict(hidden_size=4096, intermediate_size=14336, num_hidden_layers=32,
                                                num_attention_heads=32, num_key_value_heads=8, vocab_size=128256,
                                                rope_theta=500000.0, max_position_embeddings=8192, eos_token_id=128009,
                                                rms_norm_eps=1e-5),
    "meta-llama/Llama-3.1-8B-Instruct": dict(hidden_size=4096, intermediate_size=14336, num_hidden_layers=32,
                                             num_attention_heads=32, num_key_value_heads=8, vocab_size=128256,
                                             rope_theta=500000.0, rope_scaling=_LLAMA3_SCALING,
                                             max_position_embeddings=131072, eos_token_id=128009, rms_norm_eps=1e-5),
    "meta-llama/Llama-3.3-70B-Instruct": dict(hidden_size=8192, intermediate_size=28672, num_hidden_layers=80,
                                              num_attention_heads=64, num_key_value_heads=8, vocab_size=128256,
                                              rope_theta=500000.0, rope_scaling=_LLAMA3_SCALING,
                                              max_position_embeddings=131072, eos_token_id=128009, rms_norm_eps=1e-5),
}
# Gemma-1 (C5).  head_dim 256 is NOT hidden_size / heads for the 7B model (3072 / 16 = 192)
TEXT_PRESETS["google/gemma-7b"] = dict(model_type="gemma", hidden_size=3072, intermediate_size=24576, num_hidden_layers=28,
                                       num_attention_heads=16, num_key_value_heads=16, head_dim=256, vocab_size=256000,
                                       rms_norm_eps=1e-6, rope_theta=10000.0, max_position_embeddings=8192, eos_token_id=1)
TEXT_PRESETS["google/gemma-2b"] = dict(model_type="gemma", hidden_size=2048, intermediate_size=16384, num_hidden_layers=18,
                                       num_attention_heads=8, num_key_value_heads=1, head_dim=256, vocab_size=256000,
                                       rms_norm_eps=1e-6, rope_theta=10000.0, max_position_embeddings=8192, eos_token_id=1)
# Qwen (public config.json values; the reference's v0.6 recipe names Qwen/Qwen3-32B).  head_dim 128 is NOT hidden_size / heads for
# Qwen3-32B (5120 / 64 = 80)
TEXT_PRESETS["Qwen/Qwen3-32B"] = dict(model_type="qwen3", hidden_size=5120, intermediate_size=25600, num_hidden_layers=64,
                                      num_attention_heads=64, num_key_value_heads=8, head_dim=128, vocab_size=151936,
                                      rms_norm_eps=1e-6, rope_theta=1000000.0, max_position_embeddings=40960, eos_token_id=151645)
TEXT_PRESETS["Qwen/Qwen3-8B"] = dict(model_type="qwen3", hidden_size=4096, intermediate_size=12288, num_hidden_layers=36,
                                     num_attention_heads=32, num_key_value_heads=8, head_dim=128, vocab_size=151936,
                                     rms_norm_eps=1e-6, rope_theta=1000000.0, max_position_embeddings=40960, eos_token_id=151645)
TEXT_PRESETS["Qwen/Qwen2.5-7B-Instruct"] = dict(model_type="qwen2", hidden_size=3584, intermediate_size=18944, num_hidden_layers=28,
                                                num_attention_heads=28, num_key_value_heads=4, vocab_size=152064,
                                                rms_norm_eps=1e-6, rope_theta=1000000.0, max_position_embeddings=32768,
                                                eos_token_id=151645)
# the other ids the reference's recipes name (ultravox/training/configs/*.yaml)
TEXT_PRESETS["meta-llama/Llama-3.2-1B-Instruct"] = dict(hidden_size=2048, intermediate_size=8192, num_hidden_layers=16,
                                                        num_attention_heads=32, num_key_value_heads=8, head_dim=64, vocab_size=128256,
                                                        rope_theta=500000.0, rope_scaling=dict(_LLAMA3_SCALING, factor=32.0),
                                                        max_position_embeddings=131072, eos_token_id=128009, rms_norm_eps=1e-5,
                                                        tie_word_embeddings=True)
TEXT_PRESETS["meta-llama/Meta-Llama-3.1-8B-Instruct"] = TEXT_PRESETS["meta-llama/Llama-3.1-8B-Instruct"]
TEXT_PRESETS["meta-llama/Meta-Llama-3.1-70B-Instruct"] = TEXT_PRESETS["meta-llama/Llama-3.3-70B-Instruct"]      # same architecture
TEXT_PRESETS["meta-llama/Llama-3-8B-Instruct"] = TEXT_PRESETS["meta-llama/Meta-Llama-3-8B-Instruct"]
# the reference's other v0.6 backbone (v0.6_config_gemma3_27b.yaml): the text stack of google/gemma-3-27b-it (public config.json)
TEXT_PRESETS["google/gemma-3-27b-it"] = dict(model_type="gemma3", hidden_size=5376, intermediate_size=21504, num_hidden_layers=62,
                                             num_attention_heads=32, num_key_value_heads=16, head_dim=128, vocab_size=262208,
                                             rms_norm_eps=1e-6, rope_theta=1000000.0, rope_scaling=dict(rope_type="linear", factor=8.0),
                                             rope_local_base_freq=10000.0, query_pre_attn_scalar=168, sliding_window=1024,
                                             sliding_window_pattern=6, max_position_embeddings=131072, eos_token_id=1)
# Mistral backbones (ultravox_config.py:68 names MistralConfig; README.md:27; the published config.json files): v0.1 carries the 4096-position
# sliding window on every layer, v0.3 and Nemo say "sliding_window": null (0 here: plain causal attention)
TEXT_PRESETS["mistralai/Mistral-7B-Instruct-v0.1"] = dict(model_type="mistral", hidden_size=4096, intermediate_size=14336, num_hidden_layers=32,
                                                          num_attention_heads=32, num_key_value_heads=8, vocab_size=32000, rms_norm_eps=1e-5,
                                                          rope_theta=10000.0, max_position_embeddings=32768, sliding_window=4096, eos_token_id=2)
TEXT_PRESETS["mistralai/Mistral-7B-Instruct-v0.3"] = dict(TEXT_PRESETS["mistralai/Mistral-7B-Instruct-v0.1"], vocab_size=32768,
                                                          rope_theta=1000000.0, sliding_window=0)
TEXT_PRESETS["mistralai/Mistral-Nemo-Instruct-2407"] = dict(model_type="mistral", hidden_size=5120, intermediate_size=14336, num_hidden_layers=40,
                                                            num_attention_heads=32, num_key_value_heads=8, head_dim=128, vocab_size=131072,
                                                            rms_norm_eps=1e-5, rope_theta=1000000.0, max_position_embeddings=1024000,
                                                            sliding_window=0, eos_token_id=2)
TEXT_PRESETS["TinyLlama/TinyLlama-1.1B-Chat"] = TEXT_PRESETS["TinyLlama/TinyLlama-1.1B-Chat-v1.0"]
TEXT_PRESETS["meta-llama/Meta-Llama-3-8B"] = TEXT_PRESETS["meta-llama/Meta-Llama-3-8B-Instruct"]


def _mk(cls, value, presets, model_id):
    if model_id is not None:
        if model_id not in presets:
            raise ValueError(f"unknown model id {model_id!r}: no network here, known ids: {sorted(presets)}")
        return cls(**presets[model_id])
    if value is None:
        return cls()
    if isinstance(value, cls):
        return value
    names = {f.name for f in dataclasses.fields(cls)}
    get = (lambda k: value.get(k)) if isinstance(value, dict) else (lambda k: getattr(value, k, None))   # dict or HF config object
    if cls is TextConfig and get("model_type") == "gemma3" and get("text_config") is not None:
        # google/gemma-3-*-it are Gemma3ForConditionalGeneration checkpoints: AutoModelForCausalLM (ultravox_model.py:507-523) loads
        # the whole model and the LLM path runs its text stack - the nested text_config
        value = get("text_config")
        get = (lambda k: value.get(k)) if isinstance(value, dict) else (lambda k: getattr(value, k, None))
    _check_supported(cls, get)
    kw = ({k: v for k, v in value.items() if k in names} if isinstance(value, dict)
          else {k: getattr(value, k) for k in names if hasattr(value, k)})
    if cls is TextConfig:
        # transformers 5.x moved rope_theta / rope_scaling into one `rope_parameters` dict (the reference pins 4.51.3, whose configs
        # carry the two top-level fields): read both spellings - a config object or config.json saved by a newer transformers must not
        # fall back to the dataclass default of 10000 (Llama-3: 500000, Qwen: 1000000)
        rp = get("rope_parameters")
        if isinstance(rp, dict) and isinstance(rp.get("full_attention"), dict):      # gemma3 in transformers 5.x: one entry per layer type
            if get("rope_local_base_freq") is None and isinstance(rp.get("sliding_attention"), dict):
                kw["rope_local_base_freq"] = float(rp["sliding_attention"].get("rope_theta", 10000.0))
            rp = rp["full_attention"]
        if isinstance(rp, dict):
            if rp.get("rope_theta") is not None and get("rope_theta") is None:
                kw["rope_theta"] = float(rp["rope_theta"])
            if rp.get("rope_type", rp.get("type", "default")) != "default" and get("rope_scaling") is None:
                kw["rope_scaling"] = {k: v for k, v in rp.items() if k != "rope_theta"}
        if get("model_type") == "mistral" and get("sliding_window") is None and (
                ("sliding_window" in value) if isinstance(value, dict) else hasattr(value, "sliding_window")):
            kw["sliding_window"] = 0      # "sliding_window": null (Mistral v0.2 / v0.3 / Nemo): plain causal attention, NOT the family default
        rs = kw.get("rope_scaling")      # (5.x config objects alias rope_scaling to the rope_parameters dict)
        if isinstance(rs, dict) and isinstance(rs.get("full_attention"), dict):
            rs = rs["full_attention"]
        if isinstance(rs, dict):
            rs = {k: v for k, v in rs.items() if k != "rope_theta"}
            kw["rope_scaling"] = None if rs.get("rope_type", rs.get("type", "default")) == "default" else rs
    if cls is AudioConfig and get("model_type") == "wav2vec2":        # Wav2Vec2Config's names for the transformer dimensions
        for mine, theirs in (("d_model", "hidden_size"), ("encoder_layers", "num_hidden_layers"),
                             ("encoder_attention_heads", "num_attention_heads"), ("encoder_ffn_dim", "intermediate_size")):
            if get(theirs) is not None and mine not in (value if isinstance(value, dict) else ()):
                kw[mine] = get(theirs)
    return cls(**kw)


def _check_supported(cls, get) -> None:
    """Fields outside the dataclass are dropped by _mk, so anything that would change the arithmetic must be refused here:
    a Qwen2 (q/k/v biases), Mistral (sliding window) or Gemma config would otherwise run silently as a bias-free Llama."""
    want = (*TextConfig.FAMILIES, "gemma3_text") if cls is TextConfig else ("whisper", "wav2vec2")
    mt = get("model_type")
    if mt is not None and mt not in want:
        raise ValueError(f"{cls.__name__}: model_type {mt!r} is not built (this path implements {', '.join(want)})")
    if cls is TextConfig:
        for flag in ("attention_bias", "mlp_bias"):
            if get(flag):
                raise ValueError(f"text_config.{flag} = True is not built (only Qwen2's own q / k / v biases are; o_proj and the MLP "
                                 "are bias-free in every family here)")
        # Qwen2 / Qwen3 configs always carry a sliding_window VALUE; it is live only with use_sliding_window ([3P] Qwen2Config:
        # "sliding_window if use_sliding_window else None") - layer_types other than full attention likewise
        if mt in ("gemma3", "gemma3_text"):      # Gemma-3: the local layers' window is part of the family (model.py checks T against it)
            for flag in ("attn_logit_softcapping", "final_logit_softcapping", "use_bidirectional_attention"):
                if get(flag):
                    raise ValueError(f"text_config.{flag} is not built")
            return
        if mt == "mistral":      # the window is the family's own: every layer (TextConfig.window_layers)
            return
        live_window = get("sliding_window") and (mt not in ("qwen2", "qwen3") or get("use_sliding_window"))
        if live_window or any(lt != "full_attention" for lt in (get("layer_types") or ())):
            raise ValueError("text_config.sliding_window is not built (full causal attention only)")


# projector_act -> uvx_config_t.proj_act (include/uvx.h UVX_PROJ_*); the names are transformers' ACT2FN keys whose modules are ONE fused
# torch op (one rounding of the result in bf16, which is what the kernel does): nn.SiLU ("swish" is the same class), F.gelu exact and
# tanh-approximated, nn.ReLU.  ("gelu_new" is the same tanh formula evaluated op by op in the activation dtype - a different rounding
# sequence in bf16 - and is left out rather than approximated.)
PROJECTOR_ACTS = {"swiglu": 0, "silu": 1, "swish": 1, "gelu_pytorch_tanh": 2, "gelu": 3, "relu": 4}


# the nn.Linear leaves of a tower's layers, in the order of uvx_enc_lora_layer_t's fields (q, k, v, o | g, u, d).  "audio_w2v": the wav2vec2 tower, whose
# feed-forward linears (intermediate_dense / output_dense) have no adapter slot
LORA_ATTN_MODULES = {"audio": ("q_proj", "k_proj", "v_proj", "out_proj"), "audio_w2v": ("q_proj", "k_proj", "v_proj", "out_proj"),
                     "text": ("q_proj", "k_proj", "v_proj", "o_proj")}
LORA_MLP_MODULES = {"audio": ("fc1", "fc2"), "audio_w2v": (), "text": ("gate_proj", "up_proj", "down_proj")}
LORA_UNBUILT_MODULES = {"audio": (), "audio_w2v": ("intermediate_dense", "output_dense", "projection"), "text": ("lm_head",)}
LORA_DEFAULT_TARGETS = ("k_proj", "q_proj", "linear_k", "linear_q")      # LoraConfigSimplified.target_modules, ultravox_config.py:19-21


def lora_target_modules(lora_config, tower: str):
    """Which linears of `tower` ("audio": WhisperEncoder, "audio_w2v": Wav2Vec2Model, "text": the LLM) peft adapts for this LoraConfigSimplified
    dict: target_modules is matched by module-name suffix ([3P] peft check_target_module_exists, restated in tests/peft_stub.py), so names a tower
    does not have (linear_k, o_proj in Whisper, out_proj in Llama) are ignored as long as one name hits.  Returned in the field order of
    uvx_enc_lora_layer_t: the attention projections, then the MLP's linears (ABI 18)."""
    tm = list((lora_config or {}).get("target_modules") or LORA_DEFAULT_TARGETS)
    unbuilt = [m for m in LORA_UNBUILT_MODULES[tower] if m in tm]
    if unbuilt:
        raise ValueError(f"{tower}_model_lora_config.target_modules = {tm}: adapters on {unbuilt} are not built "
                         f"({list(LORA_ATTN_MODULES[tower] + LORA_MLP_MODULES[tower])} are)")
    hit = tuple(m for m in LORA_ATTN_MODULES[tower] + LORA_MLP_MODULES[tower] if m in tm)
    if not hit:
        raise ValueError(f"Target modules {tm} not found in the base model.")      # peft's message
    return hit


class UltravoxConfig:
    """Same constructor arguments and attributes as the reference's UltravoxConfig."""

    model_type = "ultravox"

    def __init__(self, audio_config=None, text_config=None, audio_model_id: Optional[str] = None,
                 text_model_id: Optional[str] = None, llm_only_training: bool = False, ignore_index: int = -100,
                 audio_token_index: Optional[int] = None, hidden_size: int = 4096, stack_factor: int = 8,
                 norm_init: float = 0.4, projector_act: str = "swiglu", projector_ln_mid: bool = False,
                 text_model_lora_config=None, audio_model_lora_config=None,
                 audio_latency_block_size: Optional[int] = None, torch_dtype: str = "bfloat16", **kwargs):
        self.ignore_index = ignore_index
        self.audio_model_id = audio_model_id
        self.text_model_id = text_model_id
        self.audio_token_index = audio_token_index
        self.hidden_size = hidden_size
        self.stack_factor = stack_factor
        self.norm_init = norm_init
        self.projector_act = projector_act
        self.projector_ln_mid = projector_ln_mid
        self.text_config: TextConfig = _mk(TextConfig, text_config, TEXT_PRESETS, text_model_id)
        self.audio_config: AudioConfig = _mk(AudioConfig, audio_config, AUDIO_PRESETS, audio_model_id)
        self.llm_only_training = llm_only_training
        as_dict = lambda c: c if isinstance(c, dict) else dataclasses.asdict(c or LoraConfigSimplified())
        self.text_model_lora_config = as_dict(text_model_lora_config)
        self.audio_model_lora_config = as_dict(audio_model_lora_config)
        self.audio_latency_block_size = audio_latency_block_size
        self.vocab_size = self.text_config.vocab_size
        self.initializer_range = self.text_config.initializer_range
        self.torch_dtype = torch_dtype
        # UltravoxProjector: transformers.activations.get_activation(projector_act) (ultravox_model.py:754); "swiglu" (the default, registered
        # by the reference at :742) halves the width, every other activation keeps it (:755)
        if projector_act not in PROJECTOR_ACTS:
            raise ValueError(f"projector_act={projector_act!r} is not built: {sorted(PROJECTOR_ACTS)} are (the reference default is 'swiglu', "
                             "ultravox_config.py:126)")
        # LoRA (apply_lora, ultravox_model.py:690-709): rank-r adapters on the attention projections target_modules names (peft's
        # suffix match; the default list hits q_proj + k_proj in Whisper / Llama) of the encoder (the release configs: r = 8) and / or
        # the LLM.  lora_target_modules() above resolves the list (attention projections and the MLP's linears; what is not built is refused by name).
        for name, lc in (("audio", self.audio_model_lora_config), ("text", self.text_model_lora_config)):
            ar = int(lc.get("r", 0) or 0)
            if lc.get("unfreeze_layers"):
                # apply_lora with r = 0 unfreezes the matching layers (ultravox_model.py:694-703): full fine-tuning of tower
                # layers is not built - refuse instead of leaving them silently frozen
                raise ValueError(f"{name}_model_lora_config.unfreeze_layers is not built (tower layers stay frozen)")
            if ar == 0:
                continue
            if not 0 < ar <= 64:
                raise ValueError(f"{name}_model_lora_config.r = {ar}: ranks 1..64 are built")
            lora_target_modules(lc, "audio_w2v" if name == "audio" and getattr(self.audio_config, "is_wav2vec2", False) else name)
        self.extra = kwargs

    @property
    def projector_mid_dim(self) -> int:
        """dim_mid of UltravoxProjector (ultravox_model.py:753-755): hidden_size // 2 behind SwiGLU, hidden_size otherwise."""
        return self.hidden_size // 2 if self.projector_act == "swiglu" else self.hidden_size

    def to_dict(self) -> Dict[str, Any]:
        d = {k: v for k, v in self.__dict__.items() if k not in ("text_config", "audio_config", "extra")}
        d["text_config"] = dataclasses.asdict(self.text_config)
        d["audio_config"] = dataclasses.asdict(self.audio_config)
        d["model_type"] = self.model_type
        return d

    def to_diff_dict(self) -> Dict[str, Any]:  # ultravox_config.py:188-203
        d = self.to_dict()
        if self.text_model_id is not None:
            d.pop("text_config", None)
        if self.audio_model_id is not None:
            d.pop("audio_config", None)
        return d
