"""Checkpoint I/O: diff-state-dict semantics pinned against the reference
(tests/golden/diff_state_dict.json from ultravox_model.py:565-584), safetensors round trip, merge rules."""
import json
import os

import pytest
import torch

from ultravox_amd import checkpoint
from ultravox_amd.config import UltravoxConfig


def test_diff_state_dict_matches_reference_fixture(golden_dir):
    for c in json.load(open(os.path.join(golden_dir, "diff_state_dict.json"))):
        sd = {k: torch.zeros(1) for k in c["state_dict_keys"]}
        kept = checkpoint.diff_state_dict(sd, c["trainable"], c["keep_params"])
        assert sorted(kept) == c["kept"], c["name"]


def test_save_load_round_trip_is_bit_exact(tmp_path):
    from test_model_gpu import SMALL
    cfg = UltravoxConfig(**SMALL)
    g = torch.Generator().manual_seed(0)
    P = "multi_modal_projector."
    sd = {P + "ln_pre.weight": torch.randn(64, generator=g).bfloat16(), P + "linear_1.weight": torch.randn(32, 64, generator=g).bfloat16(),
          "audio_tower.conv1.weight": torch.randn(8, 4, 3, generator=g).bfloat16(),
          "language_model.model.norm.weight": torch.randn(16, generator=g).bfloat16()}
    kept = checkpoint.save_pretrained(str(tmp_path), cfg, sd, trainable_params=[k for k in sd if k.startswith(P)],
                                      keep_params=["audio_tower.conv1.weight"])
    assert sorted(kept) == sorted([P + "ln_pre.weight", P + "linear_1.weight", "audio_tower.conv1.weight"])
    assert sorted(os.listdir(tmp_path)) == ["config.json", "model.safetensors"]          # the HF layout
    cfg2, ck = checkpoint.load_pretrained(str(tmp_path))
    assert sorted(ck) == sorted(kept)
    for k in ck:
        assert ck[k].dtype == torch.bfloat16 and torch.equal(ck[k], sd[k])
    assert cfg2.to_dict() == cfg.to_dict()
    # config.json is the reference's to_diff_dict: sub-configs with a model id are not inlined (ultravox_config.py:188-203)
    cj = json.load(open(tmp_path / "config.json"))
    assert ("text_config" in cj) == (cfg.text_model_id is None) and ("audio_config" in cj) == (cfg.audio_model_id is None)


def test_merge_state_dict_rules():
    base = {"a": torch.zeros(2, 3), "b": torch.zeros(4)}
    merged, keep = checkpoint.merge_state_dict(base, {"b": torch.ones(4)})
    assert torch.equal(merged["b"], torch.ones(4)) and merged["a"] is base["a"] and keep == {"b"}
    with pytest.raises(KeyError):
        checkpoint.merge_state_dict(base, {"c": torch.ones(1)})
    with pytest.raises(ValueError):
        checkpoint.merge_state_dict(base, {"b": torch.ones(5)})


def test_trainer_state_round_trip(tmp_path):
    t = {"exp_avg": torch.arange(8, dtype=torch.float32).bfloat16(), "exp_avg_sq": torch.ones(8), "master": None}
    checkpoint.save_trainer_state(str(tmp_path), 7, t, {"lr": 2e-3})
    step, tt, extra = checkpoint.load_trainer_state(str(tmp_path))
    assert step == 7 and extra["lr"] == 2e-3 and "master" not in tt
    assert torch.equal(tt["exp_avg"], t["exp_avg"]) and tt["exp_avg"].dtype == torch.bfloat16


def test_pack_llm_consume_frees_the_sources_and_packs_the_same_operands():
    """pack_llm(consume=True): identical packed operands, and the q/k/v/gate/up sources are popped layer by layer (the
    70B-parameter LLM then loads at one copy of the model plus a layer instead of 1.65 copies)."""
    import torch
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.weights import pack_llm, random_state_dict
    cfg = UltravoxConfig(audio_config=dict(d_model=64, encoder_layers=1, encoder_attention_heads=2, encoder_ffn_dim=128),
                         text_config=dict(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=4,
                                          num_key_value_heads=2, vocab_size=96), hidden_size=64)
    sd = random_state_dict(cfg, seed=1, dtype=torch.float32)
    keep = pack_llm(dict(sd), cfg, torch.float32, "cpu", with_transposes=True, rope_len=64)
    eaten = dict(sd)
    got = pack_llm(eaten, cfg, torch.float32, "cpu", with_transposes=True, rope_len=64, consume=True)
    for k in ("embed", "norm", "lm_head", "lm_head_t", "rope"):
        assert torch.equal(got[k], keep[k]), k
    for a, b in zip(got["layers"], keep["layers"]):
        assert set(a) == set(b) and all(torch.equal(a[k], b[k]) for k in a)
    gone = [k for k in sd if k not in eaten]
    assert len(gone) == 3 * 5 and all(any(t in k for t in ("q_proj", "k_proj", "v_proj", "gate_proj", "up_proj")) for k in gone)
    assert all(k in eaten for k in sd if "o_proj" in k or "down_proj" in k or "layernorm" in k or "embed" in k)


def test_tower_weights_from_local_hf_checkpoints(tmp_path):
    """audio_tower_state_dict / language_model_state_dict: a full Whisper file contributes only its encoder (the reference's
    base_model_prefix = "model.encoder"), a sharded causal-LM checkpoint is stitched from its index, tied embeddings supply
    the LM head; the result packs (pack_encoder / pack_llm) exactly like the state dict it was written from."""
    import json
    import torch
    from safetensors.torch import save_file
    from ultravox_amd import checkpoint
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.weights import pack_encoder, pack_llm, random_state_dict
    cfg = UltravoxConfig(audio_config=dict(d_model=64, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=128),
                         text_config=dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                                          num_key_value_heads=2, vocab_size=96), hidden_size=64)
    sd = random_state_dict(cfg, seed=4, dtype=torch.float32)
    # Whisper: a WhisperForConditionalGeneration-shaped file (encoder + decoder + proj_out)
    wdir = tmp_path / "whisper"
    wdir.mkdir()
    whisper = {"model.encoder." + k[len("audio_tower."):]: v.contiguous() for k, v in sd.items() if k.startswith("audio_tower.")}
    whisper["model.decoder.embed_tokens.weight"] = torch.zeros(4, 4)
    whisper["proj_out.weight"] = torch.zeros(4, 4)
    save_file(whisper, str(wdir / "model.safetensors"))
    a = checkpoint.audio_tower_state_dict(str(wdir))
    assert set(a) == {k for k in sd if k.startswith("audio_tower.")} and all(torch.equal(a[k], sd[k]) for k in a)
    # LLM: two shards + index, tied embeddings (no lm_head.weight in the files)
    ldir = tmp_path / "llm"
    ldir.mkdir()
    llm = {k[len("language_model."):]: v.contiguous() for k, v in sd.items() if k.startswith("language_model.") and "lm_head" not in k}
    names = sorted(llm)
    shards = {"model-00001-of-00002.safetensors": names[::2], "model-00002-of-00002.safetensors": names[1::2]}
    for fn, ks in shards.items():
        save_file({k: llm[k] for k in ks}, str(ldir / fn))
    json.dump({"weight_map": {k: fn for fn, ks in shards.items() for k in ks}}, open(ldir / "model.safetensors.index.json", "w"))
    m = checkpoint.language_model_state_dict(str(ldir))
    assert torch.equal(m["language_model.lm_head.weight"], sd["language_model.model.embed_tokens.weight"])
    assert all(torch.equal(m[k], sd[k]) for k in sd if k.startswith("language_model.") and "lm_head" not in k)
    tied = dict(sd)
    tied["language_model.lm_head.weight"] = sd["language_model.model.embed_tokens.weight"]
    got_e, want_e = pack_encoder({**a, **m}, cfg, torch.float32, "cpu"), pack_encoder(tied, cfg, torch.float32, "cpu")
    got_l, want_l = pack_llm({**a, **m}, cfg, torch.float32, "cpu", rope_len=32), pack_llm(tied, cfg, torch.float32, "cpu", rope_len=32)
    assert torch.equal(got_e["conv2_w"], want_e["conv2_w"]) and torch.equal(got_e["layers"][1]["wqkv"], want_e["layers"][1]["wqkv"])
    assert torch.equal(got_l["lm_head"], want_l["lm_head"]) and torch.equal(got_l["layers"][1]["wgu"], want_l["layers"][1]["wgu"])
    with pytest.raises(FileNotFoundError):
        checkpoint.load_hf_weights(str(tmp_path))
    with pytest.raises(KeyError):
        checkpoint.audio_tower_state_dict(str(ldir))


def test_configs_outside_the_built_arithmetic_are_refused_not_run_as_llama():
    from ultravox_amd.config import UltravoxConfig
    ok_text = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=1, vocab_size=128)
    UltravoxConfig(text_config={**ok_text, "model_type": "llama", "attention_bias": False, "tie_word_embeddings": False})
    for bad in ({"model_type": "mixtral"}, {"model_type": "gemma2"}, {"attention_bias": True}, {"mlp_bias": True}, {"sliding_window": 4096},
                {"hidden_act": "gelu_pytorch_tanh"},
                {"model_type": "qwen2", "sliding_window": 4096, "use_sliding_window": True}, {"model_type": "qwen3", "attention_bias": True},
                {"model_type": "qwen3", "layer_types": ["sliding_attention"]}):
        with pytest.raises(ValueError):
            UltravoxConfig(text_config={**ok_text, **bad})
    # Mistral (ultravox_config.py:68 names MistralConfig): built - a Llama block with the sliding window on EVERY layer, or none under an explicit null
    mi = UltravoxConfig(text_config={**ok_text, "model_type": "mistral", "sliding_window": 64}).text_config
    assert mi.window_layers == [1] and mi.sliding_window == 64 and not mi.has_qk_norm and not mi.has_qkv_bias and mi.hidden_act == "silu"
    assert UltravoxConfig(text_config={**ok_text, "model_type": "mistral", "sliding_window": None}).text_config.window_layers is None
    assert UltravoxConfig(text_config={**ok_text, "model_type": "llama"}).text_config.window_layers is None
    # the Qwen families (the reference's v0.6 recipe: Qwen/Qwen3-32B): built; their configs always carry a sliding_window VALUE,
    # which is live only with use_sliding_window
    q3 = UltravoxConfig(text_config={**ok_text, "model_type": "qwen3", "head_dim": 64, "sliding_window": 4096, "use_sliding_window": False,
                                     "layer_types": ["full_attention"]}).text_config
    assert q3.has_qk_norm and not q3.has_qkv_bias and q3.head_dim == 64 and q3.hidden_act == "silu" and not q3.ties_head
    q2 = UltravoxConfig(text_config={**ok_text, "model_type": "qwen2", "tie_word_embeddings": True}).text_config
    assert q2.has_qkv_bias and not q2.has_qk_norm and q2.ties_head and q2.head_dim == 32
    tied = UltravoxConfig(text_config={**ok_text, "tie_word_embeddings": True}).text_config          # Llama-3.2-1B / 3B style
    assert tied.model_type == "llama" and tied.ties_head
    from ultravox_amd.weights import pack_llm, random_state_dict
    import torch
    cfg_t = UltravoxConfig(text_config={**ok_text, "tie_word_embeddings": True})
    sd_t = random_state_dict(cfg_t, seed=1)
    assert "language_model.lm_head.weight" not in sd_t
    packed = pack_llm(sd_t, cfg_t, torch.float32, "cpu")
    assert packed["lm_head"] is packed["embed"] and torch.equal(packed["lm_head_t"], packed["embed"].t())
    cfg_u = UltravoxConfig(text_config=ok_text)
    sd_u = {k: v for k, v in random_state_dict(cfg_u, seed=1).items() if k != "language_model.lm_head.weight"}
    with pytest.raises(KeyError, match="tie"):
        pack_llm(sd_u, cfg_u, torch.float32, "cpu")
    big = UltravoxConfig(text_model_id="Qwen/Qwen3-32B").text_config
    assert (big.hidden_size, big.num_attention_heads, big.num_key_value_heads, big.head_dim, big.intermediate_size) == (5120, 64, 8, 128, 25600)
    with pytest.raises(ValueError, match="model_type"):
        UltravoxConfig(audio_config={"model_type": "hubert", "d_model": 64})
    # the families that ARE built (BASELINE config 5): Gemma backbone, wav2vec2-large-960h-style tower - and their unbuilt variants
    g = UltravoxConfig(text_config={**ok_text, "model_type": "gemma", "head_dim": 32, "tie_word_embeddings": True}).text_config
    assert g.is_gemma and g.hidden_act == "gelu_pytorch_tanh" and g.head_dim == 32
    w = UltravoxConfig(audio_config={"model_type": "wav2vec2", "hidden_size": 64, "num_hidden_layers": 2, "num_attention_heads": 2,
                                     "intermediate_size": 128, "conv_dim": [64] * 7}).audio_config
    assert w.is_wav2vec2 and (w.d_model, w.encoder_layers, w.encoder_ffn_dim) == (64, 2, 128) and w.feat_extract_output_length(16000) == 49
    # round 5: the layer-norm (-lv60) family is built too - the three Wav2Vec2Config switches are read, singly and together
    for ok in ({"feat_extract_norm": "layer"}, {"do_stable_layer_norm": True}, {"conv_bias": True},
               {"feat_extract_norm": "layer", "do_stable_layer_norm": True, "conv_bias": True}):
        v = UltravoxConfig(audio_config={"model_type": "wav2vec2", "hidden_size": 64, **ok}).audio_config
        assert all(getattr(v, k) == val for k, val in ok.items())
    lv = UltravoxConfig(audio_model_id="facebook/wav2vec2-large-960h-lv60-self").audio_config
    assert (lv.feat_extract_norm, lv.conv_bias, lv.do_stable_layer_norm, lv.d_model, lv.encoder_layers) == ("layer", True, True, 1024, 24)
    with pytest.raises(ValueError, match="feat_extract_norm"):
        UltravoxConfig(audio_config={"model_type": "wav2vec2", "hidden_size": 64, "feat_extract_norm": "batch"})
    with pytest.raises(ValueError, match="hidden_act"):
        UltravoxConfig(text_config={**ok_text, "model_type": "gemma", "hidden_act": "silu"})
    # apply_lora with r = 0 + unfreeze_layers (ultravox_model.py:694-703) is not built: refused, not silently frozen
    for r in (0, 8):
        with pytest.raises(ValueError, match="unfreeze_layers"):
            UltravoxConfig(audio_model_lora_config={"r": r, "unfreeze_layers": ["layers.3"]})


def test_rope_parameters_of_newer_transformers_configs_are_read():
    """transformers 5.x keeps rope_theta / scaling in `rope_parameters`; the reference's pin (4.51.3) has top-level rope_theta /
    rope_scaling.  Both spellings must give the same TextConfig - never the silent 10000 default."""
    import transformers
    from ultravox_amd.config import UltravoxConfig
    ok_text = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=1, num_key_value_heads=1, vocab_size=128)
    scaling = dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0, original_max_position_embeddings=8192)
    old = UltravoxConfig(text_config={**ok_text, "rope_theta": 500000.0, "rope_scaling": scaling}).text_config
    new = UltravoxConfig(text_config={**ok_text, "rope_parameters": {"rope_theta": 500000.0, **scaling}}).text_config
    assert old.rope_theta == new.rope_theta == 500000.0 and old.rope_scaling == new.rope_scaling == scaling
    plain = UltravoxConfig(text_config={**ok_text, "rope_parameters": {"rope_theta": 1000000.0, "rope_type": "default"}}).text_config
    assert plain.rope_theta == 1000000.0 and plain.rope_scaling is None
    hf = transformers.Qwen3Config(**ok_text, head_dim=64, rope_theta=1000000.0)              # whatever the installed version stores
    got = UltravoxConfig(text_config=hf).text_config
    assert got.model_type == "qwen3" and got.rope_theta == 1000000.0 and got.head_dim == 64
    got = UltravoxConfig(text_config=hf.to_dict()).text_config
    assert got.rope_theta == 1000000.0


def test_a_text_config_without_rope_theta_takes_its_familys_default():
    """a 4.51.3-style gemma3 text_config dict (the published google/gemma-3-27b-it config.json leaves rope_theta out)
    must build the global layers' rotary table with Gemma3TextConfig's 1e6, not a flat 1e4; every family is pinned against the
    installed transformers config class."""
    import transformers
    from ultravox_amd.config import UltravoxConfig
    small = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=6, num_attention_heads=2, num_key_value_heads=1, vocab_size=128)

    def hf_theta(cfg):
        rp = getattr(cfg, "rope_parameters", None)
        if isinstance(rp, dict):
            rp = rp.get("full_attention", rp)
            return float(rp["rope_theta"])
        return float(cfg.rope_theta)

    for mt, cls in (("gemma3_text", transformers.Gemma3TextConfig), ("llama", transformers.LlamaConfig), ("gemma", transformers.GemmaConfig),
                    ("qwen2", transformers.Qwen2Config), ("qwen3", transformers.Qwen3Config)):
        got = UltravoxConfig(text_config={"model_type": mt, **small}).text_config
        assert got.rope_theta == hf_theta(cls()), mt
    g3 = UltravoxConfig(text_config={"model_type": "gemma3_text", **small}).text_config
    assert g3.rope_theta == 1000000.0 and g3.rope_local_base_freq == 10000.0
    nested = UltravoxConfig(text_config={"model_type": "gemma3", "text_config": {"model_type": "gemma3_text", **small}}).text_config
    assert nested.rope_theta == 1000000.0
    assert UltravoxConfig(text_config={"model_type": "gemma3_text", **small, "rope_theta": 5000.0}).text_config.rope_theta == 5000.0


def test_from_pretrained_base_gets_adapter_keys_before_the_checkpoint_is_merged():
    """The merge refuses unknown keys; a LoRA checkpoint's adapter keys must therefore already exist in the base."""
    from ultravox_amd import checkpoint
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.weights import init_lora_state_dict, random_state_dict
    cfg = UltravoxConfig(audio_config=dict(d_model=64, encoder_layers=1, encoder_attention_heads=2, encoder_ffn_dim=128),
                         text_config=dict(hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2,
                                          num_key_value_heads=1, vocab_size=128), hidden_size=64, audio_model_lora_config={"r": 4})
    base = random_state_dict(cfg, seed=1)
    ckpt = {k: v + 1 for k, v in init_lora_state_dict(cfg, seed=1).items()}
    with pytest.raises(KeyError):
        checkpoint.merge_state_dict(base, ckpt)
    seeded = dict(base)
    for k, v in init_lora_state_dict(cfg, seed=0).items():
        seeded.setdefault(k, v)
    merged, keep = checkpoint.merge_state_dict(seeded, ckpt)
    assert keep == set(ckpt) and all(torch.equal(merged[k], ckpt[k]) for k in ckpt)


def test_config_matches_the_reference_config_fixture():
    """UltravoxConfig against tests/golden/config.json, recorded from the IMPORTED reference UltravoxConfig
    (ultravox_config.py:56-203; tests/golden/make_golden.py `config_cases`): every field the hot path reads, the [3P] family
    defaults that a PARTIAL text_config / audio_config dict resolves to (a config.json that omits rms_norm_eps means 1e-6 to
    LlamaConfig, not the 1e-5 Llama-3 checkpoints happen to carry), LoRA sub-configs, and the to_diff_dict key set."""
    import dataclasses
    import json
    import os
    from ultravox_amd.config import LossConfig, UltravoxConfig
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "config.json")))
    alias = {"hidden_size": "d_model", "num_hidden_layers": "encoder_layers", "num_attention_heads": "encoder_attention_heads",
             "intermediate_size": "encoder_ffn_dim"}          # Wav2Vec2Config / WhisperConfig names for the same four numbers
    n = 0
    for name, case in fx.items():
        if name.startswith("_"):
            continue
        cfg, exp = UltravoxConfig(**json.loads(json.dumps(case["kwargs"]))), case["expect"]
        for k, v in exp.items():
            if k in ("text", "audio", "diff_keys", "loss"):
                continue
            got = getattr(cfg, k)
            got = dataclasses.asdict(got) if dataclasses.is_dataclass(got) else got
            assert got == v, (name, k, got, v)
            n += 1
        for k, v in exp["text"].items():
            assert getattr(cfg.text_config, k) == v, (name, "text." + k, getattr(cfg.text_config, k), v)
            n += 1
        for k, v in exp["audio"].items():
            mine = k if hasattr(cfg.audio_config, k) else alias.get(k)
            if mine is None or not hasattr(cfg.audio_config, mine):
                continue                                        # (fields the device path does not read)
            got = getattr(cfg.audio_config, mine)
            got = list(got) if isinstance(got, tuple) else got
            if got is None:
                continue                                        # whisper: no conv stack
            assert got == v, (name, "audio." + k, got, v)
            n += 1
        # same keys in the saved config (torch_dtype aside: this framework always records the dtype it runs in)
        assert sorted(k for k in cfg.to_diff_dict() if k not in ("transformers_version", "torch_dtype")) == exp["diff_keys"], name
    assert n > 150
    lc, ref = LossConfig(), fx["_loss_config_defaults"]
    assert lc.loss_function.value == ref["loss_function"] and lc.kl_temperature == ref["kl_temperature"]
    assert lc.requires_alt_fields == ref["requires_alt_fields"]


@pytest.mark.parametrize("family", ["llama", "qwen2", "qwen3", "gemma", "gemma3"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_unpack_inverts_the_packed_tower_layouts(family, dtype):
    """weights.unpack_encoder / unpack_llm (what re-exports a merged tower, ultravox_model.py:528-559 + :565-591): the packed
    device layouts (fused q|k|v with the pre-scaled q rows, 16-row interleaved gate|up, im2col conv weights, family extras) go
    back to the reference's checkpoint names bit for bit - every tower key, nothing else."""
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.weights import pack_encoder, pack_llm, random_state_dict, unpack_encoder, unpack_llm
    text = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                head_dim=32, vocab_size=256)
    if family != "llama":
        text["model_type"] = {"gemma3": "gemma3_text"}.get(family, family)
    if family == "gemma3":
        text.update(sliding_window=64, query_pre_attn_scalar=32, layer_types=["sliding_attention", "full_attention"])
    cfg = UltravoxConfig(audio_config=dict(d_model=128, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=256),
                         text_config=text, hidden_size=128)
    sd = random_state_dict(cfg, seed=3, dtype=dtype)
    back = {**unpack_encoder(pack_encoder(sd, cfg, dtype, "cpu"), cfg), **unpack_llm(pack_llm(sd, cfg, dtype, "cpu", with_transposes=False), cfg)}
    towers = {k for k in sd if not k.startswith("multi_modal_projector.")}
    assert set(back) == towers
    # the name-only helpers merge_and_unload uses (no tensor is touched: ADVICE r5) list exactly the unpacked keys, in the same order
    from ultravox_amd.weights import encoder_param_names, llm_param_names
    assert encoder_param_names(cfg) + llm_param_names(pack_llm(sd, cfg, dtype, "cpu", with_transposes=False), cfg) == list(back)
    for k in sorted(towers):
        assert back[k].dtype == dtype and back[k].shape == sd[k].shape and torch.equal(back[k], sd[k]), k


@pytest.mark.parametrize("family", ["group_norm", "layer_norm"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_unpack_wav2vec2_inverts_the_packed_tower_layouts(family, dtype):
    """weights.unpack_wav2vec2 (what re-exports a LoRA-merged wav2vec2 tower): the im2col conv weights, the fused q|k|v with the pre-scaled q rows and
    the rest go back to HF Wav2Vec2Model's checkpoint names bit for bit; the positional conv's weight-norm tensors and masked_spec_embed (a parameter
    no kernel reads) come back as the checkpoint had them - every tower key, nothing else, in the order the name-only helper lists."""
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.weights import pack_wav2vec2, random_state_dict, unpack_wav2vec2, wav2vec2_param_names
    ac = {"model_type": "wav2vec2", "hidden_size": 128, "num_hidden_layers": 2, "num_attention_heads": 2, "intermediate_size": 256, "conv_dim": [64] * 7,
          "num_conv_pos_embeddings": 16, "num_conv_pos_embedding_groups": 4}
    if family == "layer_norm":
        ac.update(feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True)
    cfg = UltravoxConfig(audio_config=ac, text_config=dict(hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2,
                                                           num_key_value_heads=2, head_dim=32, vocab_size=64), hidden_size=64)
    sd = random_state_dict(cfg, seed=3, dtype=dtype)
    sd["audio_tower.masked_spec_embed"] = torch.randn(128).to(dtype)
    enc = pack_wav2vec2(sd, cfg, dtype, "cpu")
    back = unpack_wav2vec2(enc, cfg)
    tower = {k for k in sd if k.startswith("audio_tower.")}
    assert set(back) == tower and wav2vec2_param_names(enc, cfg) == list(back)
    for k in sorted(tower):
        assert back[k].dtype == dtype and back[k].shape == sd[k].shape and torch.equal(back[k], sd[k]), k


def test_unpack_refuses_a_scale_it_cannot_undo_exactly():
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.weights import pack_encoder, random_state_dict, unpack_encoder
    cfg = UltravoxConfig(audio_config=dict(d_model=96, encoder_layers=1, encoder_attention_heads=3, encoder_ffn_dim=128),
                         text_config=dict(hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2,
                                          num_key_value_heads=2, head_dim=32, vocab_size=64), hidden_size=64)
    sd = random_state_dict(cfg, seed=3, dtype=torch.float32)
    with pytest.raises(ValueError, match="power of two"):
        unpack_encoder(pack_encoder(sd, cfg, torch.float32, "cpu"), cfg)      # head_dim 32: scale 2^-2.5
    from ultravox_amd.weights import check_encoder_exportable
    with pytest.raises(ValueError, match="power of two"):
        check_encoder_exportable(cfg)                # what merge_and_unload asks BEFORE it folds any adapter
