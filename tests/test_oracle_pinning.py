"""Pins the CPU oracle (oracle/reference_cpu.py) before anything is compared against it:
  * against fixtures produced by the REFERENCE's own classes (tests/golden/*.npz, make_golden.py);
  * against the installed HF blocks (the third-party arithmetic the reference calls);
  * directly against /root/reference when it is present (build container only)."""
import os
import sys
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import reference_cpu as O
from ultravox_amd.config import UltravoxConfig
from ultravox_amd.weights import random_state_dict

TINY = dict(
    audio_config=dict(d_model=64, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=128, num_mel_bins=80,
                      max_source_positions=1500),
    text_config=dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                     num_key_value_heads=2, vocab_size=512, rope_theta=10000.0, max_position_embeddings=512),
    hidden_size=128, stack_factor=8, projector_ln_mid=True)


def tiny_cfg(**kw):
    d = {**TINY, **kw}
    return UltravoxConfig(**d)


@pytest.mark.parametrize("variant", ["mid", "post"])
def test_projector_matches_reference_fixture(golden_dir, variant):
    z = np.load(os.path.join(golden_dir, f"projector_ln_{variant}.npz"))
    cfg = tiny_cfg(projector_ln_mid=(variant == "mid"), hidden_size=256)
    p = {k[2:]: torch.from_numpy(z[k]).clone().requires_grad_(True) for k in z.files if k.startswith("w.")}
    x = torch.from_numpy(z["x"]).clone().requires_grad_(True)
    y = O.projector_ref(p, cfg, x)
    np.testing.assert_allclose(y.detach().numpy(), z["y"], rtol=1e-5, atol=1e-6)
    y.backward(torch.from_numpy(z["gy"]))
    np.testing.assert_allclose(x.grad.numpy(), z["gx"], rtol=1e-4, atol=1e-6)
    for k in p:
        np.testing.assert_allclose(p[k].grad.numpy(), z["g." + k], rtol=1e-4, atol=1e-5, err_msg=k)


@pytest.mark.parametrize("act", ["gelu", "silu", "relu", "gelu_pytorch_tanh"])
def test_projector_with_a_plain_activation_matches_reference_fixture(golden_dir, act):
    """projector_act != "swiglu" (ultravox_model.py:754-755): the reference's UltravoxProjector keeps the width (linear_2 [D, hidden])
    and applies transformers' ACT2FN[projector_act]; fixture projector_act.npz = outputs and gradients of the imported reference."""
    z = np.load(os.path.join(golden_dir, "projector_act.npz"))
    cfg = tiny_cfg(projector_ln_mid=bool(z[f"{act}.ln_mid"]), hidden_size=128, projector_act=act)
    assert cfg.projector_mid_dim == 128
    pre = f"{act}.w."
    p = {k[len(pre):]: torch.from_numpy(z[k]).clone().requires_grad_(True) for k in z.files if k.startswith(pre)}
    assert tuple(p["linear_2.weight"].shape) == (64, 128)
    x = torch.from_numpy(z[f"{act}.x"]).clone().requires_grad_(True)
    y = O.projector_ref(p, cfg, x)
    np.testing.assert_allclose(y.detach().numpy(), z[f"{act}.y"], rtol=1e-5, atol=1e-6)
    y.backward(torch.from_numpy(z[f"{act}.gy"]))
    np.testing.assert_allclose(x.grad.numpy(), z[f"{act}.gx"], rtol=1e-4, atol=1e-6)
    for k in p:
        np.testing.assert_allclose(p[k].grad.numpy(), z[f"{act}.g.{k}"], rtol=1e-4, atol=1e-5, err_msg=k)
    with pytest.raises(ValueError, match="projector_act"):
        tiny_cfg(projector_act="gelu_new")


def test_latency_mask_matches_reference_fixture(golden_dir):
    z = np.load(os.path.join(golden_dir, "latency_mask.npz"))
    for block in (100, 300, 1500):
        m = O.latency_mask_ref(3000, block, torch.float32)[0, 0, :400, :400]
        assert np.array_equal((m == 0).numpy(), z[f"allowed_{block}"])
    assert int(z["err13"]) == 1
    with pytest.raises(AssertionError, match="must divide 3000 evenly"):
        O.latency_mask_ref(3000, 13, torch.float32)


@pytest.mark.parametrize("n_mels", [80, 128])
def test_logmel_matches_hf_fixture(golden_dir, n_mels):
    z = np.load(os.path.join(golden_dir, "logmel.npz"))
    mel = O.logmel_ref(torch.from_numpy(z[f"pcm_{n_mels}"]), n_mels).numpy()
    assert mel.shape == z[f"mel_{n_mels}"].shape
    np.testing.assert_allclose(mel, z[f"mel_{n_mels}"], rtol=0, atol=2e-5)


@pytest.mark.parametrize("n_mels", [80, 128])
def test_logmel_matches_hf_on_speech_like_audio(golden_dir, n_mels):
    """30 s of harmonic, speech-like audio with pauses (forward_fixture_util.speech_like_pcm): most bins sit at or near the
    per-clip `max - 8` floor, which white noise never reaches.  Fixture: the installed HF WhisperFeatureExtractor."""
    import forward_fixture_util as U
    z = np.load(os.path.join(golden_dir, "logmel_speech.npz"))
    pcm = U.speech_like_pcm()
    assert abs(float(np.abs(pcm.astype(np.float64)).sum()) - float(z["pcm_checksum"])) < 1e-6 * float(z["pcm_checksum"])
    mel = O.logmel_ref(torch.from_numpy(pcm)[None], n_mels)[0].numpy()
    assert mel.shape == (n_mels, 3000)
    np.testing.assert_allclose(mel[:, ::4], z[f"mel_{n_mels}_every4"], rtol=0, atol=5e-5)
    assert abs(float(mel.max()) - float(z[f"max_{n_mels}"])) < 1e-5


def test_feature_extractor_contract_matches_hf():
    import transformers
    rng = np.random.RandomState(3)
    clips = [rng.randn(n).astype(np.float32) * 0.1 for n in (16000, 4321, 700)]
    hf = transformers.WhisperFeatureExtractor()(clips, sampling_rate=16000, padding="longest", pad_to_multiple_of=160,
                                                truncation=False, return_attention_mask=True, return_tensors="pt")
    ours = O.FeatureExtractorRef(80)(clips, sampling_rate=16000, padding="longest", pad_to_multiple_of=160)
    assert torch.equal(torch.as_tensor(hf["attention_mask"]).long(), ours["attention_mask"].long())
    np.testing.assert_allclose(ours["input_features"].numpy(), hf["input_features"].numpy(), atol=2e-5)


def test_encoder_matches_hf_whisper_blocks():
    """[3P] check: the restated encoder == installed HF WhisperEncoder sub-modules driven the way
    ModifiedWhisperEncoder.forward drives them (ultravox_model.py:893-899, :915-936, :944-980)."""
    from transformers import WhisperConfig
    from transformers.models.whisper.modeling_whisper import WhisperEncoder
    cfg = tiny_cfg()
    a = cfg.audio_config
    hf = WhisperEncoder(WhisperConfig(d_model=a.d_model, encoder_layers=a.encoder_layers,
                                      encoder_attention_heads=a.encoder_attention_heads, encoder_ffn_dim=a.encoder_ffn_dim,
                                      num_mel_bins=a.num_mel_bins, max_source_positions=a.max_source_positions,
                                      attn_implementation="eager")).eval()
    sd = random_state_dict(cfg, seed=5)
    enc_sd = {k[len("audio_tower."):]: v for k, v in sd.items() if k.startswith("audio_tower.")}
    missing, unexpected = hf.load_state_dict(enc_sd, strict=False)
    assert not unexpected and all("k_proj.bias" in m for m in missing), (missing, unexpected)
    torch.manual_seed(0)
    x = torch.randn(2, 80, 200)
    audio_len = torch.tensor([200, 123])
    with torch.no_grad():
        h = torch.nn.functional.gelu(hf.conv1(x))
        h = torch.nn.functional.gelu(hf.conv2(h)).permute(0, 2, 1)
        h = h + hf.embed_positions.weight[: h.size(-2)]
        feat_len = (audio_len - 1) // 2 + 1
        keep = torch.arange(h.shape[1])[None, :].lt(feat_len.view(-1, 1))
        mask = (1.0 - keep[:, None, None, :].float()) * torch.finfo(torch.float32).min
        for layer in hf.layers:
            out = layer(h, mask)
            h = out[0] if isinstance(out, tuple) else out
        want = hf.layer_norm(h)
        got = O.whisper_encoder_ref(sd, cfg, x, audio_len)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-4, atol=2e-5)


def test_llama_and_loss_match_hf_blocks():
    from transformers import LlamaConfig, LlamaForCausalLM
    cfg = tiny_cfg()
    t = cfg.text_config
    hf = LlamaForCausalLM(LlamaConfig(hidden_size=t.hidden_size, intermediate_size=t.intermediate_size,
                                      num_hidden_layers=t.num_hidden_layers, num_attention_heads=t.num_attention_heads,
                                      num_key_value_heads=t.num_key_value_heads, vocab_size=t.vocab_size,
                                      rms_norm_eps=t.rms_norm_eps, rope_theta=t.rope_theta,
                                      max_position_embeddings=t.max_position_embeddings, tie_word_embeddings=False,
                                      attn_implementation="eager")).eval()
    sd = random_state_dict(cfg, seed=6)
    llm_sd = {k[len("language_model."):]: v for k, v in sd.items() if k.startswith("language_model.")}
    missing, unexpected = hf.load_state_dict(llm_sd, strict=False)
    assert not unexpected and not [m for m in missing if "rotary" not in m and "inv_freq" not in m], (missing, unexpected)
    torch.manual_seed(1)
    B, T = 2, 37
    emb = torch.randn(B, T, t.hidden_size) * 0.5
    labels = torch.randint(0, t.vocab_size, (B, T))
    labels[:, :20] = -100
    am = torch.ones(B, T, dtype=torch.long)
    am[1, -9:] = 0  # right padding
    with torch.no_grad():
        out = hf(inputs_embeds=emb, attention_mask=am, labels=labels)
        logits = O.llama_ref(sd, cfg, emb, am)
        loss = O.causal_lm_loss_ref(logits, labels)
    keep = am.bool()
    np.testing.assert_allclose(logits[keep].numpy(), out.logits[keep].numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(loss.item(), out.loss.item(), rtol=1e-5)


def test_merge_is_sequential_overwrite():
    emb = torch.zeros(2, 10, 4)
    audio = torch.arange(3 * 5 * 4, dtype=torch.float32).view(3, 5, 4) + 1
    out = O.merge_ref(emb, audio, torch.tensor([1, 3, 0]), torch.tensor([4, 3, 2], dtype=torch.int32), torch.tensor([2, 1]))
    assert torch.equal(out[0, 1:3], audio[0, :2]) and torch.equal(out[0, 3:6], audio[1, :3])  # item 1 overwrote rows 3-4
    assert torch.equal(out[1, 0:2], audio[2, :2]) and out[0, 6:].abs().sum() == 0


@pytest.mark.skipif(not os.path.isdir("/root/reference/ultravox"), reason="reference tree only exists in the build container")
def test_projector_against_live_reference():
    sys.path.insert(0, "/root/reference")
    stubbed = "peft" not in sys.modules
    if stubbed:
        peft = types.ModuleType("peft")
        peft.LoraConfig = lambda **kw: types.SimpleNamespace(r=kw.get("r", 0))
        peft.PeftModel = type("PeftModel", (), {})
        peft.get_peft_model = lambda m, c: m
        peft.peft_model = types.ModuleType("peft.peft_model")
        peft.peft_model.PeftModel = peft.PeftModel
        sys.modules["peft"], sys.modules["peft.peft_model"] = peft, peft.peft_model
    try:
        from ultravox.model import ultravox_config, ultravox_model
    finally:
        if stubbed:      # a spec-less stub left behind breaks transformers' own `find_spec("peft")` probes in later tests
            sys.modules.pop("peft", None)
            sys.modules.pop("peft.peft_model", None)
        sys.path.remove("/root/reference")
    rcfg = ultravox_config.UltravoxConfig(
        audio_config={"model_type": "whisper", "d_model": 64, "encoder_layers": 1, "encoder_attention_heads": 2,
                      "encoder_ffn_dim": 64}, text_config={"model_type": "llama", "hidden_size": 128,
                                                           "intermediate_size": 64, "num_hidden_layers": 1,
                                                           "num_attention_heads": 2, "vocab_size": 64},
        hidden_size=128, projector_ln_mid=True)
    torch.manual_seed(3)
    proj = ultravox_model.UltravoxProjector(rcfg).float()
    x = torch.randn(2, 19, 64)
    want = proj(x)
    got = O.projector_ref({k: v for k, v in proj.state_dict().items()}, tiny_cfg(), x)
    np.testing.assert_allclose(got.detach().numpy(), want.detach().numpy(), rtol=1e-5, atol=1e-6)


def test_greedy_generate_matches_hf_generate_with_left_padding():
    """[3P] check of the oracle's generate restatement: HF LlamaForCausalLM.generate (greedy, KV cache, position ids
    from the attention mask) on a LEFT-padded batch == the cache-free restatement, token for token."""
    from transformers import LlamaConfig, LlamaForCausalLM
    cfg = tiny_cfg()
    t = cfg.text_config
    hf = LlamaForCausalLM(LlamaConfig(hidden_size=t.hidden_size, intermediate_size=t.intermediate_size,
                                      num_hidden_layers=t.num_hidden_layers, num_attention_heads=t.num_attention_heads,
                                      num_key_value_heads=t.num_key_value_heads, vocab_size=t.vocab_size,
                                      rms_norm_eps=t.rms_norm_eps, rope_theta=t.rope_theta,
                                      max_position_embeddings=t.max_position_embeddings, tie_word_embeddings=False,
                                      attn_implementation="eager", eos_token_id=3, pad_token_id=3)).eval()
    sd = random_state_dict(cfg, seed=8)
    sd["language_model.model.embed_tokens.weight"] = sd["language_model.model.embed_tokens.weight"] * 0.3
    hf.load_state_dict({k[len("language_model."):]: v for k, v in sd.items() if k.startswith("language_model.")}, strict=False)
    torch.manual_seed(4)
    B, T = 3, 11
    ids = torch.randint(4, t.vocab_size, (B, T))
    am = torch.ones(B, T, dtype=torch.long)
    am[0, :4] = 0
    am[2, :1] = 0
    ids[am == 0] = 3
    with torch.no_grad():
        want = hf.generate(input_ids=ids, attention_mask=am, max_new_tokens=6, do_sample=False, eos_token_id=3, pad_token_id=3)
    om = O.OracleModel(cfg, sd)
    got = om.generate_greedy(6, eos_token_id=3, pad_token_id=3, input_ids=ids, attention_mask=am)
    n = min(got.shape[1], want.shape[1])
    assert torch.equal(got[:, :n], want[:, :n]), (got, want)


BEAM_CASES = [dict(num_beams=3), dict(num_beams=4, length_penalty=0.0), dict(num_beams=2, early_stopping=True),
              dict(num_beams=3, num_return_sequences=2, length_penalty=2.0), dict(num_beams=3, early_stopping="never"),
              dict(num_beams=5, num_return_sequences=3, early_stopping=True), dict(num_beams=3, repetition_penalty=1.7),
              # HF's other score processors inside the beam search (they see the log-probabilities of prompt + hypothesis)
              dict(num_beams=3, no_repeat_ngram_size=2), dict(num_beams=2, min_new_tokens=4, bad_words_ids=[[40], [41, 42]], repetition_penalty=1.3)]
PROC_KEYS = ("no_repeat_ngram_size", "min_new_tokens", "bad_words_ids")


def hf_processor_list(case, prompt_len, eos):
    """(the case without the processor keywords, a [3P] HF LogitsProcessorList for them - HF's own classes, for the oracle's loops)"""
    from transformers.generation import logits_process as LP
    eos = list(eos) if isinstance(eos, (list, tuple)) else [eos]
    procs = LP.LogitsProcessorList()
    if "no_repeat_ngram_size" in case:
        procs.append(LP.NoRepeatNGramLogitsProcessor(case["no_repeat_ngram_size"]))
    if "bad_words_ids" in case:
        procs.append(LP.NoBadWordsLogitsProcessor(case["bad_words_ids"], eos_token_id=eos))
    if "min_new_tokens" in case:
        procs.append(LP.MinNewTokensLengthLogitsProcessor(prompt_len, case["min_new_tokens"], eos, device=torch.device("cpu")))
    return {k: v for k, v in case.items() if k not in PROC_KEYS}, (procs if len(procs) else None)


@pytest.mark.parametrize("n_eos", [1, 85, 200])
def test_beam_search_matches_hf_generate(n_eos):
    """[3P] check of the oracle's beam-search restatement (OracleModel.generate_beam: explicit per-item hypothesis lists, no KV cache)
    against HF LlamaForCausalLM.generate(num_beams=..) on a LEFT-padded batch: token for token, for several beam counts, length
    penalties, early_stopping settings and num_return_sequences; with 85 / 200 terminator ids (of 512) hypotheses finish early and the
    finished-list / stopping-heuristic logic is exercised."""
    from transformers import LlamaConfig, LlamaForCausalLM
    cfg = tiny_cfg()
    t = cfg.text_config
    hf = LlamaForCausalLM(LlamaConfig(hidden_size=t.hidden_size, intermediate_size=t.intermediate_size,
                                      num_hidden_layers=t.num_hidden_layers, num_attention_heads=t.num_attention_heads,
                                      num_key_value_heads=t.num_key_value_heads, vocab_size=t.vocab_size,
                                      rms_norm_eps=t.rms_norm_eps, rope_theta=t.rope_theta,
                                      max_position_embeddings=t.max_position_embeddings, tie_word_embeddings=False,
                                      attn_implementation="eager", eos_token_id=3, pad_token_id=3)).eval()
    sd = random_state_dict(cfg, seed=8)
    sd["language_model.model.embed_tokens.weight"] = sd["language_model.model.embed_tokens.weight"] * 0.3
    hf.load_state_dict({k[len("language_model."):]: v for k, v in sd.items() if k.startswith("language_model.")}, strict=False)
    torch.manual_seed(4)
    B, T = 3, 11
    ids = torch.randint(4, t.vocab_size, (B, T))
    am = torch.ones(B, T, dtype=torch.long)
    am[0, :4] = 0
    am[2, :1] = 0
    ids[am == 0] = 3
    eos = 3 if n_eos == 1 else list(range(3, 3 + n_eos))
    om = O.OracleModel(cfg, sd)
    short = 0
    for case in BEAM_CASES:
        with torch.no_grad():
            want = hf.generate(input_ids=ids, attention_mask=am, max_new_tokens=7, do_sample=False, eos_token_id=eos, pad_token_id=3, **case)
        plain, procs = hf_processor_list(case, T, eos)
        got = om.generate_beam(7, eos_token_id=eos, pad_token_id=3, input_ids=ids, attention_mask=am, logits_processor=procs, **plain)
        assert got.shape == want.shape and torch.equal(got, want), (case, got[:, T:], want[:, T:])
        short += int(want.shape[1] < T + 7 or bool((want[:, -1] == 3).any()))
    assert n_eos == 1 or short > 0          # (the many-terminator settings must actually end hypotheses early)


def test_beam_search_with_a_callers_stopping_criterion_matches_hf_generate():
    """generate(num_beams > 1, stopping_criteria=[..]): HF evaluates the caller's criteria on the flattened top-K candidates (prompt + hypothesis) of every
    step next to its terminator / length criteria; a hypothesis that hits one ends like one that produced a terminator."""
    from transformers import LlamaConfig, LlamaForCausalLM, StoppingCriteria, StoppingCriteriaList
    cfg = tiny_cfg()
    t = cfg.text_config
    hf = LlamaForCausalLM(LlamaConfig(hidden_size=t.hidden_size, intermediate_size=t.intermediate_size, num_hidden_layers=t.num_hidden_layers,
                                      num_attention_heads=t.num_attention_heads, num_key_value_heads=t.num_key_value_heads, vocab_size=t.vocab_size,
                                      rms_norm_eps=t.rms_norm_eps, rope_theta=t.rope_theta, max_position_embeddings=t.max_position_embeddings,
                                      tie_word_embeddings=False, attn_implementation="eager", eos_token_id=3, pad_token_id=3)).eval()
    sd = random_state_dict(cfg, seed=8)
    sd["language_model.model.embed_tokens.weight"] = sd["language_model.model.embed_tokens.weight"] * 0.3
    hf.load_state_dict({k[len("language_model."):]: v for k, v in sd.items() if k.startswith("language_model.")}, strict=False)
    torch.manual_seed(4)
    B, T = 3, 11
    ids = torch.randint(4, t.vocab_size, (B, T))
    am = torch.ones(B, T, dtype=torch.long)
    am[0, :4] = 0
    ids[am == 0] = 3

    class LastTokenBelow(StoppingCriteria):      # (about a fifth of the vocabulary ends a hypothesis)
        def __call__(self, input_ids, scores, **kw):
            return input_ids[:, -1] < 100
    om = O.OracleModel(cfg, sd)
    ended = 0
    for case in (dict(num_beams=3), dict(num_beams=4, length_penalty=0.0, num_return_sequences=2), dict(num_beams=2, early_stopping=True)):
        with torch.no_grad():
            want = hf.generate(input_ids=ids, attention_mask=am, max_new_tokens=7, do_sample=False, eos_token_id=3, pad_token_id=3,
                               stopping_criteria=StoppingCriteriaList([LastTokenBelow()]), **case)
            free = hf.generate(input_ids=ids, attention_mask=am, max_new_tokens=7, do_sample=False, eos_token_id=3, pad_token_id=3, **case)
        got = om.generate_beam(7, eos_token_id=3, pad_token_id=3, input_ids=ids, attention_mask=am, stopping_criteria=[LastTokenBelow()], **case)
        assert got.shape == want.shape and torch.equal(got, want), (case, got[:, T:], want[:, T:])
        ended += int(want.shape != free.shape or not torch.equal(want, free))
    assert ended > 0


BEAM_SAMPLE_CASES = [dict(num_beams=3, top_k=0), dict(num_beams=4, top_k=12, temperature=0.7), dict(num_beams=2, top_k=0, top_p=0.8, length_penalty=0.0),
                     dict(num_beams=3, top_k=0, temperature=1.3, num_return_sequences=2, repetition_penalty=1.4)]


@pytest.mark.parametrize("n_eos", [1, 2])
def test_beam_sampling_matches_hf_generate(n_eos):
    """[3P] HF beam SAMPLING (generate(num_beams > 1, do_sample=True): GenerationMixin._get_top_k_continuations draws the K continuations with ONE
    torch.multinomial on softmax(running score + warped log-probabilities) per step) against the oracle's restatement, both on the CPU generator after
    the same torch.manual_seed: token for token.  The warpers (temperature, top-k, top-p) are HF's own classes at the tail of the processor list."""
    from transformers import LlamaConfig, LlamaForCausalLM
    from transformers.generation import logits_process as LP
    cfg = tiny_cfg()
    t = cfg.text_config
    hf = LlamaForCausalLM(LlamaConfig(hidden_size=t.hidden_size, intermediate_size=t.intermediate_size, num_hidden_layers=t.num_hidden_layers,
                                      num_attention_heads=t.num_attention_heads, num_key_value_heads=t.num_key_value_heads, vocab_size=t.vocab_size,
                                      rms_norm_eps=t.rms_norm_eps, rope_theta=t.rope_theta, max_position_embeddings=t.max_position_embeddings,
                                      tie_word_embeddings=False, attn_implementation="eager", eos_token_id=3, pad_token_id=3)).eval()
    sd = random_state_dict(cfg, seed=8)
    sd["language_model.model.embed_tokens.weight"] = sd["language_model.model.embed_tokens.weight"] * 0.3
    hf.load_state_dict({k[len("language_model."):]: v for k, v in sd.items() if k.startswith("language_model.")}, strict=False)
    torch.manual_seed(4)
    B, T = 3, 11
    ids = torch.randint(4, t.vocab_size, (B, T))
    am = torch.ones(B, T, dtype=torch.long)
    am[0, :4] = 0
    ids[am == 0] = 3
    eos = 3 if n_eos == 1 else list(range(3, 3 + n_eos))
    om = O.OracleModel(cfg, sd)
    distinct = 0
    for i, case in enumerate(BEAM_SAMPLE_CASES):
        torch.manual_seed(100 + i)
        with torch.no_grad():
            want = hf.generate(input_ids=ids, attention_mask=am, max_new_tokens=7, do_sample=True, eos_token_id=eos, pad_token_id=3, **case)
            greedy = hf.generate(input_ids=ids, attention_mask=am, max_new_tokens=7, do_sample=False, eos_token_id=eos, pad_token_id=3,
                                 **{k: v for k, v in case.items() if k not in ("top_k", "top_p", "temperature")})
        warp = LP.LogitsProcessorList()
        if case.get("temperature", 1.0) != 1.0:
            warp.append(LP.TemperatureLogitsWarper(case["temperature"]))
        if case.get("top_k", 0):
            warp.append(LP.TopKLogitsWarper(case["top_k"], min_tokens_to_keep=n_eos + 1))      # (HF, beam methods: one non-terminator must survive)
        if case.get("top_p", 1.0) < 1.0:
            warp.append(LP.TopPLogitsWarper(case["top_p"], min_tokens_to_keep=n_eos + 1))
        plain = {k: v for k, v in case.items() if k not in ("top_k", "top_p", "temperature")}
        torch.manual_seed(100 + i)
        got = om.generate_beam(7, eos_token_id=eos, pad_token_id=3, input_ids=ids, attention_mask=am, logits_processor=warp, do_sample=True, **plain)
        assert got.shape == want.shape and torch.equal(got, want), (case, got[:, T:], want[:, T:])
        distinct += int(want.shape != greedy.shape or not torch.equal(want, greedy))
    assert distinct > 0      # (the draws did change the result against plain beam search)


KL_CASES = ["basic", "no_eot", "temp1_w05", "one_empty_row", "padded_tail"]


@pytest.mark.parametrize("case", KL_CASES)
def test_kl_loss_matches_reference_fixture(golden_dir, case):
    """kl_loss.npz holds the REFERENCE UltravoxModel._compute_kl_loss / _get_prediction_mask outputs
    (make_golden.py:kl_cases): masks bit-exact, loss and d loss / d student logits to f32 round-off."""
    z = np.load(os.path.join(golden_dir, "kl_loss.npz"))
    g = lambda k: torch.from_numpy(z[f"{case}.{k}"])
    labels, alt_labels = g("labels"), g("alt_labels")
    pm, em = O.prediction_mask_ref(labels)
    assert torch.equal(pm, g("pred_mask")) and torch.equal(em, g("eot_mask"))
    student = g("student").clone().requires_grad_(True)
    loss = O.kl_loss_ref(student, labels, g("teacher"), alt_labels, float(g("temperature")), float(g("eot_loss_weight")))
    loss.backward()
    assert abs(loss.item() - float(g("loss"))) <= 1e-6 * max(1.0, abs(float(g("loss"))))
    assert (student.grad - g("dstudent")).abs().max().item() < 1e-7


@pytest.mark.parametrize("case", KL_CASES)
def test_kl_row_pairs_reproduce_reference_masks(golden_dir, case):
    """The host-side pairing handed to uvx_llm_kl_loss, evaluated with plain torch, gives the reference loss: this is
    what pins the (pair_row, pair_w) contract of the C ABI without a GPU."""
    from ultravox_amd.model import kl_row_pairs
    z = np.load(os.path.join(golden_dir, "kl_loss.npz"))
    g = lambda k: torch.from_numpy(z[f"{case}.{k}"])
    labels, alt_labels, tau = g("labels"), g("alt_labels"), float(g("temperature"))
    pair_row, pair_w, n_pred = kl_row_pairs(labels, alt_labels, float(g("eot_loss_weight")))
    assert n_pred == int(g("pred_mask").sum())
    assert torch.equal(pair_row[0] >= 0, g("pred_mask").reshape(-1))
    if float(g("eot_loss_weight")) > 0:
        assert torch.equal(pair_row[1] >= 0, g("eot_mask").reshape(-1))
    s = g("student").reshape(-1, g("student").shape[-1])
    t = g("teacher").reshape(-1, g("teacher").shape[-1])
    tot = 0.0
    for slot in range(2):
        for r in torch.nonzero(pair_row[slot] >= 0)[:, 0].tolist():
            lt = F.log_softmax(t[pair_row[slot, r]] / tau, -1)
            tot += pair_w[slot, r].item() * (lt.exp() * (lt - F.log_softmax(s[r] / tau, -1))).sum().item()
    assert abs(tot - float(g("loss"))) <= 2e-6 * max(1.0, abs(float(g("loss"))))


def test_kl_row_pairs_rejects_unpairable_masks():
    from ultravox_amd.model import kl_row_pairs
    labels = torch.full((1, 8), -100); labels[0, 4:8] = 1
    alt = torch.full((1, 6), -100); alt[0, 3:6] = 1
    with pytest.raises(ValueError):
        kl_row_pairs(labels, alt, 1.0)


@pytest.mark.parametrize("case", KL_CASES)
def test_kl_compact_pairs_reproduce_reference_loss(golden_dir, case):
    """The compact operands of uvx_llm_fwd_rows / uvx_llm_kl_loss_rows (student rows, unique teacher rows, partner indices),
    evaluated with plain torch, give the reference loss."""
    from ultravox_amd.model import kl_compact_pairs, kl_row_pairs
    z = np.load(os.path.join(golden_dir, "kl_loss.npz"))
    g = lambda k: torch.from_numpy(z[f"{case}.{k}"])
    pr, pw, _ = kl_row_pairs(g("labels"), g("alt_labels"), float(g("eot_loss_weight")))
    rs, rt, pc, pwc = kl_compact_pairs(pr, pw)
    assert torch.equal(rs.long(), torch.nonzero((pr >= 0).any(0))[:, 0]) and torch.equal(rt, torch.unique(rt))
    s = g("student").reshape(-1, g("student").shape[-1])[rs.long()]
    t = g("teacher").reshape(-1, g("teacher").shape[-1])[rt.long()]
    tau, tot = float(g("temperature")), 0.0
    for slot in range(2):
        for r in range(len(rs)):
            if pc[slot, r] >= 0:
                lt = F.log_softmax(t[pc[slot, r]] / tau, -1)
                tot += pwc[slot, r].item() * (lt.exp() * (lt - F.log_softmax(s[r] / tau, -1))).sum().item()
    assert abs(tot - float(g("loss"))) <= 2e-6 * max(1.0, abs(float(g("loss"))))


def test_lora_restatement_matches_reference_apply_lora_fixture(golden_dir):
    """The REFERENCE's apply_lora (ultravox_model.py:690-709) run on installed-HF towers through tests/peft_stub.py
    (tests/golden/make_golden.py::lora_cases): the adapted module set, the trainable / checkpoint key names, the forward and
    the adapter gradients of the adapted encoder and LLM == the oracle's LoRA restatement and this package's key names.
    (peft's own arithmetic is restated by the stub - peft cannot be installed here - and that residue is stated there.)"""
    import json
    from ultravox_amd.weights import init_lora_state_dict, llm_lora_key, lora_key
    z = np.load(os.path.join(golden_dir, "lora_reference.npz"))
    meta = json.load(open(os.path.join(golden_dir, "lora_reference.json")))
    cfg = UltravoxConfig(**meta["tiny"], audio_model_lora_config=meta["lora_config"]["4"], text_model_lora_config=meta["lora_config"]["2"])
    # the reference's LoraConfigSimplified defaults reach peft unchanged, and r = 0 freezes everything without calling peft
    assert meta["lora_config"]["4"]["target_modules"] == ["k_proj", "q_proj", "linear_k", "linear_q"] and meta["r0_trainable"] == []
    assert meta["lora_config"]["4"]["lora_alpha"] == 8
    # which modules are adapted and under which names the checkpoint carries them
    enc_names = {"audio_tower." + n for n in meta["encoder"]["trainable"]}
    llm_names = {"language_model." + n for n in meta["llm"]["trainable"]}
    mine = set(init_lora_state_dict(cfg))
    assert mine == enc_names | llm_names
    assert lora_key(1, "q_proj", "A") in enc_names and llm_lora_key(0, "k_proj", "B") in llm_names
    assert "base_model.model.layers.0.self_attn.q_proj.base_layer.weight" in meta["encoder"]["state_dict_keys"]   # peft renames the wrapped linear
    assert "base_model.model.layers.0.self_attn.v_proj.weight" in meta["encoder"]["state_dict_keys"]               # v_proj is not adapted
    sd = random_state_dict(cfg, seed=meta["seed"])
    for tower, prefix in (("enc", "audio_tower."), ("llm", "language_model.")):
        for k in z.files:
            if k.startswith(tower + ".w."):
                sd[prefix + k[len(tower) + 3:]] = torch.from_numpy(z[k]).clone().requires_grad_(True)
    # encoder
    y = O.whisper_encoder_ref(sd, cfg, torch.from_numpy(z["enc.x"]), torch.from_numpy(z["enc.audio_len"]),
                              lora={"scaling": meta["lora_config"]["4"]["lora_alpha"] / 4})
    np.testing.assert_allclose(y.detach().numpy(), z["enc.y"], rtol=1e-4, atol=2e-5)
    (y * torch.from_numpy(z["enc.gy"])).sum().backward()
    for n in meta["encoder"]["trainable"]:
        np.testing.assert_allclose(sd["audio_tower." + n].grad.numpy(), z["enc.g." + n], rtol=2e-4, atol=2e-5, err_msg=n)
    # language model
    logits = O.llama_ref(sd, cfg, torch.from_numpy(z["llm.emb"]), torch.from_numpy(z["llm.mask"]),
                         lora={"scaling": meta["lora_config"]["2"]["lora_alpha"] / 2})
    keep = torch.from_numpy(z["llm.mask"]).bool()
    np.testing.assert_allclose(logits.detach()[keep].numpy(), z["llm.logits"][keep.numpy()], rtol=1e-4, atol=2e-5)
    (logits * torch.from_numpy(z["llm.gl"])).sum().backward()
    for n in meta["llm"]["trainable"]:
        np.testing.assert_allclose(sd["language_model." + n].grad.numpy(), z["llm.g." + n], rtol=2e-4, atol=2e-5, err_msg=n)


@pytest.mark.parametrize("case", ["all", "vo", "mlp"])
def test_lora_target_modules_beyond_the_default_match_the_reference_apply_lora(golden_dir, case):
    """target_modules is a config field the reference hands to peft unchanged (ultravox_config.py:19-21, ultravox_model.py:695, 707).
    Fixture lora_targets_reference.npz: the reference's apply_lora (tests/peft_stub.py) with q / k / v / out_proj / o_proj ("all") and a
    v + o-only list ("vo") on an HF WhisperEncoder and a GQA LlamaForCausalLM - which modules our config resolves, the key names, and the
    oracle's forward / adapter gradients with adapters on v_proj and on the output projection.  "mlp": q_proj next to the MLP's linears
    (fc1 / fc2 in Whisper, gate_proj / up_proj / down_proj in Llama; ABI 18)."""
    import json
    from ultravox_amd.config import lora_target_modules
    from ultravox_amd.weights import init_lora_state_dict, lora_targets
    z = np.load(os.path.join(golden_dir, "lora_targets_reference.npz"))
    meta = json.load(open(os.path.join(golden_dir, "lora_targets_reference.json")))
    cm = meta["cases"][case]
    lc = cm["lora_config"]
    cfg = UltravoxConfig(**meta["tiny"], audio_model_lora_config=lc, text_model_lora_config=lc)
    want_a = {"all": ("q_proj", "k_proj", "v_proj", "out_proj"), "vo": ("v_proj", "out_proj"), "mlp": ("q_proj", "fc1", "fc2")}[case]
    want_t = {"all": ("q_proj", "k_proj", "v_proj", "o_proj"), "vo": ("v_proj", "o_proj"), "mlp": ("q_proj", "gate_proj", "up_proj", "down_proj")}[case]
    assert lora_targets(cfg, "audio") == want_a and lora_targets(cfg, "text") == want_t
    enc_names = {"audio_tower." + n for n in cm["encoder"]["trainable"]}
    llm_names = {"language_model." + n for n in cm["llm"]["trainable"]}
    init = init_lora_state_dict(cfg)
    assert set(init) == enc_names | llm_names
    for k, v in init.items():      # shapes: lora_A [r, in], lora_B [out, r] (GQA: k / v are narrower than q; o_proj maps heads * head_dim -> hidden)
        tower, name = ("enc", k[len("audio_tower."):]) if k.startswith("audio_tower.") else ("llm", k[len("language_model."):])
        assert tuple(v.shape) == z[f"{case}.{tower}.w.{name}"].shape, k
    # peft's error for a list that hits no module is the text the config check raises
    with pytest.raises(ValueError) as e:
        lora_target_modules({"r": 2, "target_modules": ["linear_k"]}, "audio")
    assert meta["no_hit_error"] is not None and str(e.value) == meta["no_hit_error"]
    with pytest.raises(ValueError, match="lm_head"):      # what is not built is refused by name, not silently left un-adapted
        UltravoxConfig(**meta["tiny"], text_model_lora_config={"r": 2, "target_modules": ["q_proj", "lm_head"]})
    sd = random_state_dict(cfg, seed=meta["seed"])
    for tower, prefix in (("enc", "audio_tower."), ("llm", "language_model.")):
        for k in z.files:
            if k.startswith(f"{case}.{tower}.w."):
                sd[prefix + k[len(case) + len(tower) + 4:]] = torch.from_numpy(z[k]).clone().requires_grad_(True)
    scaling = lc["lora_alpha"] / lc["r"]
    y = O.whisper_encoder_ref(sd, cfg, torch.from_numpy(z[f"{case}.enc.x"]), torch.from_numpy(z[f"{case}.enc.audio_len"]), lora={"scaling": scaling})
    np.testing.assert_allclose(y.detach().numpy(), z[f"{case}.enc.y"], rtol=1e-4, atol=2e-5)
    (y * torch.from_numpy(z[f"{case}.enc.gy"])).sum().backward()
    for n in cm["encoder"]["trainable"]:
        np.testing.assert_allclose(sd["audio_tower." + n].grad.numpy(), z[f"{case}.enc.g." + n], rtol=2e-4, atol=2e-5, err_msg=n)
    logits = O.llama_ref(sd, cfg, torch.from_numpy(z[f"{case}.llm.emb"]), torch.from_numpy(z[f"{case}.llm.mask"]), lora={"scaling": scaling})
    keep = torch.from_numpy(z[f"{case}.llm.mask"]).bool()
    np.testing.assert_allclose(logits.detach()[keep].numpy(), z[f"{case}.llm.logits"][keep.numpy()], rtol=1e-4, atol=2e-5)
    (logits * torch.from_numpy(z[f"{case}.llm.gl"])).sum().backward()
    for n in cm["llm"]["trainable"]:
        np.testing.assert_allclose(sd["language_model." + n].grad.numpy(), z[f"{case}.llm.g." + n], rtol=2e-4, atol=2e-5, err_msg=n)


@pytest.mark.parametrize("hidden_act", ["gelu_pytorch_tanh", "gelu"])
def test_gemma_backbone_matches_hf_blocks(hidden_act):
    """hidden_act "gelu": [3P] GemmaMLP applies ACT2FN[config.hidden_act], so a checkpoint whose config.json says "gelu" runs the
    EXACT erf GELU (not the tanh approximation) - the oracle and the HIP path (UVX_ACT_GELU_ERF) follow the config.
    BASELINE config 5's backbone.  [3P] check: the oracle's Gemma flavour (GemmaRMSNorm, GeGLU, head_dim from the config,
    tied head, sqrt(hidden) embedding scale) == the installed HF GemmaForCausalLM.  The reference pins transformers 4.51.3,
    whose GemmaModel.forward multiplies whatever inputs_embeds it receives by the normalizer; the installed 5.x moved that
    scale into the embedding module, so the installed stack is fed PRE-SCALED embeddings."""
    from transformers import GemmaConfig, GemmaForCausalLM
    cfg = UltravoxConfig(audio_config=TINY["audio_config"], hidden_size=64,
                         text_config=dict(model_type="gemma", hidden_size=96, intermediate_size=256, num_hidden_layers=2,
                                          num_attention_heads=4, num_key_value_heads=2, head_dim=32, vocab_size=160, rms_norm_eps=1e-6,
                                          **({} if hidden_act == "gelu_pytorch_tanh" else {"hidden_act": hidden_act})))
    t = cfg.text_config
    assert t.is_gemma and t.hidden_act == hidden_act and t.head_dim * t.num_attention_heads != t.hidden_size
    hf = GemmaForCausalLM(GemmaConfig(hidden_size=96, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                                      num_key_value_heads=2, head_dim=32, vocab_size=160, rms_norm_eps=1e-6, rope_theta=t.rope_theta,
                                      max_position_embeddings=t.max_position_embeddings, hidden_act=hidden_act,
                                      attn_implementation="eager")).eval()
    assert type(hf.model.layers[0].mlp.act_fn).__name__ == ("GELUTanh" if hidden_act == "gelu_pytorch_tanh" else "GELUActivation"), \
        type(hf.model.layers[0].mlp.act_fn)
    sd = random_state_dict(cfg, seed=4)
    assert "language_model.lm_head.weight" not in sd                      # tied: the head is the embedding matrix
    llm_sd = {k[len("language_model."):]: v for k, v in sd.items() if k.startswith("language_model.")}
    missing, unexpected = hf.load_state_dict(llm_sd, strict=False)
    assert not unexpected and missing == ["lm_head.weight"]
    hf.tie_weights()
    torch.manual_seed(0)
    B, T = 2, 19
    emb = torch.randn(B, T, 96) * 0.1
    labels = torch.randint(0, 160, (B, T))
    labels[:, :9] = -100
    am = torch.ones(B, T, dtype=torch.long)
    am[1, -4:] = 0
    with torch.no_grad():
        out = hf(inputs_embeds=emb * (96 ** 0.5), attention_mask=am, labels=labels)
        logits = O.llama_ref(sd, cfg, emb, am)
        loss = O.causal_lm_loss_ref(logits, labels)
    keep = am.bool()
    np.testing.assert_allclose(logits[keep].numpy(), out.logits[keep].numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(loss.item(), out.loss.item(), rtol=1e-5)
    # bf16: the normalizer is rounded to the model dtype before it multiplies (4.51.3: torch.tensor(sqrt(H), dtype=...))
    assert float(torch.tensor(3072 ** 0.5, dtype=torch.bfloat16)) == 55.5


@pytest.mark.parametrize("family", ["qwen3", "qwen2"])
def test_qwen_backbones_match_hf_blocks(family):
    """The reference's v0.6 recipe trains on Qwen/Qwen3-32B (ultravox/training/configs/v0.6_config_qwen3_32b.yaml), reached through
    the same AutoModelForCausalLM call as Llama (ultravox_model.py:499-526).  [3P] check: the oracle's qwen3 flavour (RMSNorm over
    head_dim on every q / k head before RoPE, head_dim independent of hidden_size / heads) and qwen2 flavour (q / k / v biases)
    == the installed HF Qwen3ForCausalLM / Qwen2ForCausalLM on the same weights: logits, loss, and the gradient that reaches
    inputs_embeds (what the adapter-training path back-propagates through the frozen LLM)."""
    import transformers
    kw = dict(hidden_size=96, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
              vocab_size=160, rms_norm_eps=1e-6, rope_theta=1000000.0, max_position_embeddings=512)
    if family == "qwen3":
        kw["head_dim"] = 32
    cfg = UltravoxConfig(audio_config=TINY["audio_config"], hidden_size=64, text_config=dict(model_type=family, **kw))
    t = cfg.text_config
    assert t.has_qk_norm == (family == "qwen3") and t.has_qkv_bias == (family == "qwen2") and t.hidden_act == "silu"
    assert (t.head_dim * t.num_attention_heads != t.hidden_size) == (family == "qwen3")
    Cfg, LM = ((transformers.Qwen3Config, transformers.Qwen3ForCausalLM) if family == "qwen3"
               else (transformers.Qwen2Config, transformers.Qwen2ForCausalLM))
    hf = LM(Cfg(**kw, tie_word_embeddings=False, attn_implementation="eager")).eval()
    sd = random_state_dict(cfg, seed=6)
    llm_sd = {k[len("language_model."):]: v for k, v in sd.items() if k.startswith("language_model.")}
    missing, unexpected = hf.load_state_dict(llm_sd, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    extra = [k for k in llm_sd if k.endswith(("q_norm.weight", "k_norm.weight", "_proj.bias"))]
    assert len(extra) == (4 if family == "qwen3" else 6)                  # the family's extras exist and were consumed
    torch.manual_seed(0)
    B, T = 2, 19
    labels = torch.randint(0, 160, (B, T))
    labels[:, :9] = -100
    am = torch.ones(B, T, dtype=torch.long)
    am[1, -4:] = 0
    emb_hf = (torch.randn(B, T, 96) * 0.1).requires_grad_(True)
    emb_or = emb_hf.detach().clone().requires_grad_(True)
    out = hf(inputs_embeds=emb_hf, attention_mask=am, labels=labels)
    out.loss.backward()
    logits = O.llama_ref(sd, cfg, emb_or, am)
    loss = O.causal_lm_loss_ref(logits, labels)
    loss.backward()
    keep = am.bool()
    np.testing.assert_allclose(logits.detach()[keep].numpy(), out.logits.detach()[keep].numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(loss.item(), out.loss.item(), rtol=1e-5)
    np.testing.assert_allclose(emb_or.grad[keep].numpy(), emb_hf.grad[keep].numpy(), rtol=2e-4, atol=1e-7)


@pytest.mark.parametrize("window", [7, None])
def test_mistral_backbone_matches_hf_blocks(window):
    """`text_config` "can be any of LlamaConfig or MistralConfig" (ultravox_config.py:68; README.md:27 "trained versions on Llama 3, Mistral, and
    Gemma"), reached through the same AutoModelForCausalLM call (ultravox_model.py:499-526).  [3P] check: the oracle's mistral flavour - a Llama
    block, every layer behind the sliding-window causal mask when config.sliding_window is set (Mistral-7B-v0.1), plain causal when it is null
    (v0.2 / v0.3 / Nemo) - == the installed HF MistralForCausalLM on the same weights with a window SHORTER than the sequence: logits, loss,
    the gradient reaching inputs_embeds; and that the two settings differ (the window is live)."""
    import transformers
    kw = dict(hidden_size=96, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
              vocab_size=160, rms_norm_eps=1e-5, max_position_embeddings=512)
    hf_cfg = transformers.MistralConfig(**kw, sliding_window=window, tie_word_embeddings=False, attn_implementation="eager")
    cfg = UltravoxConfig(audio_config=TINY["audio_config"], hidden_size=64, text_config=hf_cfg)      # the HF config object, as the reference passes it
    t = cfg.text_config
    assert t.model_type == "mistral" and t.head_dim == 24 and t.rope_theta == 10000.0 and t.hidden_act == "silu"
    assert t.window_layers == ([1, 1] if window else None) and (t.sliding_window or None) == window
    # a config.json dict with an explicit null means "no window", not the family default of 4096
    assert UltravoxConfig(text_config=dict(model_type="mistral", sliding_window=None, **kw)).text_config.window_layers is None
    assert UltravoxConfig(text_config=dict(model_type="mistral", **kw)).text_config.sliding_window == 4096
    hf = transformers.MistralForCausalLM(hf_cfg).eval()
    sd = random_state_dict(cfg, seed=8)
    llm_sd = {k[len("language_model."):]: v for k, v in sd.items() if k.startswith("language_model.")}
    missing, unexpected = hf.load_state_dict(llm_sd, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    torch.manual_seed(0)
    B, T = 2, 19
    labels = torch.randint(0, 160, (B, T))
    labels[:, :9] = -100
    am = torch.ones(B, T, dtype=torch.long)
    am[1, -4:] = 0
    emb_hf = (torch.randn(B, T, 96) * 0.1).requires_grad_(True)
    emb_or = emb_hf.detach().clone().requires_grad_(True)
    out = hf(inputs_embeds=emb_hf, attention_mask=am, labels=labels)
    out.loss.backward()
    logits = O.llama_ref(sd, cfg, emb_or, am)
    loss = O.causal_lm_loss_ref(logits, labels)
    loss.backward()
    keep = am.bool()
    np.testing.assert_allclose(logits.detach()[keep].numpy(), out.logits.detach()[keep].numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(loss.item(), out.loss.item(), rtol=1e-5)
    np.testing.assert_allclose(emb_or.grad[keep].numpy(), emb_hf.grad[keep].numpy(), rtol=2e-4, atol=1e-7)
    if window:      # the same weights without the window give other logits beyond position `window`
        plain = O.llama_ref(sd, UltravoxConfig(audio_config=TINY["audio_config"], hidden_size=64,
                                               text_config=dict(model_type="mistral", sliding_window=None, **kw)), emb_or.detach(), am)
        assert (plain[:, :window] - logits.detach()[:, :window]).abs().max().item() < 1e-6
        assert (plain[0, window:] - logits.detach()[0, window:]).abs().max().item() > 1e-3


def test_gemma3_backbone_matches_hf_blocks():
    """The reference's other v0.6 recipe trains on google/gemma-3-27b-it (ultravox/training/configs/v0.6_config_gemma3_27b.yaml).
    [3P] check: the oracle's gemma3 flavour - four Gemma norms per layer (post-norms before each residual add), q / k norms over
    head_dim, query_pre_attn_scalar scaling, sliding-window layers with their own un-scaled rotary table and the global layers with
    linear rope scaling, GeGLU, tied head, the sqrt(hidden) scale inside the EMBEDDING module - == the installed HF
    Gemma3ForCausalLM on the same weights, with a window shorter than the sequence: logits, loss, the gradient reaching
    inputs_embeds, and the scaled embedding lookup."""
    import transformers
    kw = dict(hidden_size=96, intermediate_size=256, num_hidden_layers=7, num_attention_heads=4, num_key_value_heads=2, head_dim=32,
              vocab_size=160, rms_norm_eps=1e-6, query_pre_attn_scalar=24, sliding_window=8, max_position_embeddings=512)
    hfc = transformers.Gemma3TextConfig(**kw, rope_parameters={"sliding_attention": {"rope_type": "default", "rope_theta": 10000.0},
                                                               "full_attention": {"rope_type": "linear", "factor": 8.0, "rope_theta": 1000000.0}},
                                        attn_implementation="eager")
    cfg = UltravoxConfig(audio_config=TINY["audio_config"], hidden_size=64, text_config=hfc)
    t = cfg.text_config
    assert t.is_gemma3 and t.has_qk_norm and t.ties_head and t.rope_scaling == {"rope_type": "linear", "factor": 8.0}
    assert t.rope_local_base_freq == 10000.0 and t.layer_types.count("full_attention") == 1 and t.sliding_window == 8
    hf = transformers.Gemma3ForCausalLM(hfc).eval()
    sd = random_state_dict(cfg, seed=8)
    llm_sd = {k[len("language_model."):]: v for k, v in sd.items() if k.startswith("language_model.")}
    missing, unexpected = hf.load_state_dict(llm_sd, strict=False)
    assert not unexpected and missing == ["lm_head.weight"]
    hf.tie_weights()
    torch.manual_seed(0)
    B, T = 2, 21                                        # > sliding_window: the local layers really mask
    labels = torch.randint(0, 160, (B, T))
    labels[:, :9] = -100
    am = torch.ones(B, T, dtype=torch.long)
    am[1, -4:] = 0
    emb_hf = (torch.randn(B, T, 96) * 0.1).requires_grad_(True)
    emb_or = emb_hf.detach().clone().requires_grad_(True)
    out = hf(inputs_embeds=emb_hf, attention_mask=am, labels=labels)
    out.loss.backward()
    logits = O.llama_ref(sd, cfg, emb_or, am)
    loss = O.causal_lm_loss_ref(logits, labels)
    loss.backward()
    keep = am.bool()
    np.testing.assert_allclose(logits.detach()[keep].numpy(), out.logits.detach()[keep].numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(loss.item(), out.loss.item(), rtol=1e-5)
    np.testing.assert_allclose(emb_or.grad[keep].numpy(), emb_hf.grad[keep].numpy(), rtol=2e-4, atol=1e-7)
    ids = torch.randint(0, 160, (2, 5))
    np.testing.assert_allclose(O.OracleModel(cfg, sd, dtype=torch.float32).embed(ids).numpy(), hf.get_input_embeddings()(ids).detach().numpy(), rtol=1e-6)
    # a window that covers the sequence is plain causal attention (what the HIP training path requires: T <= sliding_window)
    wide = UltravoxConfig(audio_config=TINY["audio_config"], hidden_size=64, text_config=dict(hfc.to_dict(), sliding_window=64))
    with torch.no_grad():
        a = O.llama_ref(sd, wide, emb_or.detach(), am)
        hf_w = transformers.Gemma3ForCausalLM(transformers.Gemma3TextConfig(**{**kw, "sliding_window": 64}, rope_parameters=hfc.rope_parameters,
                                                                            attn_implementation="eager")).eval()
        hf_w.load_state_dict(llm_sd, strict=False); hf_w.tie_weights()
        b = hf_w(inputs_embeds=emb_or.detach(), attention_mask=am).logits
    np.testing.assert_allclose(a[keep].numpy(), b[keep].numpy(), rtol=1e-4, atol=2e-5)


W2V_TINY = {"model_type": "wav2vec2", "hidden_size": 64, "num_hidden_layers": 2, "num_attention_heads": 2, "intermediate_size": 128,
            "conv_dim": [64] * 7, "num_conv_pos_embeddings": 16, "num_conv_pos_embedding_groups": 4}


@pytest.mark.parametrize("norm,bias,stable", [("group", False, False), ("layer", True, True), ("layer", False, False), ("group", True, True)])
def test_wav2vec2_tower_matches_hf_model(norm, bias, stable):
    """BASELINE config 5's tower.  [3P] check: the restated Wav2Vec2Model.forward (conv stem, feature projection, weight-normed grouped
    positional conv, encoder layers) == the installed HF Wav2Vec2Model on the same weights - the group-norm family (wav2vec2-large-960h:
    GroupNorm after the first conv, bias-free convs, post-LN layers), the layer-norm family (round 5; the -lv60 checkpoints: LayerNorm after
    every conv, conv biases, pre-LN "stable" layers with the encoder's layer_norm at the end) and the two mixed settings of the three
    independent Wav2Vec2Config switches."""
    from transformers import Wav2Vec2Config, Wav2Vec2Model
    cfg = UltravoxConfig(audio_config={**W2V_TINY, "feat_extract_norm": norm, "conv_bias": bias, "do_stable_layer_norm": stable},
                         text_config=TINY["text_config"], hidden_size=64)
    a = cfg.audio_config
    hf = Wav2Vec2Model(Wav2Vec2Config(hidden_size=64, num_hidden_layers=2, num_attention_heads=2, intermediate_size=128,
                                      conv_dim=[64] * 7, num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4,
                                      feat_extract_norm=norm, conv_bias=bias, do_stable_layer_norm=stable,
                                      attn_implementation="eager")).eval()
    sd = random_state_dict(cfg, seed=2)
    missing, unexpected = hf.load_state_dict({k[len("audio_tower."):]: v for k, v in sd.items() if k.startswith("audio_tower.")}, strict=False)
    assert not unexpected and not missing      # (masked_spec_embed, which no kernel reads, is part of the random tower since round 6)
    torch.manual_seed(1)
    x = O.wav2vec2_normalize_ref(torch.randn(2, 6000) * 0.1 + 0.02)
    with torch.no_grad():
        want = hf(x).last_hidden_state
        got = O.wav2vec2_encoder_ref(sd, cfg, x)
    assert got.shape == want.shape == (2, a.feat_extract_output_length(6000), 64)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-4, atol=2e-5)
    assert hf._get_feat_extract_output_lengths(480000) == a.feat_extract_output_length(480000) == 1499


@pytest.mark.parametrize("case", ["post_ln_default", "stable_all", "post_ln_all"])
def test_wav2vec2_tower_under_apply_lora_matches_the_reference(golden_dir, case):
    """`apply_lora(audio_tower, audio_model_lora_config)` wraps whatever AutoModel tower was loaded (ultravox_model.py:460-467, 690-709).  Fixture
    lora_w2v_reference.npz: the reference's apply_lora (tests/peft_stub.py) on an installed-HF Wav2Vec2Model - post-LN family with the default
    target_modules, both families with q / k / v / out_proj: adapted modules, key names, last_hidden_state, adapter gradients."""
    import json
    from ultravox_amd.weights import init_lora_state_dict, lora_targets, w2v_lora_key
    z = np.load(os.path.join(golden_dir, "lora_w2v_reference.npz"))
    meta = json.load(open(os.path.join(golden_dir, "lora_w2v_reference.json")))
    cm = meta["cases"][case]
    cfg = UltravoxConfig(audio_config={**meta["w2v_tiny"], **cm["family"]}, text_config=meta["text_tiny"], hidden_size=64,
                         audio_model_lora_config=cm["lora_config"])
    assert lora_targets(cfg, "audio") == (("q_proj", "k_proj") if case == "post_ln_default" else ("q_proj", "k_proj", "v_proj", "out_proj"))
    names = {"audio_tower." + n for n in cm["trainable"]}
    init = init_lora_state_dict(cfg)
    assert set(init) == names and w2v_lora_key(1, "k_proj", "B") in names
    assert all(tuple(v.shape) == z[f"{case}.w." + k[len("audio_tower."):]].shape for k, v in init.items())
    sd = random_state_dict(cfg, seed=meta["seed"])
    for k in z.files:
        if k.startswith(case + ".w."):
            sd["audio_tower." + k[len(case) + 3:]] = torch.from_numpy(z[k]).clone().requires_grad_(True)
    lc = cm["lora_config"]
    y = O.wav2vec2_encoder_ref(sd, cfg, torch.from_numpy(z[f"{case}.x"]), lora={"scaling": lc["lora_alpha"] / lc["r"]})
    np.testing.assert_allclose(y.detach().numpy(), z[f"{case}.y"], rtol=1e-4, atol=2e-5)
    (y * torch.from_numpy(z[f"{case}.gy"])).sum().backward()
    for n in cm["trainable"]:
        np.testing.assert_allclose(sd["audio_tower." + n].grad.numpy(), z[f"{case}.g." + n], rtol=3e-4, atol=2e-5, err_msg=n)
    # without the adapters the tower gives another output (lora_B is non-zero in the fixture)
    with torch.no_grad():
        plain = O.wav2vec2_encoder_ref(sd, cfg, torch.from_numpy(z[f"{case}.x"]))
    assert (plain - y.detach()).abs().max().item() > 1e-3


def test_wav2vec2_feature_extractor_contract_matches_hf():
    from transformers import Wav2Vec2FeatureExtractor as HF
    from ultravox_amd.frontend import Wav2Vec2FeatureExtractor
    rng = np.random.RandomState(3)
    clips = [rng.randn(4000).astype(np.float32) * 0.3 + 0.1, rng.randn(2500).astype(np.float32)]
    want = HF(feature_size=1, sampling_rate=16000, padding_value=0.0, do_normalize=True, return_attention_mask=True)(
        clips, sampling_rate=16000, padding="longest", return_attention_mask=True, return_tensors="pt")
    got = Wav2Vec2FeatureExtractor()(clips, sampling_rate=16000, padding="longest", return_attention_mask=True)
    np.testing.assert_allclose(got["input_values"].numpy(), want["input_values"].numpy(), rtol=1e-5, atol=1e-6)
    assert torch.equal(got["attention_mask"].long(), want["attention_mask"].long())
    np.testing.assert_allclose(O.wav2vec2_normalize_ref(torch.from_numpy(clips[0])[None]).numpy(), want["input_values"][:1].numpy(),
                               rtol=1e-5, atol=1e-6)
    with pytest.raises(ValueError, match="sampling rate"):
        Wav2Vec2FeatureExtractor()(clips, sampling_rate=8000)


def load_forward_fixture(name):
    """tests/golden/forward_reference.npz (outputs of the REFERENCE forward) + the seeded weights / inputs it was run on
    (tests/forward_fixture_util.py) -> (cfg, state dict incl. a random audio tower, batch, tower output, expected)."""
    import json
    import os
    import forward_fixture_util as U
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.weights import random_state_dict
    here = os.path.join(os.path.dirname(__file__), "golden")
    z = np.load(os.path.join(here, "forward_reference.npz"))
    meta = json.load(open(os.path.join(here, "forward_reference.json")))["cases"][name]
    cfg = UltravoxConfig(**U.config_kwargs(name.startswith("ln_mid")))
    sd = random_state_dict(cfg, seed=1)
    for key in meta["weight_names"]:          # the reference's parameter names ARE this framework's state-dict keys
        assert key in sd, key
        sd[key] = U.param(key, sd[key].shape)
    assert {k for k in sd if k.startswith(("multi_modal_projector.", "language_model."))} == set(meta["weight_names"])
    exp = {"logits": torch.from_numpy(z[f"{name}.logits"]), "loss": float(z[f"{name}.loss"]),
           "grads": {k[len(name) + 3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(name + ".g.")}}
    return cfg, sd, U.batch(mixed=name.endswith("_mixed")), U.tower_output(), exp


def load_real_tower_fixture():
    """tests/golden/real_tower_reference.npz: outputs of the REFERENCE ModifiedWhisperEncoder.forward itself
    (ultravox_model.py:865-994, masks :915-936 built by the reference, latency mask by its init_latency_mask :834-863) and of
    the REFERENCE UltravoxModel.forward + backward with that tower in place; weights / inputs are the seeded tensors of
    tests/forward_fixture_util.py.  -> (arrays, meta, make(kind, latency) -> (cfg, state dict))."""
    import json
    import os
    import forward_fixture_util as U
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.weights import random_state_dict
    here = os.path.join(os.path.dirname(__file__), "golden")
    z = np.load(os.path.join(here, "real_tower_reference.npz"))
    meta = json.load(open(os.path.join(here, "real_tower_reference.json")))

    def make(names, latency):
        cfg = UltravoxConfig(**U.real_config_kwargs(True, latency))
        sd = random_state_dict(cfg, seed=1)
        for key in names:                         # the reference's parameter names ARE this framework's state-dict keys
            assert key in sd, key
            sd[key] = U.param(key, sd[key].shape)
        return cfg, sd
    return z, meta, make


@pytest.mark.parametrize("name", ["key_padding", "latency_and_padding", "latency_only", "odd_frames"])
def test_oracle_encoder_matches_the_reference_encoder_forward(name):
    """whisper_encoder_ref against a RUN OF THE REFERENCE's ModifiedWhisperEncoder.forward (fixture: make_golden.py
    `real_tower_cases`) - conv stem, positional slice, the reference-built key-padding mask, its latency mask, the merge of
    the two, final LayerNorm.  Rows of padded keys are compared too: with a finite finfo.min mask they are well defined."""
    import forward_fixture_util as U
    from oracle import reference_cpu as O
    z, meta, make = load_real_tower_fixture()
    case = meta["encoder_cases"][name]
    cfg, sd = make(case["weight_names"], case["audio_latency_block_size"])
    assert set(case["weight_names"]) == {k for k in sd if k.startswith("audio_tower.")}
    n = 3 if case["audio_len"] is None else len(case["audio_len"])
    lens = None if case["audio_len"] is None else torch.tensor(case["audio_len"])
    got = O.whisper_encoder_ref(sd, cfg, U.mel(n, case["frames"]), lens)
    np.testing.assert_allclose(got.numpy(), z[f"enc.{name}"], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("name", ["real_tower", "real_tower_latency"])
def test_oracle_forward_matches_the_reference_forward_with_its_own_tower(name):
    """OracleModel end to end (mel -> encoder -> projector -> merge -> Llama -> loss -> projector gradients) against the
    REFERENCE UltravoxModel.forward + loss.backward() with NOTHING stubbed (its ModifiedWhisperEncoder.forward included)."""
    import forward_fixture_util as U
    from oracle import reference_cpu as O
    z, meta, make = load_real_tower_fixture()
    case = meta["model_cases"][name]
    cfg, sd = make(case["weight_names"], case["audio_latency_block_size"])
    assert set(case["weight_names"]) == set(sd)
    batch = U.batch()
    oracle = O.OracleModel(cfg, sd)
    mel = U.mel(U.N_AUDIO, 3000)
    tower = O.whisper_encoder_ref(sd, cfg, mel, batch["audio_lens"])
    np.testing.assert_allclose(tower[:, :80].numpy(), z[f"{name}.tower_rows"], rtol=1e-4, atol=2e-5)
    out = oracle.forward(audio_values=mel, **batch)
    keep = batch["attention_mask"].bool()
    assert (out["logits"].detach()[keep] - torch.from_numpy(z[f"{name}.logits"])[keep]).abs().max().item() < 2e-5
    assert abs(out["loss"].item() - float(z[f"{name}.loss"])) < 1e-6
    out["loss"].backward()
    for k in z.files:
        if k.startswith(name + ".g."):
            key = k[len(name) + 3:]
            np.testing.assert_allclose(oracle.sd[key].grad.numpy(), z[k], rtol=2e-4, atol=2e-6, err_msg=key)


@pytest.mark.parametrize("name", ["ln_mid", "ln_post", "ln_mid_mixed"])
def test_oracle_forward_matches_the_reference_forward_end_to_end(name):
    """OracleModel.forward + backward against the REFERENCE UltravoxModel.forward run end to end (fixture: make_golden.py
    `forward_cases`): merge of two audio items into one sample, token_len truncation, left / right padding, loss shift and
    ignore index, projector gradients - the glue the piecewise fixtures do not cover."""
    from oracle import reference_cpu as O
    cfg, sd, batch, enc, exp = load_forward_fixture(name)
    oracle = O.OracleModel(cfg, sd)
    out = oracle.forward(audio_values=torch.zeros(len(enc), 80, 3000), tower_output=enc, **batch)
    keep = batch["attention_mask"].bool()
    assert (out["logits"].detach()[keep] - exp["logits"][keep]).abs().max().item() < 2e-5
    assert abs(out["loss"].item() - exp["loss"]) < 1e-6
    out["loss"].backward()
    for k, g in exp["grads"].items():
        np.testing.assert_allclose(oracle.sd[k].grad.numpy(), g.numpy(), rtol=2e-4, atol=2e-6, err_msg=k)


def test_oracle_generate_matches_the_reference_generate():
    """OracleModel.generate_greedy against the REFERENCE UltravoxModel.generate (fixture generate_reference.json: HF greedy
    search on the seeded tiny model): audio merged once, a left-padded prompt next to an unpadded one (mask-based position
    ids), EOS stops one row early and pads it."""
    import json
    import os
    import forward_fixture_util as U
    from oracle import reference_cpu as O
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "generate_reference.json")))
    cfg, sd, _, enc, _ = load_forward_fixture("ln_mid")
    oracle = O.OracleModel(cfg, sd)
    b = U.generate_batch()
    T = fx["prompt_len"]
    kw = dict(audio_values=torch.zeros(3, 80, 3000), tower_output=enc[U.GEN_AUDIO_ROWS], **b)
    free = oracle.generate_greedy(10, eos_token_id=-1, pad_token_id=fx["pad_token_id"], **kw)
    assert free[:, T:].tolist() == fx["free"]
    stop = oracle.generate_greedy(10, eos_token_id=fx["eos"], pad_token_id=fx["pad_token_id"], **kw)
    got = stop[:, T:].tolist()
    want = [row[: len(got[0])] for row in fx["with_eos"]]      # (the oracle may stop the loop once every row has finished)
    assert [r + [fx["pad_token_id"]] * (10 - len(r)) for r in got] == fx["with_eos"] and want == got


@pytest.mark.parametrize("tag", ["t2_eot1", "t1_eot0"])
def test_oracle_kl_step_matches_the_reference_forward_in_training_mode(tag):
    """OracleModel.forward(kl=...) + backward against the REFERENCE forward in training mode under KL_Divergence (fixture
    kl_forward_reference.npz): the teacher pass over alt_* inside the model, prediction / end-of-turn masks, batchmean."""
    import json
    import os
    import forward_fixture_util as U
    from oracle import reference_cpu as O
    here = os.path.join(os.path.dirname(__file__), "golden")
    z = np.load(os.path.join(here, "kl_forward_reference.npz"))
    meta = json.load(open(os.path.join(here, "kl_forward_reference.json")))[tag]
    cfg, sd, batch, enc, _ = load_forward_fixture("ln_mid")
    oracle = O.OracleModel(cfg, sd)
    out = oracle.forward(audio_values=torch.zeros(len(enc), 80, 3000), tower_output=enc, **batch, **U.alt_batch(),
                         kl={"temperature": meta["kl_temperature"], "eot_loss_weight": meta["eot_loss_weight"]})
    assert abs(out["loss"].item() - float(z[f"{tag}.loss"])) < 1e-6
    out["loss"].backward()
    for k in z.files:
        if k.startswith(tag + ".g."):
            np.testing.assert_allclose(oracle.sd[k[len(tag) + 3:]].grad.numpy(), z[k], rtol=3e-4, atol=1e-7, err_msg=k)


def load_lora_forward_fixture(stem="lora_forward_reference"):
    """tests/golden/lora_forward_reference.* (or kl_lora_forward_reference.*) + the seeded weights (by the REFERENCE's parameter names: under peft the base
    weights are `language_model.base_model.model.<...>.base_layer.weight`; this framework keeps HF names for them)."""
    import json
    import os
    import forward_fixture_util as U
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.weights import random_state_dict
    here = os.path.join(os.path.dirname(__file__), "golden")
    z = np.load(os.path.join(here, stem + ".npz"))
    meta = json.load(open(os.path.join(here, stem + ".json")))
    from ultravox_amd.weights import init_lora_state_dict
    cfg = UltravoxConfig(**U.config_kwargs(True), text_model_lora_config=meta["lora_config"])
    sd = random_state_dict(cfg, seed=1)
    sd.update(init_lora_state_dict(cfg))          # peft's key names for the adapter matrices (pinned by lora_reference.json)
    seen = set()
    for ref in meta["weight_names"]:
        key = ref if ".lora_" in ref else ref.replace("language_model.base_model.model.", "language_model.").replace(".base_layer", "")
        assert key in sd, (ref, key)
        sd[key] = U.param(ref, sd[key].shape)
        seen.add(key)
    assert {k for k in sd if k.startswith(("multi_modal_projector.", "language_model."))} == seen
    exp = {"logits": torch.from_numpy(z["logits"]) if "logits" in z.files else None, "loss": float(z["loss"]), "meta": meta,
           "grads": {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("g.")}}
    assert sorted(exp["grads"]) == sorted(meta["trainable"])
    return cfg, sd, U.batch(), U.tower_output(), exp


def test_oracle_llm_lora_step_matches_the_reference_model_with_text_lora():
    """OracleModel (text_model_lora_config r = 4) against the REFERENCE model with apply_lora on the LLM (fixture
    lora_forward_reference.npz, peft restated by tests/peft_stub.py): logits, loss, projector and adapter gradients."""
    from oracle import reference_cpu as O
    cfg, sd, batch, enc, exp = load_lora_forward_fixture()
    oracle = O.OracleModel(cfg, sd)
    assert sorted(oracle.trainable) == sorted(exp["grads"])
    out = oracle.forward(audio_values=torch.zeros(len(enc), 80, 3000), tower_output=enc, **batch)
    keep = batch["attention_mask"].bool()
    assert (out["logits"].detach()[keep] - exp["logits"][keep]).abs().max().item() < 3e-5
    assert abs(out["loss"].item() - exp["loss"]) < 1e-6
    out["loss"].backward()
    for k, g in exp["grads"].items():
        np.testing.assert_allclose(oracle.sd[k].grad.numpy(), g.numpy(), rtol=3e-4, atol=3e-6, err_msg=k)


def test_oracle_kl_step_under_llm_lora_runs_the_teacher_through_the_adapted_model():
    """KL distillation with text_model_lora_config r = 4 against the REFERENCE (fixture kl_lora_forward_reference.npz): the teacher
    pass of _compute_kl_loss is `self.language_model.forward` (ultravox_model.py:212-222) - the peft-wrapped model, adapters
    ACTIVE, no_grad.  A teacher with the adapters switched off gives a different loss (asserted: the fixture tells them apart)."""
    import forward_fixture_util as U
    from oracle import reference_cpu as O
    cfg, sd, batch, enc, exp = load_lora_forward_fixture("kl_lora_forward_reference")
    kl = {"temperature": exp["meta"]["kl_temperature"], "eot_loss_weight": exp["meta"]["eot_loss_weight"]}
    oracle = O.OracleModel(cfg, sd)
    out = oracle.forward(audio_values=torch.zeros(len(enc), 80, 3000), tower_output=enc, **batch, **U.alt_batch(), kl=kl)
    assert abs(out["loss"].item() - exp["loss"]) < 1e-6
    out["loss"].backward()
    for k, g in exp["grads"].items():
        np.testing.assert_allclose(oracle.sd[k].grad.numpy(), g.numpy(), rtol=3e-4, atol=3e-6, err_msg=k)
    plain_teacher = O.llama_ref(oracle.sd, cfg, oracle.embed(U.alt_batch()["alt_input_ids"]), U.alt_batch()["alt_attention_mask"])
    wrong = O.kl_loss_ref(out["logits"].detach(), batch["labels"], plain_teacher.detach(), U.alt_batch()["alt_labels"],
                          kl["temperature"], kl["eot_loss_weight"])
    assert abs(wrong.item() - exp["loss"]) > 1e-4
