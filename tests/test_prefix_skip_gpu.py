"""uvx_llm_bwd_train_from (ABI 19): the LLM backward below the first audio token is skipped - the text prefix feeds later positions only
(causal mask), so nothing the adapter training updates is reachable from it (reference: the frozen LLM of apply_lora r = 0,
ultravox_model.py:697-703; the only consumer of d inputs_embeds is the scatter back to the audio rows, _prepare_audio_embeds :354-396).
The gradient tensors below the last layer are row-compacted to the positions >= first_pos (rounded down to a multiple of 16); every remaining
row goes through the same arithmetic, so the loss, the projector gradients and the kept rows of d inputs_embeds are BIT-identical to the
full backward."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

# head_dim 128 (the fused attention backward the compacted form needs), GQA 4 : 2
HD128 = dict(
    audio_config=dict(d_model=128, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=256, num_mel_bins=80, max_source_positions=1500),
    text_config=dict(hidden_size=512, intermediate_size=1024, num_hidden_layers=3, num_attention_heads=4, num_key_value_heads=2, vocab_size=512,
                     rope_theta=10000.0, max_position_embeddings=512, eos_token_id=2),
    hidden_size=256, stack_factor=8, projector_ln_mid=True)


def _setup(B=3, audio_start=21, n_text=60, seed=3, pad=None, **cfg_kw):
    from oracle.reference_cpu import synthetic_batch
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.frontend import WhisperFeatureExtractor
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import random_state_dict
    cfg = UltravoxConfig(**dict(HD128, **cfg_kw))
    sd = {k: v.bfloat16() for k, v in random_state_dict(cfg, seed=seed).items()}
    if cfg_kw.get("audio_model_lora_config"):      # peft initialises lora_B to zero: random values so that every adapter gradient is non-zero
        from ultravox_amd.weights import init_lora_state_dict
        sd.update(init_lora_state_dict(cfg, seed=seed, dtype=torch.bfloat16, random_b=True))
    b = synthetic_batch(cfg, B, 3.0, n_text=n_text, audio_start=audio_start, n_supervised=12)
    if pad:      # left / right padding in the attention mask (the labels of padded positions are ignored)
        for i, (lo, hi) in pad.items():
            b["attention_mask"][i, :lo] = 0
            b["labels"][i, :lo] = -100
            if hi:
                b["attention_mask"][i, -hi:] = 0
                b["labels"][i, -hi:] = -100
    pcm = b.pop("pcm")
    mel = WhisperFeatureExtractor(cfg.audio_config.num_mel_bins).logmel_device(pcm.to(DEV))
    model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=torch.bfloat16)
    model.train()
    return cfg, model, b, mel


def _step(model, b, mel, skip, on_device=True):
    model.skip_prefix_backward = skip
    gb = {k: (v.to(DEV) if on_device or k != "audio_token_start_idx" else v) for k, v in b.items()}
    loss = model.forward_backward(audio_values=mel, **gb)
    torch.cuda.synchronize()
    return loss.clone(), {k: v.clone() for k, v in model.projector_grads().items()}, model._last_d_embeds.clone()


@pytest.mark.parametrize("case", ["uniform", "ragged", "padded", "one_layer", "unaligned_T"])
def test_backward_from_the_first_audio_token_is_bit_identical(case):
    kw, cfg_kw = {}, {}
    if case == "padded":
        kw = dict(pad={0: (5, 0), 2: (0, 9)}, audio_start=26)
    if case == "one_layer":
        cfg_kw = dict(text_config=dict(HD128["text_config"], num_hidden_layers=1))
    if case == "unaligned_T":
        kw = dict(n_text=71, audio_start=40, B=2)
    cfg, model, b, mel = _setup(**kw, **cfg_kw)
    if case == "ragged":      # the audio starts at another position in every sample: the batch minimum counts
        ids, st = b["input_ids"], b["audio_token_start_idx"]
        na = int(b["audio_token_len"][0])
        for i, s in enumerate([21, 37, 50]):      # move sample i's placeholder run to position s
            text = torch.cat([ids[i, :21], ids[i, 21 + na:]])
            ids[i] = torch.cat([text[:s], ids[i, 21:21 + na], text[s:]])
            st[i] = s
        b["labels"] = ids.clone()
        b["labels"][:, : ids.shape[1] - 12] = -100
    first = int(b["audio_token_start_idx"].min())
    s16 = first // 16 * 16
    assert s16 >= 16
    l0, g0, d0 = _step(model, b, mel, skip=False)
    l1, g1, d1 = _step(model, b, mel, skip=True)
    assert torch.equal(l0, l1)
    for k in g0:
        assert g0[k].abs().max().item() > 0
        assert torch.equal(g0[k], g1[k]), (k, int((g0[k] != g1[k]).sum()))
    assert torch.equal(d0[:, s16:], d1[:, s16:])
    assert d0[:, :s16].abs().max().item() > 0 and d1[:, :s16].abs().max().item() == 0      # the full backward did write the prefix rows
    # the host-tensor route (collator output on the CPU: no read-back) takes the same path
    l2, g2, d2 = _step(model, b, mel, skip=True, on_device=False)
    assert torch.equal(l1, l2) and torch.equal(d1, d2)


def test_backward_from_a_position_below_16_is_the_full_backward():
    """first_pos < 16 (nothing to skip after rounding down to the 16-row tiles): the entry point IS uvx_llm_bwd_train - every row of d inputs_embeds is written."""
    cfg, model, b, mel = _setup(audio_start=9)
    _, _, d = _step(model, b, mel, skip=True)
    assert d[:, :9].abs().max().item() > 0


@pytest.mark.parametrize("case", ["long", "head_dim_64", "window"])
def test_backward_from_the_first_audio_token_on_the_unfused_attention_kernels(case):
    """Sequences beyond the fused attention backward's 320 positions, head_dim 64 and sliding-window layers take the dQ + dK/dV kernel pair, which reads /
    writes the row-compacted gradients the same way (AttnBwdDesc::d_first)."""
    if case == "long":
        cfg, model, b, mel = _setup(audio_start=20, n_text=330, B=2)      # T = 330 + 19 audio tokens > 320
        s16 = 16
    elif case == "window":      # Mistral: a 64-position window on every layer, shorter than the sequence
        cfg, model, b, mel = _setup(audio_start=37, n_text=90, B=2, text_config=dict(HD128["text_config"], model_type="mistral", sliding_window=64))
        assert model._c.llm_window == 64
        s16 = 32
    else:
        from test_model_gpu import SMALL
        from oracle.reference_cpu import synthetic_batch
        from ultravox_amd.config import UltravoxConfig
        from ultravox_amd.frontend import WhisperFeatureExtractor
        from ultravox_amd.model import UltravoxModel
        from ultravox_amd.weights import random_state_dict
        cfg = UltravoxConfig(**SMALL)
        sd = {k: v.bfloat16() for k, v in random_state_dict(cfg, seed=5).items()}
        b = synthetic_batch(cfg, 2, 3.0, n_text=60, audio_start=33, n_supervised=12)
        mel = WhisperFeatureExtractor(80).logmel_device(b.pop("pcm").to(DEV))
        model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=torch.bfloat16)
        model.train()
        s16 = 32
    l0, g0, d0 = _step(model, b, mel, skip=False)
    l1, g1, d1 = _step(model, b, mel, skip=True)
    assert torch.equal(l0, l1)
    for k in g0:
        assert g0[k].abs().max().item() > 0 and torch.equal(g0[k], g1[k]), (k, int((g0[k] != g1[k]).sum()))
    assert torch.equal(d0[:, s16:], d1[:, s16:]) and d1[:, :s16].abs().max().item() == 0 and d0[:, :s16].abs().max().item() > 0


def test_trainer_steps_with_and_without_the_prefix_backward_give_the_same_weights():
    from ultravox_amd.model import UltravoxTrainer
    out = []
    for skip in (False, True):
        cfg, model, b, mel = _setup(seed=9)
        model.skip_prefix_backward = skip
        tr = UltravoxTrainer(model, lr=2e-3, master_weights=True)
        gb = {k: v.to(DEV) for k, v in b.items()}
        losses = [tr.train_step(audio_values=mel, **gb).item() for _ in range(3)]
        torch.cuda.synchronize()
        out.append((losses, {k: v.clone() for k, v in model.projector_state_dict().items()}))
    (la, wa), (lb, wb) = out
    assert la == lb and la[2] < la[0]
    for k in wa:
        assert torch.equal(wa[k], wb[k]), k


@pytest.mark.parametrize("audio_lora", [False, True])
def test_kl_step_backward_from_the_first_audio_token_is_bit_identical(audio_lora):
    """The KL recipe's student backward (uvx_llm_bwd_rows_from) - with and without the rank-r encoder adapters of the release configs, whose
    gradients also enter through the audio rows only: same loss, same projector / adapter gradients, bit for bit."""
    from test_kl_gpu import _alt_fields
    from ultravox_amd.config import LossConfig, LossFunction
    kw = dict(audio_model_lora_config={"r": 4, "lora_alpha": 8}) if audio_lora else {}
    cfg, model, b, mel = _setup(audio_start=21, **kw)
    model.set_loss_config(LossConfig(loss_function=LossFunction.KL_Divergence))
    b.update(_alt_fields(b, cfg, 21, 12))
    l0, g0, d0 = _step(model, b, mel, skip=False)
    l1, g1, d1 = _step(model, b, mel, skip=True)
    assert torch.equal(l0, l1)
    assert set(g0) == set(g1) and (not audio_lora or any("lora" in k for k in g0))
    for k in g0:
        assert g0[k].abs().max().item() > 0, k
        assert torch.equal(g0[k], g1[k]), (k, int((g0[k] != g1[k]).sum()))
    assert torch.equal(d0[:, 16:], d1[:, 16:]) and d1[:, :16].abs().max().item() == 0 and d0[:, :16].abs().max().item() > 0


@pytest.mark.parametrize("family", ["qwen3", "qwen2", "gemma3"])
def test_backward_from_the_first_audio_token_on_the_other_backbones(family):
    """The v0.6 recipes' backbones at head_dim 128: Qwen3 (per-head q / k norms: their backward reads the raw rows through the map), Qwen2 (q|k|v
    bias) and Gemma-3 (no compact last layer - the final norm's backward runs on every row and the kept rows move to the front; post norms on both
    branches, GeGLU, local layers whose window covers the sequence) - bit-identical to the full backward."""
    from oracle.reference_cpu import synthetic_batch
    from ultravox_amd.frontend import WhisperFeatureExtractor
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import random_state_dict
    if family == "gemma3":      # (window 512 > T: the local layers' window covers the sequence)
        from test_gemma3_gpu import _cfg
        cfg = _cfg(head_dim=128, layers=4)
    elif family == "qwen3":
        from test_qwen_gpu import _cfg
        cfg = _cfg(family, head_dim=128)
    else:      # qwen2: head_dim = hidden / heads
        from ultravox_amd.config import UltravoxConfig
        tc = dict(model_type="qwen2", hidden_size=512, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                  vocab_size=512, rms_norm_eps=1e-6, rope_theta=1000000.0, eos_token_id=1)
        cfg = UltravoxConfig(audio_config=dict(d_model=128, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=256), text_config=tc,
                             hidden_size=256, projector_ln_mid=True)
    sd = {k: v.bfloat16() for k, v in random_state_dict(cfg, seed=13).items()}
    b = synthetic_batch(cfg, 2, 3.0, n_text=70, audio_start=35, n_supervised=12)
    mel = WhisperFeatureExtractor(80).logmel_device(b.pop("pcm").to(DEV))
    model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=torch.bfloat16)
    model.train()
    l0, g0, d0 = _step(model, b, mel, skip=False)
    l1, g1, d1 = _step(model, b, mel, skip=True)
    assert torch.equal(l0, l1)
    for k in g0:
        assert g0[k].abs().max().item() > 0 and torch.equal(g0[k], g1[k]), (k, int((g0[k] != g1[k]).sum()))
    assert torch.equal(d0[:, 32:], d1[:, 32:]) and d1[:, :32].abs().max().item() == 0 and d0[:, :32].abs().max().item() > 0


def test_text_only_sample_with_labels_in_front_of_the_first_audio_token():
    """A batch that mixes audio samples with a text-only one (audio_batch_size 0) whose supervised positions lie BELOW the batch's first audio position: their
    gradient can only reach text rows, so they drop out of the compacted backward (compact_row_list sends them to an unused row) - the loss still counts
    them, and the projector gradients are those of the full backward, bit for bit."""
    cfg, model, b, mel = _setup(B=3, audio_start=37, n_text=70)
    T = b["input_ids"].shape[1]
    g = torch.Generator().manual_seed(5)
    b["input_ids"][1] = torch.randint(3, 500, (T,), generator=g)      # sample 1: text only (no placeholder run), supervised on positions 4 .. 19
    b["labels"][1] = -100
    b["labels"][1, 4:20] = b["input_ids"][1, 4:20]
    b["attention_mask"][1, 40:] = 0
    keep = torch.tensor([0, 2])
    for k in ("audio_token_start_idx", "audio_lens", "audio_token_len"):
        b[k] = b[k][keep]
    b["audio_batch_size"] = torch.tensor([1, 0, 1])
    mel = mel[keep.to(mel.device)]
    l0, g0, d0 = _step(model, b, mel, skip=False)
    l1, g1, d1 = _step(model, b, mel, skip=True)
    assert torch.equal(l0, l1)
    for k in g0:
        assert g0[k].abs().max().item() > 0 and torch.equal(g0[k], g1[k]), (k, int((g0[k] != g1[k]).sum()))
    assert torch.equal(d0[:, 32:], d1[:, 32:]) and d1[:, :32].abs().max().item() == 0
    assert d0[1, :20].abs().max().item() > 0      # the full backward did carry the text-only sample's gradient
