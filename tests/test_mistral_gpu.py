"""Mistral backbones (`text_config` "can be any of LlamaConfig or MistralConfig", ultravox_config.py:68; README.md:27 "Llama 3, Mistral, and Gemma"),
reached through the same AutoModelForCausalLM call as Llama (ultravox_model.py:499-526): a Llama block whose EVERY layer attends to the last
`sliding_window` positions when the config sets one ([3P] MistralModel.forward: create_sliding_window_causal_mask), plain causal attention when it
is null (v0.2 / v0.3 / Nemo).  The windowed attention kernels are the ones Gemma-3's local layers run (uvx_config_t.llm_window +
uvx_llm_weights_t.layer_local, here all ones); the oracle's mistral flavour is pinned to the installed HF MistralForCausalLM in
tests/test_oracle_pinning.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel_l2(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _cfg(window, head_dim=64, layers=3, **kw):
    from ultravox_amd.config import UltravoxConfig
    tc = dict(model_type="mistral", hidden_size=192, intermediate_size=512, num_hidden_layers=layers, num_attention_heads=4,
              num_key_value_heads=2, head_dim=head_dim, vocab_size=512, rms_norm_eps=1e-5, eos_token_id=2, sliding_window=window, rope_theta=10000.0)
    return UltravoxConfig(audio_config=dict(d_model=128, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=256),
                          text_config=tc, hidden_size=256, projector_ln_mid=True, **kw)


def _batch(cfg):
    from oracle.reference_cpu import logmel_ref, synthetic_batch
    b = synthetic_batch(cfg, 2, 2.0, n_text=24, audio_start=5, n_supervised=8)
    b["audio_values"] = logmel_ref(b.pop("pcm"), 80)
    b["attention_mask"][1, -3:] = 0
    b["labels"][1, -3:] = -100
    return b


@pytest.mark.parametrize("head_dim", [64, 128])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_mistral_train_step_beyond_the_sliding_window(dtype, head_dim):
    """37 positions behind a window of 16 on every layer: forward, loss and the projector gradients (through the windowed attention backward) against
    the oracle, which masks the window in full; and the window is live (a null window gives other logits)."""
    from oracle.reference_cpu import OracleModel
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import random_state_dict
    cfg = _cfg(16, head_dim)
    sd = {k: v.to(dtype) for k, v in random_state_dict(cfg, seed=71).items()}
    model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=dtype)
    assert model._llm["layer_local"] == [1, 1, 1] and model._c.llm_window == 16 and model._c.llm_flavor == 0
    oracle = OracleModel(cfg, sd, dtype=torch.float32)
    b = _batch(cfg)
    assert b["input_ids"].shape[1] == 37
    gb = {k: v.to(DEV) for k, v in b.items()}
    gb["audio_values"] = gb["audio_values"].to(dtype)
    ref, grads, _ = oracle.train_step({**b, "audio_values": b["audio_values"].to(dtype).float()})
    out = model.forward(**gb)
    model.train()
    loss = model.forward_backward(**gb)
    keep = b["attention_mask"].bool()
    mine = model.projector_grads()
    if dtype == torch.float32:
        assert (out.logits.cpu() - ref["logits"])[keep].abs().max().item() < 1e-3
        assert abs(loss.item() - ref["loss"].item()) < 1e-4
        for k, g in grads.items():
            assert rel_l2(mine[k], g) < 2e-3, k
    else:
        assert rel_l2(out.logits.cpu()[keep], ref["logits"][keep]) < 3e-2
        assert abs(loss.item() - ref["loss"].item()) < 2e-2 * abs(ref["loss"].item())
        for k, g in grads.items():
            assert rel_l2(mine[k], g) < 8e-2, k
    wide = OracleModel(_cfg(None, head_dim), sd, dtype=torch.float32)
    assert (wide.forward(**{**b, "audio_values": b["audio_values"].to(dtype).float()})["logits"] - ref["logits"])[keep].abs().max().item() > 1e-2


def test_mistral_without_a_window_is_the_llama_path_bit_for_bit():
    """sliding_window: null (Mistral v0.2 / v0.3 / Nemo) - and any window the sequence fits in - run exactly what a LlamaConfig of the same sizes runs."""
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import random_state_dict
    mi = _cfg(None)
    tc = {k: v for k, v in mi.text_config.__dict__.items() if k in ("hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads",
                                                                      "num_key_value_heads", "head_dim", "vocab_size", "rms_norm_eps", "eos_token_id", "rope_theta")}
    ll = UltravoxConfig(audio_config=dict(d_model=128, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=256),
                        text_config=dict(model_type="llama", **tc), hidden_size=256, projector_ln_mid=True)
    sd = random_state_dict(ll, seed=73, dtype=torch.bfloat16)
    b = _batch(ll)
    gb = {k: v.to(DEV) for k, v in b.items()}
    gb["audio_values"] = gb["audio_values"].bfloat16()
    outs = []
    for cfg in (ll, mi, _cfg(4096)):
        m = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=torch.bfloat16)
        logits = m.forward(**gb).logits.clone()
        m.train()
        loss = m.forward_backward(**gb).clone()
        outs.append((logits, loss, {k: v.clone() for k, v in m.projector_grads().items()}))
    for logits, loss, g in outs[1:]:
        assert torch.equal(logits, outs[0][0]) and torch.equal(loss, outs[0][1])
        assert all(torch.equal(g[k], outs[0][2][k]) for k in g)


@pytest.mark.parametrize("window", [40, 24])
def test_mistral_generate_runs_past_the_sliding_window(window):
    """Generation across the window: the 33-position prompt (left padding on one row) is prefilled with the windowed kernel where it is longer than
    the window (24), the decode steps clamp the first visible cache slot on EVERY layer (40: crossed while decoding).  Token-exact in f32
    against the oracle's cache-free greedy loop, and different from the un-windowed continuation."""
    from oracle.reference_cpu import OracleModel, logmel_ref, synthetic_batch
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import random_state_dict
    cfg = _cfg(window)
    sd = random_state_dict(cfg, seed=75)
    model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=torch.float32, with_backward=False)
    oracle = OracleModel(cfg, sd, dtype=torch.float32)
    b = synthetic_batch(cfg, 2, 2.0, n_text=20, audio_start=4, n_supervised=4)
    b.pop("labels")
    b["audio_values"] = logmel_ref(b.pop("pcm"), 80)
    b["attention_mask"][1, :3] = 0
    b["input_ids"][1, :3] = 1
    assert b["input_ids"].shape[1] == 33
    N = 12
    got = model.generate(max_new_tokens=N, eos_token_id=-1, **{k: v.to(DEV) for k, v in b.items()}).cpu()
    want = oracle.generate_greedy(N, -1, pad_token_id=0, **b)
    assert torch.equal(got, want)
    wide = OracleModel(_cfg(None), sd, dtype=torch.float32).generate_greedy(N, -1, pad_token_id=0, **b)
    assert not torch.equal(wide, want)


def test_mistral_presets_and_bf16_decode_batch():
    """The published Mistral configs the reference's model cards name resolve offline; a bf16 generate() with a live window (8-row decode batch: the
    grouped decode-attention kernel's clamp) agrees with the f32 path on the first tokens' logits."""
    from ultravox_amd.config import UltravoxConfig
    v01 = UltravoxConfig(text_model_id="mistralai/Mistral-7B-Instruct-v0.1").text_config
    assert (v01.sliding_window, v01.num_key_value_heads, v01.head_dim, v01.vocab_size) == (4096, 8, 128, 32000) and v01.window_layers == [1] * 32
    nemo = UltravoxConfig(text_model_id="mistralai/Mistral-Nemo-Instruct-2407").text_config
    assert (nemo.hidden_size, nemo.head_dim, nemo.num_hidden_layers, nemo.vocab_size) == (5120, 128, 40, 131072) and nemo.window_layers is None
    from oracle.reference_cpu import logmel_ref, synthetic_batch
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import random_state_dict
    cfg = _cfg(24, 128)
    sd = random_state_dict(cfg, seed=77)
    b = synthetic_batch(cfg, 8, 2.0, n_text=20, audio_start=4, n_supervised=4)
    b.pop("labels")
    b["audio_values"] = logmel_ref(b.pop("pcm"), 80)
    outs = {}
    for dtype in (torch.float32, torch.bfloat16):
        m = UltravoxModel(cfg, state_dict={k: v.to(dtype) for k, v in sd.items()}, device=DEV, dtype=dtype, with_backward=False)
        gb = {k: v.to(DEV) for k, v in b.items()}
        gb["audio_values"] = gb["audio_values"].to(dtype)
        outs[dtype] = m.generate(max_new_tokens=3, eos_token_id=-1, return_dict_in_generate=True, output_logits=True, **gb)
    f32, b16 = outs[torch.float32], outs[torch.bfloat16]
    assert rel_l2(b16.logits[0], f32.logits[0]) < 4e-2                         # the windowed prefill
    T = b["input_ids"].shape[1]
    same = (f32.sequences[:, T] == b16.sequences[:, T])                        # rows whose first new token agrees: their decode steps are comparable
    assert same.sum().item() >= 4
    assert rel_l2(b16.logits[1][same], f32.logits[1][same]) < 4e-2             # a decode step behind the clamp
