"""The Gemma-3 text stack on the GPU - the reference's other v0.6 recipe (ultravox/training/configs/v0.6_config_gemma3_27b.yaml:
google/gemma-3-27b-it behind whisper-large-v3-turbo), reached through the same AutoModelForCausalLM call as Llama
(ultravox_model.py:499-526).  Deltas (uvx_config_t.llm_flavor = UVX_LLM_GEMMA3): a POST norm on each branch before its residual add
(four Gemma norms per layer), q_norm / k_norm in the Gemma flavour fused with RoPE, query_pre_attn_scalar scaling, a second rotary
table for the sliding-window layers (sequences stay within the window: those layers are then plain causal attention), linear rope
scaling on the global layers, the sqrt(hidden) scale on the looked-up embedding rows only, tied head.  Against the oracle's gemma3
flavour, which tests/test_oracle_pinning.py pins to the installed HF Gemma3ForCausalLM."""
import pytest
import torch

from parity_util import record, rel_l2, stage_errors

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _cfg(head_dim=64, layers=7, window=512, **kw):
    from ultravox_amd.config import UltravoxConfig
    tc = dict(model_type="gemma3", hidden_size=192, intermediate_size=512, num_hidden_layers=layers, num_attention_heads=4,
              num_key_value_heads=2, head_dim=head_dim, vocab_size=512, rms_norm_eps=1e-6, eos_token_id=1, query_pre_attn_scalar=48,
              sliding_window=window, sliding_window_pattern=3, rope_theta=1000000.0, rope_scaling=dict(rope_type="linear", factor=8.0),
              rope_local_base_freq=10000.0)
    return UltravoxConfig(audio_config=dict(d_model=128, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=256),
                          text_config=tc, hidden_size=256, projector_ln_mid=True, **kw)


def _step(cfg, dtype, seed):
    from oracle.reference_cpu import OracleModel, logmel_ref, synthetic_batch
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import random_state_dict
    sd = {k: v.to(dtype) for k, v in random_state_dict(cfg, seed=seed).items()}
    assert "language_model.lm_head.weight" not in sd
    model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=dtype)
    oracle = OracleModel(cfg, sd, dtype=torch.float32)
    b = synthetic_batch(cfg, 2, 2.0, n_text=24, audio_start=5, n_supervised=8)
    b["audio_values"] = logmel_ref(b.pop("pcm"), 80).to(dtype)
    b["attention_mask"][1, -3:] = 0
    b["labels"][1, -3:] = -100
    gb = {k: v.to(DEV) for k, v in b.items()}
    ref, grads, _ = oracle.train_step({**b, "audio_values": b["audio_values"].float()})
    out = model.forward(**gb)
    model.train()
    loss = model.forward_backward(**gb)
    return model, out, loss, ref, grads, b


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemma_flavoured_qk_norm_kernels(dtype):
    """qk_norm_rope / qk_norm_bwd with flavor 1: x_hat * (1 + w) in f32, one rounding (Gemma3RMSNorm), then RoPE."""
    from test_qwen_gpu import _rope_table
    from ultravox_amd import ops
    torch.manual_seed(4)
    B, T, Hq, Hkv, D, eps = 2, 29, 4, 2, 128, 1e-6
    ld = (Hq + 2 * Hkv) * D
    qkv = (torch.randn(B, T, ld, device=DEV) * 1.5).to(dtype)
    wq, wk = ((0.2 * torch.randn(D, device=DEV)).to(dtype) for _ in range(2))       # zero-centred, as Gemma stores them
    cs = _rope_table(T, D, 10000.0).to(DEV)
    x = qkv.clone()
    raw = ops.qk_norm_rope_(x.view(B * T, ld), wq, wk, cs, T, Hq, Hkv, D, eps, keep_raw=True, flavor=1)

    def ref(xh, w):
        h = xh.float()
        n = ((h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + eps)) * (1.0 + w.float())).to(dtype)
        cos = torch.cat([cs[..., 0], cs[..., 0]], -1).to(dtype)[None, :, None, :]
        sin = torch.cat([cs[..., 1], cs[..., 1]], -1).to(dtype)[None, :, None, :]
        rot = torch.cat([-n[..., D // 2:], n[..., : D // 2]], -1)
        return n * cos + rot * sin
    q = qkv[..., : Hq * D].view(B, T, Hq, D)
    k = qkv[..., Hq * D: (Hq + Hkv) * D].view(B, T, Hkv, D)
    want = torch.cat([ref(q, wq).reshape(B, T, -1), ref(k, wk).reshape(B, T, -1)], -1)
    got = x[..., : (Hq + Hkv) * D]
    assert rel_l2(got, want) < (1e-6 if dtype == torch.float32 else 2e-3)
    dy = (torch.randn(B, T, ld, device=DEV) * 0.3).to(dtype)
    g = dy.clone()
    ops.qk_norm_bwd_(g.view(B * T, ld), raw, wq, wk, Hq, Hkv, D, eps, flavor=1)
    for (lo, H, w) in ((0, Hq, wq), (Hq * D, Hkv, wk)):
        xr = qkv[..., lo: lo + H * D].float().view(B, T, H, D).requires_grad_(True)
        y = (xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + eps)) * (1.0 + w.float())
        y.backward(dy[..., lo: lo + H * D].float().view(B, T, H, D))
        assert rel_l2(g[..., lo: lo + H * D], xr.grad.reshape(B, T, -1)) < (1e-5 if dtype == torch.float32 else 6e-3)


@pytest.mark.parametrize("head_dim", [64, 128])
def test_gemma3_train_step_f32_within_1e3(head_dim):
    model, out, loss, ref, grads, b = _step(_cfg(head_dim), torch.float32, 61)
    assert model._c.llm_flavor == 2 and model._c.llm_qk_norm == 1 and abs(model._c.llm_attn_scale - 48 ** -0.5) < 1e-7
    assert sum(model._llm["layer_local"]) == 5 and model._lw.rope_cos_sin_local
    keep = b["attention_mask"].bool()
    assert (out.logits.cpu() - ref["logits"])[keep].abs().max().item() < 1e-3
    assert abs(out.loss.item() - ref["loss"].item()) < 1e-4 and abs(loss.item() - ref["loss"].item()) < 1e-4
    mine = model.projector_grads()
    for k, g in grads.items():
        assert rel_l2(mine[k], g) < 2e-3, k


@pytest.mark.parametrize("head_dim", [64, 128])
def test_gemma3_train_step_bf16(head_dim):
    model, out, loss, ref, grads, b = _step(_cfg(head_dim), torch.bfloat16, 62)
    keep = b["attention_mask"].bool()
    rec = {"logits": stage_errors(out.logits.cpu()[keep], ref["logits"][keep]), "loss": [loss.item(), ref["loss"].item()]}
    mine = model.projector_grads()
    rec["grads"] = {k: rel_l2(mine[k], g) for k, g in grads.items()}
    record(f"gemma3_small_bf16_hd{head_dim}", rec)
    assert rec["logits"]["rel_l2"] < 3e-2
    assert abs(loss.item() - ref["loss"].item()) < 2e-2 * abs(ref["loss"].item())
    for k, v in rec["grads"].items():
        assert v < 8e-2, (k, v)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemma3_sequences_beyond_the_sliding_window(dtype):
    """Sequences LONGER than the sliding window: the local layers run the windowed attention kernels (forward, dK/dV, dQ skip what
    no pair can see and mask the boundary), the global layers plain causal attention - whole train step against the oracle, which
    masks the window in full (window 16, 37 positions)."""
    cfg = _cfg(64, layers=4, window=16)
    model, out, loss, ref, grads, b = _step(cfg, dtype, 63)
    assert b["input_ids"].shape[1] == 37 and sum(model._llm["layer_local"]) == 3
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
    # the window matters here: a model whose window covers the sequence gives other logits
    from oracle.reference_cpu import OracleModel
    from ultravox_amd.weights import random_state_dict
    sd = {k: v.to(dtype) for k, v in random_state_dict(cfg, seed=63).items()}
    wide = OracleModel(_cfg(64, layers=4, window=512), sd, dtype=torch.float32)
    assert (wide.forward(**{**b, "audio_values": b["audio_values"].float()})["logits"] - ref["logits"])[keep].abs().max().item() > 1e-2


@pytest.mark.parametrize("head_dim", [64, 128])
def test_gemma3_generate_token_exact_in_f32(head_dim):
    from oracle.reference_cpu import OracleModel, logmel_ref, synthetic_batch
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import random_state_dict
    cfg = _cfg(head_dim)
    sd = random_state_dict(cfg, seed=65)
    # an explicit (untied) head for this test only: with random weights a TIED head makes every step copy the last token
    sd["language_model.lm_head.weight"] = 0.3 * torch.randn(512, 192, generator=torch.Generator().manual_seed(7))
    model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=torch.float32, with_backward=False)
    oracle = OracleModel(cfg, sd, dtype=torch.float32)
    b = synthetic_batch(cfg, 2, 2.0, n_text=20, audio_start=4, n_supervised=4)
    b.pop("labels")
    b["audio_values"] = logmel_ref(b.pop("pcm"), 80)
    b["attention_mask"][1, :3] = 0
    b["input_ids"][1, :3] = 1
    N = 6
    got = model.generate(max_new_tokens=N, eos_token_id=-1, **{k: v.to(DEV) for k, v in b.items()}).cpu()
    want = oracle.generate_greedy(N, -1, pad_token_id=0, **b)
    assert torch.equal(got, want)


def test_gemma3_decode_runs_past_the_sliding_window():
    """Generation across the sliding window: the prompt (33 positions) is prefilled with the windowed attention kernel where it is
    longer than the window, every decode step clamps the first visible cache slot of a sliding-window layer to the last `window`
    positions.  Token-exact against the oracle, which masks the window in full - window 40 (crossed while decoding) and window
    24 (crossed inside the prompt)."""
    _run_decode_window(40)
    _run_decode_window(24)


def _run_decode_window(window):
    from oracle.reference_cpu import OracleModel, logmel_ref, synthetic_batch
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import random_state_dict
    cfg = _cfg(64, layers=4, window=window)
    sd = random_state_dict(cfg, seed=69)
    sd["language_model.lm_head.weight"] = 0.3 * torch.randn(512, 192, generator=torch.Generator().manual_seed(9))
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
    # and the window matters in this set-up: with a window that covers everything the continuation differs somewhere
    wide = OracleModel(_cfg(64, layers=4, window=512), sd, dtype=torch.float32).generate_greedy(N, -1, pad_token_id=0, **b)
    assert not torch.equal(wide, want)


def test_gemma3_27b_width_train_step_matches_oracle():
    """Gemma-3-27B WIDTH (hidden 5376, intermediate 21504, 32 query / 16 key-value heads x 128, vocab 262208, query_pre_attn_scalar
    168) at depth 2 (one sliding-window layer, one global layer with the linearly scaled table) behind the whisper-medium-width
    encoder (depth 1), 2 x 30 s clips."""
    from oracle.reference_cpu import OracleModel, synthetic_batch
    from ultravox_amd.config import AUDIO_PRESETS, TEXT_PRESETS, UltravoxConfig
    from ultravox_amd.frontend import WhisperFeatureExtractor
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import random_state_dict
    from parity_util import oracle_threads
    tc = dict(TEXT_PRESETS["google/gemma-3-27b-it"], num_hidden_layers=2, sliding_window_pattern=2)
    ac = dict(AUDIO_PRESETS["openai/whisper-medium"], encoder_layers=1)
    cfg = UltravoxConfig(text_config=tc, audio_config=ac, hidden_size=4096, stack_factor=8, projector_ln_mid=True, torch_dtype="bfloat16")
    assert cfg.text_config.layer_types == ["sliding_attention", "full_attention"]
    sd = random_state_dict(cfg, seed=67, dtype=torch.bfloat16, device="cuda")
    model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=torch.bfloat16, rope_len=512)
    oracle = OracleModel(cfg, {k: v.cpu() for k, v in sd.items()}, dtype=torch.float32)
    b = synthetic_batch(cfg, 2, 30.0, n_text=128, audio_start=16, n_supervised=32)
    pcm = b.pop("pcm")
    mel = WhisperFeatureExtractor(80).logmel_device(pcm.to(DEV))
    gb = {k: v.to(DEV) for k, v in b.items()}
    oracle_threads()
    ref, grads, _ = oracle.train_step({**b, "audio_values": mel.cpu().bfloat16().float()})
    out = model.forward(audio_values=mel, **gb)
    model.train()
    loss = model.forward_backward(audio_values=mel, **gb)
    mine = model.projector_grads()
    rec = {"logits": stage_errors(out.logits, ref["logits"]), "loss": [loss.item(), ref["loss"].item()],
           "grads": {k: rel_l2(mine[k], g) for k, g in grads.items()}}
    record("gemma3_27b_width_depth2", rec)
    assert rec["logits"]["rel_l2"] < 3e-2
    assert abs(loss.item() - ref["loss"].item()) < 2e-2 * abs(ref["loss"].item())
    for k, v in rec["grads"].items():
        assert v < 8e-2, (k, v)
