"""The Qwen backbones on the GPU.  The reference's v0.6 recipe trains on Qwen/Qwen3-32B
(ultravox/training/configs/v0.6_config_qwen3_32b.yaml), reached through the same AutoModelForCausalLM call as Llama
(ultravox_model.py:499-526).  Deltas against the Llama family: qwen3 - an RMSNorm over head_dim on every q / k head before RoPE
(fused with the rotary embedding forward, its own kernel after the RoPE-inverting attention backward), head_dim independent of
hidden_size / heads; qwen2 - q / k / v projection biases.  Checked against the oracle's flavours, which tests/test_oracle_pinning.py
pins to the installed HF Qwen3ForCausalLM / Qwen2ForCausalLM: f32 mode at north_star's 1e-3, bf16 at the bf16-vs-f32 bars of
test_model_gpu.py, generate() token-exact in f32, LLM LoRA on top, and Qwen3-32B WIDTH at depth 1."""
import pytest
import torch

from parity_util import record, rel_l2, stage_errors

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rope_table(T, D, theta=1e6):
    inv = 1.0 / (theta ** (torch.arange(0, D, 2).float() / D))
    f = torch.arange(T).float()[:, None] * inv[None]
    return torch.stack([f.cos(), f.sin()], -1).contiguous()          # [T, D/2, 2]


def _norm_rope_ref(x, w, cs, eps, dtype):
    """x [B, T, H, D] (values of `dtype`) -> LlamaRMSNorm over D in `dtype`, then HF rotary embedding in `dtype`."""
    h = x.float()
    h = (h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + eps)).to(dtype)
    n = (w.to(dtype) * h)
    cos = torch.cat([cs[..., 0], cs[..., 0]], -1).to(dtype)[None, :, None, :]
    sin = torch.cat([cs[..., 1], cs[..., 1]], -1).to(dtype)[None, :, None, :]
    rot = torch.cat([-n[..., n.shape[-1] // 2:], n[..., : n.shape[-1] // 2]], -1)
    return n * cos + rot * sin


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Hq,Hkv,D", [(4, 2, 128), (3, 1, 64), (2, 2, 256)])
def test_qk_norm_rope_kernel_and_its_backward(dtype, Hq, Hkv, D):
    from ultravox_amd import ops
    torch.manual_seed(3)
    B, T, eps = 2, 37, 1e-6
    ld = (Hq + 2 * Hkv) * D
    qkv = (torch.randn(B, T, ld, device=DEV) * 1.5).to(dtype)
    wq = (1 + 0.2 * torch.randn(D, device=DEV)).to(dtype)
    wk = (1 + 0.2 * torch.randn(D, device=DEV)).to(dtype)
    cs = _rope_table(T, D).to(DEV)
    x = qkv.clone()
    raw = ops.qk_norm_rope_(x.view(B * T, ld), wq, wk, cs, T, Hq, Hkv, D, eps, keep_raw=True)
    assert torch.equal(raw.view(B, T, -1), qkv[..., : (Hq + Hkv) * D])                 # the raw q | k rows, bit for bit
    assert torch.equal(x[..., (Hq + Hkv) * D:], qkv[..., (Hq + Hkv) * D:])             # v untouched
    q = qkv[..., : Hq * D].view(B, T, Hq, D)
    k = qkv[..., Hq * D: (Hq + Hkv) * D].view(B, T, Hkv, D)
    want = torch.cat([_norm_rope_ref(q, wq, cs, eps, dtype).reshape(B, T, -1), _norm_rope_ref(k, wk, cs, eps, dtype).reshape(B, T, -1)], -1)
    got = x[..., : (Hq + Hkv) * D]
    if dtype == torch.float32:
        assert (got - want).abs().max().item() < 2e-5
    else:   # same rounding points as torch's bf16 ops: at most one bf16 ulp on a few elements
        assert rel_l2(got, want) < 2e-3 and (got.float() - want.float()).abs().max().item() <= 2 ** -6 * want.float().abs().max().item()
    # backward: autograd through the f32 norm of the raw rows (no RoPE: the attention backward hands over RoPE-inverted gradients)
    dy = (torch.randn(B, T, ld, device=DEV) * 0.3).to(dtype)
    g = dy.clone()
    ops.qk_norm_bwd_(g.view(B * T, ld), raw, wq, wk, Hq, Hkv, D, eps)
    assert torch.equal(g[..., (Hq + Hkv) * D:], dy[..., (Hq + Hkv) * D:])
    for (lo, H, w) in ((0, Hq, wq), (Hq * D, Hkv, wk)):
        xr = qkv[..., lo: lo + H * D].float().view(B, T, H, D).requires_grad_(True)
        y = w.float() * (xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + eps))
        y.backward(dy[..., lo: lo + H * D].float().view(B, T, H, D))
        assert rel_l2(g[..., lo: lo + H * D], xr.grad.reshape(B, T, -1)) < (1e-5 if dtype == torch.float32 else 6e-3)


def _cfg(family, head_dim=64, **kw):
    from ultravox_amd.config import UltravoxConfig
    tc = dict(model_type=family, hidden_size=192, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
              num_key_value_heads=2, vocab_size=512, rms_norm_eps=1e-6, rope_theta=1000000.0, eos_token_id=1)
    if family == "qwen3":
        tc["head_dim"] = head_dim                     # 4 x 64 = 256 != 192: head_dim is its own parameter
    else:
        tc.update(hidden_size=256, num_attention_heads=4)   # qwen2: head_dim = hidden / heads = 64
    return UltravoxConfig(audio_config=dict(d_model=128, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=256),
                          text_config=tc, hidden_size=256, projector_ln_mid=True, **kw)


def _step(cfg, dtype, seed, **model_kw):
    from oracle.reference_cpu import OracleModel, logmel_ref, synthetic_batch
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import random_state_dict
    sd = {k: v.to(dtype) for k, v in random_state_dict(cfg, seed=seed).items()}
    model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=dtype, **model_kw)
    oracle = OracleModel(cfg, sd, dtype=torch.float32)
    b = synthetic_batch(cfg, 2, 2.0, n_text=24, audio_start=5, n_supervised=8)
    b["audio_values"] = logmel_ref(b.pop("pcm"), 80).to(dtype)
    b["attention_mask"][1, -3:] = 0                   # right padding on one sample
    b["labels"][1, -3:] = -100
    gb = {k: v.to(DEV) for k, v in b.items()}
    ref, grads, _ = oracle.train_step({**b, "audio_values": b["audio_values"].float()})
    out = model.forward(**gb)
    model.train()
    loss = model.forward_backward(**gb)
    return model, out, loss, ref, grads, b


CASES = [("qwen3", 64), ("qwen3", 128), ("qwen2", 64)]


@pytest.mark.parametrize("family,head_dim", CASES)
def test_qwen_train_step_f32_within_1e3(family, head_dim):
    model, out, loss, ref, grads, b = _step(_cfg(family, head_dim), torch.float32, 51)
    assert model._c.llm_qk_norm == int(family == "qwen3")
    keep = b["attention_mask"].bool()
    assert (out.logits.cpu() - ref["logits"])[keep].abs().max().item() < 1e-3
    assert abs(out.loss.item() - ref["loss"].item()) < 1e-4 and abs(loss.item() - ref["loss"].item()) < 1e-4
    mine = model.projector_grads()
    for k, g in grads.items():
        assert rel_l2(mine[k], g) < 2e-3, k


@pytest.mark.parametrize("family,head_dim", CASES)
def test_qwen_train_step_bf16(family, head_dim):
    model, out, loss, ref, grads, b = _step(_cfg(family, head_dim), torch.bfloat16, 52)
    keep = b["attention_mask"].bool()
    rec = {"logits": stage_errors(out.logits.cpu()[keep], ref["logits"][keep]), "loss": [loss.item(), ref["loss"].item()]}
    mine = model.projector_grads()
    rec["grads"] = {k: rel_l2(mine[k], g) for k, g in grads.items()}
    record(f"{family}_small_bf16_hd{head_dim}", rec)
    assert rec["logits"]["rel_l2"] < 3e-2
    assert abs(loss.item() - ref["loss"].item()) < 2e-2 * abs(ref["loss"].item())
    for k, v in rec["grads"].items():
        assert v < 8e-2, (k, v)


def test_qwen3_with_llm_lora_trains_the_adapters():
    """text_model_lora_config on a Qwen3 backbone: peft's q_proj / k_proj adapters add to the projections BEFORE q_norm / k_norm
    (the norm wraps the projection's output) - gradients of projector and adapters against the oracle in f32."""
    from oracle.reference_cpu import OracleModel, logmel_ref, synthetic_batch
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import init_lora_state_dict, random_state_dict
    cfg = _cfg("qwen3", 64, text_model_lora_config=dict(r=4, lora_alpha=8))
    sd = random_state_dict(cfg, seed=53)
    sd.update(init_lora_state_dict(cfg, seed=54, random_b=True))
    model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=torch.float32)
    oracle = OracleModel(cfg, sd, dtype=torch.float32)
    b = synthetic_batch(cfg, 2, 2.0, n_text=24, audio_start=5, n_supervised=8)
    b["audio_values"] = logmel_ref(b.pop("pcm"), 80)
    ref, grads, _ = oracle.train_step(b)
    model.train()
    loss = model.forward_backward(**{k: v.to(DEV) for k, v in b.items()})
    assert abs(loss.item() - ref["loss"].item()) < 1e-4
    mine = model.projector_grads()
    assert any("lora_A" in k for k in grads) and set(grads) <= set(mine)
    for k, g in grads.items():
        assert rel_l2(mine[k], g) < 3e-3, k


@pytest.mark.parametrize("family,head_dim", CASES)
def test_qwen_generate_token_exact_in_f32(family, head_dim):
    """generate() on the Qwen backbones: prefill, chunked prefill through forward(past_key_values), KV-cache decode - all three
    run the family's q | k | v stage (biases / per-head norms at explicit positions) - token-exact against the oracle's cache-free
    greedy search in f32, with audio and a left-padded prompt."""
    from oracle.reference_cpu import OracleModel, logmel_ref, synthetic_batch
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import random_state_dict
    cfg = _cfg(family, head_dim)
    sd = random_state_dict(cfg, seed=55)
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


def test_qwen3_32b_width_train_step_matches_oracle():
    """Qwen3-32B WIDTH (hidden 5120, intermediate 25600, 64 query / 8 key-value heads x 128, vocab 151936) at depth 1 behind the
    whisper-medium-width encoder (depth 1), 2 x 30 s clips: the N = 10240 q|k|v GEMM, K = 8192 o_proj, N = 51200 gate|up GEMM with
    the fused SwiGLU, the fused head_dim-128 attention backward behind the per-head norm backward, the 151936-row head."""
    from oracle.reference_cpu import OracleModel, synthetic_batch
    from ultravox_amd.config import AUDIO_PRESETS, TEXT_PRESETS, UltravoxConfig
    from ultravox_amd.frontend import WhisperFeatureExtractor
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import random_state_dict
    from parity_util import oracle_threads
    tc = dict(TEXT_PRESETS["Qwen/Qwen3-32B"], num_hidden_layers=1)
    ac = dict(AUDIO_PRESETS["openai/whisper-medium"], encoder_layers=1)
    cfg = UltravoxConfig(text_config=tc, audio_config=ac, hidden_size=4096, stack_factor=8, projector_ln_mid=True, torch_dtype="bfloat16")
    assert cfg.text_config.head_dim * cfg.text_config.num_attention_heads == 8192 != cfg.text_config.hidden_size
    sd = random_state_dict(cfg, seed=57, dtype=torch.bfloat16, device="cuda")
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
    record("qwen3_32b_width_depth1", rec)
    assert rec["logits"]["rel_l2"] < 3e-2
    assert abs(loss.item() - ref["loss"].item()) < 2e-2 * abs(ref["loss"].item())
    for k, v in rec["grads"].items():
        assert v < 8e-2, (k, v)
