"""UVX_F32 parity mode on a real MI355X: every tensor f32 (exact-f32 matrix cores), compared with the f32
CPU oracle at the tolerance north_star states — logits within 1e-3 (absolute), integer outputs bit-exact.
Includes BASELINE.json configs[0]: TinyLlama-1.1B + whisper-tiny, 1 x 4 s clip, one adapter-train step."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def test_f32_gemm_is_exact_f32():
    from ultravox_amd import ops
    torch.manual_seed(0)
    for (M, N, K) in [(64, 64, 16), (100, 132, 192), (333, 260, 1024)]:
        a, b = torch.randn(M, K, device=DEV), torch.randn(N, K, device=DEV)
        bias, res = torch.randn(N, device=DEV), torch.randn(M, N, device=DEV)
        out = ops.gemm(a, b, bias=bias, residual=res, act="gelu")
        ref = F.gelu(a.double() @ b.double().t() + bias.double()) + res.double()
        assert (out.double() - ref).abs().max().item() < 2e-6 * math.sqrt(K) * 4


@pytest.mark.parametrize("D,Hq,Hkv,T,causal", [(64, 4, 2, 150, True), (128, 4, 1, 70, True), (64, 2, 2, 200, False)])
def test_f32_attention_fwd_bwd(D, Hq, Hkv, T, causal):
    from ultravox_amd import ops
    from test_kernels_gpu import sdpa_ref
    torch.manual_seed(1)
    B = 2
    q = torch.randn(B, T, Hq, D, device=DEV)
    k = torch.randn(B, T, Hkv, D, device=DEV)
    v = torch.randn(B, T, Hkv, D, device=DEV)
    do = torch.randn(B, T, Hq * D, device=DEV)
    kv_len = None if causal else torch.tensor([T, T - 31], device=DEV, dtype=torch.int32)
    o, lse = ops.attention(q, k, v, causal=causal, kv_len=kv_len)
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    ref, _ = sdpa_ref(qr, kr, vr, causal, 0, D ** -0.5, kv_len=kv_len)
    assert (o - ref).abs().max().item() < 2e-5
    dq, dk, dv = ops.attention_bwd(q, k, v, o, lse, do, causal=causal, kv_len=kv_len)
    ref.backward(do)
    assert rel_l2(dq, qr.grad) < 1e-5 and rel_l2(dk, kr.grad) < 1e-5 and rel_l2(dv, vr.grad) < 1e-5


def _run_step(cfg, seed, B, seconds, n_text, audio_start, n_sup):
    from oracle.reference_cpu import OracleModel, logmel_ref, synthetic_batch
    from ultravox_amd.frontend import WhisperFeatureExtractor
    from ultravox_amd.model import UltravoxModel, UltravoxTrainer
    from ultravox_amd.weights import random_state_dict
    sd = random_state_dict(cfg, seed=seed, dtype=torch.float32)
    model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=torch.float32)
    oracle = OracleModel(cfg, sd, dtype=torch.float32)
    b = synthetic_batch(cfg, B, seconds, n_text=n_text, audio_start=audio_start, n_supervised=n_sup)
    pcm = b.pop("pcm")
    mel = WhisperFeatureExtractor(cfg.audio_config.num_mel_bins).logmel_device(pcm.to(DEV))   # K1 on device
    mel_ref = logmel_ref(pcm, cfg.audio_config.num_mel_bins)
    assert (mel.cpu() - mel_ref).abs().max().item() < 2e-4
    gb = {k: v.to(DEV) for k, v in b.items()}
    out = model.forward(audio_values=mel, **gb)
    # the oracle gets the DEVICE mel so that the comparison isolates the model path (mel parity is above)
    ob = {**b, "audio_values": mel.cpu()}
    with torch.no_grad():
        ref = oracle.forward(**ob)
    return model, oracle, out, ref, gb, ob, mel, sd


def test_f32_small_model_logits_within_1e3():
    from ultravox_amd.config import UltravoxConfig
    from test_model_gpu import SMALL
    cfg = UltravoxConfig(**SMALL)
    model, oracle, out, ref, gb, ob, mel, sd = _run_step(cfg, 11, 2, 2.0, 24, 5, 8)
    assert (out.logits.cpu() - ref["logits"]).abs().max().item() < 1e-3           # north_star tolerance
    assert abs(out.loss.item() - ref["loss"].item()) < 1e-4
    _, grads, _ = oracle.train_step(ob)
    model.train()
    loss = model.forward_backward(audio_values=mel, **gb)
    mine = model.projector_grads()
    for k, g in grads.items():
        assert rel_l2(mine[k], g) < 1e-3, k


def test_c1_tinyllama_whisper_tiny_train_step():
    """BASELINE.json configs[0]: TinyLlama-1.1B + whisper-tiny, 1 x 4 s clip, adapter-only train step; the
    reference plumbing runs this on CPU in f32 (config_base.py:245-249) — so does the oracle."""
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.model import UltravoxTrainer
    cfg = UltravoxConfig(audio_model_id="openai/whisper-tiny", text_model_id="TinyLlama/TinyLlama-1.1B-Chat-v1.0",
                         hidden_size=4096, stack_factor=8, projector_ln_mid=True, torch_dtype="float32")
    model, oracle, out, ref, gb, ob, mel, sd = _run_step(cfg, 12, 1, 4.0, 128, 16, 32)
    assert tuple(out.logits.shape) == (1, 153, 32000)                               # 128 text + ceil(400/16) audio
    err = (out.logits.cpu() - ref["logits"]).abs().max().item()
    assert err < 1e-3, f"max |logit diff| = {err}"
    assert abs(out.loss.item() - ref["loss"].item()) < 1e-4
    params = [oracle.sd[k] for k in oracle.trainable]
    opt = torch.optim.AdamW(params, lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    _, grads, gn = oracle.train_step(ob, opt)
    trainer = UltravoxTrainer(model, lr=2e-3)
    loss = trainer.train_step(audio_values=mel, **gb)
    mine = model.projector_grads()
    for k, g in grads.items():
        assert rel_l2(mine[k], g) < 2e-3, k
    assert abs(trainer.grad_norm().item() - gn.item()) < 2e-3 * gn.item()
    new = model.projector_state_dict()
    for k in oracle.trainable:
        assert rel_l2(new[k], oracle.sd[k].detach()) < 1e-4, k


@pytest.mark.parametrize("name", ["ln_mid", "ln_post", "ln_mid_mixed"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_forward_matches_the_reference_forward_fixture(name, dtype):
    """The HIP path against the REFERENCE UltravoxModel.forward + loss.backward() itself (tests/golden/forward_reference.npz:
    the imported reference run end to end on tiny random towers, its audio tower stubbed by recorded hidden states - the
    tower is stubbed here the same way): two audio items merged into one sample, token_len truncation, left and right
    padding, ForCausalLMLoss, projector gradients.  f32 compute mode: north_star's 1e-3 on the logits; bf16: the usual bars."""
    from test_oracle_pinning import load_forward_fixture
    from ultravox_amd.model import UltravoxModel
    cfg, sd, batch, enc, exp = load_forward_fixture(name)
    model = UltravoxModel(cfg, state_dict={k: v.to(dtype) for k, v in sd.items()}, device=DEV, dtype=dtype)
    tower = enc.to(DEV, dtype)
    model.audio_tower_forward = lambda audio_values, audio_len: tower[: audio_values.shape[0]]
    gb = {k: v.to(DEV) for k, v in batch.items()}
    mel = torch.zeros(len(enc), 80, 3000, device=DEV, dtype=dtype)
    out = model.forward(audio_values=mel, **gb)
    keep = batch["attention_mask"].bool()
    logits = out.logits.float().cpu()
    if dtype == torch.float32:
        assert (logits[keep] - exp["logits"][keep]).abs().max().item() < 1e-3
        assert abs(out.loss.item() - exp["loss"]) < 1e-4
    else:
        assert rel_l2(logits[keep], exp["logits"][keep]) < 3e-2
        assert abs(out.loss.item() - exp["loss"]) < 2e-2 * exp["loss"]
    model.train()
    model.forward_backward(audio_values=mel, **gb)
    mine = model.projector_grads()
    for k, g in exp["grads"].items():
        assert rel_l2(mine[k], g) < (1e-3 if dtype == torch.float32 else 8e-2), k


@pytest.mark.parametrize("name", ["key_padding", "latency_and_padding", "latency_only", "odd_frames"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_encoder_matches_the_reference_encoder_forward_fixture(name, dtype):
    """uvx_encoder_fwd against a RUN OF THE REFERENCE's ModifiedWhisperEncoder.forward (ultravox_model.py:865-994; fixture
    tests/golden/real_tower_reference.npz, generator make_golden.py `real_tower_cases`): the key-padding mask from audio_len
    (:915-926), the latency mask (:834-863) and their merge (:928-936) are the reference's own.  f32 mode: 1e-3 absolute."""
    import forward_fixture_util as U
    from test_oracle_pinning import load_real_tower_fixture
    from ultravox_amd.model import UltravoxModel
    z, meta, make = load_real_tower_fixture()
    case = meta["encoder_cases"][name]
    cfg, sd = make(case["weight_names"], case["audio_latency_block_size"])
    model = UltravoxModel(cfg, state_dict={k: v.to(dtype) for k, v in sd.items()}, device=DEV, dtype=dtype)
    n = 3 if case["audio_len"] is None else len(case["audio_len"])
    lens = None if case["audio_len"] is None else torch.tensor(case["audio_len"], device=DEV)
    got = model.audio_tower_forward(U.mel(n, case["frames"]).to(DEV, dtype), lens).float().cpu()
    want = torch.from_numpy(z[f"enc.{name}"])
    assert got.shape == want.shape
    if dtype == torch.float32:
        assert (got - want).abs().max().item() < 1e-3
    else:
        assert rel_l2(got, want) < 2e-2


@pytest.mark.parametrize("name", ["real_tower", "real_tower_latency"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_forward_matches_the_reference_forward_with_its_own_tower(name, dtype):
    """The whole HIP path (mel -> encoder -> projector -> merge -> Llama -> loss -> projector gradients) against the REFERENCE
    UltravoxModel.forward + loss.backward() with nothing stubbed - the reference's ModifiedWhisperEncoder.forward included."""
    import forward_fixture_util as U
    from test_oracle_pinning import load_real_tower_fixture
    from ultravox_amd.model import UltravoxModel
    z, meta, make = load_real_tower_fixture()
    case = meta["model_cases"][name]
    cfg, sd = make(case["weight_names"], case["audio_latency_block_size"])
    model = UltravoxModel(cfg, state_dict={k: v.to(dtype) for k, v in sd.items()}, device=DEV, dtype=dtype)
    batch = U.batch()
    gb = {k: v.to(DEV) for k, v in batch.items()}
    mel = U.mel(U.N_AUDIO, 3000).to(DEV, dtype)
    out = model.forward(audio_values=mel, **gb)
    keep = batch["attention_mask"].bool()
    logits, want = out.logits.float().cpu(), torch.from_numpy(z[f"{name}.logits"])
    loss = float(z[f"{name}.loss"])
    if dtype == torch.float32:
        assert (logits[keep] - want[keep]).abs().max().item() < 1e-3
        assert abs(out.loss.item() - loss) < 1e-4
    else:
        assert rel_l2(logits[keep], want[keep]) < 3e-2
        assert abs(out.loss.item() - loss) < 2e-2 * loss
    model.train()
    model.forward_backward(audio_values=mel, **gb)
    mine = model.projector_grads()
    for k in z.files:
        if k.startswith(name + ".g."):
            key = k[len(name) + 3:]
            assert rel_l2(mine[key], torch.from_numpy(z[k])) < (1e-3 if dtype == torch.float32 else 8e-2), key
