"""Encoder LoRA (audio_model_lora_config.r > 0: the reference's release recipe, ultravox_model.py:690-709 via peft 0.11.1)
on the GPU: the new backward kernels against torch, and a whole training step (projector + LoRA gradients, one AdamW
update) against the CPU oracle's autograd."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel_l2(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layernorm_and_gelu_backward_kernels(dtype):
    from ultravox_amd import ops
    g = torch.Generator(device=DEV).manual_seed(0)
    rows, cols = 77, 384
    x = (torch.randn(rows, cols, device=DEV, generator=g) * 1.5 + 0.2).to(dtype)
    w = (1 + 0.2 * torch.randn(cols, device=DEV, generator=g)).to(dtype)
    b = torch.randn(cols, device=DEV, generator=g).to(dtype)
    dy = torch.randn(rows, cols, device=DEV, generator=g).to(dtype)
    add = torch.randn(rows, cols, device=DEV, generator=g).to(dtype)
    xr = x.float().requires_grad_(True)
    F.layer_norm(xr, (cols,), w.float(), b.float(), 1e-5).backward(dy.float())
    tol = 1e-5 if dtype == torch.float32 else 6e-3
    assert rel_l2(ops.layernorm_bwd(dy, x, w, 1e-5), xr.grad) < tol
    assert rel_l2(ops.layernorm_bwd(dy, x, w, 1e-5, dx_add=add), xr.grad + add.float()) < tol
    pre = (torch.randn(rows, 512, device=DEV, generator=g) * 2).to(dtype)
    d = torch.randn(rows, 512, device=DEV, generator=g).to(dtype)
    pr = pre.float().requires_grad_(True)
    y = F.gelu(pr)
    y.backward(d.float())
    assert rel_l2(ops.gelu(pre), y) < (1e-6 if dtype == torch.float32 else 4e-3)
    assert rel_l2(ops.gelu_bwd(d, pre), pr.grad) < (1e-5 if dtype == torch.float32 else 4e-3)


def _setup(dtype, r=4, seed=31):
    from oracle.reference_cpu import OracleModel, synthetic_batch
    from test_model_gpu import SMALL
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.frontend import WhisperFeatureExtractor
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import init_lora_state_dict, random_state_dict
    cfg = UltravoxConfig(**SMALL, audio_model_lora_config={"r": r, "lora_alpha": 8})
    sd = random_state_dict(cfg, seed=seed, dtype=dtype)
    sd.update(init_lora_state_dict(cfg, seed=seed, dtype=dtype, random_b=True))    # non-zero B: A gets a gradient too
    model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=dtype)
    oracle = OracleModel(cfg, sd, dtype=torch.float32)
    b = synthetic_batch(cfg, 2, 3.0, n_text=24, audio_start=5, n_supervised=8)
    b["audio_lens"] = torch.tensor([300, 230])                                    # key-padding mask on the second clip
    pcm = b.pop("pcm")
    mel = WhisperFeatureExtractor(cfg.audio_config.num_mel_bins).logmel_device(pcm.to(DEV))
    gb = {k: v.to(DEV) for k, v in b.items()}
    ob = {**b, "audio_values": mel.cpu().to(dtype).float()}
    return cfg, sd, model, oracle, gb, ob, mel.to(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_lora_train_step_matches_oracle(dtype):
    cfg, sd, model, oracle, gb, ob, mel = _setup(dtype)
    assert len(oracle.trainable) == 4 + 2 * 2 * cfg.audio_config.encoder_layers
    ref, grads, _ = oracle.train_step(ob)
    out = model.forward(audio_values=mel, **gb)                      # the LoRA terms are part of the forward pass
    if dtype == torch.float32:
        assert (out.logits.cpu() - ref["logits"]).abs().max().item() < 1e-3
    else:
        assert rel_l2(out.logits, ref["logits"]) < 3e-2
    model.train()
    loss = model.forward_backward(audio_values=mel, **gb)
    assert abs(loss.item() - ref["loss"].item()) < (1e-4 if dtype == torch.float32 else 2e-2) * abs(ref["loss"].item())
    mine = model.projector_grads()
    assert set(mine) == set(grads)
    tol = 2e-3 if dtype == torch.float32 else 8e-2
    for k, g in grads.items():
        assert g.abs().max().item() > 0, k
        assert rel_l2(mine[k], g) < tol, k


def test_lora_trainer_step_and_checkpoint_keys(tmp_path):
    from ultravox_amd.model import UltravoxTrainer
    cfg, sd, model, oracle, gb, ob, mel = _setup(torch.float32)
    params = [oracle.sd[k] for k in oracle.trainable]
    opt = torch.optim.AdamW(params, lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    _, _, gn = oracle.train_step(ob, opt)
    trainer = UltravoxTrainer(model, lr=2e-3)
    trainer.train_step(audio_values=mel, **gb)
    assert abs(trainer.grad_norm().item() - gn.item()) < 3e-3 * gn.item()
    new = model.projector_state_dict()
    for k in oracle.trainable:
        delta_ref = oracle.sd[k].detach() - sd[k].float()
        assert rel_l2(new[k].float().cpu() - sd[k].float(), delta_ref) < 2e-2, k
    # the diff state dict carries the adapter under peft's names (what the reference's save_pretrained writes)
    model.save_pretrained(str(tmp_path))
    from ultravox_amd import checkpoint
    _, ck = checkpoint.load_pretrained(str(tmp_path))
    assert "audio_tower.base_model.model.layers.0.self_attn.q_proj.lora_A.default.weight" in ck
    assert sum(".lora_" in k for k in ck) == 4 * cfg.audio_config.encoder_layers


def test_lora_with_zero_b_equals_frozen_tower():
    """peft initialises lora_B to zero: the adapted tower must reproduce the frozen one bit for bit."""
    from test_model_gpu import SMALL
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import random_state_dict
    sd = random_state_dict(UltravoxConfig(**SMALL), seed=4, dtype=torch.bfloat16)
    m0 = UltravoxModel(UltravoxConfig(**SMALL), state_dict=sd, device=DEV)
    m1 = UltravoxModel(UltravoxConfig(**SMALL, audio_model_lora_config={"r": 8}), state_dict=sd, device=DEV)
    mel = torch.randn(2, 80, 200, device=DEV).bfloat16()
    lens = torch.tensor([200, 150], device=DEV)
    assert torch.equal(m0.audio_tower_forward(mel, lens), m1.audio_tower_forward(mel, lens))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_llm_and_encoder_lora_train_step_matches_oracle(dtype):
    """text_model_lora_config + audio_model_lora_config together: gradients of the projector, the encoder adapters and the
    LLM adapters (GQA: k_proj is narrower than q_proj; adapters sit before RoPE) against the oracle's autograd."""
    from oracle.reference_cpu import OracleModel, synthetic_batch
    from test_model_gpu import SMALL
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.frontend import WhisperFeatureExtractor
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import init_lora_state_dict, random_state_dict
    cfg = UltravoxConfig(**SMALL, audio_model_lora_config={"r": 4}, text_model_lora_config={"r": 8, "lora_alpha": 16})
    sd = random_state_dict(cfg, seed=41, dtype=dtype)
    sd.update(init_lora_state_dict(cfg, seed=41, dtype=dtype, random_b=True))
    model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=dtype)
    oracle = OracleModel(cfg, sd, dtype=torch.float32)
    nl, ne = cfg.text_config.num_hidden_layers, cfg.audio_config.encoder_layers
    assert len(oracle.trainable) == 4 + 4 * ne + 4 * nl
    b = synthetic_batch(cfg, 2, 3.0, n_text=24, audio_start=5, n_supervised=8)
    pcm = b.pop("pcm")
    mel = WhisperFeatureExtractor(cfg.audio_config.num_mel_bins).logmel_device(pcm.to(DEV)).to(dtype)
    gb = {k: v.to(DEV) for k, v in b.items()}
    ref, grads, _ = oracle.train_step({**b, "audio_values": mel.cpu().float()})
    model.train()
    loss = model.forward_backward(audio_values=mel, **gb)
    assert abs(loss.item() - ref["loss"].item()) < (1e-4 if dtype == torch.float32 else 2e-2) * abs(ref["loss"].item())
    mine = model.projector_grads()
    assert set(mine) == set(grads)
    tol = 2e-3 if dtype == torch.float32 else 8e-2
    for k, g in grads.items():
        assert g.abs().max().item() > 0, k
        assert rel_l2(mine[k], g) < tol, k
    # a text-only batch still trains the LLM adapters; projector / encoder-adapter gradients are exactly zero
    tb = {k: v for k, v in gb.items() if k in ("input_ids", "attention_mask", "labels")}
    model.forward_backward(**tb)
    mine = model.projector_grads()
    assert mine["multi_modal_projector.linear_1.weight"].abs().max().item() == 0
    assert mine["language_model.base_model.model.model.layers.0.self_attn.q_proj.lora_B.default.weight"].abs().max().item() > 0
    # generate() under the un-merged LLM adapter (the reference's peft-wrapped LLM decodes with the adapters active): the q / k rows
    # are folded for the call and put back bit for bit; the tokens are those of the merged model (same folded weights)
    rows_before = [L["wqkv"].clone() for L in model._llm["layers"]]
    gen = {k: v for k, v in gb.items() if k != "labels"}
    model.eval()
    out = model.generate(audio_values=mel, max_new_tokens=4, eos_token_id=-1, return_dict_in_generate=True, **gen)
    assert model.text_lora_r == cfg.text_model_lora_config["r"]
    assert all(torch.equal(L["wqkv"], w) for L, w in zip(model._llm["layers"], rows_before))
    step = model.forward(input_ids=out.sequences[:, -1:], past_key_values=out.past_key_values)      # the cache path folds per call
    assert all(torch.equal(L["wqkv"], w) for L, w in zip(model._llm["layers"], rows_before))
    model.merge_and_unload()
    merged = model.generate(audio_values=mel, max_new_tokens=5, eos_token_id=-1, return_dict_in_generate=True, output_logits=True, **gen)
    assert torch.equal(merged.sequences[:, :-1], out.sequences)
    assert rel_l2(step.logits[:, -1], merged.logits[4]) < (1e-4 if dtype == torch.float32 else 3e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_merge_and_unload_matches_adapter_forward(dtype):
    """merge_and_unload folds W += scaling * B A into the packed tower weights: same logits as the adapter forward (up to
    the rounding of the merged weights in bf16), and generate() works afterwards."""
    from oracle.reference_cpu import synthetic_batch
    from test_model_gpu import SMALL
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.frontend import WhisperFeatureExtractor
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import init_lora_state_dict, random_state_dict
    cfg = UltravoxConfig(**SMALL, audio_model_lora_config={"r": 4}, text_model_lora_config={"r": 8})
    sd = random_state_dict(cfg, seed=43, dtype=dtype)
    sd.update(init_lora_state_dict(cfg, seed=43, dtype=dtype, random_b=True))
    model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=dtype)
    b = synthetic_batch(cfg, 2, 3.0, n_text=24, audio_start=5, n_supervised=8)
    pcm = b.pop("pcm")
    mel = WhisperFeatureExtractor(cfg.audio_config.num_mel_bins).logmel_device(pcm.to(DEV)).to(dtype)
    gb = {k: v.to(DEV) for k, v in b.items()}
    before = model.forward(audio_values=mel, **gb).logits.float()
    model.merge_and_unload()
    assert model.lora_r == 0 and model.text_lora_r == 0 and set(model.projector_state_dict()) == {k for k in model.projector_state_dict() if k.startswith("multi_modal_projector.")}
    after = model.forward(audio_values=mel, **gb).logits.float()
    assert rel_l2(after, before) < (1e-5 if dtype == torch.float32 else 2e-2)
    gen = {k: v for k, v in gb.items() if k != "labels"}
    out = model.generate(audio_values=mel, max_new_tokens=4, eos_token_id=-1, **gen)
    assert out.shape[1] == gb["input_ids"].shape[1] + 4
    # the reference's book-keeping of a merged model (ultravox_model.py:528-559): base ids cleared, LoRA configs gone, every tower
    # parameter a keep_param - and the export that follows from it: save_pretrained writes the merged towers whole (read back from
    # the packed device weights), a reload from that directory ALONE computes the same logits bit for bit
    import tempfile
    from ultravox_amd.weights import random_state_dict as rsd
    assert model.config.audio_model_id is None and model.config.text_model_id is None
    assert not hasattr(model.config, "audio_model_lora_config") and not hasattr(model.config, "text_model_lora_config")
    towers = {k for k in rsd(cfg, seed=0, dtype=dtype) if not k.startswith("multi_modal_projector.")}
    assert towers <= model.keep_params and not any("lora_" in k for k in model.keep_params)
    with tempfile.TemporaryDirectory() as d:
        saved = model.save_pretrained(d)
        assert set(saved) == towers | set(model.projector_state_dict())
        again = UltravoxModel.from_pretrained(d, device=DEV, dtype=dtype, seed=999)      # (a different seed: nothing may come from the random base)
    assert again.lora_r == 0 and again.text_lora_r == 0
    assert torch.equal(again.forward(audio_values=mel, **gb).logits, model.forward(audio_values=mel, **gb).logits)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_llm_lora_step_matches_the_reference_model_fixture(dtype):
    """uvx_llm_fwd_lora / uvx_llm_bwd_lora inside a whole training step against the REFERENCE model with
    text_model_lora_config r = 4 itself (fixture lora_forward_reference.npz: imported reference, apply_lora through
    tests/peft_stub.py, seeded tiny model with non-zero adapters, audio tower stubbed): loss, logits, the projector's and
    every adapter matrix's gradient."""
    from test_oracle_pinning import load_lora_forward_fixture
    from ultravox_amd.model import UltravoxModel
    cfg, sd, batch, enc, exp = load_lora_forward_fixture()
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
    model.train()
    model.forward_backward(audio_values=mel, **gb)
    mine = model.projector_grads()
    assert sorted(mine) == sorted(exp["grads"])
    for k, g in exp["grads"].items():
        assert rel_l2(mine[k], g) < (2e-3 if dtype == torch.float32 else 0.1), k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_kl_step_under_llm_lora_matches_the_reference_model_fixture(dtype):
    """KL distillation WITH an LLM LoRA adapter: the reference's teacher pass is `self.language_model.forward`
    (ultravox_model.py:212-222) - the same adapted model, adapters active, no_grad - so both passes run uvx_llm_fwd_lora.  Fixture
    kl_lora_forward_reference.npz is the imported reference in training mode (text_model_lora_config r = 4, KL_Divergence,
    non-zero lora_B): loss, projector and adapter gradients."""
    import forward_fixture_util as U
    from test_oracle_pinning import load_lora_forward_fixture
    from ultravox_amd.config import LossConfig, LossFunction
    from ultravox_amd.model import UltravoxModel
    cfg, sd, batch, enc, exp = load_lora_forward_fixture("kl_lora_forward_reference")
    model = UltravoxModel(cfg, state_dict={k: v.to(dtype) for k, v in sd.items()}, device=DEV, dtype=dtype)
    model.set_loss_config(LossConfig(loss_function=LossFunction.KL_Divergence, kl_temperature=exp["meta"]["kl_temperature"],
                                     eot_loss_weight=exp["meta"]["eot_loss_weight"]))
    tower = enc.to(DEV, dtype)
    model.audio_tower_forward = lambda audio_values, audio_len: tower[: audio_values.shape[0]]
    gb = {k: v.to(DEV) for k, v in {**batch, **U.alt_batch()}.items()}
    model.train()
    loss = model.forward_backward(audio_values=torch.zeros(len(enc), 80, 3000, device=DEV, dtype=dtype), **gb)
    want = exp["loss"]
    assert abs(loss.item() - want) < (1e-5 + 1e-4 * want if dtype == torch.float32 else 0.05 * want + 1e-4)
    mine = model.projector_grads()
    assert sorted(mine) == sorted(exp["grads"])
    for k, g in exp["grads"].items():
        assert rel_l2(mine[k], g) < (2e-3 if dtype == torch.float32 else 0.1), k


@pytest.mark.parametrize("r", [4, 8])
def test_gelu_epilogues_and_paired_adapter_launches_are_bit_identical_to_the_separate_kernels(r):
    """Round 6: the training tower's GELU / GELU backward in the fc1 / fc2-dgrad GEMM epilogues (tuning option 21 = 0, the default) against the
    separate gelu_* launches of rounds 3-5 (option 21 = 1): same loss and the same projector + adapter gradients, bit for bit (the epilogues
    restate the kernels' arithmetic and rounding points)."""
    from ultravox_amd import _lib
    L = _lib.lib()
    cfg, sd, model, oracle, gb, ob, mel = _setup(torch.bfloat16, r=r)
    model.train()

    def run():
        loss = model.forward_backward(audio_values=mel, **gb)
        torch.cuda.synchronize()
        return loss.clone(), {k: v.clone() for k, v in model.projector_grads().items()}

    try:
        L.uvx_set_option(21, 1)
        L.uvx_set_option(22, 1)      # (and the q_proj / k_proj adapter products as separate launches: option 22 = 1; paired by default)
        loss0, g0 = run()
        for o21 in (0, 1, 0):
            L.uvx_set_option(21, o21)
            L.uvx_set_option(22, o21)
            loss, g = run()
            assert torch.equal(loss, loss0), o21
            for k in g0:      # (every gradient, the RMSNorm weights' included: their block partials are summed in a fixed order since round 6)
                assert torch.equal(g[k], g0[k]), (o21, k, (g[k] != g0[k]).sum().item())
        assert sum(v.abs().sum().item() for v in g0.values()) > 0
    finally:
        L.uvx_set_option(21, 0)
        L.uvx_set_option(22, 0)


def test_llm_only_training_builds_the_language_model_alone(tmp_path):
    """config.llm_only_training (ultravox_config.py:120; ultravox_model.py:62-67; the LLMOnlyModelPack of training/model_types.py:139-163: the
    reference's text-only LoRA pre-stage): no audio tower, no projector - the state dict needs (and the model holds) the language model and
    its adapters only; a text batch trains exactly as in the full model (same loss, same adapter gradients, bit for bit); audio inputs hit the
    missing attribute; without adapters there is nothing to optimise; save / load round-trips the adapter keys."""
    from oracle.reference_cpu import synthetic_batch
    from test_model_gpu import SMALL
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.model import UltravoxModel, UltravoxTrainer
    from ultravox_amd.weights import init_lora_state_dict, random_state_dict
    dtype = torch.bfloat16
    full_cfg = UltravoxConfig(**SMALL, text_model_lora_config={"r": 8, "lora_alpha": 16})
    sd = random_state_dict(full_cfg, seed=43, dtype=dtype)
    sd.update(init_lora_state_dict(full_cfg, seed=43, dtype=dtype, random_b=True))
    cfg = UltravoxConfig(**SMALL, text_model_lora_config={"r": 8, "lora_alpha": 16}, llm_only_training=True)
    text_sd = {k: v for k, v in sd.items() if k.startswith("language_model.")}
    model = UltravoxModel(cfg, state_dict=text_sd, device=DEV, dtype=dtype)           # tower / projector keys are not needed
    full = UltravoxModel(full_cfg, state_dict=sd, device=DEV, dtype=dtype)
    names = model.trainable_parameter_names()
    assert names and all(k.startswith("language_model.") and "lora_" in k for k in names)
    assert len(names) == 4 * cfg.text_config.num_hidden_layers
    b = synthetic_batch(full_cfg, 2, 1.0, n_text=24, audio_start=5, n_supervised=8)
    tb = {k: b[k].to(DEV) for k in ("input_ids", "attention_mask", "labels")}
    model.train(); full.train()
    loss, ref = model.forward_backward(**tb), full.forward_backward(**tb)
    assert torch.equal(loss, ref)
    mine, theirs = model.projector_grads(), full.projector_grads()
    assert set(mine) == set(names)
    for k in names:
        assert torch.equal(mine[k], theirs[k]), k
    assert max(g.abs().max().item() for g in mine.values()) > 0
    with pytest.raises(AttributeError, match="audio_tower"):
        model.forward(audio_values=torch.zeros(2, 80, 100, device=DEV), audio_token_start_idx=torch.zeros(2, dtype=torch.long),
                      audio_token_len=torch.ones(2, dtype=torch.long), audio_lens=torch.full((2,), 100), audio_batch_size=torch.ones(2, dtype=torch.long),
                      **{k: v for k, v in tb.items() if k != "labels"})
    trainer = UltravoxTrainer(model, lr=1e-3)
    before = {k: v.clone() for k, v in model.projector_state_dict().items()}
    trainer.train_step(**tb)
    assert any(not torch.equal(before[k], v) for k, v in model.projector_state_dict().items())
    model.save_pretrained(str(tmp_path))
    from safetensors.torch import load_file
    saved = load_file(str(tmp_path / "model.safetensors"))
    assert set(saved) == set(names)
    frozen = UltravoxModel(UltravoxConfig(**SMALL, llm_only_training=True), state_dict=text_sd, device=DEV, dtype=dtype)
    assert frozen.trainable_parameter_names() == []
    with pytest.raises(ValueError, match="empty parameter list"):
        UltravoxTrainer(frozen)
    out = frozen.generate(tb["input_ids"][:, :10], max_new_tokens=3, eos_token_id=-1)      # inference works as for any text prompt
    assert out.shape == (2, 13)


ALL_A, ALL_T = ["q_proj", "k_proj", "v_proj", "out_proj", "o_proj"], ["q_proj", "k_proj", "v_proj", "out_proj", "o_proj"]
VO = ["v_proj", "out_proj", "o_proj"]


def _targets_setup(dtype, targets, seed=47):
    from oracle.reference_cpu import OracleModel, synthetic_batch
    from test_model_gpu import SMALL
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.frontend import WhisperFeatureExtractor
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import init_lora_state_dict, random_state_dict
    cfg = UltravoxConfig(**SMALL, audio_model_lora_config={"r": 4, "lora_alpha": 8, "target_modules": targets},
                         text_model_lora_config={"r": 8, "lora_alpha": 16, "target_modules": targets})
    sd = random_state_dict(cfg, seed=seed, dtype=dtype)
    sd.update(init_lora_state_dict(cfg, seed=seed, dtype=dtype, random_b=True))
    model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=dtype)
    oracle = OracleModel(cfg, sd, dtype=torch.float32)
    b = synthetic_batch(cfg, 2, 3.0, n_text=24, audio_start=5, n_supervised=8)
    b["audio_lens"] = torch.tensor([300, 230])
    pcm = b.pop("pcm")
    mel = WhisperFeatureExtractor(cfg.audio_config.num_mel_bins).logmel_device(pcm.to(DEV)).to(dtype)
    gb = {k: v.to(DEV) for k, v in b.items()}
    return cfg, sd, model, oracle, gb, {**b, "audio_values": mel.cpu().float()}, mel


@pytest.mark.parametrize("targets", [ALL_A, VO, ["q_proj", "o_proj", "out_proj"]], ids=["qkvo", "vo", "qo"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_lora_target_modules_beyond_q_and_k_train_step_matches_oracle(dtype, targets):
    """target_modules beyond the reference's default list (ultravox_config.py:19-21 -> peft, ultravox_model.py:695-707): adapters on
    v_proj and on the output projection (out_proj in Whisper, o_proj in the LLM), alone or next to q / k, in both towers - ABI 17's
    uvx_enc_lora_layer_t.v / .o.  Loss, logits, the projector's and every adapter matrix's gradient against the oracle's autograd (the
    oracle with these adapters is pinned to the reference's apply_lora by tests/golden/lora_targets_reference.npz)."""
    cfg, sd, model, oracle, gb, ob, mel = _targets_setup(dtype, targets)
    ne, nl = cfg.audio_config.encoder_layers, cfg.text_config.num_hidden_layers
    n_a = len([m for m in targets if m != "o_proj"])
    n_t = len([m for m in targets if m != "out_proj"])
    assert len(oracle.trainable) == 4 + 2 * n_a * ne + 2 * n_t * nl
    ref, grads, _ = oracle.train_step(ob)
    out = model.forward(audio_values=mel, **gb)
    if dtype == torch.float32:
        assert (out.logits.cpu() - ref["logits"]).abs().max().item() < 1e-3
    else:
        assert rel_l2(out.logits, ref["logits"]) < 3e-2
    model.train()
    loss = model.forward_backward(audio_values=mel, **gb)
    assert abs(loss.item() - ref["loss"].item()) < (1e-4 if dtype == torch.float32 else 2e-2) * abs(ref["loss"].item())
    mine = model.projector_grads()
    assert set(mine) == set(grads)
    tol = 2e-3 if dtype == torch.float32 else 8e-2
    for k, g in grads.items():
        assert g.abs().max().item() > 0, k
        assert rel_l2(mine[k], g) < tol, (k, rel_l2(mine[k], g))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_lora_on_v_and_o_merges_and_decodes_like_the_adapter_forward(dtype):
    """merge_and_unload and the per-call fold of generate() / cached forward() with adapters on v_proj / o_proj: the folded wqkv rows and wo
    give the adapter forward's logits; the un-merged model decodes the merged model's tokens and its weights come back bit for bit."""
    cfg, sd, model, oracle, gb, ob, mel = _targets_setup(dtype, ALL_A, seed=53)
    before = model.forward(audio_values=mel, **gb).logits.float()
    keep = [(L["wqkv"].clone(), L["wo"].clone()) for L in model._llm["layers"]]
    gen = {k: v for k, v in gb.items() if k != "labels"}
    model.eval()
    out = model.generate(audio_values=mel, max_new_tokens=4, eos_token_id=-1, **gen)
    assert all(torch.equal(L["wqkv"], a) and torch.equal(L["wo"], b) for L, (a, b) in zip(model._llm["layers"], keep))
    model.merge_and_unload()
    assert any(not torch.equal(L["wo"], b) for L, (a, b) in zip(model._llm["layers"], keep))
    after = model.forward(audio_values=mel, **gb).logits.float()
    assert rel_l2(after, before) < (1e-5 if dtype == torch.float32 else 2e-2)
    merged = model.generate(audio_values=mel, max_new_tokens=4, eos_token_id=-1, **gen)
    assert torch.equal(merged, out)


MLP_T = ["q_proj", "fc1", "fc2", "gate_proj", "up_proj", "down_proj"]


@pytest.mark.parametrize("targets", [MLP_T, ["fc2", "up_proj"], ["fc1", "gate_proj", "down_proj", "v_proj"]], ids=["q+mlp", "fc2+up", "fc1+gate+down+v"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_lora_on_the_mlp_linears_train_step_matches_oracle(dtype, targets):
    """target_modules naming the MLP's linears (ultravox_config.py:19-21 hands any list to peft, ultravox_model.py:695-707): fc1 / fc2 in the
    Whisper encoder, gate_proj / up_proj / down_proj in the LLM - ABI 18's uvx_enc_lora_layer_t.g / .u / .d.  The fc1 / gate / up terms join the
    pre-activation (so the fused GELU / SwiGLU epilogues step aside), the fc2 / down terms the residual branch.  Loss, logits, the projector's and
    every adapter matrix's gradient against the oracle's autograd (pinned to the reference's apply_lora by lora_targets_reference.npz, case "mlp")."""
    from ultravox_amd.config import LORA_ATTN_MODULES, LORA_MLP_MODULES
    cfg, sd, model, oracle, gb, ob, mel = _targets_setup(dtype, targets)
    ne, nl = cfg.audio_config.encoder_layers, cfg.text_config.num_hidden_layers
    n_a = len([m for m in targets if m in LORA_ATTN_MODULES["audio"] + LORA_MLP_MODULES["audio"]])
    n_t = len([m for m in targets if m in LORA_ATTN_MODULES["text"] + LORA_MLP_MODULES["text"]])
    assert len(oracle.trainable) == 4 + 2 * n_a * ne + 2 * n_t * nl
    ref, grads, _ = oracle.train_step(ob)
    out = model.forward(audio_values=mel, **gb)
    if dtype == torch.float32:
        assert (out.logits.cpu() - ref["logits"]).abs().max().item() < 1e-3
    else:
        assert rel_l2(out.logits, ref["logits"]) < 3e-2
    model.train()
    loss = model.forward_backward(audio_values=mel, **gb)
    assert abs(loss.item() - ref["loss"].item()) < (1e-4 if dtype == torch.float32 else 2e-2) * abs(ref["loss"].item())
    mine = model.projector_grads()
    assert set(mine) == set(grads)
    tol = 2e-3 if dtype == torch.float32 else 8e-2
    for k, g in grads.items():
        assert g.abs().max().item() > 0, k
        assert rel_l2(mine[k], g) < tol, (k, rel_l2(mine[k], g))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_lora_on_the_mlp_linears_merges_and_decodes_like_the_adapter_forward(dtype):
    """merge_and_unload and the per-call fold of generate() with MLP adapters: the folded fc1 / fc2 / gate|up rows (alternating 16-row blocks of
    the packed matrix) / down_proj give the adapter forward's logits; the un-merged model decodes the merged model's tokens and its weights come
    back bit for bit."""
    cfg, sd, model, oracle, gb, ob, mel = _targets_setup(dtype, MLP_T, seed=61)
    before = model.forward(audio_values=mel, **gb).logits.float()
    keep = [(L["wgu"].clone(), L["wd"].clone()) for L in model._llm["layers"]]
    gen = {k: v for k, v in gb.items() if k != "labels"}
    model.eval()
    out = model.generate(audio_values=mel, max_new_tokens=4, eos_token_id=-1, **gen)
    assert all(torch.equal(L["wgu"], a) and torch.equal(L["wd"], b) for L, (a, b) in zip(model._llm["layers"], keep))
    model.merge_and_unload()
    assert all(not torch.equal(L["wgu"], a) and not torch.equal(L["wd"], b) for L, (a, b) in zip(model._llm["layers"], keep))
    after = model.forward(audio_values=mel, **gb).logits.float()
    assert rel_l2(after, before) < (1e-5 if dtype == torch.float32 else 2e-2)
    merged = model.generate(audio_values=mel, max_new_tokens=4, eos_token_id=-1, **gen)
    assert torch.equal(merged, out)


def test_lora_on_the_mlp_of_a_gemma3_backbone_joins_the_branch_before_its_post_norm():
    """Gemma-3: down_proj's output is normalised (post_feedforward_layernorm) before the residual add, the activation is GeGLU - the down_proj
    adapter's term and its gradient sit BEHIND that norm.  f32 against the oracle's autograd."""
    from oracle.reference_cpu import OracleModel, logmel_ref, synthetic_batch
    from test_gemma3_gpu import _cfg
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import init_lora_state_dict, random_state_dict
    cfg = _cfg(layers=3, text_model_lora_config={"r": 4, "lora_alpha": 8, "target_modules": ["q_proj", "gate_proj", "up_proj", "down_proj"]})
    sd = random_state_dict(cfg, seed=67)
    sd.update(init_lora_state_dict(cfg, seed=67, random_b=True))
    model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=torch.float32)
    oracle = OracleModel(cfg, sd, dtype=torch.float32)
    assert len(oracle.trainable) == 4 + 8 * 3
    b = synthetic_batch(cfg, 2, 2.0, n_text=24, audio_start=5, n_supervised=8)
    b["audio_values"] = logmel_ref(b.pop("pcm"), 80)
    gb = {k: v.to(DEV) for k, v in b.items()}
    ref, grads, _ = oracle.train_step(b)
    model.train()
    loss = model.forward_backward(**gb)
    assert abs(loss.item() - ref["loss"].item()) < 1e-4 * abs(ref["loss"].item())
    mine = model.projector_grads()
    assert set(mine) == set(grads)
    for k, g in grads.items():
        assert g.abs().max().item() > 0, k
        assert rel_l2(mine[k], g) < 2e-3, (k, rel_l2(mine[k], g))


def test_lora_on_o_proj_of_a_gemma3_backbone_joins_the_branch_before_its_post_norm():
    """Gemma-3's decoder layer normalises o_proj's output (post_attention_layernorm) before the residual add: the o_proj adapter's term and
    its gradient sit BEHIND that norm (and q / k adapters in front of q_norm / k_norm).  f32 against the oracle's autograd."""
    from oracle.reference_cpu import OracleModel, logmel_ref, synthetic_batch
    from test_gemma3_gpu import _cfg
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import init_lora_state_dict, random_state_dict
    cfg = _cfg(layers=3, text_model_lora_config={"r": 4, "lora_alpha": 8, "target_modules": ["q_proj", "k_proj", "v_proj", "o_proj"]})
    sd = random_state_dict(cfg, seed=59)
    sd.update(init_lora_state_dict(cfg, seed=59, random_b=True))
    model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=torch.float32)
    oracle = OracleModel(cfg, sd, dtype=torch.float32)
    assert len(oracle.trainable) == 4 + 8 * 3
    b = synthetic_batch(cfg, 2, 2.0, n_text=24, audio_start=5, n_supervised=8)
    b["audio_values"] = logmel_ref(b.pop("pcm"), 80)
    gb = {k: v.to(DEV) for k, v in b.items()}
    ref, grads, _ = oracle.train_step(b)
    model.train()
    loss = model.forward_backward(**gb)
    assert abs(loss.item() - ref["loss"].item()) < 1e-4 * abs(ref["loss"].item())
    mine = model.projector_grads()
    assert set(mine) == set(grads)
    for k, g in grads.items():
        assert g.abs().max().item() > 0, k
        assert rel_l2(mine[k], g) < 2e-3, (k, rel_l2(mine[k], g))
