"""dtype-for-dtype parity of the PRODUCTION (bf16) path: the HIP kernels against the oracle run in bf16 on the CPU.

test_model_gpu.py holds the bf16 kernels to a few percent against an f32 oracle - a bound wide enough to hide a
systematic rounding-point error of a few 1e-3.  Here both sides compute in bf16 with the same rounding points (torch's
module-by-module rounding: every linear / norm / activation / residual output is rounded once; attention as a fused kernel
does it - f32 scores and statistics, bf16 probabilities, see oracle.reference_cpu.FUSED_ATTENTION), so what is left is
accumulation order and the flash kernels' running-max rescaling.  Bounds (SURVEY.md §7 "Tolerance" plan ii), >= 5x tighter
than the f32-oracle bars:   logits rel-L2 <= 6e-3 (was 3e-2), projector gradients <= 1.2e-2 (was 6e-2 / 8e-2),
per-stage activations <= 4e-3 (was 2e-2), loss within 2e-3 relative (was 2e-2).
max-abs and rel-L2 per stage are recorded to gpurun_out/parity/ (committed under profiles/ each round).

The optimizer is pinned ELEMENTWISE: clip + AdamW in f32-master mode applied to the gradients the device produced must equal
torch.optim.AdamW + clip_grad_norm_ on those same gradients to f32 round-off (the whole-step comparison in test_model_gpu.py
can only bound the update loosely, because Adam's first steps are +-lr * sign(g) and near-zero gradients flip sign)."""
import pytest
import torch

from parity_util import oracle_threads, record, rel_l2, stage_errors, width_config

pytestmark = pytest.mark.gpu
DEV = "cuda"

SMALL = dict(
    audio_config=dict(d_model=128, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=256, num_mel_bins=80,
                      max_source_positions=1500),
    text_config=dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                     num_key_value_heads=2, vocab_size=512, rope_theta=10000.0, max_position_embeddings=512, eos_token_id=2),
    hidden_size=256, stack_factor=8, projector_ln_mid=True)

BOUND = {"stage": 4e-3, "logits": 6e-3, "grads": 1.2e-2, "loss": 2e-3}


def _compare(cfg, sd, B, seconds, n_text, audio_start, n_sup, name):
    from oracle.reference_cpu import OracleModel, fused_attention, logmel_ref, synthetic_batch
    from ultravox_amd.frontend import WhisperFeatureExtractor
    from ultravox_amd.model import UltravoxModel
    model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=torch.bfloat16, rope_len=512)
    o16 = OracleModel(cfg, {k: v.cpu() for k, v in sd.items()}, dtype=torch.bfloat16)
    b = synthetic_batch(cfg, B, seconds, n_text=n_text, audio_start=audio_start, n_supervised=n_sup)
    pcm = b.pop("pcm")
    mel = WhisperFeatureExtractor(cfg.audio_config.num_mel_bins).logmel_device(pcm.to(DEV))
    rec = {"mel_f32": stage_errors(mel, logmel_ref(pcm, cfg.audio_config.num_mel_bins))}
    gb = {k: v.to(DEV) for k, v in b.items()}
    ob = {**b, "audio_values": mel.cpu().bfloat16()}          # both sides start from the SAME bf16 mel
    oracle_threads()
    with fused_attention():
        ref, grads, _ = o16.train_step(ob)
        with torch.no_grad():
            enc_ref, emb_ref = o16.audio_embeds(ob["audio_values"], ob["audio_lens"])
    enc = model.audio_tower_forward(mel, gb["audio_lens"])
    rec["encoder_out"] = stage_errors(enc, enc_ref)
    rec["audio_embeds"] = stage_errors(model.multi_modal_projector_forward(enc), emb_ref)
    out = model.forward(audio_values=mel, **gb)
    rec["logits"] = stage_errors(out.logits, ref["logits"])
    rec["loss"] = {"hip": out.loss.item(), "oracle_bf16": ref["loss"].item()}
    model.train()
    loss = model.forward_backward(audio_values=mel, **gb)
    mine = model.projector_grads()
    rec["grads"] = {k: stage_errors(mine[k], g) for k, g in grads.items()}
    rec["bounds"] = BOUND
    record(name, rec)
    assert rec["mel_f32"]["max_abs"] < 1e-3
    assert rec["encoder_out"]["rel_l2"] < BOUND["stage"] and rec["audio_embeds"]["rel_l2"] < BOUND["stage"], rec
    assert rec["logits"]["rel_l2"] < BOUND["logits"], rec["logits"]
    for l in (out.loss.item(), loss.item()):
        assert abs(l - ref["loss"].item()) < BOUND["loss"] * abs(ref["loss"].item()), rec["loss"]
    for k, v in rec["grads"].items():
        assert v["rel_l2"] < BOUND["grads"], (k, v)
    return model


def test_small_config_stage_by_stage_against_the_bf16_oracle():
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.weights import random_state_dict
    cfg = UltravoxConfig(**SMALL)
    sd = {k: v.bfloat16() for k, v in random_state_dict(cfg, seed=21).items()}
    _compare(cfg, sd, 3, 3.0, 32, 5, 12, "bf16_points_small")


def test_c2_width_stage_by_stage_against_the_bf16_oracle():
    from ultravox_amd.weights import random_state_dict
    cfg = width_config("meta-llama/Meta-Llama-3-8B-Instruct", "openai/whisper-medium", 2, 2)
    sd = random_state_dict(cfg, seed=3, dtype=torch.bfloat16, device="cuda")
    _compare(cfg, sd, 2, 30.0, 128, 16, 32, "bf16_points_c2_width")


@pytest.mark.parametrize("clip", [1.0, 1e9], ids=["clipped", "unclipped"])
def test_adamw_update_is_pinned_elementwise_on_the_device_gradients(clip):
    from oracle.reference_cpu import logmel_ref, synthetic_batch
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.model import UltravoxModel, UltravoxTrainer
    from ultravox_amd.weights import random_state_dict
    cfg = UltravoxConfig(**SMALL)
    sd = {k: v.bfloat16() for k, v in random_state_dict(cfg, seed=22).items()}
    model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=torch.bfloat16)
    trainer = UltravoxTrainer(model, lr=2e-3, master_weights=True, max_grad_norm=clip)
    b = synthetic_batch(cfg, 3, 3.0, n_text=32, audio_start=5, n_supervised=12)
    b["audio_values"] = logmel_ref(b.pop("pcm"), 80)
    gb = {k: v.to(DEV) for k, v in b.items()}
    p = trainer.master.detach().cpu().clone().requires_grad_(True)          # torch.optim.AdamW on the same flat f32 vector
    opt = torch.optim.AdamW([p], lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    for step in range(3):
        trainer.train_step(**gb)
        torch.cuda.synchronize()
        p.grad = model.proj_grad.detach().cpu().clone()                     # the gradients the device produced
        gn = torch.nn.utils.clip_grad_norm_([p], clip)
        opt.step()
        assert abs(trainer.grad_norm().item() - gn.item()) <= 1e-4 * gn.item()     # f32 sum of ~1e6 squares, two summation orders
        d = (trainer.master.cpu() - p.detach()).abs().max().item()
        # one AdamW step moves a weight by <= lr = 2e-3; what may differ is f32 round-off (bias corrections computed in f32
        # here and in double by torch; one ulp of a 0.4-sized weight is 3e-8)
        assert d <= 1.5e-7, (step, d)
        assert torch.equal(model.proj_flat.cpu(), trainer.master.bfloat16().cpu())       # bf16 parameters mirror the master
