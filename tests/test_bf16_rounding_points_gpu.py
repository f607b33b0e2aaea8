"""dtype-for-dtype parity of the PRODUCTION (bf16) path: are the kernels' bf16 ROUNDING POINTS the reference's?

test_model_gpu.py holds the bf16 kernels to a few percent against an f32 oracle - wide enough to hide a systematic
rounding-point error of a few 1e-3.  Two tighter statements are made here (SURVEY.md §7 "Tolerance", plan ii).

1. PER KERNEL, bit level.  Each production kernel against torch's own bf16 CPU op on the SAME bf16 inputs (what the reference
   executes module by module: every linear / norm / activation / residual / RoPE output rounded once, attention as a fused
   flash kernel rounds it - oracle.reference_cpu.FUSED_ATTENTION).  With identical rounding points the two can differ only
   where f32 accumulation ORDER moves a value across a bf16 rounding boundary: a small fraction of elements, by exactly one
   bf16 ulp.  Asserted: mismatching elements <= 1 % (measured 1e-5 ... 2e-3, recorded; RoPE and SwiGLU are bit-identical),
   never more than 1 ulp apart at the tensor's scale (2 for kernels with two consecutive roundings), rel-L2 <= 5e-4 (measured
   <= 1.2e-4) - a misplaced rounding point would move ~half of all elements by an ulp, rel-L2 ~ 2e-3.

2. WHOLE PATH, calibrated.  Rounding noise is chaotic: one element that rounds the other way perturbs every accumulation
   downstream by ~2^-8 / sqrt(K) and flips a few percent of the NEXT op's roundings, so within 4-5 GEMMs two bf16
   implementations that agree bit-for-bit per kernel are as decorrelated as two independent noise realisations (measured
   here: the same kernel path against the bf16 oracle is NOT closer than against the f32 oracle, at any depth).  A
   multi-layer bound ">= 5x tighter than the f32 bars" is therefore unattainable for ANY implementation that does not
   reproduce torch's accumulation order bit for bit.  What can be asserted is calibration against the truth: the distance of
   the HIP path to the f32 oracle must not exceed the distance of TORCH'S OWN bf16 arithmetic (the bf16 oracle) to the f32
   oracle by more than 25 % - logits, audio embeddings, every projector gradient.  A systematic kernel error of the size
   the f32 bars could hide (a few 1e-3 on 1e-2) fails this.

The optimizer is pinned ELEMENTWISE: clip + AdamW in f32-master mode applied to the gradients the device produced must equal
torch.optim.AdamW + clip_grad_norm_ on those same gradients to f32 round-off (the whole-step comparison in test_model_gpu.py
can only bound the update loosely, because Adam's first steps are +-lr * sign(g) and near-zero gradients flip sign).
Every number goes to gpurun_out/parity/ (committed under profiles/ each round)."""
import math

import pytest
import torch
import torch.nn.functional as F

from parity_util import oracle_threads, record, rel_l2, stage_errors, width_config

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16

SMALL = dict(
    audio_config=dict(d_model=128, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=256, num_mel_bins=80,
                      max_source_positions=1500),
    text_config=dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                     num_key_value_heads=2, vocab_size=512, rope_theta=10000.0, max_position_embeddings=512, eos_token_id=2),
    hidden_size=256, stack_factor=8, projector_ln_mid=True)


def ulp_stats(got: torch.Tensor, want: torch.Tensor) -> dict:
    """bf16 tensors -> fraction of elements that are not bit-equal, and the largest difference in bf16 ulps.  One ulp of x is
    taken as 2^-7 |x| (the spacing at the top of x's binade), with |x| floored at the tensor's RMS: an output that is small
    because its terms cancel (a residual add, a near-zero dot product) carries the rounding of full-sized INTERMEDIATES, and
    counting that in the ulps of the small result (or across a sign change) would measure nothing."""
    g, w = got.detach().float().cpu(), want.detach().float().cpu()
    bits = lambda t: t.to(torch.bfloat16).contiguous().view(torch.int16)
    differ = bits(g) != bits(w)
    floor = w.pow(2).mean().sqrt()
    ulps = (g - w).abs() / (2.0 ** -7 * torch.maximum(w.abs(), floor))
    return {"mismatch_frac": differ.float().mean().item(), "max_ulp": ulps.max().item(), "rel_l2": rel_l2(got, want), "n": w.numel()}


def check(name, got, want, rec, max_ulp=1, max_frac=1e-2, max_rel=5e-4):
    st = ulp_stats(got, want)
    rec[name] = st
    return st["mismatch_frac"] <= max_frac and st["max_ulp"] <= max_ulp + 0.01 and st["rel_l2"] <= max_rel


def test_every_kernel_rounds_where_torch_bf16_rounds():
    from oracle import reference_cpu as O
    from ultravox_amd import ops
    torch.manual_seed(0)
    oracle_threads()
    rec, ok = {}, {}
    g = lambda t: t.to(DEV)
    # ---- GEMM epilogues: nn.Linear (+ bias), + GELU, + residual -------------------------------------------------------
    for (M, N, K) in ((1500, 1024, 1024), (632, 4096, 4096), (300, 1280, 5120)):
        a, w = (0.5 * torch.randn(M, K)).to(BF), (torch.randn(N, K) / math.sqrt(K)).to(BF)
        bias, res = torch.randn(N).to(BF), torch.randn(M, N).to(BF)
        tag = f"gemm_{M}x{N}x{K}"
        ok[tag] = check(tag, ops.gemm(g(a), g(w)), F.linear(a, w), rec)
        ok[tag + "_bias"] = check(tag + "_bias", ops.gemm(g(a), g(w), bias=g(bias)), F.linear(a, w, bias), rec)
        ok[tag + "_bias_gelu"] = check(tag + "_bias_gelu", ops.gemm(g(a), g(w), bias=g(bias), act="gelu"), F.gelu(F.linear(a, w, bias)), rec, max_ulp=2)
        ok[tag + "_bias_residual"] = check(tag + "_bias_residual", ops.gemm(g(a), g(w), bias=g(bias), residual=g(res)), res + F.linear(a, w, bias), rec, max_ulp=2)
    # ---- norms --------------------------------------------------------------------------------------------------------
    for cols in (1024, 4096):
        x, w, b = (2.0 * torch.randn(700, cols) + 0.3).to(BF), (1 + 0.1 * torch.randn(cols)).to(BF), (0.1 * torch.randn(cols)).to(BF)
        ok[f"layernorm_{cols}"] = check(f"layernorm_{cols}", ops.layernorm(g(x), g(w), g(b), 1e-5), F.layer_norm(x, (cols,), w, b, 1e-5), rec)
        ok[f"rmsnorm_{cols}"] = check(f"rmsnorm_{cols}", ops.rmsnorm(g(x), g(w), 1e-5), O.rmsnorm_ref(x, w, 1e-5), rec)
    # ---- SwiGLU (Llama MLP: silu(gate) * up; projector: first half = value) --------------------------------------------
    x = (1.5 * torch.randn(500, 2 * 1024)).to(BF)
    ok["swiglu_gate_first"] = check("swiglu_gate_first", ops.swiglu(g(x), gate_first=True), F.silu(x[:, :1024]) * x[:, 1024:], rec, max_ulp=2)
    ok["swiglu_value_first"] = check("swiglu_value_first", ops.swiglu(g(x), gate_first=False), F.silu(x[:, 1024:]) * x[:, :1024], rec, max_ulp=2)
    # ---- RoPE (three roundings per element in the reference: two products, one sum) -------------------------------------
    from ultravox_amd.config import TextConfig
    tc = TextConfig(hidden_size=1024, num_attention_heads=8, num_key_value_heads=2, head_dim=128, rope_theta=500000.0)
    T, H, dh = 316, 8, 128
    q = torch.randn(2, T, H, dh).to(BF)
    cos, sin = O.rope_cos_sin_ref(tc, T, BF)
    want = (q.transpose(1, 2) * cos + O._rotate_half(q.transpose(1, 2)) * sin).transpose(1, 2)
    cos32, sin32 = O.rope_cos_sin_ref(tc, T, torch.float32)
    table = torch.stack([cos32[:, : dh // 2], sin32[:, : dh // 2]], -1).reshape(T, dh).contiguous()      # [T, dh/2] pairs (cos, sin)
    got = ops.rope_(g(q).reshape(2 * T, H * dh).clone(), g(table), T, H, dh).reshape(2, T, H, dh)
    ok["rope"] = check("rope", got, want, rec)
    # ---- attention forward vs the flash-rounding restatement (encoder: D 64 with key padding; LLM: D 128 causal GQA) ----
    for (D, Hq, Hkv, Tn, causal) in ((64, 4, 4, 1500, False), (128, 8, 2, 316, True)):
        qkv = [torch.randn(2, Tn, h, D).to(BF) for h in (Hq, Hkv, Hkv)]
        kv_len = torch.tensor([Tn, Tn - 37], dtype=torch.int32)
        o, _ = ops.attention(*(g(t) for t in qkv), causal=causal, kv_len=g(kv_len))
        keep = torch.arange(Tn)[None, :] < kv_len[:, None]
        fmin = torch.finfo(torch.float32).min
        mask = (~keep)[:, None, None, :].float() * fmin
        if causal:
            mask = torch.clamp(mask + torch.full((Tn, Tn), fmin).triu(1)[None, None], min=fmin)
        qh, kh, vh = (t.transpose(1, 2) for t in qkv)
        kh, vh = (t.repeat_interleave(Hq // Hkv, dim=1) for t in (kh, vh))
        with O.fused_attention():
            want = O._attend(qh, kh, vh, mask, D ** -0.5).transpose(1, 2).reshape(2, Tn, Hq * D)
        rows = keep if causal else torch.ones_like(keep)            # padded QUERY rows of a causal sequence see no key: undefined
        ok[f"attention_D{D}"] = check(f"attention_D{D}", o.cpu()[rows], want[rows], rec, max_frac=2e-2, max_rel=1e-3)
    record("bf16_rounding_points_per_kernel", rec)
    bad = {k: rec[k] for k, v in ok.items() if not v}
    assert not bad, bad


def _whole_path(cfg, sd, B, seconds, n_text, audio_start, n_sup, name):
    from oracle.reference_cpu import OracleModel, fused_attention, logmel_ref, synthetic_batch
    from ultravox_amd.frontend import WhisperFeatureExtractor
    from ultravox_amd.model import UltravoxModel
    model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=BF, rope_len=512)
    cpu_sd = {k: v.cpu() for k, v in sd.items()}
    b = synthetic_batch(cfg, B, seconds, n_text=n_text, audio_start=audio_start, n_supervised=n_sup)
    pcm = b.pop("pcm")
    mel = WhisperFeatureExtractor(cfg.audio_config.num_mel_bins).logmel_device(pcm.to(DEV))
    rec = {"mel_f32": stage_errors(mel, logmel_ref(pcm, cfg.audio_config.num_mel_bins))}
    assert rec["mel_f32"]["max_abs"] < 1e-3
    gb = {k: v.to(DEV) for k, v in b.items()}
    mel16 = mel.cpu().bfloat16()                                   # all three start from the SAME bf16-rounded mel
    oracle_threads()
    o32 = OracleModel(cfg, cpu_sd, dtype=torch.float32)
    r32, g32, _ = o32.train_step({**b, "audio_values": mel16.float()})
    del o32
    o16 = OracleModel(cfg, cpu_sd, dtype=BF)
    with fused_attention():
        r16, g16, _ = o16.train_step({**b, "audio_values": mel16})
    out = model.forward(audio_values=mel, **gb)
    model.train()
    loss = model.forward_backward(audio_values=mel, **gb)
    mine = model.projector_grads()
    pairs = {"logits": (out.logits, r16["logits"], r32["logits"]), "audio_embeds": (
        model.multi_modal_projector_forward(model.audio_tower_forward(mel, gb["audio_lens"])), r16["audio_embeds"], r32["audio_embeds"])}
    pairs.update({"grad." + k.split(".", 1)[1]: (mine[k], g16[k], g32[k]) for k in g32})
    for k, (hip, t16, t32) in pairs.items():
        rec[k] = {"hip_vs_f32": rel_l2(hip, t32.detach()), "torch_bf16_vs_f32": rel_l2(t16.detach(), t32.detach()),
                  "hip_vs_torch_bf16": rel_l2(hip, t16.detach()),
                  "max_abs_hip_vs_f32": (hip.float().cpu() - t32.detach().float()).abs().max().item()}
    rec["loss"] = {"hip": loss.item(), "torch_bf16": r16["loss"].item(), "f32": r32["loss"].item()}
    record(name, rec)
    for k in pairs:
        assert rec[k]["hip_vs_f32"] <= 1.25 * rec[k]["torch_bf16_vs_f32"] + 1e-4, (k, rec[k])
    assert abs(loss.item() - r32["loss"].item()) <= 1.25 * abs(r16["loss"].item() - r32["loss"].item()) + 2e-3 * abs(r32["loss"].item()), rec["loss"]
    return rec


def test_small_config_is_as_close_to_f32_as_torch_bf16_is():
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.weights import random_state_dict
    cfg = UltravoxConfig(**SMALL)
    sd = {k: v.bfloat16() for k, v in random_state_dict(cfg, seed=21).items()}
    _whole_path(cfg, sd, 3, 3.0, 32, 5, 12, "bf16_calibrated_small")


@pytest.mark.parametrize("depth", [2, 8])
def test_c2_width_is_as_close_to_f32_as_torch_bf16_is(depth):
    """Llama-3-8B + whisper-medium WIDTH at 2 and at 8 + 8 layers: error growth with depth, against both oracles."""
    from ultravox_amd.weights import random_state_dict
    cfg = width_config("meta-llama/Meta-Llama-3-8B-Instruct", "openai/whisper-medium", depth, depth)
    sd = random_state_dict(cfg, seed=3, dtype=BF, device="cuda")
    rec = _whole_path(cfg, sd, 2, 30.0, 128, 16, 32, f"bf16_calibrated_c2_width_depth{depth}")
    assert rec["logits"]["hip_vs_f32"] < 3e-2          # the absolute bars of test_model_gpu.py hold at depth too
    for k, v in rec.items():
        if k.startswith("grad."):
            assert v["hip_vs_f32"] < 8e-2, (k, v)


@pytest.mark.parametrize("clip", [1.0, 1e9], ids=["clipped", "unclipped"])
def test_adamw_update_is_pinned_elementwise_on_the_device_gradients(clip):
    from oracle.reference_cpu import logmel_ref, synthetic_batch
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.model import UltravoxModel, UltravoxTrainer
    from ultravox_amd.weights import random_state_dict
    cfg = UltravoxConfig(**SMALL)
    sd = {k: v.bfloat16() for k, v in random_state_dict(cfg, seed=22).items()}
    model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=BF)
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
