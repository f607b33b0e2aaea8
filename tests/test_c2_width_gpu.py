"""The C2 shapes at full WIDTH and reduced depth: Llama-3-8B dimensions (4096 / 14336 / 32:8 heads / 128256 vocab) with
2 layers, whisper-medium width (1024 / 4096 / 16 heads) with 2 layers, 4 x 30 s clips + 128 text tokens (M = 1264 LLM rows,
6000 encoder rows).  The small-config tests exercise the
arithmetic; this one exercises what only appears at real sizes - the eight-phase GEMM variants and their fused epilogues
at N = 28672, the tile picker's choices, the split-K (12-way) head dgrad on the supervised rows, head_dim 128 attention -
against the f32 CPU oracle on the same bf16-rounded weights (tolerances as in test_model_gpu.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def test_c2_width_train_step_matches_oracle():
    from oracle.reference_cpu import OracleModel, synthetic_batch
    from ultravox_amd.config import AUDIO_PRESETS, TEXT_PRESETS, UltravoxConfig
    from ultravox_amd.frontend import WhisperFeatureExtractor
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import random_state_dict
    tc = dict(TEXT_PRESETS["meta-llama/Meta-Llama-3-8B-Instruct"], num_hidden_layers=2)
    ac = dict(AUDIO_PRESETS["openai/whisper-medium"], encoder_layers=2)
    cfg = UltravoxConfig(text_config=tc, audio_config=ac, hidden_size=4096, stack_factor=8, projector_ln_mid=True,
                         torch_dtype="bfloat16")
    sd = random_state_dict(cfg, seed=3, dtype=torch.bfloat16, device="cuda")
    model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=torch.bfloat16, rope_len=512)
    oracle = OracleModel(cfg, {k: v.cpu() for k, v in sd.items()}, dtype=torch.float32)
    b = synthetic_batch(cfg, 4, 30.0, n_text=128, audio_start=16, n_supervised=32)
    pcm = b.pop("pcm")
    mel = WhisperFeatureExtractor(cfg.audio_config.num_mel_bins).logmel_device(pcm.to(DEV))
    gb = {k: v.to(DEV) for k, v in b.items()}
    ob = {**b, "audio_values": mel.cpu().bfloat16().float()}
    torch.set_num_threads(32)
    ref, grads, _ = oracle.train_step(ob)
    out = model.forward(audio_values=mel, **gb)                       # full logits + loss
    assert rel_l2(out.logits, ref["logits"]) < 3e-2
    assert abs(out.loss.item() - ref["loss"].item()) < 2e-2 * abs(ref["loss"].item())
    model.train()
    loss = model.forward_backward(audio_values=mel, **gb)             # supervised-row head, split-K head dgrad
    assert abs(loss.item() - ref["loss"].item()) < 2e-2 * abs(ref["loss"].item())
    mine = model.projector_grads()
    for k, g in grads.items():
        assert rel_l2(mine[k], g) < 8e-2, k
    # the tile picker takes the eight-phase kernels at these sizes (no silent fall-back to the small-tile kernel)
    from ultravox_amd import _lib
    L = _lib.lib()
    M = gb["input_ids"].numel()
    assert M == 4 * 316
    for n, k in ((28672, 4096), (4096, 14336), (6144, 4096), (4096, 4096)):
        assert L.uvx_gemm_pick_variant(M, n, k, 1) in (31, 32, 33, 34), (n, k)
