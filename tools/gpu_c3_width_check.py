"""GPU check (not yet a test: written after the round-1 GPU budget was spent, run it first next round and then move it into
tests/): the C3 shapes at full WIDTH and reduced depth - whisper-large-v3 width (1280 / 5120 / 20 heads, 128 mel bins) with
2 layers + Llama-3-8B width with 1 layer, 2 x 30 s clips - one train step (projector + rank-8 encoder LoRA) against the
f32 CPU oracle.  What C3 adds over the C2 width test: K = 1280 / N = 3840 / 5120 GEMM shapes, 128-bin log-mel and conv1
im2col (K = 384), 20-head D = 64 attention, the 8 x 1280 = 10240-wide projector input.
usage: PYTHONPATH=. python tools/gpu_c3_width_check.py"""
import sys
import torch
from oracle.reference_cpu import OracleModel, synthetic_batch
from ultravox_amd.config import AUDIO_PRESETS, TEXT_PRESETS, UltravoxConfig
from ultravox_amd.frontend import WhisperFeatureExtractor
from ultravox_amd.model import UltravoxModel
from ultravox_amd.weights import init_lora_state_dict, random_state_dict

DEV = "cuda"


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


tc = dict(TEXT_PRESETS["meta-llama/Meta-Llama-3-8B-Instruct"], num_hidden_layers=1)
ac = dict(AUDIO_PRESETS["openai/whisper-large-v3"], encoder_layers=2)
ok = True
for lora in (None, {"r": 8}):
    cfg = UltravoxConfig(text_config=tc, audio_config=ac, hidden_size=4096, stack_factor=8, projector_ln_mid=True,
                         torch_dtype="bfloat16", audio_model_lora_config=lora)
    sd = random_state_dict(cfg, seed=3, dtype=torch.bfloat16, device="cuda")
    if lora:
        sd.update(init_lora_state_dict(cfg, seed=3, dtype=torch.bfloat16, random_b=True))
    model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=torch.bfloat16, rope_len=512)
    oracle = OracleModel(cfg, {k: v.cpu() for k, v in sd.items()}, dtype=torch.float32)
    b = synthetic_batch(cfg, 2, 30.0, n_text=64, audio_start=8, n_supervised=16)
    pcm = b.pop("pcm")
    mel = WhisperFeatureExtractor(cfg.audio_config.num_mel_bins).logmel_device(pcm.to(DEV))
    assert mel.shape[1] == 128
    gb = {k: v.to(DEV) for k, v in b.items()}
    ob = {**b, "audio_values": mel.cpu().bfloat16().float()}
    torch.set_num_threads(32)
    ref, grads, _ = oracle.train_step(ob)
    model.train()
    loss = model.forward_backward(audio_values=mel, **gb)
    d_loss = abs(loss.item() - ref["loss"].item()) / abs(ref["loss"].item())
    mine = model.projector_grads()
    worst = max((rel_l2(mine[k], g), k) for k, g in grads.items())
    good = d_loss < 2e-2 and worst[0] < 8e-2
    ok &= good
    print(f"lora={lora}: loss {loss.item():.5f} vs oracle {ref['loss'].item():.5f} (rel {d_loss:.2e}); worst gradient rel-L2 "
          f"{worst[0]:.3e} ({worst[1]}) over {len(grads)} tensors -> {'OK' if good else 'FAIL'}", flush=True)
sys.exit(0 if ok else 1)
