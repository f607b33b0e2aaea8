"""GPU record (round 6): the C2 configuration at FULL size (Llama-3-8B + whisper-medium, 8 x 30 s, 128 text tokens, random-init weights, the benchmark's synthetic
batch) trained for N optimizer steps twice - the LLM backward on every row (uvx_llm_bwd_train) and from the first audio token (uvx_llm_bwd_train_from) - and the
two loss sequences compared bit for bit: the row-compacted backward changes no gradient bit, so the trajectories are identical."""
import sys
import torch
from ultravox_amd.config import UltravoxConfig
from ultravox_amd.frontend import WhisperFeatureExtractor
from ultravox_amd.model import UltravoxModel, UltravoxTrainer
from ultravox_amd.synthetic import synthetic_batch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda", 0)
cfg = UltravoxConfig(audio_model_id="openai/whisper-medium", text_model_id="meta-llama/Meta-Llama-3-8B-Instruct", hidden_size=4096, stack_factor=8,
                     projector_ln_mid=True, torch_dtype="bfloat16")
curves = []
for skip in (False, True):
    model = UltravoxModel(cfg, device=str(dev), dtype=torch.bfloat16, seed=0, rope_len=1024)
    model.skip_prefix_backward = skip
    trainer = UltravoxTrainer(model, lr=2e-3, max_grad_norm=1.0)
    fe = WhisperFeatureExtractor(cfg.audio_config.num_mel_bins, device=str(dev))
    batch = synthetic_batch(cfg, 8, 30.0, n_text=128, audio_start=16, n_supervised=32)
    pcm = batch.pop("pcm").to(dev)
    batch = {k: v.to(dev) for k, v in batch.items()}
    losses = []
    for step in range(N):
        losses.append(trainer.train_step(audio_values=fe.logmel_device(pcm), **batch).clone())
    torch.cuda.synchronize()
    curves.append(torch.stack(losses).float().cpu())
    flat = model.proj_flat.clone()
    curves.append(flat)
    del model, trainer
    torch.cuda.empty_cache()
full, w_full, skipped, w_skip = curves
print("step   loss (full backward)   loss (from the first audio token)")
for i in range(N):
    print(f"{i:4d}   {full[i].item():.6f}               {skipped[i].item():.6f}")
print("losses bit-identical:", torch.equal(full, skipped), " trained projector bit-identical:", torch.equal(w_full, w_skip),
      " loss", round(full[0].item(), 4), "->", round(full[-1].item(), 4))
