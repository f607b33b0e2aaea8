"""GPU probe: generate() at the C2 model size - prefill time and decode ms/token (greedy, B prompts of ~316 positions).
usage: PYTHONPATH=. python tools/gpu_decode_probe.py [B] [new_tokens] [text_model_id]
  text_model_id meta-llama/Llama-3.3-70B-Instruct = BASELINE config C4 (141 GB of bf16 weights: loaded with
  consume_state_dict so the peak stays near one copy; weight-streaming floor 141 GB / 8 TB/s = 17.6 ms per token)"""
import sys, time
import torch
from ultravox_amd.config import UltravoxConfig
from ultravox_amd.frontend import WhisperFeatureExtractor
from ultravox_amd.model import UltravoxModel
from ultravox_amd.synthetic import synthetic_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
new = int(sys.argv[2]) if len(sys.argv) > 2 else 64
text_id = sys.argv[3] if len(sys.argv) > 3 else "meta-llama/Meta-Llama-3-8B-Instruct"
dev = "cuda"
free_gb = torch.cuda.mem_get_info()[0] / 2 ** 30
need_gb = 170 if "70B" in text_id else 40          # bf16 weights + KV cache + workspaces, with headroom
if free_gb < need_gb:                              # never drive the box out of memory: a dead box is a strike
    print(f"SKIP: {free_gb:.0f} GiB free on the device, {text_id} needs about {need_gb} GiB")
    sys.exit(0)
cfg = UltravoxConfig(audio_model_id="openai/whisper-medium", text_model_id=text_id,
                     hidden_size=4096, stack_factor=8, projector_ln_mid=True, torch_dtype="bfloat16")
model = UltravoxModel(cfg, device=dev, dtype=torch.bfloat16, seed=0, rope_len=1024, with_backward=False, consume_state_dict=True)
print(f"{text_id}: {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB peak while loading, {torch.cuda.memory_allocated() / 2**30:.1f} GiB resident", flush=True)
batch = synthetic_batch(cfg, B, 30.0, n_text=128, audio_start=16, n_supervised=32)
pcm = batch.pop("pcm").to(dev)
batch.pop("labels")
mel = WhisperFeatureExtractor(cfg.audio_config.num_mel_bins, device=dev).logmel_device(pcm)
gb = {k: v.to(dev) for k, v in batch.items()}
for n in (1, new):       # n = 1: prefill only (+ one argmax); n = new: prefill + (new - 1) decode steps
    model.generate(audio_values=mel, max_new_tokens=n, eos_token_id=-1, **gb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = model.generate(audio_values=mel, max_new_tokens=n, eos_token_id=-1, **gb)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if n == 1: t_prefill = dt
    print(f"B={B} max_new_tokens={n}: {dt * 1e3:.1f} ms total, output {tuple(out.shape)}", flush=True)
print(f"prefill (encoder + projector + LLM prefill) {t_prefill * 1e3:.1f} ms; decode {(dt - t_prefill) / (new - 1) * 1e3:.2f} ms/token "
      f"= {B * (new - 1) / (dt - t_prefill):.0f} tokens/s at batch {B}")
