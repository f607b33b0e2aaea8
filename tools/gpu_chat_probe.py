"""GPU probe: a multi-turn conversation at the C2 model size - per-turn prefill latency with the KV cache carried between
turns (uvx_llm_prefill_chunk) against re-prefilling the whole dialogue.  Turn 1 holds 30 s of audio; later turns are text.
usage: PYTHONPATH=. python tools/gpu_chat_probe.py [turns] [reply_tokens] [user_tokens]"""
import sys, time
import torch
from ultravox_amd.config import UltravoxConfig
from ultravox_amd.frontend import WhisperFeatureExtractor
from ultravox_amd.model import UltravoxModel
from ultravox_amd.synthetic import synthetic_batch

turns = int(sys.argv[1]) if len(sys.argv) > 1 else 6
reply = int(sys.argv[2]) if len(sys.argv) > 2 else 48
user = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = "cuda"
cfg = UltravoxConfig(audio_model_id="openai/whisper-medium", text_model_id="meta-llama/Meta-Llama-3-8B-Instruct",
                     hidden_size=4096, stack_factor=8, projector_ln_mid=True, torch_dtype="bfloat16")
model = UltravoxModel(cfg, device=dev, dtype=torch.bfloat16, seed=0, rope_len=4096, with_backward=False)
batch = synthetic_batch(cfg, 1, 30.0, n_text=128, audio_start=16, n_supervised=32)
pcm = batch.pop("pcm").to(dev)
batch.pop("labels")
mel = WhisperFeatureExtractor(cfg.audio_config.num_mel_bins, device=dev).logmel_device(pcm)
first = {k: v.to(dev) for k, v in batch.items()}
g = torch.Generator().manual_seed(0)


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) * 1e3


for mode in ("warm-up", "cached", "re-prefill"):
    state, ids, rows = None, None, []
    for t in range(turns if mode != "warm-up" else 2):
        if t == 0:
            kw = dict(audio_values=mel, **first)
        else:       # the dialogue so far (audio positions are plain ids now, as LocalInference rewrites them) + a new user turn
            ids = torch.cat([ids, torch.randint(3, 32000, (1, user), generator=g).to(dev)], 1)
            kw = dict(input_ids=ids)
        past = state if mode != "re-prefill" else None
        # max_new_tokens = 1: the turn's prefill (+ one argmax); then the reply itself
        _, t_prefill = timed(lambda: model.generate(max_new_tokens=1, eos_token_id=-1, past_key_values=past, **kw))
        out, t_all = timed(lambda: model.generate(max_new_tokens=reply, eos_token_id=-1, past_key_values=past,
                                                  return_dict_in_generate=True, **kw))
        rows.append((t, kw["input_ids"].shape[1], model.last_prefill_reused, t_prefill, t_all))
        state, ids = out.past_key_values, out.sequences
    if mode != "warm-up":
        print(f"--- {mode}")
        for t, n, reused, tp, ta in rows:
            print(f"turn {t}: prompt {n:5d} tokens, {reused:5d} from the cache: prefill {tp:7.1f} ms, turn total {ta:7.1f} ms "
                  f"({(ta - tp) / max(1, reply - 1):.2f} ms/token)", flush=True)
