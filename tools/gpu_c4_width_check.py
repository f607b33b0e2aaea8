"""GPU check (queued at the end of round 1, not yet run): the C4 shapes at full WIDTH and reduced depth - Llama-3.3-70B
dimensions (8192 / 28672 / 64:8 heads x 128 / 128256 vocab) with 2 layers + whisper-medium width with 1 layer - through
generate(): the KV-cache decode path (skinny GEMMs at K = 8192 / 28672, grouped decode attention with 8 query heads per KV
head) must agree with the model's own teacher-forced forward on prompt + generated tokens wherever the arg-max margin
exceeds bf16 noise, and the first generated token with the f32 CPU oracle.  Memory: 2 layers of 70B width = 3.4 GB bf16 +
2.1 GB embeddings / head; the full 80-layer model needs a layer-streaming loader first.
usage: PYTHONPATH=. python tools/gpu_c4_width_check.py"""
import sys
import torch
from oracle.reference_cpu import OracleModel
from ultravox_amd.config import AUDIO_PRESETS, TEXT_PRESETS, UltravoxConfig
from ultravox_amd.model import UltravoxModel
from ultravox_amd.weights import random_state_dict

DEV = "cuda"
tc = dict(TEXT_PRESETS["meta-llama/Llama-3.3-70B-Instruct"], num_hidden_layers=2)
ac = dict(AUDIO_PRESETS["openai/whisper-medium"], encoder_layers=1)
cfg = UltravoxConfig(text_config=tc, audio_config=ac, hidden_size=4096, stack_factor=8, projector_ln_mid=True, torch_dtype="bfloat16")
sd = random_state_dict(cfg, seed=9, dtype=torch.bfloat16, device="cuda")
sd["language_model.model.embed_tokens.weight"] *= 0.3
model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=torch.bfloat16, rope_len=512, with_backward=False)
oracle = OracleModel(cfg, {k: v.cpu() for k, v in sd.items()}, dtype=torch.float32)
torch.manual_seed(5)
ok = True
for B in (1, 8):
    T, N = 40, 8
    ids = torch.randint(3, cfg.vocab_size - 1, (B, T))
    am = torch.ones(B, T, dtype=torch.long)
    if B > 1:
        am[1, :7] = 0
        ids[am == 0] = 2
    out = model.generate(ids.to(DEV), attention_mask=am.to(DEV), max_new_tokens=N, eos_token_id=-1)
    am_full = torch.cat([am, torch.ones(B, N, dtype=torch.long)], 1).to(DEV)
    logits = model.forward(input_ids=out, attention_mask=am_full).logits.float()
    top2 = logits.topk(2, -1).values
    margin, pred = top2[..., 0] - top2[..., 1], logits.argmax(-1)
    agree = all(bool(((pred[:, t] == out[:, t + 1]) | (margin[:, t] < 5e-2)).all()) for t in range(T - 1, T + N - 1))
    torch.set_num_threads(32)
    with torch.no_grad():
        ref = oracle.forward(input_ids=ids, attention_mask=am)["logits"][:, -1]
    rm = ref.topk(2, -1).values
    clear = (rm[:, 0] - rm[:, 1]) > 5e-2
    first = torch.equal(out[:, T].cpu()[clear], ref.argmax(-1)[clear])
    good = agree and first
    ok &= good
    print(f"B={B}: decode vs teacher-forced forward {'agree' if agree else 'DISAGREE'}; first token vs oracle on {int(clear.sum())}/{B} "
          f"clear rows {'equal' if first else 'DIFFERENT'} -> {'OK' if good else 'FAIL'}", flush=True)
sys.exit(0 if ok else 1)
