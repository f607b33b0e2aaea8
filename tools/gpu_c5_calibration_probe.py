"""GPU probe (round 4): is the Gemma LLM path as close to the f32 oracle as torch-ROCm's own bf16 arithmetic is?  The C5 full-depth
record of call 7 had the HIP path at 2x torch's distance (logits 0.028 vs 0.015) - but that second opinion was given the f32 tower
output.  Here: TEXT-ONLY forward (no tower, no projector) at full width and a few depths, Gemma-7B and (control) Llama-3-8B:
rel-L2 of the logits to the f32 CPU oracle for (a) the HIP bf16 path, (b) the oracle restatement run by torch in bf16 on the GPU."""
import sys
import torch
sys.path.insert(0, "tests")
from oracle.reference_cpu import OracleModel, fused_attention
from parity_util import oracle_threads, rel_l2, width_config
from ultravox_amd.model import UltravoxModel
from ultravox_amd.weights import random_state_dict

DEV = "cuda"
oracle_threads()
for text_id in ("google/gemma-7b", "meta-llama/Meta-Llama-3-8B-Instruct"):
    for depth in (1, 4):
        cfg = width_config(text_id, "openai/whisper-medium", depth, 1)
        sd = random_state_dict(cfg, seed=7, dtype=torch.bfloat16, device="cuda")
        model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=torch.bfloat16, rope_len=512)
        o32 = OracleModel(cfg, sd, dtype=torch.float32)
        o16 = OracleModel(cfg, sd, dtype=torch.bfloat16, device=DEV)
        del sd
        torch.manual_seed(1)
        B, T = 2, 160
        ids = torch.randint(3, 30000, (B, T))
        labels = ids.clone(); labels[:, : T - 32] = -100
        am = torch.ones(B, T, dtype=torch.long)
        with torch.no_grad():
            r32 = o32.forward(input_ids=ids, labels=labels, attention_mask=am)
            with torch.device(DEV), fused_attention():
                r16 = o16.forward(input_ids=ids.to(DEV), labels=labels.to(DEV), attention_mask=am.to(DEV))
            with torch.device(DEV):
                r16e = o16.forward(input_ids=ids.to(DEV), labels=labels.to(DEV), attention_mask=am.to(DEV))
        out = model.forward(input_ids=ids.to(DEV), labels=labels.to(DEV), attention_mask=am.to(DEV))
        f = r32["logits"]
        print(f"{text_id} depth {depth}: logits rel-L2 to f32: hip {rel_l2(out.logits, f):.5f}  torch bf16 (flash-rounded attn) {rel_l2(r16['logits'].cpu(), f):.5f}  "
              f"torch bf16 (eager attn) {rel_l2(r16e['logits'].cpu(), f):.5f}  hip vs torch-bf16 {rel_l2(out.logits, r16['logits'].cpu()):.5f}   "
              f"loss hip {out.loss.item():.5f} t16 {r16['loss'].item():.5f} f32 {r32['loss'].item():.5f}", flush=True)
        # embeddings alone (Gemma: x sqrt(hidden) inside the model)
        del model, o32, o16
        torch.cuda.empty_cache()
