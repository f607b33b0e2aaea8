import ctypes as C, sys, torch
sys.path.insert(0, ".")
from ultravox_amd import _lib
from ultravox_amd.config import UltravoxConfig
from ultravox_amd.model import UltravoxModel
DEV = "cuda"
for family in ("gemma", "llama"):
    text = dict(hidden_size=1024, intermediate_size=2816, num_hidden_layers=3, num_attention_heads=8, num_key_value_heads=2, head_dim=128,
                vocab_size=2048, eos_token_id=2, max_position_embeddings=1024)
    if family != "llama":
        text["model_type"] = family
    cfg = UltravoxConfig(audio_config=dict(d_model=128, encoder_layers=1, encoder_attention_heads=2, encoder_ffn_dim=256), text_config=text, hidden_size=256, projector_ln_mid=True)
    model = UltravoxModel(cfg, device=DEV, dtype=torch.bfloat16, seed=13, rope_len=256)
    L = _lib.lib()
    torch.manual_seed(4)
    B, T, new = 20, 12, 3
    ids = torch.randint(3, 2048, (B, T)); am = torch.ones(B, T, dtype=torch.long)
    for r in range(1, B, 4): am[r, :r % 5 + 1] = 0
    ids[am == 0] = 2
    runs = []
    for opt in (1, 1, 0, 0):
        L.uvx_set_option(17, opt)
        out = model.generate(ids.to(DEV), attention_mask=am.to(DEV), max_new_tokens=new, eos_token_id=-1, return_dict_in_generate=True, output_logits=True)
        more_ids = torch.randint(3, 2048, (B, 5), generator=torch.Generator().manual_seed(6)).to(DEV)
        more = model.forward(input_ids=more_ids, past_key_values=out.past_key_values)
        runs.append(more.logits.float())
    L.uvx_set_option(17, 1)
    def d(a, b): return [(a[i] - b[i]).abs().max().item() for i in range(a.shape[0])]
    bad = (runs[0] != runs[2]).nonzero()
    print("   first differing (b, t, v):", bad[:5].tolist(), "rows with a difference:", sorted(set((int(i), int(j)) for i, j, _ in bad.tolist()))[:20])
    print(family, "1 vs 1", d(runs[0], runs[1]), "0 vs 0", d(runs[2], runs[3]), "1 vs 0", d(runs[0], runs[2]), "nonfinite", [(~torch.isfinite(r)).sum().item() for r in runs])
    # per-step count of differing elements
    print("   differing elements per step (1 vs 0):", [(runs[0][i] != runs[2][i]).sum().item() for i in range(runs[0].shape[0])], "of", runs[0][0].numel())
