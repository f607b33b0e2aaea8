"""GPU check (queued at the end of round 1, not yet run; promote into tests/test_generate_gpu.py once green): generate() with
`KVState.partial_ok` reuses the longest common prefix when the new prompt departs from the cached ids part-way — same tokens
(f32) / same cache rows (bf16, rel-L2) as a fresh full prefill.  usage: PYTHONPATH=.:tests python tools/gpu_lcp_check.py"""
import sys
import torch
sys.path.insert(0, "tests")
from test_generate_gpu import _build, rel_l2  # noqa: E402

DEV = "cuda"
ok = True
for dtype in (torch.float32, torch.bfloat16):
    cfg, model, _ = _build(dtype, 31)
    torch.manual_seed(11)
    B, T1 = 2, 90
    ids = torch.randint(3, 512, (B, T1))
    am = torch.ones(B, T1, dtype=torch.long)
    am[1, :5] = 0
    ids[am == 0] = 2
    out = model.generate(ids.to(DEV), attention_mask=am.to(DEV), max_new_tokens=20, eos_token_id=-1, return_dict_in_generate=True)
    st = out.past_key_values
    st.partial_ok = True
    nxt = torch.cat([out.sequences.cpu()[:, :100], torch.randint(3, 512, (B, 40))], 1)       # departs at index 70 / 100
    nxt[0, 70] = (nxt[0, 70] + 1) % 500 + 3
    am2 = torch.cat([am, torch.ones(B, nxt.shape[1] - T1, dtype=torch.long)], 1)
    gen = dict(attention_mask=am2.to(DEV), max_new_tokens=6, eos_token_id=-1, return_dict_in_generate=True)
    fresh = model.generate(nxt.to(DEV), **gen)
    got = model.generate(nxt.to(DEV), past_key_values=st, **gen)
    reused = model.last_prefill_reused
    kvd = cfg.text_config.num_key_value_heads * cfg.text_config.head_dim
    T2 = nxt.shape[1]

    def rows(s):
        v = s.cache.view(dtype).view(-1, s.Tmax, kvd)[:, :T2].float().clone()
        v.view(-1, B, T2, kvd)[:, 1, :5] = 0
        return v
    err = rel_l2(rows(got.past_key_values), rows(fresh.past_key_values))
    same = torch.equal(got.sequences, fresh.sequences)
    good = reused == 70 and err < (1e-5 if dtype == torch.float32 else 2e-2) and (same or dtype != torch.float32)
    ok &= good
    print(f"{dtype}: reused {reused} rows (want 70), cache rel-L2 {err:.2e}, tokens equal {same} -> {'OK' if good else 'FAIL'}", flush=True)
sys.exit(0 if ok else 1)
