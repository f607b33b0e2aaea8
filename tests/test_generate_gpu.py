"""generate(): prefill + KV-cache greedy decode through the C ABI against the oracle's
cache-free restatement (itself pinned against HF generate in tests/test_oracle_pinning.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _build(dtype, seed):
    from oracle.reference_cpu import OracleModel
    from test_model_gpu import SMALL
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import random_state_dict
    cfg = UltravoxConfig(**SMALL)
    sd = random_state_dict(cfg, seed=seed, dtype=torch.float32)
    sd["language_model.model.embed_tokens.weight"] *= 0.3
    if dtype == torch.bfloat16:
        sd = {k: v.bfloat16() for k, v in sd.items()}
    return cfg, UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=dtype, rope_len=512), OracleModel(cfg, sd, dtype=torch.float32)


def test_f32_generate_matches_oracle_tokens_with_audio_and_left_padding():
    from oracle.reference_cpu import logmel_ref, synthetic_batch
    cfg, model, oracle = _build(torch.float32, 21)
    b = synthetic_batch(cfg, 3, 2.0, n_text=20, audio_start=4, n_supervised=4)
    b.pop("labels")
    b["audio_values"] = logmel_ref(b.pop("pcm"), 80)
    # left padding as the inference collator produces it (ultravox_processing.py:53-63)
    T = b["input_ids"].shape[1]
    # simpler and explicit: sequence i is right-aligned after `lp[i]` pad tokens
    lp = [0, 5, 2]
    width = T + max(lp)
    ids = torch.full((3, width), 2, dtype=torch.long)
    am = torch.zeros(3, width, dtype=torch.long)
    for i, p in enumerate(lp):
        ids[i, width - T:] = b["input_ids"][i]
        am[i, width - T:] = 1
    b["input_ids"], b["attention_mask"] = ids, am
    b["audio_token_start_idx"] = b["audio_token_start_idx"] + (width - T)
    want = oracle.generate_greedy(8, eos_token_id=2, **b)
    got = model.generate(max_new_tokens=8, eos_token_id=2, **{k: v.to(DEV) for k, v in b.items()}).cpu()
    n = min(got.shape[1], want.shape[1])
    assert n > ids.shape[1] and torch.equal(got[:, :n], want[:, :n]), (got[:, ids.shape[1]:], want[:, ids.shape[1]:])


def test_f32_generate_text_only_left_padded_matches_oracle():
    cfg, model, oracle = _build(torch.float32, 22)
    torch.manual_seed(3)
    B, T = 4, 17
    ids = torch.randint(3, 512, (B, T))
    am = torch.ones(B, T, dtype=torch.long)
    am[1, :6] = 0
    am[3, :2] = 0
    ids[am == 0] = 2
    want = oracle.generate_greedy(10, eos_token_id=2, input_ids=ids, attention_mask=am)
    got = model.generate(ids.to(DEV), attention_mask=am.to(DEV), max_new_tokens=10, eos_token_id=2).cpu()
    n = min(got.shape[1], want.shape[1])
    assert torch.equal(got[:, :n], want[:, :n])


def test_bf16_decode_is_consistent_with_teacher_forced_forward():
    """bf16: the KV-cache decode path must agree with our own full forward on prompt + generated tokens wherever
    the arg-max margin exceeds bf16 noise."""
    cfg, model, oracle = _build(torch.bfloat16, 23)
    torch.manual_seed(5)
    B, T, N = 2, 24, 6
    ids = torch.randint(3, 512, (B, T))
    out = model.generate(ids.to(DEV), max_new_tokens=N, eos_token_id=-1)
    assert out.shape == (B, T + N)
    logits = model.forward(input_ids=out).logits.float()          # teacher forcing, no cache
    top2 = logits.topk(2, -1).values
    margin = top2[..., 0] - top2[..., 1]
    pred = logits.argmax(-1)
    for t in range(T - 1, T + N - 1):
        ok = (pred[:, t] == out[:, t + 1]) | (margin[:, t] < 5e-2)
        assert bool(ok.all()), (t, pred[:, t], out[:, t + 1], margin[:, t])
    # and the first generated token agrees with the oracle where its margin is clear
    with torch.no_grad():
        ref = oracle.forward(input_ids=ids)["logits"][:, -1]
    rm = ref.topk(2, -1).values
    clear = (rm[:, 0] - rm[:, 1]) > 5e-2
    assert torch.equal(out[:, T].cpu()[clear], ref.argmax(-1)[clear])


def test_sampling_policies():
    """do_sample: top_k = 1 degenerates to greedy; a seeded generator reproduces the draw; with top_k = 5 every sampled token
    is among the 5 most likely ones of the teacher-forced distribution at its position."""
    from test_model_gpu import SMALL
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.model import UltravoxModel
    cfg = UltravoxConfig(**SMALL)
    model = UltravoxModel(cfg, device=DEV, dtype=torch.bfloat16, seed=3)
    ids = torch.randint(3, 500, (2, 12), generator=torch.Generator().manual_seed(16)).to(DEV)      # (seeded - bf16 logits tie often: top_k = 1 keeps every tied token and a tied 5th / 6th rank breaks the top-5 check; about a third of the seeds do)
    greedy = model.generate(input_ids=ids, max_new_tokens=8, eos_token_id=-1)
    k1 = model.generate(input_ids=ids, max_new_tokens=8, eos_token_id=-1, do_sample=True, top_k=1, temperature=0.7)
    assert torch.equal(greedy, k1)
    g = torch.Generator(device=DEV)
    a = model.generate(input_ids=ids, max_new_tokens=8, eos_token_id=-1, do_sample=True, top_k=5, top_p=0.95, generator=g.manual_seed(1))
    b = model.generate(input_ids=ids, max_new_tokens=8, eos_token_id=-1, do_sample=True, top_k=5, top_p=0.95, generator=g.manual_seed(1))
    assert torch.equal(a, b)
    logits = model.forward(input_ids=a[:, :-1]).logits.float()
    for t in range(12, a.shape[1]):
        top5 = torch.topk(logits[:, t - 1], 5, dim=-1).indices
        assert bool((top5 == a[:, t:t + 1]).any(dim=-1).all()), t
    with pytest.raises(ValueError):
        model.generate(input_ids=ids, max_new_tokens=2, do_sample=True, temperature=0.0)


def test_terminator_list_and_streamer_protocol():
    """eos_token_id as a list of terminators (infer.py:326-328) and the HF streamer protocol: put(prompt), put(token) per
    step, end() — and the LocalInference wrapper streaming from the real generate()."""
    cfg, model, oracle = _build(torch.float32, 23)
    torch.manual_seed(5)
    ids = torch.randint(3, 512, (1, 12))
    free = model.generate(ids.to(DEV), max_new_tokens=8, eos_token_id=-1).cpu()[0, 12:].tolist()
    stop = free[3]                                    # make the 4th generated token a terminator
    first = free.index(stop)

    class Rec:
        def __init__(self):
            self.puts, self.ended = [], False

        def put(self, v):
            self.puts.append(v.clone())

        def end(self):
            self.ended = True
    rec = Rec()
    got = model.generate(ids.to(DEV), max_new_tokens=8, eos_token_id=[600, stop], streamer=rec).cpu()
    assert got[0, 12:].tolist() == free[:first + 1]
    assert rec.ended and torch.equal(rec.puts[0], ids) and [int(p) for p in rec.puts[1:]] == free[:first + 1]
    with pytest.raises(TypeError):          # only the KVState generate() itself returned is accepted as a cache
        model.generate(ids.to(DEV), past_key_values=object())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_chunked_prefill_over_a_cached_prefix(dtype):
    """generate(past_key_values=state): only input_ids[:, cur_len:] run (uvx_llm_prefill_chunk) — same tokens (f32) and
    the same cache contents as a fresh full prefill; left padding, a prefix long enough that whole query blocks are
    skipped, a grown cache, a 1-token chunk, and the fall-back when the cached ids are no longer a prefix."""
    cfg, model, oracle = _build(dtype, 24)
    torch.manual_seed(7)
    B, T1 = 2, 90
    ids1 = torch.randint(3, 512, (B, T1))
    am1 = torch.ones(B, T1, dtype=torch.long)
    am1[1, :5] = 0
    ids1[am1 == 0] = 2
    gen = dict(eos_token_id=-1, return_dict_in_generate=True)
    out1 = model.generate(ids1.to(DEV), attention_mask=am1.to(DEV), max_new_tokens=6, **gen)
    st1 = out1.past_key_values
    assert st1.cur_len == T1 + 5 and st1.tokens.shape == (B, T1 + 5) and model.last_prefill_reused == 0
    ids2 = torch.cat([out1.sequences.cpu(), torch.randint(3, 512, (B, 70))], 1)
    am2 = torch.cat([am1, torch.ones(B, ids2.shape[1] - T1, dtype=torch.long)], 1)
    fresh = model.generate(ids2.to(DEV), attention_mask=am2.to(DEV), max_new_tokens=7, **gen)
    assert model.last_prefill_reused == 0
    out2 = model.generate(ids2.to(DEV), attention_mask=am2.to(DEV), max_new_tokens=7, past_key_values=st1, **gen)
    assert model.last_prefill_reused == T1 + 5
    st2, stf = out2.past_key_values, fresh.past_key_values
    assert st2.cur_len == stf.cur_len == ids2.shape[1] + 6 and st2.Tmax == stf.Tmax
    assert torch.equal(st2.pos_next, stf.pos_next) and torch.equal(st2.kv_start, stf.kv_start)

    def planes(st, upto):   # [L * 2 * B, Tmax, kv width] -> rows [0, upto); row 1's left padding holds don't-care values
        v = st.cache.view(dtype).view(-1, st.Tmax, cfg.text_config.num_key_value_heads * cfg.text_config.head_dim)
        v = v[:, :upto].float().clone()
        v.view(-1, B, upto, v.shape[-1])[:, 1, :5] = 0
        return v
    T2 = ids2.shape[1]
    if dtype == torch.float32:
        assert torch.equal(out2.sequences, fresh.sequences)
        assert rel_l2(planes(st2, st2.cur_len), planes(stf, st2.cur_len)) < 1e-5
    else:                    # bf16: the decoded tokens may part ways on a near-tie, the prompt's rows may not
        assert rel_l2(planes(st2, T2), planes(stf, T2)) < 2e-2
    # one single further token on top of the (grown) cache: the last generated id, whose keys / values are not cached yet
    ids3 = out2.sequences.cpu()
    T3 = ids3.shape[1]
    am3 = torch.cat([am2, torch.ones(B, T3 - T2, dtype=torch.long)], 1)
    fresh3 = model.generate(ids3.to(DEV), attention_mask=am3.to(DEV), max_new_tokens=4, **gen)
    out3 = model.generate(ids3.to(DEV), attention_mask=am3.to(DEV), max_new_tokens=4, past_key_values=st2, **gen)
    assert model.last_prefill_reused == st2.cur_len == T3 - 1
    if dtype == torch.float32:
        assert torch.equal(out3.sequences, fresh3.sequences)
    assert rel_l2(planes(out3.past_key_values, T3), planes(fresh3.past_key_values, T3)) < (1e-5 if dtype == torch.float32 else 2e-2)
    # a prompt that no longer starts with the cached ids: the cache is dropped, not trusted
    bad = ids2.clone()
    bad[0, 40] = (bad[0, 40] + 1) % 500 + 3
    want = model.generate(bad.to(DEV), attention_mask=am2.to(DEV), max_new_tokens=3, eos_token_id=-1)
    got = model.generate(bad.to(DEV), attention_mask=am2.to(DEV), max_new_tokens=3, eos_token_id=-1, past_key_values=st1)
    assert model.last_prefill_reused == 0 and torch.equal(got, want)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_longest_common_prefix_reuse_on_the_device(dtype):
    """KVState.partial_ok (set by LocalInference in conversation mode): a prompt that departs from the cached ids part-way
    reuses the rows of the longest common prefix - same tokens (f32) / same cache rows as a fresh full prefill."""
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
    assert model.last_prefill_reused == 70
    kvd = cfg.text_config.num_key_value_heads * cfg.text_config.head_dim
    T2 = nxt.shape[1]

    def rows(s):
        v = s.cache.view(dtype).view(-1, s.Tmax, kvd)[:, :T2].float().clone()
        v.view(-1, B, T2, kvd)[:, 1, :5] = 0
        return v
    assert rel_l2(rows(got.past_key_values), rows(fresh.past_key_values)) < (1e-5 if dtype == torch.float32 else 2e-2)
    if dtype == torch.float32:
        assert torch.equal(got.sequences, fresh.sequences)


def test_f32_generate_matches_the_reference_generate_fixture():
    """generate() against the REFERENCE UltravoxModel.generate itself (tests/golden/generate_reference.json: the imported
    reference + HF greedy search on the seeded tiny model, audio tower stubbed by recorded hidden states): new tokens of an
    unpadded prompt with two audio items and a left-padded prompt, without EOS and with an EOS that stops one row early."""
    import json
    import os
    import forward_fixture_util as U
    from test_oracle_pinning import load_forward_fixture
    from ultravox_amd.model import UltravoxModel
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "generate_reference.json")))
    cfg, sd, _, enc, _ = load_forward_fixture("ln_mid")
    model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=torch.float32, rope_len=512)
    tower = enc[U.GEN_AUDIO_ROWS].to(DEV)
    model.audio_tower_forward = lambda audio_values, audio_len: tower[: audio_values.shape[0]]
    b = {k: v.to(DEV) for k, v in U.generate_batch().items()}
    mel = torch.zeros(3, 80, 3000, device=DEV)
    T = fx["prompt_len"]
    free = model.generate(audio_values=mel, max_new_tokens=10, eos_token_id=None, pad_token_id=fx["pad_token_id"], **b).cpu()
    assert free[:, T:].tolist() == fx["free"]
    stop = model.generate(audio_values=mel, max_new_tokens=10, eos_token_id=fx["eos"], pad_token_id=fx["pad_token_id"], **b).cpu()
    got = stop[:, T:].tolist()
    assert [r + [fx["pad_token_id"]] * (10 - len(r)) for r in got] == fx["with_eos"]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_forward_with_past_key_values_drives_a_decode_loop(dtype):
    """forward(input_ids=new tokens, past_key_values=KVState) - the call the reference forwards to the language model
    (ultravox_model.py:328-334) and HF's generation loop issues every step: a hand-rolled greedy loop over it reproduces
    generate() token for token (f32), appends to / grows the cache, and returns [B, Tn, V] logits + the extended state
    ([B, 1, V] with logits_to_keep=1, HF's own prefill setting)."""
    cfg, model, _ = _build(dtype, 37)
    torch.manual_seed(3)
    B, T = 2, 40
    ids = torch.randint(3, 512, (B, T))
    am = torch.ones(B, T, dtype=torch.long)
    am[1, :4] = 0
    ids[am == 0] = 2
    n_new = 9
    want = model.generate(ids.to(DEV), attention_mask=am.to(DEV), max_new_tokens=n_new, eos_token_id=-1).cpu()
    # prompt -> cache (one generated token), then step by step through forward()
    first = model.generate(ids.to(DEV), attention_mask=am.to(DEV), max_new_tokens=1, eos_token_id=-1, return_dict_in_generate=True)
    state, seq = first.past_key_values, first.sequences
    assert state.cur_len == T and state.Tmax == T + 1
    for step in range(n_new - 1):
        out = model.forward(input_ids=seq[:, -1:], past_key_values=state)
        assert tuple(out.logits.shape) == (B, 1, 512) and out.past_key_values.cur_len == T + 1 + step
        state = out.past_key_values
        seq = torch.cat([seq, out.logits[:, 0].float().argmax(-1, keepdim=True)], 1)
    assert state.Tmax >= T + n_new - 1                       # grown past the buffer generate() sized for one token
    if dtype == torch.float32:
        assert torch.equal(seq.cpu(), want)
    else:
        assert (seq.cpu() == want).float().mean().item() > 0.9
    # several new tokens at once: logits of EVERY new position [B, Tn, V] as from the HF language model (ultravox_model.py:328-334)
    # - against the teacher-forced forward over the whole sequence - or the last one only on request (logits_to_keep=1)
    full = model.generate(ids.to(DEV), attention_mask=am.to(DEV), max_new_tokens=1, eos_token_id=-1, return_dict_in_generate=True)
    outN = model.forward(input_ids=want[:, T:T + 3].to(DEV), past_key_values=full.past_key_values)
    assert tuple(outN.logits.shape) == (B, 3, 512) and outN.past_key_values.cur_len == T + 3
    am_full = torch.cat([am, torch.ones(B, 3, dtype=torch.long)], 1)
    tf = model.forward(input_ids=want[:, :T + 3].to(DEV), attention_mask=am_full.to(DEV)).logits[:, T:T + 3].float()
    if dtype == torch.float32:
        assert (outN.logits.float() - tf).abs().max().item() < 1e-4
        assert torch.equal(outN.logits.float().argmax(-1).cpu(), want[:, T + 1:T + 4])
    else:
        assert rel_l2(outN.logits.float(), tf) < 2e-2
    two = model.generate(ids.to(DEV), attention_mask=am.to(DEV), max_new_tokens=1, eos_token_id=-1, return_dict_in_generate=True)
    out2 = model.forward(input_ids=want[:, T:T + 3].to(DEV), past_key_values=two.past_key_values, logits_to_keep=2)
    assert tuple(out2.logits.shape) == (B, 2, 512) and torch.equal(out2.logits, outN.logits[:, -2:]) and out2.past_key_values.cur_len == T + 3
    again = model.generate(ids.to(DEV), attention_mask=am.to(DEV), max_new_tokens=1, eos_token_id=-1, return_dict_in_generate=True)
    out3 = model.forward(input_ids=want[:, T:T + 3].to(DEV), past_key_values=again.past_key_values, logits_to_keep=1)
    assert out3.past_key_values.cur_len == T + 3 and tuple(out3.logits.shape) == (B, 1, 512)
    assert torch.equal(out3.logits[:, 0], outN.logits[:, -1]) or rel_l2(out3.logits[:, 0].float(), outN.logits[:, -1].float()) < 1e-2
    if dtype == torch.float32:
        assert torch.equal(out3.logits[:, 0].float().argmax(-1).cpu(), want[:, T + 3])
    with pytest.raises(ValueError, match="inference call"):
        model.forward(input_ids=seq[:, -1:], past_key_values=state, labels=seq[:, -1:])


@pytest.mark.parametrize("heads,kv_heads,head_dim,T", [(8, 1, 128, 1400), (8, 2, 64, 700), (4, 4, 128, 300)])
def test_bf16_decode_attention_over_long_caches_and_group_sizes(heads, kv_heads, head_dim, T):
    """The grouped decode-attention kernel (one block of 1024 threads per sequence and KV head; scores in dynamic LDS next to ~36 KB of
    static LDS) at the edges of its launch conditions: 8 query heads per KV head over 1400 cached keys (45 KB of scores), head_dim 64,
    no grouping; left padding in one row.  Decode logits against the teacher-forced forward of the same model."""
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.model import UltravoxModel
    cfg = UltravoxConfig(
        audio_config=dict(d_model=128, encoder_layers=1, encoder_attention_heads=2, encoder_ffn_dim=256),
        text_config=dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=heads,
                         num_key_value_heads=kv_heads, head_dim=head_dim, vocab_size=512, eos_token_id=2, max_position_embeddings=4096),
        hidden_size=256, projector_ln_mid=True)
    model = UltravoxModel(cfg, device=DEV, dtype=torch.bfloat16, seed=5, rope_len=2048)
    torch.manual_seed(9)
    B = 2
    ids = torch.randint(3, 512, (B, T))
    am = torch.ones(B, T, dtype=torch.long)
    am[1, :37] = 0
    ids[am == 0] = 2
    new = 5
    seq = model.generate(ids.to(DEV), attention_mask=am.to(DEV), max_new_tokens=new, eos_token_id=-1)
    am_full = torch.cat([am, torch.ones(B, new, dtype=torch.long)], 1).to(DEV)
    tf = model.forward(input_ids=seq, attention_mask=am_full).logits.float()
    # step by step through the cache: the logits that produced each new token
    first = model.generate(ids.to(DEV), attention_mask=am.to(DEV), max_new_tokens=1, eos_token_id=-1, return_dict_in_generate=True)
    state = first.past_key_values
    for step in range(new - 1):
        out = model.forward(input_ids=seq[:, T + step:T + step + 1], past_key_values=state)
        state = out.past_key_values
        got, want = out.logits[:, 0].float(), tf[:, T + step]
        assert rel_l2(got, want) < 3e-2, (step, rel_l2(got, want))


@pytest.mark.parametrize("B", [17, 40, 64])
def test_bf16_decode_batches_beyond_16_rows_take_the_tiled_split_path(B):
    """Round 5: a decode batch of more than 16 sequences runs its linears on the tiled kernels with split-K (the weight-streaming
    kernels serve 16-row tiles one after the other; gemm.hip gemm_nt) - Llama-shaped layers wide enough for the split to engage
    (K = 1024 / 2816: 16 / 44 K-tiles), left padding in some rows.  Every decode step's logits (generate(output_logits=True)) against the
    teacher-forced forward of the same model, and the greedy tokens they imply; a B = 16 batch (weight-streaming kernels) of the same
    prompts gives the same tokens wherever the two paths' logits are not within rounding of a tie."""
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.model import UltravoxModel
    cfg = UltravoxConfig(
        audio_config=dict(d_model=128, encoder_layers=1, encoder_attention_heads=2, encoder_ffn_dim=256),
        text_config=dict(hidden_size=1024, intermediate_size=2816, num_hidden_layers=3, num_attention_heads=8,
                         num_key_value_heads=2, head_dim=128, vocab_size=2048, eos_token_id=2, max_position_embeddings=1024),
        hidden_size=256, projector_ln_mid=True)
    model = UltravoxModel(cfg, device=DEV, dtype=torch.bfloat16, seed=7, rope_len=256)
    torch.manual_seed(B)
    T, new = 41, 6
    ids = torch.randint(3, 2048, (B, T))
    am = torch.ones(B, T, dtype=torch.long)
    for r in range(1, B, 3):
        am[r, :r % 11 + 1] = 0
    ids[am == 0] = 2
    out = model.generate(ids.to(DEV), attention_mask=am.to(DEV), max_new_tokens=new, eos_token_id=-1, return_dict_in_generate=True, output_logits=True)
    seq = out.sequences
    assert len(out.logits) == new and tuple(out.logits[0].shape) == (B, 2048)
    am_full = torch.cat([am, torch.ones(B, new, dtype=torch.long)], 1).to(DEV)
    tf = model.forward(input_ids=seq, attention_mask=am_full).logits.float()
    for step in range(new):
        got, want = out.logits[step].float(), tf[:, T - 1 + step]
        assert rel_l2(got, want) < 3e-2, (step, rel_l2(got, want))
        assert torch.equal(got.argmax(-1), seq[:, T + step])
    # The two paths round differently (split-K partial sums vs one f32 chain), so a row may part ways at a near-tie and then - fed its own
    # token - stay apart: compare each row up to its first differing token only, and require that token to BE a near-tie in this path's
    # logits (the two candidates closer than the bf16 logit spacing the 3e-2 bar above allows).
    small = model.generate(ids[:16].to(DEV), attention_mask=am[:16].to(DEV), max_new_tokens=new, eos_token_id=-1)
    same = small[:, T:] == seq[:16, T:]
    assert same[:, 0].float().mean().item() >= 0.75
    for r in range(16):
        miss = (~same[r]).nonzero()
        if miss.numel() == 0:
            continue
        step = int(miss[0])
        row = out.logits[step][r].float()
        gap = (row[seq[r, T + step]] - row[small[r, T + step]]).abs().item()
        assert gap <= 3e-2 * row.norm().item() / row.numel() ** 0.5 * 4, (r, step, gap)


@pytest.mark.parametrize("family", ["llama", "qwen2"])
def test_prefill_rope_and_cache_append_in_one_launch_is_bit_identical(family):
    """Round 5: the prefill (and the chunked prefill behind forward(past_key_values)) rotate q / k and write the new cache rows in
    one launch per layer (rope_kv_append_k with Tn positions per sequence; option 16) instead of rope_k + kv_append_k: same
    arithmetic, so tokens, every step's logits and the cache itself are bit-identical.  Left padding, GQA; qwen2 adds q / k / v biases."""
    from ultravox_amd import _lib
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.model import UltravoxModel
    text = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=3, num_attention_heads=4, num_key_value_heads=2, head_dim=64,
                vocab_size=512, eos_token_id=2, max_position_embeddings=512)
    if family == "qwen2":
        text["model_type"] = "qwen2"
    cfg = UltravoxConfig(audio_config=dict(d_model=128, encoder_layers=1, encoder_attention_heads=2, encoder_ffn_dim=256),
                         text_config=text, hidden_size=256, projector_ln_mid=True)
    model = UltravoxModel(cfg, device=DEV, dtype=torch.bfloat16, seed=11, rope_len=256)
    torch.manual_seed(3)
    B, T, new = 3, 37, 4
    ids = torch.randint(3, 512, (B, T))
    am = torch.ones(B, T, dtype=torch.long)
    am[1, :9] = 0
    am[2, :2] = 0
    ids[am == 0] = 2
    L = _lib.lib()
    runs = []
    try:
        for opt in (1, 0):
            L.uvx_set_option(16, opt)
            out = model.generate(ids.to(DEV), attention_mask=am.to(DEV), max_new_tokens=new, eos_token_id=-1, return_dict_in_generate=True,
                                 output_logits=True)
            def rows(state):      # the cache rows written so far ([L][2][B][Tmax][kv_heads * head_dim]; rows >= cur_len were never written)
                kvd = text["num_key_value_heads"] * text["head_dim"]
                return state.cache.view(torch.bfloat16).view(text["num_hidden_layers"], 2, B, state.Tmax, kvd)[:, :, :, :state.cur_len].clone()
            cache = rows(out.past_key_values)
            more = model.forward(input_ids=torch.randint(3, 512, (B, 5), generator=torch.Generator().manual_seed(5)).to(DEV),
                                 past_key_values=out.past_key_values)      # chunked prefill: five new positions per sequence
            runs.append((out.sequences, torch.stack(out.logits), cache, more.logits, rows(more.past_key_values)))
    finally:
        L.uvx_set_option(16, 1)
    assert L.uvx_get_option(16) == 1
    for a, b in zip(*runs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("family", ["llama", "gemma", "qwen3"])
def test_rmsnorm_inside_the_splitk_reduce_is_bit_identical(family):
    """Round 5: where a linear of generate() is split over K, the RMSNorm that follows it (o_proj -> post_attention_layernorm,
    down_proj -> the next layer's input_layernorm) is computed by the reduce kernel (splitk_reduce_norm_k, option 17) with
    rmsnorm_fwd_k's thread mapping and summation order: tokens, logits and cache rows are bit-identical to the separate launches.
    Layers wide enough for the split to engage at these row counts (K = 1024 / 2816; uvx_gemm_pick_split says so below): a prefill of
    240 rows, decode steps of 20 rows (beyond 16: the tiled path), a chunked prefill; Gemma = the (1 + w) flavour, Qwen3 = q / k norms."""
    import ctypes as C
    from ultravox_amd import _lib
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.model import UltravoxModel
    # (hidden 1536, not 1024: at 512 / 1024 / 2048 columns rmsnorm_fwd runs its one-wave-per-row kernel - another summation order - and
    #  the fusion stays off there; the LLMs this path serves are 3584 ... 8192 wide)
    text = dict(hidden_size=1536, intermediate_size=4096, num_hidden_layers=3, num_attention_heads=12, num_key_value_heads=4, head_dim=128,
                vocab_size=2048, eos_token_id=2, max_position_embeddings=1024)
    if family != "llama":
        text["model_type"] = family
    cfg = UltravoxConfig(audio_config=dict(d_model=128, encoder_layers=1, encoder_attention_heads=2, encoder_ffn_dim=256),
                         text_config=text, hidden_size=256, projector_ln_mid=True)
    model = UltravoxModel(cfg, device=DEV, dtype=torch.bfloat16, seed=13, rope_len=256)
    L = _lib.lib()
    L.uvx_gemm_pick_split.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_size_t, C.POINTER(C.c_int32)]
    L.uvx_gemm_splitk_ws_bytes.restype = C.c_size_t
    for M in (240, 20, 100):          # the o / down projections of the three phases below are split
        for K in (1536, 4096):
            assert L.uvx_gemm_pick_split(M, 1536, K, L.uvx_gemm_splitk_ws_bytes(M, 1536), None) > 1, (M, K)
    torch.manual_seed(4)
    B, T, new = 20, 12, 3
    ids = torch.randint(3, 2048, (B, T))
    am = torch.ones(B, T, dtype=torch.long)
    for r in range(1, B, 4):
        am[r, :r % 5 + 1] = 0
    ids[am == 0] = 2
    more_ids = torch.randint(3, 2048, (B, 5), generator=torch.Generator().manual_seed(6)).to(DEV)
    runs = []
    try:
        for opt in (1, 0):
            L.uvx_set_option(17, opt)
            out = model.generate(ids.to(DEV), attention_mask=am.to(DEV), max_new_tokens=new, eos_token_id=-1, return_dict_in_generate=True,
                                 output_logits=True)
            st = out.past_key_values
            kvd = text["num_key_value_heads"] * text["head_dim"]
            rows = lambda s_: s_.cache.view(torch.bfloat16).view(3, 2, B, s_.Tmax, kvd)[:, :, :, :s_.cur_len].clone()
            cache = rows(st)
            more = model.forward(input_ids=more_ids, past_key_values=st)      # chunked prefill: 100 rows
            runs.append((out.sequences, torch.stack(out.logits), cache, more.logits, rows(more.past_key_values)))
    finally:
        L.uvx_set_option(17, 1)
    for a, b in zip(*runs):
        assert torch.equal(a, b)
    assert torch.isfinite(runs[0][1].float()).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_greedy_bookkeeping_in_one_launch_equals_the_generic_loop(dtype, monkeypatch):
    """Round 6: uvx_greedy_select (argmax + pad-after-finish + EOS test + next RoPE position + unfinished count in ONE launch) against the
    generic loop's torch bookkeeping (UVX_GREEDY_SELECT=0): same sequences when the rows of a batch stop at DIFFERENT steps (the finished
    ones are padded), when none stops, with a list of terminators and a pad id that is not an EOS id, and the same KV state."""
    cfg, model, oracle = _build(dtype, 29)
    torch.manual_seed(9)
    B, T, N = 4, 14, 12
    ids = torch.randint(3, 512, (B, T)).to(DEV)
    am = torch.ones(B, T, dtype=torch.long)
    am[2, :4] = 0
    monkeypatch.setenv("UVX_GREEDY_SELECT", "0")
    free = model.generate(ids, attention_mask=am.to(DEV), max_new_tokens=N, eos_token_id=-1).cpu()
    stops = [int(free[0, T + 2]), int(free[1, T + 6])]           # row 0 stops at step 2 (or earlier), row 1 at step 6 (or earlier)
    for kw in (dict(eos_token_id=-1), dict(eos_token_id=stops, pad_token_id=1), dict(eos_token_id=stops[0])):
        monkeypatch.setenv("UVX_GREEDY_SELECT", "0")
        want = model.generate(ids, attention_mask=am.to(DEV), max_new_tokens=N, return_dict_in_generate=True, **kw)
        monkeypatch.setenv("UVX_GREEDY_SELECT", "1")
        got = model.generate(ids, attention_mask=am.to(DEV), max_new_tokens=N, return_dict_in_generate=True, **kw)
        assert torch.equal(got.sequences, want.sequences), kw
        assert got.past_key_values.cur_len == want.past_key_values.cur_len
        assert torch.equal(got.past_key_values.pos_next, want.past_key_values.pos_next)
        if "pad_token_id" in kw:           # a finished row is padded with pad_token_id from the step after its terminator on
            row = got.sequences[0, T:].tolist()
            first = next(i for i, t in enumerate(row) if t in stops)
            assert all(t == 1 for t in row[first + 1:])


@pytest.mark.parametrize("heads,kv_heads,head_dim", [(4, 2, 64), (8, 2, 128), (8, 1, 128), (4, 4, 128)])
def test_rope_and_cache_append_inside_the_decode_attention_is_bit_identical(heads, kv_heads, head_dim):
    """Round 6 (tuning option 23): the grouped decode-attention kernel rotates the new token's q / k and appends its k / v rows to the cache itself
    instead of after a rope_kv_append_k launch per layer - same step logits at every decoded position and the same KV cache, bit for bit, for
    group sizes 1 / 2 / 4 / 8 (incl. the head-split blocks that write the same cache row), left padding, B = 3."""
    from test_model_gpu import SMALL
    from ultravox_amd import _lib
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.model import UltravoxModel
    from ultravox_amd.weights import random_state_dict
    L = _lib.lib()
    small = dict(SMALL)
    small["text_config"] = dict(small["text_config"], num_attention_heads=heads, num_key_value_heads=kv_heads, head_dim=head_dim,
                                hidden_size=heads * head_dim if heads * head_dim >= 256 else 256)
    cfg = UltravoxConfig(**small)
    sd = {k: v.bfloat16() for k, v in random_state_dict(cfg, seed=37).items()}
    model = UltravoxModel(cfg, state_dict=sd, device=DEV, dtype=torch.bfloat16, rope_len=512, with_backward=False)
    torch.manual_seed(2)
    ids = torch.randint(3, cfg.vocab_size - 1, (3, 21)).to(DEV)
    am = torch.ones(3, 21, dtype=torch.long)
    am[1, :5] = 0
    res = []
    try:
        for opt in (1, 0):
            L.uvx_set_option(23, opt)
            out = model.generate(ids, attention_mask=am.to(DEV), max_new_tokens=9, eos_token_id=-1, return_dict_in_generate=True, output_logits=True)
            torch.cuda.synchronize()
            st = out.past_key_values
            t = cfg.text_config
            planes = t.num_hidden_layers * 2 * 3
            rows = st.cache.view(torch.bfloat16)[:planes * st.Tmax * kv_heads * head_dim].view(planes, st.Tmax, kv_heads * head_dim)[:, :st.cur_len]
            res.append((out.sequences.clone(), [x.clone() for x in out.logits], rows.clone(), st.cur_len))      # (rows beyond cur_len are never written)
    finally:
        L.uvx_set_option(23, 0)
    (s0, l0, c0, n0), (s1, l1, c1, n1) = res
    assert torch.equal(s0, s1) and n0 == n1
    for a, b in zip(l0, l1):
        assert torch.equal(a, b)
    assert torch.equal(c0, c1)


@pytest.mark.parametrize("B,family", [(5, "llama"), (12, "llama"), (8, "gemma")])
def test_decode_rmsnorm_inside_the_staged_skinny_kernel(B, family):
    """Round 6 (opt-in, option 24 = 1: measured slower, profiles/r06_decode_norm_in_skinny_ab.txt): at 3..16 decode rows the input_layernorm /
    post_attention_layernorm are applied to the activation fragments inside the staged weight-streaming kernel (gemm_skinny_bf16_k<.., NORM>,
    K = hidden size a multiple of 2048) instead of 2 rmsnorm launches per layer.  Same arithmetic and rounding points; the sum of squares is folded in another order, so rstd may move by an ulp: every
    step's logits within 5e-3 rel-L2 of the two-launch path (the bf16 path's own distance to f32 is 2e-2), the greedy tokens equal wherever
    the two candidates are not a near-tie, and both paths consistent with the teacher-forced forward."""
    from ultravox_amd import _lib
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.model import UltravoxModel
    text = dict(hidden_size=2048, intermediate_size=4096, num_hidden_layers=3, num_attention_heads=16, num_key_value_heads=4, head_dim=128,
                vocab_size=1024, eos_token_id=2, max_position_embeddings=512)
    if family == "gemma":
        text.update(model_type="gemma", hidden_act="gelu_pytorch_tanh", head_dim=128)
    cfg = UltravoxConfig(audio_config=dict(d_model=128, encoder_layers=1, encoder_attention_heads=2, encoder_ffn_dim=256),
                         text_config=text, hidden_size=256, projector_ln_mid=True)
    model = UltravoxModel(cfg, device=DEV, dtype=torch.bfloat16, seed=21, rope_len=256)
    torch.manual_seed(B)
    T, new = 29, 5
    ids = torch.randint(3, 1024, (B, T))
    am = torch.ones(B, T, dtype=torch.long)
    for r in range(1, B, 2):
        am[r, :r + 2] = 0
    ids[am == 0] = 2
    L = _lib.lib()
    runs = []
    try:
        for opt in (1, 0):
            L.uvx_set_option(24, opt)
            runs.append(model.generate(ids.to(DEV), attention_mask=am.to(DEV), max_new_tokens=new, eos_token_id=-1, return_dict_in_generate=True,
                                       output_logits=True))
    finally:
        L.uvx_set_option(24, 0)
    fused, two = runs
    for step in range(new):
        a, b = fused.logits[step].float(), two.logits[step].float()
        if step == 0 or torch.equal(fused.sequences[:, :T + step], two.sequences[:, :T + step]):
            assert rel_l2(a, b) < 5e-3, (step, rel_l2(a, b))
    same = fused.sequences == two.sequences
    for r in range(B):
        miss = (~same[r]).nonzero()
        if miss.numel() == 0:
            continue
        step = int(miss[0]) - T
        row = two.logits[step][r].float()
        gap = (row[two.sequences[r, T + step]] - row[fused.sequences[r, T + step]]).abs().item()
        assert gap <= 2e-2 * row.abs().max().item(), (r, step, gap)
    am_full = torch.cat([am, torch.ones(B, new, dtype=torch.long)], 1).to(DEV)
    tf = model.forward(input_ids=fused.sequences, attention_mask=am_full).logits.float()
    for step in range(new):
        assert rel_l2(fused.logits[step].float(), tf[:, T - 1 + step]) < 3e-2


@pytest.mark.parametrize("case", [dict(num_beams=3), dict(num_beams=4, length_penalty=0.0, num_return_sequences=2), dict(num_beams=2, early_stopping=True),
                                  dict(num_beams=3, repetition_penalty=1.5), dict(num_beams=3, no_repeat_ngram_size=2, min_new_tokens=4, bad_words_ids=[[40], [41, 42]])])
@pytest.mark.parametrize("n_eos", [1, 120])
def test_f32_beam_search_matches_the_oracle_tokens(case, n_eos):
    """generate(num_beams > 1) (the reference forwards the keyword to HF's generate, ultravox_model.py:422-426): the HIP path (one
    prefill, B * beams decode rows, KV-cache planes re-gathered to follow the surviving beams) in f32 against the oracle's cache-free
    restatement of HF's beam search (pinned token for token to HF in tests/test_oracle_pinning.py), audio + left padding, with one
    terminator and with 120 of 512 ids ending a hypothesis (finished-set and stopping logic)."""
    from oracle.reference_cpu import logmel_ref, synthetic_batch
    cfg, model, oracle = _build(torch.float32, 23)
    b = synthetic_batch(cfg, 3, 2.0, n_text=20, audio_start=4, n_supervised=4)
    b.pop("labels")
    b["audio_values"] = logmel_ref(b.pop("pcm"), 80)
    T = b["input_ids"].shape[1]
    lp = [0, 5, 2]
    width = T + max(lp)
    ids = torch.full((3, width), 2, dtype=torch.long)
    am = torch.zeros(3, width, dtype=torch.long)
    for i, p in enumerate(lp):
        ids[i, width - T:] = b["input_ids"][i]
        am[i, width - T:] = 1
    b["input_ids"], b["attention_mask"] = ids, am
    b["audio_token_start_idx"] = b["audio_token_start_idx"] + (width - T)
    eos = 2 if n_eos == 1 else list(range(7, 7 + n_eos))
    from test_oracle_pinning import hf_processor_list
    plain, procs = hf_processor_list(case, width, eos)      # (HF's own processor classes for the oracle; the keywords themselves for the HIP path)
    want = oracle.generate_beam(7, eos_token_id=eos, pad_token_id=1, logits_processor=procs, **plain, **b)
    got = model.generate(max_new_tokens=7, eos_token_id=eos, pad_token_id=1, **case, **{k: v.to(DEV) for k, v in b.items()}).cpu()
    assert got.shape == want.shape and torch.equal(got, want), (got[:, width:], want[:, width:])


def test_bf16_beam_search_is_consistent_with_its_own_scores():
    """bf16 production path: the returned hypotheses of a 4-beam search re-scored by the teacher-forced forward of the same model
    (sum of next-token log-probabilities / length) reproduce sequences_scores and the best-first order; the hypotheses of a prompt are distinct."""
    cfg, model, _ = _build(torch.bfloat16, 24)
    torch.manual_seed(5)
    B, T, new, nb = 3, 15, 6, 4
    ids = torch.randint(3, 512, (B, T))
    am = torch.ones(B, T, dtype=torch.long)
    am[1, :4] = 0
    ids[am == 0] = 2
    out = model.generate(ids.to(DEV), attention_mask=am.to(DEV), max_new_tokens=new, eos_token_id=-1, num_beams=nb, num_return_sequences=nb,
                         return_dict_in_generate=True)
    seq, sc = out.sequences, out.sequences_scores.float().view(B, nb)
    assert tuple(seq.shape) == (B * nb, T + new)
    am_full = torch.cat([am.repeat_interleave(nb, 0), torch.ones(B * nb, new, dtype=torch.long)], 1).to(DEV)
    lp = torch.log_softmax(model.forward(input_ids=seq, attention_mask=am_full).logits.float(), -1)
    tok_lp = lp[:, T - 1:-1].gather(-1, seq[:, T:, None])[..., 0].sum(-1) / new
    assert (tok_lp.view(B, nb) - sc).abs().max().item() < 3e-2
    assert bool((sc[:, :-1] >= sc[:, 1:]).all())
    for b in range(B):
        assert len({tuple(r.tolist()) for r in seq[b * nb:(b + 1) * nb]}) == nb


def test_bf16_beam_sampling_runs_on_the_device_generator():
    """generate(num_beams > 1, do_sample=True) - HF's beam sampling (token-exact against HF on the CPU generator: tests/test_generate_host_cpu.py,
    tests/test_oracle_pinning.py) - on the production path with a CUDA generator: the same seed reproduces the hypotheses, another seed or plain beam
    search gives others; the scores are the returned hypotheses' own length-normalised log-probabilities under the teacher-forced forward."""
    cfg, model, _ = _build(torch.bfloat16, 24)
    torch.manual_seed(5)
    B, T, new, nb = 3, 15, 6, 3
    ids = torch.randint(3, 512, (B, T))
    am = torch.ones(B, T, dtype=torch.long)
    am[1, :4] = 0
    ids[am == 0] = 2
    run = lambda seed, **kw: model.generate(ids.to(DEV), attention_mask=am.to(DEV), max_new_tokens=new, eos_token_id=-1, num_beams=nb, num_return_sequences=nb,
                                            return_dict_in_generate=True, **({"do_sample": True, "temperature": 1.2, "top_k": 40,
                                                                              "generator": torch.Generator(device=DEV).manual_seed(seed)} if seed else {}), **kw)
    a, b, c, plain = run(7), run(7), run(8), run(0)
    assert tuple(a.sequences.shape) == (B * nb, T + new) and int(a.sequences.min()) >= 0 and int(a.sequences.max()) < 512
    assert torch.equal(a.sequences, b.sequences) and not torch.equal(a.sequences, c.sequences) and not torch.equal(a.sequences, plain.sequences)
    seq, sc = a.sequences, a.sequences_scores.float().view(B, nb)
    am_full = torch.cat([am.repeat_interleave(nb, 0), torch.ones(B * nb, new, dtype=torch.long)], 1).to(DEV)
    lp = torch.log_softmax(model.forward(input_ids=seq, attention_mask=am_full).logits.float(), -1)
    tok_lp = lp[:, T - 1:-1].gather(-1, seq[:, T:, None])[..., 0].sum(-1) / new
    # (HF's beam scores accumulate the WARPED log-probabilities: log p / temperature; top-k only removes candidates)
    assert (tok_lp.view(B, nb) / 1.2 - sc).abs().max().item() < 3e-2 and bool((sc[:, :-1] >= sc[:, 1:]).all())
    with pytest.raises(NotImplementedError, match="min_p"):
        model.generate(ids.to(DEV), attention_mask=am.to(DEV), max_new_tokens=2, num_beams=2, min_p=0.1)


def test_f32_generate_with_hf_generation_keywords_matches_the_oracle_loop_over_hf_processors():
    """generate(**kwargs): the reference forwards every keyword to HF `generate` (ultravox_model.py:422-426).  no_repeat_ngram_size / bad_words_ids /
    min_new_tokens / suppress_tokens / repetition_penalty through the KV-cache decode loop (ultravox_amd/generation.py) against the oracle's cache-free
    loop applying HF's OWN LogitsProcessorList (built here from transformers.generation.logits_process) - token for token, audio and left padding
    included; the keywords are chosen from the plain continuation so that each one changes it."""
    from transformers.generation import logits_process as LP
    from oracle.reference_cpu import logmel_ref, synthetic_batch
    cfg, model, oracle = _build(torch.float32, 23)
    b = synthetic_batch(cfg, 2, 2.0, n_text=20, audio_start=4, n_supervised=4)
    b.pop("labels")
    b["audio_values"] = logmel_ref(b.pop("pcm"), 80)
    b["attention_mask"][1, :3] = 0
    b["input_ids"][1, :3] = 2
    gb = {k: v.to(DEV) for k, v in b.items()}
    T, N, eos = b["input_ids"].shape[1], 8, 2
    plain = model.generate(max_new_tokens=N, eos_token_id=-1, **gb).cpu()
    p0, p1 = plain[0, T:].tolist(), plain[1, T:].tolist()
    kw = dict(no_repeat_ngram_size=2, bad_words_ids=[[p0[1]], [p1[0], p1[1]]], min_new_tokens=5, suppress_tokens=[p1[3]], repetition_penalty=1.3)
    cpu = torch.device("cpu")
    procs = LP.LogitsProcessorList([LP.RepetitionPenaltyLogitsProcessor(1.3), LP.NoRepeatNGramLogitsProcessor(2),
                                    LP.NoBadWordsLogitsProcessor(kw["bad_words_ids"], eos_token_id=[eos]), LP.MinNewTokensLengthLogitsProcessor(T, 5, [eos], device=cpu),
                                    LP.SuppressTokensLogitsProcessor(kw["suppress_tokens"], device=cpu)])
    want = oracle.generate_greedy(N, eos_token_id=eos, logits_processor=procs, **b)
    got = model.generate(max_new_tokens=N, eos_token_id=eos, **kw, **gb).cpu()
    n = min(got.shape[1], want.shape[1])
    assert n >= T + 5 and torch.equal(got[:, :n], want[:, :n]), (got[:, T:], want[:, T:])
    assert not torch.equal(got[:, :plain.shape[1]], plain[:, :got.shape[1]])
    with pytest.raises(NotImplementedError, match="penalty_alpha"):
        model.generate(max_new_tokens=2, penalty_alpha=0.6, top_k=4, **gb)
