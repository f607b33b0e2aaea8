"""generate()'s HOST logic — KV-state bookkeeping, cache growth, prefix check, terminators, padding of finished rows, the
streamer protocol, repetition penalty — exercised on the CPU against a tiny SIMULATED device: the C-ABI calls generate()
makes are replaced by torch stand-ins whose "logits" depend on every cached row, its position and the left padding, so a
row that is missing, misplaced or stale changes the generated tokens.  (This is a test double for the library, not a
fallback: the product never runs without libuvx; the device kernels themselves are checked in tests/test_generate_gpu.py.)"""
import types

import pytest
import torch

import ultravox_amd.model as M
from ultravox_amd import _lib
from ultravox_amd.model import KVState, UltravoxModel

V, D = 97, 4


class FakeLib:
    """cache: f32 [1 layer][k | v][B][Tmax][1]: k = the token feature (embedding[0]), v = its RoPE position."""

    def __init__(self):
        self.calls = []

    @staticmethod
    def _planes(cache, B, Tmax):
        return cache[:2 * B * Tmax * 4].view(torch.float32).view(2, B, Tmax)

    @staticmethod
    def _logits(k, v, lo, hi, out):
        for b in range(k.shape[0]):
            s = float((k[b, lo[b]:hi] * (v[b, lo[b]:hi] + 1.0)).sum())
            target = int(round(s)) % V
            out[b] = -(torch.arange(V, dtype=torch.float32) - target).abs()

    def uvx_kv_cache_bytes(self, cfg, B, Tmax):
        return 2 * B * Tmax * 4

    def uvx_llm_infer_ws_bytes(self, cfg, B, T):
        return 64

    def uvx_llm_prefill_chunk_ws_bytes(self, cfg, B, Tn, P):
        return 64

    def uvx_llm_prefill(self, st, cfg, lw, embeds, am, B, T, cache, Tmax, next_pos, kv_start, logits, ws, nb):
        self.calls.append(("prefill", T))
        kv = self._planes(cache, B, Tmax)
        for b in range(B):
            keep = torch.ones(T, dtype=torch.bool) if am is None else am[b].bool()
            pos = torch.cumsum(keep.long(), 0) - 1
            pos[~keep] = 1
            kv[0, b, :T], kv[1, b, :T] = embeds[b, :, 0], pos.float()
            idx = torch.nonzero(keep)[:, 0]
            kv_start[b], next_pos[b] = int(idx[0]) if len(idx) else 0, int(keep.sum())
        self._logits(kv[0], kv[1], kv_start.tolist(), T, logits)
        return 0

    def uvx_llm_prefill_chunk(self, st, cfg, lw, chunk, B, Tn, cache, Tmax, cur_len, pos0, kv_start, logits, ws, nb):
        self.calls.append(("chunk", cur_len, Tn))
        kv = self._planes(cache, B, Tmax)
        for b in range(B):
            kv[0, b, cur_len:cur_len + Tn] = chunk[b, :, 0]
            kv[1, b, cur_len:cur_len + Tn] = float(pos0[b]) + torch.arange(Tn, dtype=torch.float32)
        self._logits(kv[0], kv[1], kv_start.tolist(), cur_len + Tn, logits)
        return 0

    def uvx_llm_decode(self, st, cfg, lw, emb, pos, kv_start, cache, Tmax, cur_len, B, logits, ws, nb):
        self.calls.append(("decode", cur_len))
        kv = self._planes(cache, B, Tmax)
        kv[0, :, cur_len], kv[1, :, cur_len] = emb[:, 0], pos.float()
        self._logits(kv[0], kv[1], kv_start.tolist(), cur_len + 1, logits)
        return 0

    def uvx_embed_merge(self, st, cfg, table, tok, a, b, c, d, B, T, e, f, out, offs):
        out.copy_(table[tok])
        return 0

    def uvx_argmax(self, st, code, logits, B, Vv, out):
        out.copy_(logits.argmax(-1))
        return 0

    def uvx_greedy_select(self, st, code, logits, B, Vv, eos_ids, n_eos, pad, live, nxt, seq, stride, col, pos0, pos, step, counter):
        """include/uvx.h: tok = unfinished ? argmax : pad; appended; unfinished &= tok not an EOS id; positions = positions0 + step; the count of
        unfinished rows goes to counter[step & 1] and the other slot is cleared."""
        tok = torch.where(live.bool(), logits.argmax(-1), torch.full((B,), int(pad.value), dtype=torch.int64))
        nxt.copy_(tok)
        seq[:, int(col.value)] = tok
        live.copy_((live.bool() & ~torch.isin(tok, eos_ids[:n_eos])).int())
        if pos is not None:
            pos.copy_(pos0 + step)
        counter[step & 1] = int(live.sum())
        counter[(step & 1) ^ 1] = 0
        return 0


@pytest.fixture()
def model(monkeypatch):
    fake = FakeLib()
    monkeypatch.setattr(_lib, "lib", lambda: fake)
    monkeypatch.setattr(M, "ptr", lambda t: t)
    monkeypatch.setattr(M, "stream_ptr", lambda: None)
    monkeypatch.setattr(M, "check", lambda rc, what="": None)
    monkeypatch.setattr(M.C, "byref", lambda x: x)
    monkeypatch.setattr(M.C, "c_size_t", lambda x: x)
    m = UltravoxModel.__new__(UltravoxModel)
    m.device, m.dtype, m.code, m.text_lora_r = torch.device("cpu"), torch.float32, 0, 0
    m.config = types.SimpleNamespace(vocab_size=V, text_config=types.SimpleNamespace(eos_token_id=2, hidden_size=D))
    table = (torch.arange(V, dtype=torch.float32)[:, None] + 1.0).repeat(1, D)
    m._llm = {"rope_len": 4096, "embed": table}
    m._c = types.SimpleNamespace(llm_layers=1)
    m._lw, m._ws = None, {}
    m._embed_merge = lambda emb, ids, *a: table[ids]
    m.fake = fake
    return m


def left_padded(B, T, pads, seed=0):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, V, (B, T), generator=g)
    am = torch.ones(B, T, dtype=torch.long)
    for b, p in enumerate(pads):
        am[b, :p] = 0
        ids[b, :p] = 2
    return ids, am


def test_cached_continuation_equals_a_fresh_prefill(model):
    ids1, am1 = left_padded(2, 9, [0, 3])
    out1 = model.generate(ids1, attention_mask=am1, max_new_tokens=5, eos_token_id=-1, return_dict_in_generate=True)
    st = out1.past_key_values
    assert isinstance(st, KVState) and st.cur_len == 13 and st.Tmax == 14 and st.tokens.shape == (2, 13)
    assert st.pos_next.tolist() == [13, 10] and st.kv_start.tolist() == [0, 3]
    ids2 = torch.cat([out1.sequences, torch.randint(3, V, (2, 6), generator=torch.Generator().manual_seed(1))], 1)
    am2 = torch.cat([am1, torch.ones(2, ids2.shape[1] - 9, dtype=torch.long)], 1)
    fresh = model.generate(ids2, attention_mask=am2, max_new_tokens=4, eos_token_id=-1, return_dict_in_generate=True)
    assert model.last_prefill_reused == 0
    model.fake.calls.clear()
    out2 = model.generate(ids2, attention_mask=am2, max_new_tokens=4, eos_token_id=-1, past_key_values=st, return_dict_in_generate=True)
    assert model.last_prefill_reused == 13 and model.fake.calls[0] == ("chunk", 13, ids2.shape[1] - 13)     # only the new tokens ran
    assert torch.equal(out2.sequences, fresh.sequences)
    s2, sf = out2.past_key_values, fresh.past_key_values
    assert s2.cur_len == sf.cur_len == ids2.shape[1] + 3 and s2.Tmax == sf.Tmax and torch.equal(s2.pos_next, sf.pos_next)
    rows = lambda s: FakeLib._planes(s.cache, 2, s.Tmax)[:, :, :s.cur_len]
    a, b = rows(s2).clone(), rows(sf).clone()
    a[:, 1, :3] = b[:, 1, :3] = 0                      # left padding of row 1: don't-care positions
    assert torch.equal(a, b)                           # the grown cache holds exactly what a fresh prefill + decode wrote
    assert st.cur_len == 13 and torch.equal(rows(st)[0, 0], rows(s2)[0, 0, :13])       # the consumed state is still readable


def test_cache_is_dropped_when_it_is_not_a_prefix_and_reused_in_place_when_it_fits(model):
    ids, _ = left_padded(1, 8, [0], seed=3)
    out = model.generate(ids, max_new_tokens=40, eos_token_id=-1, return_dict_in_generate=True)
    st = out.past_key_values
    assert st.Tmax == 48 and st.cur_len == 47
    nxt = torch.cat([out.sequences[:, :20], torch.tensor([[5, 6]])], 1)           # shorter than the cache: not a prefix of it
    want = model.generate(nxt, max_new_tokens=3, eos_token_id=-1)
    got = model.generate(nxt, max_new_tokens=3, eos_token_id=-1, past_key_values=st)
    assert model.last_prefill_reused == 0 and torch.equal(got, want)
    bad = out.sequences.clone()
    bad[0, 5] = (bad[0, 5] + 1) % V
    bad = torch.cat([bad, torch.tensor([[7]])], 1)
    model.generate(bad, max_new_tokens=2, eos_token_id=-1, past_key_values=st)
    assert model.last_prefill_reused == 0
    # a continuation that fits the existing buffer appends in place (no growth copy)
    st_small = model.generate(ids, max_new_tokens=6, eos_token_id=-1, return_dict_in_generate=True)
    big = KVState(cache=torch.zeros(2 * 1 * 64 * 4, dtype=torch.uint8), Tmax=64, cur_len=st_small.past_key_values.cur_len,
                  pos_next=st_small.past_key_values.pos_next, kv_start=st_small.past_key_values.kv_start,
                  tokens=st_small.past_key_values.tokens)
    FakeLib._planes(big.cache, 1, 64)[:, :, :big.cur_len] = FakeLib._planes(st_small.past_key_values.cache, 1, 14)[:, :, :big.cur_len]
    cont = torch.cat([st_small.sequences, torch.tensor([[9, 11]])], 1)
    o = model.generate(cont, max_new_tokens=3, eos_token_id=-1, past_key_values=big, return_dict_in_generate=True)
    assert o.past_key_values.cache.data_ptr() == big.cache.data_ptr() and o.past_key_values.Tmax == 64
    assert torch.equal(o.sequences, model.generate(cont, max_new_tokens=3, eos_token_id=-1))
    with pytest.raises(TypeError):
        model.generate(ids, past_key_values=object())


def test_terminators_padding_streamer_and_repetition_penalty(model):
    ids, am = left_padded(2, 7, [0, 2], seed=5)
    free = model.generate(ids, attention_mask=am, max_new_tokens=6, eos_token_id=-1)[:, 7:]
    stop = int(free[0, 2])                                 # row 0 stops at its 3rd token, row 1 runs on
    assume_other = stop not in free[1].tolist()

    class Rec:
        def __init__(self):
            self.puts, self.ended = [], False

        def put(self, v):
            self.puts.append(v.clone())

        def end(self):
            self.ended = True
    rec = Rec()
    out = model.generate(ids, attention_mask=am, max_new_tokens=6, eos_token_id=[stop, 96], pad_token_id=1, streamer=rec)
    first = free[0].tolist().index(stop)
    assert out[0, 7:7 + first + 1].tolist() == free[0, :first + 1].tolist()
    if assume_other:
        assert out.shape[1] == 13 and out[0, 7 + first + 1:].tolist() == [1] * (5 - first)       # finished row padded
        assert out[1, 7:].tolist() == free[1].tolist()
    assert rec.ended and torch.equal(rec.puts[0], ids) and len(rec.puts) == out.shape[1] - 7 + 1
    with pytest.warns(UserWarning, match="no effect"):      # reporting / cache-management keywords: no effect on the tokens
        model.generate(ids, attention_mask=am, max_new_tokens=1, eos_token_id=-1, use_cache=True, output_attentions=False)
    with pytest.raises(NotImplementedError, match="typical_p"):      # HF would act on it: refused, not ignored
        model.generate(ids, attention_mask=am, max_new_tokens=1, eos_token_id=-1, typical_p=0.9)
    with pytest.raises(ValueError, match="not used by the model"):   # HF's own message for an unknown keyword
        model.generate(ids, attention_mask=am, max_new_tokens=1, eos_token_id=-1, no_repeat_ngram=2)
    with pytest.raises(NotImplementedError, match="min_p"):      # (a sampling warper without do_sample; beam sampling and stopping criteria are built: tests below)
        model.generate(ids, num_beams=4, min_p=0.1)
    with pytest.raises(ValueError):
        model.generate(ids, num_beams=2, num_return_sequences=3)
    with pytest.raises(ValueError):
        model.generate(ids, num_return_sequences=2)
    with pytest.raises(ValueError):
        model.generate(ids, num_beams=2, streamer=rec)
    with pytest.raises(ValueError):
        model.generate(ids, do_sample=True, temperature=0.0)
    with pytest.raises(ValueError):
        model.generate(ids, max_new_tokens=5000)
    # repetition penalty: the scores of seen ids are damped before the arg-max (here: the un-penalised winner is a seen id)
    pen = model.generate(ids, attention_mask=am, max_new_tokens=6, eos_token_id=-1, repetition_penalty=50.0)
    assert pen.shape == (2, 13)
    g = torch.Generator().manual_seed(0)
    a = model.generate(ids, attention_mask=am, max_new_tokens=4, eos_token_id=-1, do_sample=True, temperature=0.8, top_k=5, generator=g)
    g = torch.Generator().manual_seed(0)
    b = model.generate(ids, attention_mask=am, max_new_tokens=4, eos_token_id=-1, do_sample=True, temperature=0.8, top_k=5, generator=g)
    assert torch.equal(a, b)


@pytest.mark.parametrize("case", [dict(num_beams=3), dict(num_beams=4, length_penalty=0.0, num_return_sequences=2), dict(num_beams=2, early_stopping=True),
                                  dict(num_beams=3, early_stopping="never", length_penalty=2.0), dict(num_beams=3, repetition_penalty=3.0),
                                  dict(num_beams=3, no_repeat_ngram_size=1, min_new_tokens=3, bad_words_ids=[[50], [51, 52]])])
@pytest.mark.parametrize("n_eos", [1, 30])
def test_beam_search_host_logic_against_the_oracle_restatement(model, case, n_eos):
    """generate(num_beams > 1) on the simulated device against the oracle's list-based restatement of HF's beam search
    (oracle.reference_cpu.beam_search_ref, pinned to HF in tests/test_oracle_pinning.py) driven by a CACHE-FREE evaluation of the same
    fake "model": its logits depend on every cached row of a hypothesis, so a cache plane that did not follow its beam changes the
    tokens.  Tie-free logits (a triangular bump with different slopes left and right of the target)."""
    from oracle.reference_cpu import beam_search_ref

    def bump(target):
        d = torch.arange(V, dtype=torch.float32) - target
        return -(d.abs() * 0.31 + (d > 0) * 0.17)
    model.fake._logits = lambda k, v, lo, hi, out: [out.__setitem__(b, bump(int(round(float((k[b, lo[b]:hi] * (v[b, lo[b]:hi] + 1.0)).sum()))) % V))
                                                    for b in range(k.shape[0])]
    ids, am = left_padded(3, 8, [0, 3, 1], seed=11)
    eos = -1 if n_eos == 1 else list(range(5, 5 + n_eos))

    def next_logits(hyps):
        out = torch.empty(len(hyps), len(hyps[0]), V)
        for b, item in enumerate(hyps):
            keep = am[b].bool()
            lo = int(torch.nonzero(keep)[0, 0])
            for j, toks in enumerate(item):
                feat = torch.cat([ids[b].float() + 1.0, torch.tensor(toks, dtype=torch.float32) + 1.0])
                pos = torch.cumsum(keep.long(), 0) - 1
                pos = torch.cat([pos, int(keep.sum()) + torch.arange(len(toks))]).float()
                out[b, j] = bump(int(round(float((feat[lo:] * (pos[lo:] + 1.0)).sum()))) % V)
        return out
    from test_oracle_pinning import hf_processor_list
    kw, procs = hf_processor_list({k: v for k, v in case.items() if k != "num_beams"}, 8, eos)
    want = beam_search_ref(next_logits, ids, 6, eos, 1, case["num_beams"], kw.get("length_penalty", 1.0), kw.get("early_stopping", False),
                           kw.get("num_return_sequences", 1), kw.get("repetition_penalty"), procs)
    model.fake.calls.clear()
    got = model.generate(ids, attention_mask=am, max_new_tokens=6, eos_token_id=eos, pad_token_id=1, **case)
    assert got.shape == want.shape and torch.equal(got, want), (got[:, 8:], want[:, 8:])
    assert model.fake.calls[0] == ("prefill", 8) and all(c[0] == "decode" for c in model.fake.calls[1:])      # one prefill of the B prompts
    out = model.generate(ids, attention_mask=am, max_new_tokens=6, eos_token_id=eos, pad_token_id=1, return_dict_in_generate=True, **case)
    assert torch.equal(out.sequences, want) and out.sequences_scores.shape == (want.shape[0],) and out.past_key_values is None


@pytest.mark.parametrize("case", [dict(num_beams=3), dict(num_beams=2, temperature=0.8, top_k=20), dict(num_beams=3, top_p=0.9, repetition_penalty=2.0, num_return_sequences=2),
                                  dict(num_beams=4, temperature=1.5, top_k=30, top_p=0.95, length_penalty=0.0)])
@pytest.mark.parametrize("n_eos", [1, 2])
def test_beam_sampling_host_logic_against_the_oracle_restatement(model, case, n_eos):
    """generate(num_beams > 1, do_sample=True) on the simulated device against the oracle's restatement of HF's beam sampling (pinned to HF generate in
    tests/test_oracle_pinning.py), both drawing from the CPU generator after the same torch.manual_seed: one multinomial on the [B, beams * V] matrix per
    step, the warpers ([3P] HF's own classes on the oracle's side, UltravoxModel._warp on ours) behind the score processors with 1 + n_terminators tokens
    kept - token for token, and not what plain beam search returns."""
    from oracle.reference_cpu import beam_search_ref
    from transformers.generation import logits_process as LP

    def bump(target):
        d = torch.arange(V, dtype=torch.float32) - target
        return -(d.abs() * 0.11 + (d > 0) * 0.07)
    model.fake._logits = lambda k, v, lo, hi, out: [out.__setitem__(b, bump(int(round(float((k[b, lo[b]:hi] * (v[b, lo[b]:hi] + 1.0)).sum()))) % V))
                                                    for b in range(k.shape[0])]
    ids, am = left_padded(3, 8, [0, 3, 1], seed=11)
    eos = -1 if n_eos == 1 else [5, 9]

    def next_logits(hyps):
        out = torch.empty(len(hyps), len(hyps[0]), V)
        for b, item in enumerate(hyps):
            keep = am[b].bool()
            lo = int(torch.nonzero(keep)[0, 0])
            for j, toks in enumerate(item):
                feat = torch.cat([ids[b].float() + 1.0, torch.tensor(toks, dtype=torch.float32) + 1.0])
                pos = torch.cumsum(keep.long(), 0) - 1
                pos = torch.cat([pos, int(keep.sum()) + torch.arange(len(toks))]).float()
                out[b, j] = bump(int(round(float((feat[lo:] * (pos[lo:] + 1.0)).sum()))) % V)
        return out
    warp = LP.LogitsProcessorList()
    if case.get("temperature", 1.0) != 1.0:
        warp.append(LP.TemperatureLogitsWarper(case["temperature"]))
    if case.get("top_k"):
        warp.append(LP.TopKLogitsWarper(case["top_k"], min_tokens_to_keep=n_eos + 1))
    if case.get("top_p", 1.0) < 1.0:
        warp.append(LP.TopPLogitsWarper(case["top_p"], min_tokens_to_keep=n_eos + 1))
    torch.manual_seed(31)
    want = beam_search_ref(next_logits, ids, 6, eos, 1, case["num_beams"], case.get("length_penalty", 1.0), False, case.get("num_return_sequences", 1),
                           case.get("repetition_penalty"), warp, do_sample=True)
    torch.manual_seed(31)
    got = model.generate(ids, attention_mask=am, max_new_tokens=6, eos_token_id=eos, pad_token_id=1, do_sample=True, **case)
    assert got.shape == want.shape and torch.equal(got, want), (got[:, 8:], want[:, 8:])
    plain = model.generate(ids, attention_mask=am, max_new_tokens=6, eos_token_id=eos, pad_token_id=1,
                           **{k: v for k, v in case.items() if k not in ("temperature", "top_k", "top_p")})
    assert plain.shape != got.shape or not torch.equal(plain, got)


@pytest.mark.parametrize("case", [dict(num_beams=3), dict(num_beams=2, early_stopping=True), dict(num_beams=4, length_penalty=0.0, num_return_sequences=2)])
def test_beam_search_with_stopping_criteria_against_the_oracle_restatement(model, case):
    """generate(num_beams > 1, stopping_criteria=[..]) on the simulated device against the oracle's restatement (pinned to HF generate in
    tests/test_oracle_pinning.py): the caller's criteria see prompt + candidate of the K continuations of every step and end hypotheses like a terminator."""
    from oracle.reference_cpu import beam_search_ref

    def bump(target):
        d = torch.arange(V, dtype=torch.float32) - target
        return -(d.abs() * 0.31 + (d > 0) * 0.17)
    model.fake._logits = lambda k, v, lo, hi, out: [out.__setitem__(b, bump(int(round(float((k[b, lo[b]:hi] * (v[b, lo[b]:hi] + 1.0)).sum()))) % V))
                                                    for b in range(k.shape[0])]
    ids, am = left_padded(3, 8, [0, 3, 1], seed=11)

    def next_logits(hyps):
        out = torch.empty(len(hyps), len(hyps[0]), V)
        for b, item in enumerate(hyps):
            keep = am[b].bool()
            lo = int(torch.nonzero(keep)[0, 0])
            for j, toks in enumerate(item):
                feat = torch.cat([ids[b].float() + 1.0, torch.tensor(toks, dtype=torch.float32) + 1.0])
                pos = torch.cumsum(keep.long(), 0) - 1
                pos = torch.cat([pos, int(keep.sum()) + torch.arange(len(toks))]).float()
                out[b, j] = bump(int(round(float((feat[lo:] * (pos[lo:] + 1.0)).sum()))) % V)
        return out
    crit = [lambda seq, scores: (seq[:, -1] % 3) == 0, lambda seq, scores: seq[:, 8:].sum(-1) > 250]      # (the second reads the whole hypothesis behind the 8-token prompt)
    want = beam_search_ref(next_logits, ids, 6, -1, 1, case["num_beams"], case.get("length_penalty", 1.0), case.get("early_stopping", False),
                           case.get("num_return_sequences", 1), None, None, False, crit)
    got = model.generate(ids, attention_mask=am, max_new_tokens=6, eos_token_id=-1, pad_token_id=1, stopping_criteria=crit, **case)
    assert got.shape == want.shape and torch.equal(got, want), (got[:, 8:], want[:, 8:])
    free = model.generate(ids, attention_mask=am, max_new_tokens=6, eos_token_id=-1, pad_token_id=1, **case)
    assert free.shape != got.shape or not torch.equal(free, got)


def test_longest_common_prefix_reuse_is_opt_in_and_keeps_the_old_state_intact(model):
    ids, am = left_padded(2, 10, [0, 2], seed=7)
    out = model.generate(ids, attention_mask=am, max_new_tokens=8, eos_token_id=-1, return_dict_in_generate=True)
    st = out.past_key_values
    snapshot = st.cache.clone()
    # the new prompt follows the cached ids up to index 12 (row 0) / 14 (row 1), then departs
    nxt = torch.cat([out.sequences[:, :15], torch.randint(3, V, (2, 5), generator=torch.Generator().manual_seed(2))], 1)
    nxt[0, 12] = (nxt[0, 12] + 1) % V
    nxt[1, 14] = (nxt[1, 14] + 1) % V
    am2 = torch.cat([am, torch.ones(2, 10, dtype=torch.long)], 1)
    want = model.generate(nxt, attention_mask=am2, max_new_tokens=4, eos_token_id=-1)
    model.generate(nxt, attention_mask=am2, max_new_tokens=4, eos_token_id=-1, past_key_values=st)
    assert model.last_prefill_reused == 0                                  # default: all or nothing
    st.partial_ok = True
    model.fake.calls.clear()
    got = model.generate(nxt, attention_mask=am2, max_new_tokens=4, eos_token_id=-1, past_key_values=st, return_dict_in_generate=True)
    assert model.last_prefill_reused == 12 and model.fake.calls[0] == ("chunk", 12, 8)          # min over the rows
    assert torch.equal(got.sequences, want)
    new = got.past_key_values
    assert new.partial_ok and new.cache.data_ptr() != st.cache.data_ptr() and torch.equal(st.cache, snapshot)
    # a mismatch inside the left padding region cannot be reused (positions would not line up)
    early = nxt.clone()
    early[1, 1] = 5
    model.generate(early, attention_mask=am2, max_new_tokens=2, eos_token_id=-1, past_key_values=st)
    assert model.last_prefill_reused == 0
    # a shorter prompt that is a prefix of the cached ids: everything but its last token comes from the cache
    short = out.sequences[:, :11]
    am3 = torch.cat([am, torch.ones(2, 1, dtype=torch.long)], 1)
    w = model.generate(short, attention_mask=am3, max_new_tokens=3, eos_token_id=-1)
    g = model.generate(short, attention_mask=am3, max_new_tokens=3, eos_token_id=-1, past_key_values=st)
    assert model.last_prefill_reused == 10 and torch.equal(g, w)


def test_local_inference_marks_its_states_partial_ok():
    from fake_tokenizer import FakeChatTokenizer
    from oracle.reference_cpu import FeatureExtractorRef
    from ultravox_amd.inference import LocalInference, VoiceSample
    from ultravox_amd.processing import UltravoxProcessor
    tok = FakeChatTokenizer()
    state = KVState(cache=torch.zeros(8, dtype=torch.uint8), Tmax=1, cur_len=1, pos_next=torch.zeros(1, dtype=torch.int32),
                    kv_start=torch.zeros(1, dtype=torch.int32), tokens=torch.zeros(1, 1, dtype=torch.long))

    class Stub:
        device, dtype = torch.device("cpu"), torch.float32

        def generate(self, **kw):
            return types.SimpleNamespace(sequences=torch.cat([kw["input_ids"], torch.tensor([[128009]])], 1), past_key_values=state)
    inf = LocalInference(Stub(), UltravoxProcessor(FeatureExtractorRef(80), tokenizer=tok), tok, dtype=torch.float32, conversation_mode=True)
    assert not state.partial_ok
    inf.infer(VoiceSample.from_prompt("Hi"))
    assert inf.past_key_values is state and state.partial_ok


def _hf():
    from transformers.generation import logits_process as LP
    return LP


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_score_processors_match_the_hf_classes(seed):
    """generation.py restates [3P] transformers.generation.logits_process; every function against the class it cites, on random rows with repeats
    (prompt + generated ids, left padding included - HF hands the processors the whole row)."""
    from ultravox_amd import generation as G
    LP = _hf()
    g = torch.Generator().manual_seed(seed)
    B, L, Vv = 5, 17, 23
    ids = torch.randint(0, 9, (B, L), generator=g)                  # small alphabet: n-grams repeat
    scores = torch.randn(B, Vv, generator=g) * 3
    for n in (1, 2, 3, 4, 18, 19):
        want = LP.NoRepeatNGramLogitsProcessor(n)(ids, scores.clone())
        assert torch.equal(G.no_repeat_ngram_(scores.clone(), ids, n), want), n
    assert torch.equal(G.repetition_penalty_(scores.clone(), ids, 1.3), LP.RepetitionPenaltyLogitsProcessor(1.3)(ids, scores.clone()))
    bad = [[3], [22], [int(ids[0, -1]), 5], [int(ids[1, -2]), int(ids[1, -1]), 7], [1, 2, 3, 4, 5, 6, 7, 8, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9], [2]]
    want = LP.NoBadWordsLogitsProcessor(bad, eos_token_id=[2, 11])(ids, scores.clone())
    assert torch.equal(G.bad_words_(scores.clone(), ids, bad, [2, 11]), want)
    dev = torch.device("cpu")
    assert torch.equal(G.suppress_(scores.clone(), [4, 6]), LP.SuppressTokensLogitsProcessor([4, 6], device=dev)(ids, scores.clone()))
    assert torch.equal(G.force_(scores.clone(), [6]), LP.ForcedEOSTokenLogitsProcessor(L + 1, 6, device=dev)(ids, scores.clone()))
    for mp in (0.05, 0.3, 0.9):
        assert torch.equal(G.min_p_(scores.clone(), mp), LP.MinPLogitsWarper(mp)(ids, scores.clone())), mp
    # the assembled list in HF's order
    kw = dict(no_repeat_ngram_size=2, bad_words_ids=bad, min_length=L + 3, min_new_tokens=4, suppress_tokens=[4], begin_suppress_tokens=[8], forced_eos_token_id=6)
    for prompt_len, cur in ((L, ids), (L - 3, ids), (L - 6, ids)):
        for max_new in (L - prompt_len + 1, 30):
            procs = G.ScoreProcessors(kw, prompt_len, max_new, [2, 11], repetition_penalty=1.2)
            hf = LP.LogitsProcessorList([LP.RepetitionPenaltyLogitsProcessor(1.2), LP.NoRepeatNGramLogitsProcessor(2),
                                         LP.NoBadWordsLogitsProcessor(bad, eos_token_id=[2, 11]), LP.MinLengthLogitsProcessor(L + 3, [2, 11], device=dev),
                                         LP.MinNewTokensLengthLogitsProcessor(prompt_len, 4, [2, 11], device=dev),
                                         LP.ForcedEOSTokenLogitsProcessor(prompt_len + max_new, 6, device=dev), LP.SuppressTokensLogitsProcessor([4], device=dev),
                                         LP.SuppressTokensAtBeginLogitsProcessor([8], prompt_len, device=dev)])
            assert procs.active and torch.equal(procs(cur, scores.clone()), hf(cur, scores.clone())), (prompt_len, max_new)
    assert not G.ScoreProcessors({}, 5, 5, [2]).active


@pytest.mark.parametrize("kind", ["ngram", "bad_words", "min_length", "begin_forced_rep"])
def test_generate_with_score_processors_equals_a_loop_over_the_hf_classes(model, kind):
    """The decode loop with HF's generation keywords (ultravox_model.py:422-426 forwards them all) against a plain loop that builds HF's OWN
    LogitsProcessorList and applies it to the same step logits (the simulated device): same tokens, EOS / padding handling included.  The keyword
    values are chosen from the un-processed continuation so that every processor bites."""
    LP = _hf()
    ids, am = left_padded(2, 7, [0, 2], seed=9)
    N, pad = 8, 1
    plain = model.generate(ids, attention_mask=am, max_new_tokens=N, eos_token_id=-1)
    p0, p1 = plain[0, 7:].tolist(), plain[1, 7:].tolist()
    eos = [p0[3]]                                                   # row 0 would stop at its 4th token
    kw = {"ngram": dict(no_repeat_ngram_size=1),                     # every id of the prompt (and every earlier pick) is banned
          "bad_words": dict(bad_words_ids=[[p1[0]], [p0[0], p0[1]]], min_new_tokens=6),
          "min_length": dict(min_length=7 + 6, suppress_tokens=[p0[1], p1[2]]),
          "begin_forced_rep": dict(begin_suppress_tokens=[p0[0], p1[0]], forced_eos_token_id=eos[0], repetition_penalty=1.4)}[kind]
    got = model.generate(ids, attention_mask=am, max_new_tokens=N, eos_token_id=eos, pad_token_id=pad, **kw)
    dev = torch.device("cpu")
    procs = LP.LogitsProcessorList()
    if "repetition_penalty" in kw:
        procs.append(LP.RepetitionPenaltyLogitsProcessor(kw["repetition_penalty"]))
    if "no_repeat_ngram_size" in kw:
        procs.append(LP.NoRepeatNGramLogitsProcessor(kw["no_repeat_ngram_size"]))
    if "bad_words_ids" in kw:
        procs.append(LP.NoBadWordsLogitsProcessor(kw["bad_words_ids"], eos_token_id=eos))
    if "min_length" in kw:
        procs.append(LP.MinLengthLogitsProcessor(kw["min_length"], eos, device=dev))
    if "min_new_tokens" in kw:
        procs.append(LP.MinNewTokensLengthLogitsProcessor(7, kw["min_new_tokens"], eos, device=dev))
    if "forced_eos_token_id" in kw:
        procs.append(LP.ForcedEOSTokenLogitsProcessor(7 + N, kw["forced_eos_token_id"], device=dev))
    if "suppress_tokens" in kw:
        procs.append(LP.SuppressTokensLogitsProcessor(kw["suppress_tokens"], device=dev))
    if "begin_suppress_tokens" in kw:
        procs.append(LP.SuppressTokensAtBeginLogitsProcessor(kw["begin_suppress_tokens"], 7, device=dev))
    # reference loop on the same simulated device: step logits come from a processor-free generate() that is FORCED along the reference's own tokens
    seq, unfinished = ids.clone(), torch.ones(2, dtype=torch.bool)
    for step in range(N):
        cur_am = torch.cat([am, torch.ones(2, seq.shape[1] - 7, dtype=torch.long)], 1)
        logits = model.generate(seq, attention_mask=cur_am, max_new_tokens=1, eos_token_id=-1, return_dict_in_generate=True, output_logits=True).logits[0]
        nxt = procs(seq, logits.float().clone()).argmax(-1)
        tok = torch.where(unfinished, nxt, torch.full_like(nxt, pad))
        seq = torch.cat([seq, tok[:, None]], 1)
        unfinished = unfinished & ~torch.isin(tok, torch.tensor(eos))
        if not bool(unfinished.any()):
            break
    assert torch.equal(got, seq)
    assert not torch.equal(model.generate(ids, attention_mask=am, max_new_tokens=N, eos_token_id=eos, pad_token_id=pad), got)      # the keywords bite


def test_custom_logits_processors_and_stopping_criteria(model):
    """`logits_processor=` / `stopping_criteria=` callables, HF's calling convention: (input_ids incl. the prompt, f32 scores)."""
    ids, am = left_padded(2, 7, [0, 2], seed=11)
    seen = []

    def only_even(input_ids, scores):
        seen.append(input_ids.shape[1])
        scores[:, 1::2] = float("-inf")
        return scores
    out = model.generate(ids, attention_mask=am, max_new_tokens=5, eos_token_id=-1, logits_processor=[only_even])
    assert seen == [7, 8, 9, 10, 11] and bool((out[:, 7:] % 2 == 0).all())
    stop_at = lambda input_ids, scores: torch.tensor([input_ids.shape[1] >= 10, False])       # row 0 stops once it is 10 long
    out = model.generate(ids, attention_mask=am, max_new_tokens=6, eos_token_id=-1, pad_token_id=1, stopping_criteria=[stop_at])
    assert out.shape[1] == 13 and out[0, 10:].tolist() == [1, 1, 1] and 1 not in out[1, 7:].tolist()
    assert model.generate(ids, attention_mask=am, max_new_tokens=6, eos_token_id=-1, stopping_criteria=[lambda i, s: i.shape[1] >= 9]).shape[1] == 9
    # (with beams the criteria see the B * K candidates of a step: test_beam_search_with_stopping_criteria_against_the_oracle_restatement)
