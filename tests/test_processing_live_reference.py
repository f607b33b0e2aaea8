"""Differential test against the LIVE reference processor (only where /root/reference exists, i.e. the build container; the
committed fixtures in tests/golden/ cover the GPU box): hypothesis draws audio counts / lengths / prompt shapes, both
processors run on the same inputs, every integer output and the audio tensor shape must agree exactly."""
import os
import sys
import types

import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from fake_tokenizer import FakeTokenizer
from oracle.reference_cpu import FeatureExtractorRef
from ultravox_amd.processing import UltravoxProcessor

REF = "/root/reference"
pytestmark = [pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "ultravox")), reason="reference tree only exists in the build container"),
              pytest.mark.filterwarnings("ignore::DeprecationWarning")]      # the reference's own numpy-2 deprecations

# lengths in samples: around the 2-hop minimum, hop boundaries, 1 s, and the 30 s chunk boundary (480000) incl. multi-chunk
LENGTHS = [0, 1, 159, 160, 161, 319, 320, 321, 4000, 16000, 16001, 47999, 479840, 480000, 480001, 480160, 560000, 960000, 960001]
WORDS = ["Transcribe", "this", "please", "and", "then", "answer", ":", "ok", "\n", "what", "follows", "?"]


def reference_processor(tok):
    import transformers
    sys.path.insert(0, REF)
    try:
        from ultravox.model import ultravox_processing
    finally:
        sys.path.remove(REF)
    fe = transformers.WhisperFeatureExtractor()
    ap = type("AP", (), {"feature_extractor": fe, "model_input_names": fe.model_input_names,
                         "__call__": staticmethod(lambda *a, **k: fe(*a, **k))})()
    proc = ultravox_processing.UltravoxProcessor.__new__(ultravox_processing.UltravoxProcessor)
    proc.audio_padding, proc.encoder_ds_factor, proc.stack_factor = "longest", 2, 8
    proc.audio_placeholder, proc.audio_context_size = "<|audio|>", 3000
    proc.vocab = tok.get_vocab()
    proc.audio_token_replacement = tok.eos_token
    proc.audio_processor, proc.tokenizer = ap, tok
    return proc


@pytest.fixture(scope="module")
def both():
    tok = FakeTokenizer()
    tok.pad_token_id = tok.eos_token_id
    return reference_processor(tok), UltravoxProcessor(FeatureExtractorRef(80), tokenizer=tok)


@settings(max_examples=30, deadline=None, derandomize=True, database=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(lengths=st.lists(st.sampled_from(LENGTHS), max_size=3), words=st.lists(st.sampled_from(WORDS), min_size=1, max_size=8),
       placeholder_delta=st.sampled_from([0, 0, 0, 0, -1, 1]), chunks=st.booleans(), seed=st.integers(0, 3))
def test_processor_agrees_with_the_live_reference(both, lengths, words, placeholder_delta, chunks, seed):
    ref, mine = both
    rng = np.random.RandomState(seed)
    audios = [rng.randn(n).astype(np.float32) * 0.1 for n in lengths]
    n_ph = max(0, len(audios) + placeholder_delta)            # sometimes one placeholder too few / too many
    parts = [" ".join(words[i::n_ph + 1]) or "x" for i in range(n_ph + 1)]
    text = " <|audio|> ".join(parts)
    kw = dict(audios=audios, sampling_rate=16000, include_audio_num_chunks=chunks) if audios else {}

    def run(p):
        try:
            return p(text, **kw), None
        except ValueError as e:
            return None, str(e)
    want, want_err = run(ref)
    got, got_err = run(mine)
    assert got_err == want_err, (text, lengths)
    if want is None:
        return
    assert set(got.keys()) == set(want.keys()), (sorted(got.keys()), sorted(want.keys()))
    for k in want.keys():
        w, g = torch.as_tensor(want[k]), torch.as_tensor(got[k])
        if k == "audio_values":
            assert g.shape == w.shape and g.dtype == w.dtype
            assert torch.allclose(g, w, atol=2e-4), (g - w).abs().max()
        else:
            assert g.dtype == w.dtype and torch.equal(g, w), (k, text, lengths)


def reference_dataproc_cls():
    sys.path.insert(0, REF)
    try:
        import ultravox
        saved = sys.modules.get("ultravox.data")
        stub = types.ModuleType("ultravox.data")      # the real package needs librosa / soundfile; only these names are used
        stub.Dataproc = type("Dataproc", (), {"__init__": lambda self, dataset: setattr(self, "_dataset", dataset)})
        stub.SizedIterableDataset = stub.VoiceSample = stub.Augmentation = object
        sys.modules["ultravox.data"] = ultravox.data = stub
        try:
            from ultravox.model import ultravox_config, ultravox_data_proc
        finally:
            if saved is None:
                sys.modules.pop("ultravox.data", None)
            else:
                sys.modules["ultravox.data"] = saved
    finally:
        sys.path.remove(REF)
    return ultravox_data_proc.UltravoxDataproc, ultravox_config.LossMaskType


@settings(max_examples=40, deadline=None, derandomize=True, database=None, suppress_health_check=[HealthCheck.too_slow])
@given(n_samples=st.sampled_from([None, 0, 200, 16000, 40000, 560000]), mask=st.sampled_from(["last_assistant", "after_audio", "all"]),
       alt=st.booleans(), max_resp=st.sampled_from([None, 1, 3, 50]), inference=st.booleans(), system=st.booleans(),
       words=st.lists(st.sampled_from(WORDS), min_size=1, max_size=6), reply=st.lists(st.sampled_from(WORDS), min_size=1, max_size=6),
       transcript=st.sampled_from([None, "", "hello there", "a b c d e f g"]), tail=st.booleans())
def test_dataproc_agrees_with_the_live_reference(n_samples, mask, alt, max_resp, inference, system, words, reply, transcript, tail):
    from fake_tokenizer import FakeChatTokenizer
    from ultravox_amd.config import LossMaskType
    from ultravox_amd.data_proc import UltravoxDataproc
    RefDataproc, RefMask = reference_dataproc_cls()
    user = " ".join(words) + (" <|audio|>" if n_samples is not None else "") + (" thanks" if tail else "")
    messages = ([{"role": "system", "content": "Be brief ."}] if system else []) + \
        [{"role": "user", "content": user}, {"role": "assistant", "content": " ".join(reply)}]
    audio = None if n_samples is None else np.random.RandomState(1).randn(n_samples).astype(np.float32) * 0.1

    def sample():
        return types.SimpleNamespace(messages=[dict(m) for m in messages], audio=audio, sample_rate=16000, audio_transcript=transcript)
    kw = dict(inference_mode=inference, include_alt_fields=alt, max_response_tokens=max_resp)
    tok_r, tok_m = FakeChatTokenizer("right"), FakeChatTokenizer("right")
    tok_r.pad_token_id = tok_m.pad_token_id = tok_r.eos_token_id
    ref = RefDataproc([], reference_processor(tok_r), RefMask(mask), **kw)
    mine = UltravoxDataproc([], UltravoxProcessor(FeatureExtractorRef(80), tokenizer=tok_m), LossMaskType(mask), **kw)

    def run(dp):
        try:
            return dp._process(sample()), None
        except (ValueError, IndexError) as e:
            return None, (type(e).__name__, str(e))
    want, want_err = run(ref)
    got, got_err = run(mine)
    assert got_err == want_err, (messages, mask, kw)
    if want is None:
        return
    assert set(got) == set(want)
    for k in want:
        if k == "audio_values":
            assert tuple(got[k].shape) == tuple(want[k].shape)
            continue
        w = want[k].tolist() if hasattr(want[k], "tolist") else want[k]
        g = got[k].tolist() if hasattr(got[k], "tolist") else got[k]
        assert g == w, (k, messages, mask, kw)


@settings(max_examples=25, deadline=None, derandomize=True, database=None, suppress_health_check=[HealthCheck.too_slow])
@given(side=st.sampled_from(["right", "left"]), alt=st.booleans(),
       picks=st.lists(st.sampled_from(["asr", "qa", "long", "two"]), min_size=1, max_size=4))
def test_collator_agrees_with_the_live_reference(side, alt, picks):
    """DataCollatorForSeq2SeqWithAudio over data-proc features: the reference's own __call__ body runs with the [3P]
    DataCollatorForSeq2Seq padding it inherits stubbed by the same rule (pad ids / mask / labels on tokenizer.padding_side),
    including the alt fields of the KL path and the left-padding displacement of audio_token_start_idx."""
    import unittest.mock as mock
    import torch.nn.functional as F
    import transformers
    from fake_tokenizer import FakeChatTokenizer
    from ultravox_amd.config import LossMaskType
    from ultravox_amd.data_proc import UltravoxDataproc
    from ultravox_amd.processing import DataCollatorForSeq2SeqWithAudio
    sys.path.insert(0, REF)
    try:
        from ultravox.model import ultravox_processing
    finally:
        sys.path.remove(REF)
    tok = FakeChatTokenizer(side)
    tok.pad_token_id = tok.eos_token_id
    dp = UltravoxDataproc([], UltravoxProcessor(FeatureExtractorRef(80), tokenizer=tok), LossMaskType.LAST_ASSISTANT, include_alt_fields=alt)
    rng = np.random.RandomState(0)
    bank = {
        "asr": ([{"role": "user", "content": "Transcribe <|audio|>"}, {"role": "assistant", "content": "a b c"}], 16000, "a b c"),
        "qa": ([{"role": "user", "content": "Listen <|audio|> and answer now please"}, {"role": "assistant", "content": "ok then"}], 40000, "hm"),
        "long": ([{"role": "user", "content": "<|audio|>"}, {"role": "assistant", "content": "long one indeed yes"}], 500000, "x y"),
        "two": ([{"role": "system", "content": "sys"}, {"role": "user", "content": "A <|audio|> B"}, {"role": "assistant", "content": "r"}], 8000, None),
    }
    feats = []
    for name in picks:
        msgs, n, tr = bank[name]
        feats.append(dp._process(types.SimpleNamespace(messages=[dict(m) for m in msgs], audio=rng.randn(n).astype(np.float32) * 0.1,
                                                       sample_rate=16000, audio_transcript=tr)))

    def hf_pad(features):
        n = max(len(f["input_ids"]) for f in features)

        def pad(x, v):
            g = n - len(x)
            return F.pad(torch.as_tensor(x), (g, 0) if side == "left" else (0, g), value=v)
        b = {"input_ids": torch.stack([pad(f["input_ids"], tok.pad_token_id) for f in features]),
             "attention_mask": torch.stack([pad(f["attention_mask"], 0) for f in features]),
             "labels": torch.stack([pad(f["labels"], -100) for f in features])}
        if "audio_batch_size" in features[0]:
            b["audio_batch_size"] = torch.stack([f["audio_batch_size"] for f in features])
        return b
    ref = ultravox_processing.DataCollatorForSeq2SeqWithAudio.__new__(ultravox_processing.DataCollatorForSeq2SeqWithAudio)
    ref.tokenizer, ref.include_alt_fields = tok, alt
    clone = lambda fs: [{k: (v.clone() if hasattr(v, "clone") else list(v) if isinstance(v, list) else v) for k, v in f.items()} for f in fs]
    with mock.patch.object(transformers.DataCollatorForSeq2Seq, "__call__", lambda self, features, *a, **k: hf_pad(features)):
        want = ref(clone(feats))
    got = DataCollatorForSeq2SeqWithAudio(tok, include_alt_fields=alt)(clone(feats))
    assert set(got) == set(want), (sorted(got), sorted(want))
    for k in want:
        assert got[k].dtype == want[k].dtype and got[k].shape == want[k].shape and torch.equal(got[k], want[k]), (k, side, alt, picks)
