"""Host index arithmetic (UltravoxProcessor / collator mirror) replayed against fixtures produced by the
REFERENCE implementation (tests/golden/processor.json, made by tests/golden/make_golden.py) — bit-exact.
The mel values come from the oracle here (no GPU); only integer outputs and shapes are asserted."""
import json
import os

import numpy as np
import pytest
import torch

from fake_tokenizer import FakeTokenizer
from oracle.reference_cpu import FeatureExtractorRef
from ultravox_amd.processing import DataCollatorForSeq2SeqWithAudio, UltravoxProcessor

SR = 16000


@pytest.fixture(scope="module")
def golden(golden_dir):
    return json.load(open(os.path.join(golden_dir, "processor.json")))


def make_proc(side="right"):
    tok = FakeTokenizer(padding_side=side)
    return UltravoxProcessor(FeatureExtractorRef(80), tokenizer=tok)


def clips():
    rng = np.random.RandomState(0)
    return {"short": rng.randn(SR).astype(np.float32), "long": rng.randn(SR * 10).astype(np.float32),
            "overflow": rng.randn(SR * 35).astype(np.float32), "exact30": rng.randn(SR * 30).astype(np.float32),
            "s61": rng.randn(SR * 61).astype(np.float32)}


CASE_AUDIO = {"text_only": [], "single": ["short"], "overflow": ["overflow"], "two": ["short", "long"],
              "three_overflow": ["short", "overflow", "long"], "exact30": ["exact30"], "s61": ["s61"],
              "trailing_text": ["long"]}


def test_processor_matches_reference_vectors(golden):
    proc, cl = make_proc(), clips()
    for case in golden["cases"]:
        names = CASE_AUDIO[case["name"]]
        kw = dict(audios=[cl[n] for n in names], sampling_rate=SR, include_audio_num_chunks=True) if names else {}
        r = proc(case["text"], **kw)
        for k in ["audio_lens", "audio_token_len", "audio_token_start_idx", "input_ids", "attention_mask",
                  "audio_batch_size", "audio_num_chunks"]:
            if k in case:
                assert r[k].tolist() == case[k], (case["name"], k)
            else:
                assert k not in r
        if "audio_values_shape" in case:
            assert list(r["audio_values"].shape) == case["audio_values_shape"]
            assert r["audio_token_len"].dtype == torch.int32 and r["audio_lens"].dtype == torch.int64


def test_reference_test_literals():
    # the literal expectations of ultravox_processing_test.py:46-137 and infer_test.py:72-86
    proc, cl = make_proc(), clips()
    r = proc("Test with <|audio|>", audio=cl["short"], sampling_rate=SR)
    assert (r.audio_lens.tolist(), r.audio_token_len.tolist(), r.audio_token_start_idx.tolist()) == ([100], [7], [3])
    eos = proc.vocab[proc.audio_token_replacement]
    assert r.input_ids[0, 3:].tolist() == [eos] * 7 and r.audio_batch_size.tolist() == [1]
    r = proc("Test with <|audio|>", audios=[cl["overflow"]], sampling_rate=SR)
    assert r.audio_lens.tolist() == [3000, 500] and r.audio_token_len.tolist() == [188, 32]
    assert r.audio_token_start_idx.tolist() == [3, 3 + 188] and r.audio_batch_size.tolist() == [2]
    r = proc("Test with <|audio|> and <|audio|> and <|audio|>", audios=[cl["short"], cl["overflow"], cl["long"]],
             sampling_rate=SR, include_audio_num_chunks=True)
    assert r.audio_token_start_idx.tolist() == [3, 12, 200, 234] and r.audio_num_chunks.tolist() == [1, 2, 1]
    r = proc("12345678<|audio|>".replace("12345678", "a b c d e f g h"), audio=np.zeros(SR * 60, np.float32) + 0.1,
             sampling_rate=SR)
    assert tuple(r.audio_values.shape) == (2, 80, 3000) and r.audio_token_len.tolist() == [188, 188]


def test_tiny_lengths(golden):
    proc = make_proc()
    for rec in golden["tiny"]:
        r = proc("<|audio|>", audio=np.zeros(rec["n"], np.float32) + 0.01, sampling_rate=SR)
        assert r.audio_lens.tolist() == rec["audio_lens"]
        assert r.audio_values.shape[-1] == rec["frames"] == r.audio_lens.item()
        assert r.audio_token_len.tolist() == rec["audio_token_len"]


def test_errors(golden):
    proc, cl = make_proc(), clips()
    names = {1: ["short"], 2: ["short", "long"]}
    for rec in golden["errors"]:
        with pytest.raises(ValueError) as e:
            proc(rec["text"], audios=[cl[n] for n in names[rec["n_audio"]]], sampling_rate=SR)
        assert str(e.value) == rec["error"]
    with pytest.raises(ValueError, match="Only one of `audio` or `audios`"):
        proc("<|audio|>", audio=cl["short"], audios=[cl["short"]], sampling_rate=SR)
    with pytest.raises(ValueError, match="Text must be a string"):
        proc(["a", "b"])
    with pytest.raises(ValueError, match=r"too few audio placeholders. \(Expected 2 placeholders\)"):
        proc("no placeholder at all", audios=[cl["short"], cl["long"]], sampling_rate=SR)


def test_collator_right_and_left_padding(golden):
    cl = clips()
    for side in ("right", "left"):
        proc = make_proc(side)
        feats = []
        for text, names in [("Test with <|audio|>", ["short"]),
                            ("A much longer prompt with <|audio|> and <|audio|> ok", ["long", "short"]),
                            ("text only sample here", [])]:
            kw = dict(audios=[cl[n] for n in names], sampling_rate=SR) if names else {}
            r = proc(text, **kw)
            f = {k: (v[0] if k in ("input_ids", "attention_mask") else v) for k, v in r.items()}
            f["labels"] = f["input_ids"].clone()
            if "audio_batch_size" not in f:
                f["audio_batch_size"] = torch.tensor([0])
            feats.append(f)
        batch = DataCollatorForSeq2SeqWithAudio(proc.tokenizer)(feats)
        ref = golden["collator"][side]
        for k, v in ref.items():
            if k == "audio_values":
                assert list(batch[k].shape) == v
            else:
                assert batch[k].tolist() == v, (side, k)


def test_raw_waveform_tower_branch_of_the_processor():
    """BASELINE config 5 (wav2vec2): the `input_values` fallback (ultravox_processing.py:308).  The reference cannot execute it
    (feature_extractor.hop_length, :284), so this pins OUR contract: one un-chunked item per audio, audio_lens = encoder frames,
    audio_token_len = ceil(frames / stack_factor) = the projector's rows, placeholders expanded and collated as for Whisper."""
    import numpy as np
    from fake_tokenizer import FakeTokenizer
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.frontend import Wav2Vec2FeatureExtractor
    from ultravox_amd.processing import DataCollatorForSeq2SeqWithAudio, UltravoxProcessor
    cfg = UltravoxConfig(audio_model_id="facebook/wav2vec2-large-960h", text_model_id="google/gemma-7b")
    tok = FakeTokenizer()
    proc = UltravoxProcessor(Wav2Vec2FeatureExtractor(), tok, stack_factor=8, audio_frames_fn=cfg.audio_config.feat_extract_output_length)
    rng = np.random.RandomState(0)
    a30, a1 = rng.randn(480000).astype(np.float32), rng.randn(16000).astype(np.float32)
    out = proc(text="Listen <|audio|> and <|audio|> done", audios=[a30, a1], sampling_rate=16000, include_audio_num_chunks=True)
    assert tuple(out["audio_values"].shape) == (2, 480000) and out["audio_values"].dtype == torch.float32
    assert out["audio_lens"].tolist() == [1499, 49] and out["audio_token_len"].tolist() == [188, 7]
    assert out["audio_batch_size"].tolist() == [2] and out["audio_num_chunks"].tolist() == [1, 1]
    ids = out["input_ids"][0].tolist()
    s0, s1 = out["audio_token_start_idx"].tolist()
    eos = tok.eos_token_id
    assert ids[s0:s0 + 188] == [eos] * 188 and ids[s1:s1 + 7] == [eos] * 7 and s1 > s0 + 188
    assert abs(float(out["audio_values"][1, :16000].mean())) < 1e-5 and float(out["audio_values"][1, 16000:].abs().max()) == 0.0
    with pytest.raises(ValueError, match="audio placeholders"):
        proc(text="only one <|audio|>", audios=[a1, a1], sampling_rate=16000)
    with pytest.raises(ValueError, match="receptive field"):
        proc(text="<|audio|>", audios=[a1[:300]], sampling_rate=16000)
    # collation: the waveforms are right-padded to the longest one
    one = proc(text="<|audio|> hi", audios=[a1], sampling_rate=16000)
    feats = [{k: (v[0] if k in ("input_ids", "attention_mask") else v) for k, v in f.items()} for f in (out, one)]
    for f in feats:
        f.pop("audio_num_chunks", None)
    batch = DataCollatorForSeq2SeqWithAudio(tok)(feats)
    assert tuple(batch["audio_values"].shape) == (3, 480000) and batch["audio_lens"].tolist() == [1499, 49, 49]
    assert batch["audio_batch_size"].reshape(-1).tolist() == [2, 1]
