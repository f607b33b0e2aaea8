"""LocalInference host logic (the reference's ultravox/inference/infer.py, exercised like infer_test.py:36-180 does: the
model's generate() is a stub that always answers five tokens, everything else — chat template, processor, terminators,
streaming, conversation bookkeeping — is the real code)."""
import numpy as np
import pytest
import torch

from fake_tokenizer import FakeChatTokenizer
from oracle.reference_cpu import FeatureExtractorRef
from ultravox_amd.inference import InferenceChunk, InferenceStats, LocalInference, VoiceSample
from ultravox_amd.processing import UltravoxProcessor


class StubModel:
    device = torch.device("cpu")
    dtype = torch.float32

    def __init__(self, tok, reply="the answer is 42 ."):
        self.tok, self.reply, self.calls = tok, reply, []

    def generate(self, **kw):
        self.calls.append(kw)
        ids = kw["input_ids"]
        new = self.tok._tok(self.reply) + [kw["eos_token_id"][0]]
        new = new[: kw["max_new_tokens"]]
        streamer = kw.get("streamer")
        if streamer is not None:
            streamer.put(ids.cpu())
            for t in new:
                streamer.put(torch.tensor([t]))
            streamer.end()
        return torch.cat([ids, torch.tensor([new] * ids.shape[0])], dim=1)


def make(conversation_mode=False, **kw):
    tok = FakeChatTokenizer()
    proc = UltravoxProcessor(FeatureExtractorRef(80), tokenizer=tok)
    model = StubModel(tok)
    return LocalInference(model, proc, tok, dtype=torch.float32, conversation_mode=conversation_mode, **kw), model


def test_infer_with_audio_builds_the_reference_inputs():
    inf, model = make()
    out = inf.infer(VoiceSample.from_prompt_and_raw("Transcribe\n<|audio|>", np.ones(16000, dtype=np.float32), 16000))
    assert out.text == "the answer is 42 ." and out.output_tokens == 6
    kw = model.calls[0]
    assert kw["audio_values"].shape == (1, 80, 100)                     # infer_test.py:98-99
    assert kw["audio_token_len"].item() == 7                            # 1 s -> 100 frames -> 50 -> ceil(50 / 8)
    s = kw["audio_token_start_idx"].item()
    assert torch.all(kw["input_ids"][0, s:s + 7] == inf.tokenizer.eos_token_id)
    assert out.input_tokens == kw["input_ids"].shape[1]
    assert kw["do_sample"] is False and kw["eos_token_id"] == [128009] and kw["pad_token_id"] == 128009


def test_resampling_and_int16_and_temperature():
    inf, model = make()
    pcm = (np.ones(48000) * 1000).astype(np.int16)
    inf.infer(VoiceSample.from_prompt_and_raw("<|audio|>", pcm, 48000), max_tokens=3, temperature=0.7)
    kw = model.calls[0]
    assert kw["audio_values"].shape == (1, 80, 100) and kw["audio_token_len"].item() == 7     # infer_test.py:112-117
    assert kw["do_sample"] is True and kw["temperature"] == 0.7 and kw["max_new_tokens"] == 3
    with pytest.raises(ValueError):
        inf.infer(VoiceSample.from_prompt_and_raw("<|audio|>", np.ones(100, dtype=np.int32), 16000))


def test_text_only_has_no_audio_arguments():
    inf, model = make()
    out = inf.infer(VoiceSample.from_prompt("Hello?"))
    kw = model.calls[0]
    assert kw.get("audio_values") is None and kw.get("audio_token_len") is None and out.output_tokens == 6


def test_stream_yields_chunks_then_stats():
    inf, model = make()
    msgs = list(inf.infer_stream(VoiceSample.from_prompt_and_raw("<|audio|>", np.ones(16000, dtype=np.float32), 16000)))
    text = "".join(m.text for m in msgs if isinstance(m, InferenceChunk))
    stats = [m for m in msgs if isinstance(m, InferenceStats)]
    assert text == "the answer is 42 ." and len(stats) == 1 and isinstance(msgs[-1], InferenceStats)
    assert stats[0].output_tokens == 6 and stats[0].input_tokens == model.calls[0]["input_ids"].shape[1]


def test_stream_surfaces_a_failing_generate():
    inf, model = make()
    model.generate = lambda **kw: (_ for _ in ()).throw(RuntimeError("boom"))
    with pytest.raises(RuntimeError, match="boom"):
        list(inf.infer_stream(VoiceSample.from_prompt("Hi")))


def test_conversation_mode_carries_the_dialogue():
    inf, model = make(conversation_mode=True)
    inf.infer(VoiceSample.from_prompt_and_raw("Listen: <|audio|>", np.ones(16000, dtype=np.float32), 16000))
    assert [m["role"] for m in inf.past_messages] == ["user", "assistant"]
    assert inf.past_messages[0]["content"] == "Listen: " + inf.tokenizer.eos_token * 7     # infer.py:80-89
    assert inf.past_messages[1]["content"] == "the answer is 42 ."
    first_len = model.calls[0]["input_ids"].shape[1]
    inf.infer(VoiceSample.from_prompt("And then?"))
    second = model.calls[1]
    assert second.get("audio_values") is None                           # the old audio turn is now plain eos tokens
    assert second["input_ids"].shape[1] > first_len and len(inf.past_messages) == 4
    inf.infer()                                                         # answer again without a new user turn
    assert len(inf.past_messages) == 5
    inf.update_conversation([])
    with pytest.raises(ValueError):
        inf.infer()


def test_conversation_mode_threads_the_kv_state_through_generate():
    import types
    inf, model = make(conversation_mode=True)
    plain = model.generate
    model.generate = lambda **kw: types.SimpleNamespace(sequences=plain(**kw), past_key_values=("state", len(model.calls)))
    inf.infer(VoiceSample.from_prompt("One"))
    assert model.calls[0]["past_key_values"] is None and model.calls[0]["return_dict_in_generate"] is True
    assert inf.past_key_values == ("state", 1)
    list(inf.infer_stream(VoiceSample.from_prompt("Two")))
    assert model.calls[1]["past_key_values"] == ("state", 1) and inf.past_key_values == ("state", 2)
    inf.update_conversation([])
    assert inf.past_key_values is None
    single, m2 = make()
    single.infer(VoiceSample.from_prompt("One"))
    assert "past_key_values" not in m2.calls[0]


def test_thinking_content_is_split_off():
    inf, model = make(enable_thinking=True, thinking_regex=r"<think>(.*?)</think>")
    model.reply = "<think> hmm </think> yes"
    out = inf.infer(VoiceSample.from_prompt("Q"))
    assert out.text == "yes" and out.thinking_content == "hmm"
    bad, m2 = make(enable_thinking=True)
    with pytest.raises(ValueError):
        bad.infer(VoiceSample.from_prompt("Q"))


def test_batch_left_pads_and_strips_after_the_first_terminator():
    inf, model = make()
    outs = inf.infer_batch([VoiceSample.from_prompt("Hi"), VoiceSample.from_prompt("A much longer question here ?")])
    kw = model.calls[0]
    assert kw["input_ids"].shape[0] == 2 and kw["attention_mask"][0, 0].item() == 0 and kw["attention_mask"][1, 0].item() == 1
    assert [o.text for o in outs] == ["the answer is 42 ."] * 2 and outs[0].output_tokens == 6
