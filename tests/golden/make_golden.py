"""Generates tests/golden/*.json|*.npz by IMPORTING THE REFERENCE from /root/reference (build container
only; the fixtures travel, the reference does not).  Run:  python tests/golden/make_golden.py

What is recorded
  processor.json  — UltravoxProcessor / DataCollatorForSeq2SeqWithAudio integer outputs of the REFERENCE
                    implementation (ultravox/model/ultravox_processing.py) for the cases its own tests use
                    (ultravox_processing_test.py:46-229, infer_test.py:72-109), with the HF
                    WhisperFeatureExtractor and tests/fake_tokenizer.py.
  projector_*.npz — outputs AND weight/input gradients of the REFERENCE UltravoxProjector
                    (ultravox_model.py:745-800, imported with an in-memory `peft` stub) for seeded inputs,
                    both projector_ln_mid variants, T not a multiple of the stack factor.
  projector_act.npz — the same for projector_act = gelu / silu / relu / gelu_pytorch_tanh (:754-755: the width is kept, linear_2 [D, hidden]).
  latency_mask.npz — ModifiedWhisperEncoder.init_latency_mask (ultravox_model.py:834-863).
  kl_loss.npz     — the REFERENCE UltravoxModel._get_prediction_mask / _compute_kl_loss (ultravox_model.py:157-256) called
                    unbound on a stub `self` whose language model returns recorded teacher logits: masks, loss and
                    d loss / d student logits for seeded logits, several eot weights / temperatures / ragged labels.
  diff_state_dict.json — key sets kept by the REFERENCE UltravoxModel.diff_state_dict (ultravox_model.py:565-584) for stub
                    models (trainable / frozen / keep_params / FSDP-wrapped names).
  dataproc.json   — the REFERENCE UltravoxDataproc._process (ultravox_data_proc.py:46-154, imported with a stub `ultravox.data`
                    package: its real one needs librosa / soundfile) over the reference processor and FakeChatTokenizer:
                    input_ids / labels / alt_* for every loss-mask type, alt fields, response truncation, inference mode.
  logmel.npz      — HF WhisperFeatureExtractor (the [3P] K1 arithmetic) on seeded PCM, 80 and 128 mels.
  forward_reference.npz / .json — the REFERENCE UltravoxModel.forward + loss.backward() end to end (ultravox_model.py:277-396:
                    _prepare_audio_embeds, _audio_iter, projector, merge loop, HF Llama + ForCausalLMLoss) on tiny random towers,
                    the audio tower stubbed by recorded hidden states; both projector variants, two / three items in one
                    sample, a text-only sample in the batch, left and right padding: logits, loss, projector gradients.
  kl_forward_reference.npz / .json — the REFERENCE forward in training mode under LossFunction.KL_Divergence (teacher pass +
                    _compute_kl_loss inside the model) on the same seeded tiny model: loss and projector gradients, two settings.
  lora_forward_reference.npz / .json — the REFERENCE model with text_model_lora_config r = 4 (apply_lora via tests/peft_stub.py),
                    forward + backward with non-zero adapters: loss, logits, projector and adapter gradients.
  kl_lora_forward_reference.npz / .json — the same model in training mode under KL_Divergence: teacher and student both run the
                    ADAPTED language model (ultravox_model.py:212-222); loss, projector and adapter gradients.
  (No Gemma-backbone run of the reference model: the installed transformers 5.x moved Gemma's sqrt(hidden) scale into the
   embedding module, i.e. it no longer scales the MERGED inputs_embeds as the reference's pinned 4.51.3 does - a fixture
   generated here would pin the wrong semantics.  The Gemma blocks are pinned against HF with the scale applied explicitly.)
  generate_reference.json — the REFERENCE UltravoxModel.generate (greedy, HF GenerationMixin) on the same seeded tiny model:
                    new tokens for an unpadded and a left-padded prompt, without EOS and with an EOS that stops one row early.
  config.json     — the REFERENCE UltravoxConfig (ultravox_config.py:56-203) for keyword sets that need no network: every field the
                    hot path reads, the [3P] family defaults a partial sub-config dict resolves to, the to_diff_dict key set.
  lora_reference.npz / .json — the REFERENCE apply_lora (ultravox_model.py:690-709) run on an installed-HF WhisperEncoder and
                    LlamaForCausalLM with the reference's own LoraConfigSimplified defaults, through tests/peft_stub.py
                    (peft itself is not installable here: the stub restates peft 0.11.1's LoRA Linear and says so): adapted
                    module set, trainable names, state-dict key names, forward outputs and adapter gradients.
  lora_targets_reference.npz / .json — the same with target_modules beyond the default: q / k / v / out_proj (Whisper) and q / k / v /
                    o_proj (Llama, GQA), and a v + o-only list; peft's error text for a list that hits nothing.
  lora_w2v_reference.npz / .json — the REFERENCE apply_lora on an installed-HF Wav2Vec2Model (the AutoModel tower, both encoder families): adapted
                    modules and names, last_hidden_state, adapter gradients.
"""
import dataclasses
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
sys.path.insert(1, os.path.dirname(os.path.dirname(HERE)))      # repo root: ultravox_amd (weights / config only, no GPU)

import peft_stub  # noqa: E402  (tests/peft_stub.py: the three names the reference imports from peft, restated)

peft_stub.install()

import transformers  # noqa: E402
from fake_tokenizer import FakeTokenizer  # noqa: E402
from ultravox.model import ultravox_config, ultravox_model, ultravox_processing  # noqa: E402


def tolist(v):
    return v.tolist() if hasattr(v, "tolist") else v


def processor_cases():
    fe = transformers.WhisperFeatureExtractor()
    ap = types.SimpleNamespace(feature_extractor=fe, model_input_names=fe.model_input_names)
    ap_call = lambda *a, **k: fe(*a, **k)
    ap = type("AP", (), {"feature_extractor": fe, "model_input_names": fe.model_input_names,
                         "__call__": staticmethod(ap_call)})()
    tok = FakeTokenizer()
    proc = ultravox_processing.UltravoxProcessor.__new__(ultravox_processing.UltravoxProcessor)
    # bypass ProcessorMixin.__init__ type checks: set the fields the methods use
    proc.audio_padding, proc.encoder_ds_factor, proc.stack_factor = "longest", 2, 8
    proc.audio_placeholder, proc.audio_context_size = "<|audio|>", 3000
    proc.vocab = tok.get_vocab()
    proc.audio_token_replacement = tok.eos_token
    tok.pad_token_id = tok.eos_token_id
    proc.audio_processor, proc.tokenizer = ap, tok
    rng = np.random.RandomState(0)
    sr = 16000
    clips = {"short": rng.randn(sr).astype(np.float32), "long": rng.randn(sr * 10).astype(np.float32),
             "overflow": rng.randn(sr * 35).astype(np.float32), "exact30": rng.randn(sr * 30).astype(np.float32),
             "s61": rng.randn(sr * 61).astype(np.float32)}
    cases = [
        ("text_only", "Hello, how are you?", []),
        ("single", "Test with <|audio|>", ["short"]),
        ("overflow", "Test with <|audio|>", ["overflow"]),
        ("two", "Test with <|audio|> and <|audio|>", ["short", "long"]),
        ("three_overflow", "Test with <|audio|> and <|audio|> and <|audio|>", ["short", "overflow", "long"]),
        ("exact30", "A <|audio|> B", ["exact30"]),
        ("s61", "A <|audio|> B", ["s61"]),
        ("trailing_text", "x <|audio|> tail words here", ["long"]),
    ]
    out = {"cases": []}
    keys = ["audio_lens", "audio_token_len", "audio_token_start_idx", "input_ids", "attention_mask",
            "audio_batch_size", "audio_num_chunks"]
    for name, text, names in cases:
        kw = dict(audios=[clips[n] for n in names], sampling_rate=sr, include_audio_num_chunks=True) if names else {}
        r = proc(text, **kw)
        rec = {"name": name, "text": text, "seconds": [len(clips[n]) / sr for n in names]}
        for k in keys:
            if k in r:
                rec[k] = tolist(r[k])
        if "audio_values" in r:
            rec["audio_values_shape"] = list(r["audio_values"].shape)
        out["cases"].append(rec)
    # sub-2-hop edge lengths (ultravox_processing_test.py:177-186)
    out["tiny"] = []
    for n in [0, 1, 159, 160, 161, 319, 320, 321, 479, 480, 481]:
        r = proc("<|audio|>", audio=np.zeros(n, np.float32) + 0.01, sampling_rate=sr)
        out["tiny"].append({"n": n, "audio_lens": tolist(r["audio_lens"]), "frames": int(r["audio_values"].shape[-1]),
                            "audio_token_len": tolist(r["audio_token_len"])})
    # error cases (:140-174)
    errs = []
    for text, names in [("Test with <|audio|> and <|audio|>", ["short"]), ("Test with no placeholder", ["short"]),
                        ("Test <|audio|>", ["short", "long"])]:
        try:
            proc(text, audios=[clips[n] for n in names], sampling_rate=sr)
            errs.append({"text": text, "n_audio": len(names), "error": None})
        except ValueError as e:
            errs.append({"text": text, "n_audio": len(names), "error": str(e)})
    out["errors"] = errs
    # collator (:189-229 and the left-pad displacement :53-63)
    coll = {}
    for side in ("right", "left"):
        tk = FakeTokenizer(padding_side=side)
        tk.pad_token_id = tk.eos_token_id
        proc.tokenizer = tk
        feats = []
        for text, names in [("Test with <|audio|>", ["short"]), ("A much longer prompt with <|audio|> and <|audio|> ok", ["long", "short"]),
                            ("text only sample here", [])]:
            kw = dict(audios=[clips[n] for n in names], sampling_rate=sr) if names else {}
            r = proc(text, **kw)
            f = {k: (v[0] if k in ("input_ids", "attention_mask") else v) for k, v in r.items()}
            f["labels"] = f["input_ids"].clone()
            if "audio_batch_size" not in f:
                f["audio_batch_size"] = torch.tensor([0])
            feats.append(f)
        dc = ultravox_processing.DataCollatorForSeq2SeqWithAudio.__new__(ultravox_processing.DataCollatorForSeq2SeqWithAudio)
        # emulate the [3P] DataCollatorForSeq2Seq padding that super().__call__ performs
        def hf_pad(features, tk=tk):
            import torch.nn.functional as F
            n = max(len(f["input_ids"]) for f in features)
            def pad(x, v):
                g = n - len(x)
                return F.pad(torch.as_tensor(x), (g, 0) if tk.padding_side == "left" else (0, g), value=v)
            b = {"input_ids": torch.stack([pad(f["input_ids"], tk.pad_token_id) for f in features]),
                 "attention_mask": torch.stack([pad(f["attention_mask"], 0) for f in features]),
                 "labels": torch.stack([pad(f["labels"], -100) for f in features]),
                 "audio_batch_size": torch.stack([f["audio_batch_size"] for f in features])}
            return b
        dc.tokenizer, dc.include_alt_fields = tk, False
        # run the reference's own __call__ body with the HF padding stubbed in for super().__call__
        import unittest.mock as mock
        with mock.patch.object(transformers.DataCollatorForSeq2Seq, "__call__", lambda self, features, *a, **k: hf_pad(features)):
            batch = dc([dict(f) for f in feats])
        coll[side] = {k: (tolist(v) if k != "audio_values" else list(v.shape)) for k, v in batch.items()}
    out["collator"] = coll
    json.dump(out, open(os.path.join(HERE, "processor.json"), "w"), indent=0)
    print("processor.json:", [c["name"] for c in out["cases"]])


def projector_cases():
    for ln_mid in (True, False):
        cfg = ultravox_config.UltravoxConfig(
            audio_config={"model_type": "whisper", "d_model": 32, "encoder_layers": 1, "encoder_attention_heads": 2,
                          "encoder_ffn_dim": 64, "num_mel_bins": 80},
            text_config={"model_type": "llama", "hidden_size": 64, "intermediate_size": 64, "num_hidden_layers": 1,
                         "num_attention_heads": 2, "num_key_value_heads": 2, "vocab_size": 128},
            hidden_size=256, stack_factor=8, projector_ln_mid=ln_mid)
        torch.manual_seed(7)
        proj = ultravox_model.UltravoxProjector(cfg).float()
        with torch.no_grad():
            for p in proj.parameters():
                if p.dim() == 1:
                    p.mul_(1.0 + 0.2 * torch.randn_like(p))
        x = torch.randn(3, 21, 32, requires_grad=True)
        y = proj(x)
        gy = torch.randn_like(y)
        y.backward(gy)
        rec = {"x": x.detach().numpy(), "y": y.detach().numpy(), "gy": gy.numpy(), "gx": x.grad.numpy()}
        for k, v in proj.state_dict().items():
            rec["w." + k] = v.numpy()
        for k, v in proj.named_parameters():
            rec["g." + k] = v.grad.numpy()
        stacked = ultravox_model.StackAudioFrames(8)(x.detach())
        rec["stacked"] = stacked.numpy()
        np.savez_compressed(os.path.join(HERE, f"projector_ln_{'mid' if ln_mid else 'post'}.npz"), **rec)
    print("projector_*.npz written")


def projector_act_cases():
    """UltravoxProjector with projector_act != "swiglu" (ultravox_model.py:754-755: transformers.activations.get_activation, dim_mid = hidden
    for every activation but swiglu): outputs and gradients for the four fused-op activations the HIP path builds, both norm placements."""
    rec = {}
    for act, ln_mid in (("gelu", True), ("silu", False), ("relu", True), ("gelu_pytorch_tanh", False)):
        cfg = ultravox_config.UltravoxConfig(
            audio_config={"model_type": "whisper", "d_model": 32, "encoder_layers": 1, "encoder_attention_heads": 2,
                          "encoder_ffn_dim": 64, "num_mel_bins": 80},
            text_config={"model_type": "llama", "hidden_size": 64, "intermediate_size": 64, "num_hidden_layers": 1,
                         "num_attention_heads": 2, "num_key_value_heads": 2, "vocab_size": 128},
            hidden_size=128, stack_factor=8, projector_ln_mid=ln_mid, projector_act=act)
        torch.manual_seed(11)
        proj = ultravox_model.UltravoxProjector(cfg).float()
        assert proj.linear_2.weight.shape == (64, 128)                      # the width is kept (:755)
        with torch.no_grad():
            for p in proj.parameters():
                if p.dim() == 1:
                    p.mul_(1.0 + 0.2 * torch.randn_like(p))
        x = torch.randn(3, 21, 32, requires_grad=True)
        y = proj(x)
        gy = torch.randn_like(y)
        y.backward(gy)
        rec.update({f"{act}.x": x.detach().numpy(), f"{act}.y": y.detach().numpy(), f"{act}.gy": gy.numpy(), f"{act}.gx": x.grad.numpy(),
                    f"{act}.ln_mid": np.array(int(ln_mid))})
        for k, v in proj.state_dict().items():
            rec[f"{act}.w.{k}"] = v.numpy()
        for k, v in proj.named_parameters():
            rec[f"{act}.g.{k}"] = v.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "projector_act.npz"), **rec)
    print("projector_act.npz written:", sorted({k.split(".")[0] for k in rec}))


def latency_mask_cases():
    rec = {}
    for block in (100, 300, 1500):
        dummy = types.SimpleNamespace(max_context_length=3000)
        dummy.register_buffer = lambda name, t, persistent=False, d=dummy: setattr(d, name, t)
        ultravox_model.ModifiedWhisperEncoder.init_latency_mask(dummy, block, torch.float32)
        m = dummy.audio_streaming_mask[0, 0, :400, :400]
        rec[f"allowed_{block}"] = (m == 0).numpy()
    try:
        dummy = types.SimpleNamespace(max_context_length=3000)
        ultravox_model.ModifiedWhisperEncoder.init_latency_mask(dummy, 13, torch.float32)
        rec["err13"] = np.array(0)
    except AssertionError:
        rec["err13"] = np.array(1)
    np.savez_compressed(os.path.join(HERE, "latency_mask.npz"), **rec)
    print("latency_mask.npz written")


def logmel_cases():
    rec = {}
    g = torch.Generator().manual_seed(11)
    for n_mels in (80, 128):
        fe = transformers.WhisperFeatureExtractor(feature_size=n_mels)
        noise = (0.1 * torch.randn(2, 16000 * 2, generator=g)).clamp(-1, 1).numpy()
        t = np.arange(16000 * 2) / 16000.0
        tone = (0.5 * np.sin(2 * np.pi * 440 * t) + 0.05 * np.sin(2 * np.pi * 3000 * t)).astype(np.float32)
        tone[20000:] = 0.0  # trailing silence: exercises the clip-max floor
        pcm = np.stack([noise[0], noise[1] * 0.01, tone]).astype(np.float32)
        out = fe(list(pcm), sampling_rate=16000, padding="longest", pad_to_multiple_of=160, truncation=False,
                 return_attention_mask=True, return_tensors="np")
        rec[f"pcm_{n_mels}"] = pcm
        rec[f"mel_{n_mels}"] = out["input_features"].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "logmel.npz"), **rec)
    print("logmel.npz written")


def logmel_speech_cases():
    """HF WhisperFeatureExtractor on a 30 s harmonic, speech-like clip (tests/forward_fixture_util.py speech_like_pcm): every
    4th frame of the 80 x 3000 log-mel is kept (240 KB instead of 960 KB), plus the clip statistics the floor depends on."""
    import forward_fixture_util as U
    pcm = U.speech_like_pcm()
    rec = {}
    for n_mels in (80, 128):
        fe = transformers.WhisperFeatureExtractor(feature_size=n_mels)
        out = fe([pcm], sampling_rate=16000, padding="longest", pad_to_multiple_of=160, truncation=False,
                 return_attention_mask=True, return_tensors="np")["input_features"][0].astype(np.float32)
        assert out.shape == (n_mels, 3000)
        rec[f"mel_{n_mels}_every4"] = out[:, ::4]
        rec[f"max_{n_mels}"] = np.float32(out.max())
        rec[f"n_floor_{n_mels}"] = np.int64((out == out.min()).sum())
    rec["pcm_checksum"] = np.float64(np.abs(pcm.astype(np.float64)).sum())
    np.savez_compressed(os.path.join(HERE, "logmel_speech.npz"), **rec)
    print("logmel_speech.npz written; floor fraction", float(rec["n_floor_80"]) / (80 * 3000))


def kl_cases():
    """Reference KL loss on seeded logits.  `self` is a stub: get_input_embeddings().forward -> passthrough,
    language_model.forward -> the recorded teacher logits."""
    M = ultravox_model.UltravoxModel
    out = {}
    g = torch.Generator().manual_seed(7)
    V = 64
    cases = [
        # (name, student label spans per row (start, end), teacher spans, T_student, T_teacher, temperature, eot_w)
        ("basic", [(10, 16), (8, 14)], [(5, 11), (3, 9)], 16, 11, 2.0, 1.0),
        ("no_eot", [(10, 16), (8, 14)], [(5, 11), (3, 9)], 16, 11, 2.0, 0.0),
        ("temp1_w05", [(4, 9), (6, 12), (2, 5)], [(2, 7), (1, 7), (3, 6)], 12, 9, 1.0, 0.5),
        ("one_empty_row", [(10, 16), None, (9, 12)], [(5, 11), None, (4, 7)], 16, 11, 3.0, 1.0),
        ("padded_tail", [(6, 10), (3, 12)], [(2, 6), (1, 10)], 14, 12, 2.0, 2.0),
    ]
    for name, sp, tp, Ts, Tt, temp, w in cases:
        B = len(sp)
        labels = torch.full((B, Ts), -100, dtype=torch.long)
        alt_labels = torch.full((B, Tt), -100, dtype=torch.long)
        for b in range(B):
            if sp[b] is not None:
                labels[b, sp[b][0]:sp[b][1]] = torch.randint(0, V, (sp[b][1] - sp[b][0],), generator=g)
                alt_labels[b, tp[b][0]:tp[b][1]] = labels[b, sp[b][0]:sp[b][1]]
        student = (2.0 * torch.randn(B, Ts, V, generator=g)).requires_grad_(True)
        teacher = 2.0 * torch.randn(B, Tt, V, generator=g)
        stub = types.SimpleNamespace()
        stub.get_input_embeddings = lambda: types.SimpleNamespace(forward=lambda ids: ids)
        stub.language_model = types.SimpleNamespace(forward=lambda **kw: types.SimpleNamespace(logits=teacher))
        stub.loss_config = ultravox_config.LossConfig(loss_function=ultravox_config.LossFunction.KL_Divergence,
                                                      kl_temperature=temp, eot_loss_weight=w)
        stub._get_prediction_mask = lambda l: M._get_prediction_mask(stub, l)
        loss = M._compute_kl_loss(stub, lm_output=types.SimpleNamespace(logits=student), labels=labels,
                                  alt_input_ids=torch.zeros(B, Tt, dtype=torch.long),
                                  alt_attention_mask=torch.ones(B, Tt, dtype=torch.long), alt_labels=alt_labels)
        loss.backward()
        pm, em = M._get_prediction_mask(stub, labels)
        for k, v in (("labels", labels), ("alt_labels", alt_labels), ("student", student.detach()), ("teacher", teacher),
                     ("loss", loss.detach()), ("dstudent", student.grad), ("pred_mask", pm), ("eot_mask", em),
                     ("temperature", torch.tensor(temp)), ("eot_loss_weight", torch.tensor(w))):
            out[f"{name}.{k}"] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "kl_loss.npz"), **out)
    print("kl_loss.npz:", [c[0] for c in cases])


def _reference_processor(tok):
    fe = transformers.WhisperFeatureExtractor()
    ap = type("AP", (), {"feature_extractor": fe, "model_input_names": fe.model_input_names,
                         "__call__": staticmethod(lambda *a, **k: fe(*a, **k))})()
    proc = ultravox_processing.UltravoxProcessor.__new__(ultravox_processing.UltravoxProcessor)
    proc.audio_padding, proc.encoder_ds_factor, proc.stack_factor = "longest", 2, 8
    proc.audio_placeholder, proc.audio_context_size = "<|audio|>", 3000
    proc.vocab = tok.get_vocab()
    proc.audio_token_replacement = tok.eos_token
    tok.pad_token_id = tok.eos_token_id
    proc.audio_processor, proc.tokenizer = ap, tok
    return proc


DATAPROC_SAMPLES = {
    "asr": dict(messages=[{"role": "user", "content": "Transcribe\n<|audio|>"}, {"role": "assistant", "content": "the quick brown fox jumps"}],
                seconds=1.0, transcript="the quick brown fox jumps"),
    "system_qa": dict(messages=[{"role": "system", "content": "You are helpful ."}, {"role": "user", "content": "Listen to <|audio|> and answer briefly"},
                                {"role": "assistant", "content": "it says hello world again and again and again"}],
                      seconds=3.5, transcript="hello world"),
    "overflow": dict(messages=[{"role": "user", "content": "<|audio|>"}, {"role": "assistant", "content": "a long recording indeed"}],
                     seconds=35.0, transcript="thirty five seconds of speech"),
    "text_only": dict(messages=[{"role": "user", "content": "What is two plus two ?"}, {"role": "assistant", "content": "four"}],
                      seconds=None, transcript=None),
    "no_transcript": dict(messages=[{"role": "user", "content": "Hear <|audio|> now"}, {"role": "assistant", "content": "ok"}],
                          seconds=0.5, transcript=None),
}
DATAPROC_CONFIGS = [
    dict(loss_mask_type="last_assistant"), dict(loss_mask_type="after_audio"), dict(loss_mask_type="all"),
    dict(loss_mask_type="last_assistant", include_alt_fields=True),
    dict(loss_mask_type="after_audio", include_alt_fields=True, max_response_tokens=3),
    dict(loss_mask_type="last_assistant", max_response_tokens=2),
    dict(loss_mask_type="last_assistant", inference_mode=True),      # raises with audio: the mask text loses its placeholder
    dict(loss_mask_type="after_audio", inference_mode=True),
    dict(loss_mask_type="all", inference_mode=True, include_alt_fields=True),
]


def dataproc_sample(name):
    d = DATAPROC_SAMPLES[name]
    audio = None
    if d["seconds"] is not None:
        audio = np.random.RandomState(len(name)).randn(int(16000 * d["seconds"])).astype(np.float32)
    return types.SimpleNamespace(messages=[dict(m) for m in d["messages"]], audio=audio, sample_rate=16000,
                                 audio_transcript=d["transcript"])


def dataproc_cases():
    import ultravox
    stub = types.ModuleType("ultravox.data")      # the real package imports librosa / soundfile; only these names are used
    stub.Dataproc = type("Dataproc", (), {"__init__": lambda self, dataset: setattr(self, "_dataset", dataset)})
    stub.SizedIterableDataset = stub.VoiceSample = stub.Augmentation = object
    sys.modules["ultravox.data"] = ultravox.data = stub
    from fake_tokenizer import FakeChatTokenizer
    from ultravox.model import ultravox_data_proc
    out = []
    for cfg in DATAPROC_CONFIGS:
        for name in DATAPROC_SAMPLES:
            tok = FakeChatTokenizer(padding_side="right")
            kw = dict(cfg)
            kw["loss_mask_type"] = ultravox_config.LossMaskType(kw["loss_mask_type"])
            dp = ultravox_data_proc.UltravoxDataproc([], _reference_processor(tok), **kw)
            try:
                r = dp._process(dataproc_sample(name))
            except ValueError as e:      # e.g. AFTER_AUDIO on a text-only sample: recorded, the mirror must raise the same
                out.append({"config": cfg, "sample": name, "error": str(e)})
                continue
            rec = {"config": cfg, "sample": name}
            for k, v in r.items():
                if k == "audio_values":
                    rec["audio_values_shape"] = list(v.shape)
                else:
                    rec[k] = tolist(v)
            out.append(rec)
    json.dump({"cases": out}, open(os.path.join(HERE, "dataproc.json"), "w"), indent=0)
    print("dataproc.json", len(out), "cases")


def diff_state_dict_cases():
    M = ultravox_model.UltravoxModel
    cases = []
    defs = [
        ("projector_only", ["multi_modal_projector.ln_pre.weight", "multi_modal_projector.linear_1.weight"], [],
         ["audio_tower.conv1.weight", "language_model.model.norm.weight"]),
        ("with_keep", ["multi_modal_projector.linear_2.weight"], ["audio_tower.conv1.weight"],
         ["audio_tower.conv1.weight", "audio_tower.conv2.weight", "language_model.lm_head.weight"]),
        ("fsdp_names", ["audio_tower.base_model.model.layers.0._fsdp_wrapped_module.self_attn.k_proj.lora_B.default.weight"], [],
         ["audio_tower.base_model.model.layers.0.self_attn.k_proj.lora_B.default.weight", "audio_tower.conv1.weight"]),
    ]
    for name, trainable, keep, frozen in defs:
        params = [(k, types.SimpleNamespace(requires_grad=True)) for k in trainable] + \
                 [(k, types.SimpleNamespace(requires_grad=False)) for k in frozen]
        stub = types.SimpleNamespace(named_parameters=lambda params=params: params, keep_params=set(keep))
        # the state dict carries the NORMALISED names (what FSDP's full state dict returns)
        sd = {k.replace("_fsdp_wrapped_module.", ""): 0 for k, _ in params}
        kept = sorted(M.diff_state_dict(stub, sd).keys())
        cases.append({"name": name, "trainable": trainable, "keep_params": keep, "state_dict_keys": sorted(sd), "kept": kept})
    with open(os.path.join(HERE, "diff_state_dict.json"), "w") as f:
        json.dump(cases, f, indent=1)
    print("diff_state_dict.json:", [c["name"] for c in cases])


def lora_cases():
    """apply_lora on HF towers -> what is adapted, what trains, under which names, and what comes out."""
    from transformers import LlamaConfig, LlamaForCausalLM, WhisperConfig
    from transformers.models.whisper.modeling_whisper import WhisperEncoder
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.weights import random_state_dict
    tiny = dict(audio_config=dict(d_model=64, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=128, num_mel_bins=80,
                                  max_source_positions=1500),
                text_config=dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                                 num_key_value_heads=2, vocab_size=256, rope_theta=10000.0, max_position_embeddings=512,
                                 rms_norm_eps=1e-5),
                hidden_size=128, stack_factor=8, projector_ln_mid=True)
    cfg = UltravoxConfig(**tiny)
    a, t = cfg.audio_config, cfg.text_config
    sd = random_state_dict(cfg, seed=17)
    # the reference's own defaults for everything but r: LoraConfigSimplified (ultravox_config.py:8-23)
    simp = ultravox_config.LoraConfigSimplified
    arrays, meta = {}, {"tiny": tiny, "seed": 17}

    def lora_dict(r):
        d = dataclasses.asdict(simp(r=r))
        meta.setdefault("lora_config", {})[str(r)] = {k: v for k, v in d.items()}
        return d

    def randomise_b(model, seed):          # peft starts lora_B at zero: give it values so that the forward depends on it
        g = torch.Generator().manual_seed(seed)
        for n, p in model.named_parameters():
            if "lora_B" in n:
                p.data = 0.05 * torch.randn(p.shape, generator=g)

    # ---- encoder ----
    enc = WhisperEncoder(WhisperConfig(d_model=a.d_model, encoder_layers=a.encoder_layers, encoder_attention_heads=a.encoder_attention_heads,
                                       encoder_ffn_dim=a.encoder_ffn_dim, num_mel_bins=a.num_mel_bins,
                                       max_source_positions=a.max_source_positions, attn_implementation="eager")).eval()
    enc.load_state_dict({k[len("audio_tower."):]: v for k, v in sd.items() if k.startswith("audio_tower.")}, strict=False)
    wrapped = ultravox_model.apply_lora(enc, lora_dict(4))
    randomise_b(wrapped, 1)
    names = [n for n, p in wrapped.named_parameters() if p.requires_grad]
    meta["encoder"] = {"trainable": names, "state_dict_keys": sorted(wrapped.state_dict().keys()),
                       "adapted": sorted({n.split(".lora_")[0] for n in names})}
    torch.manual_seed(0)
    x, audio_len = torch.randn(2, 80, 120), torch.tensor([120, 75])
    inner = wrapped.base_model.model
    h = torch.nn.functional.gelu(inner.conv1(x))
    h = torch.nn.functional.gelu(inner.conv2(h)).permute(0, 2, 1)
    h = h + inner.embed_positions.weight[: h.size(-2)]
    keep = torch.arange(h.shape[1])[None, :].lt(((audio_len - 1) // 2 + 1).view(-1, 1))
    mask = (1.0 - keep[:, None, None, :].float()) * torch.finfo(torch.float32).min
    for layer in inner.layers:
        out = layer(h, mask)
        h = out[0] if isinstance(out, tuple) else out
    y = inner.layer_norm(h)
    gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(2))
    (y * gy).sum().backward()
    arrays.update({"enc.x": x.numpy(), "enc.audio_len": audio_len.numpy(), "enc.y": y.detach().numpy(), "enc.gy": gy.numpy()})
    for n, p in wrapped.named_parameters():
        if p.requires_grad:
            arrays["enc.w." + n], arrays["enc.g." + n] = p.detach().numpy(), p.grad.numpy()

    # ---- language model ----
    llm = LlamaForCausalLM(LlamaConfig(hidden_size=t.hidden_size, intermediate_size=t.intermediate_size, num_hidden_layers=t.num_hidden_layers,
                                       num_attention_heads=t.num_attention_heads, num_key_value_heads=t.num_key_value_heads,
                                       vocab_size=t.vocab_size, rms_norm_eps=t.rms_norm_eps, rope_theta=t.rope_theta,
                                       max_position_embeddings=t.max_position_embeddings, tie_word_embeddings=False,
                                       attn_implementation="eager")).eval()
    llm.load_state_dict({k[len("language_model."):]: v for k, v in sd.items() if k.startswith("language_model.")}, strict=False)
    wl = ultravox_model.apply_lora(llm, lora_dict(2))
    randomise_b(wl, 3)
    names = [n for n, p in wl.named_parameters() if p.requires_grad]
    meta["llm"] = {"trainable": names, "adapted": sorted({n.split(".lora_")[0] for n in names}),
                   "n_state_dict_keys": len(wl.state_dict())}
    emb = 0.5 * torch.randn(2, 21, t.hidden_size, generator=torch.Generator().manual_seed(4))
    am = torch.ones(2, 21, dtype=torch.long)
    am[1, -5:] = 0
    logits = wl(inputs_embeds=emb, attention_mask=am).logits
    gl = torch.randn(logits.shape, generator=torch.Generator().manual_seed(5)) * am[..., None]
    (logits * gl).sum().backward()
    arrays.update({"llm.emb": emb.numpy(), "llm.mask": am.numpy(), "llm.logits": logits.detach().numpy(), "llm.gl": gl.numpy()})
    for n, p in wl.named_parameters():
        if p.requires_grad:
            arrays["llm.w." + n], arrays["llm.g." + n] = p.detach().numpy(), p.grad.numpy()
    # r = 0: apply_lora freezes everything (no peft call)
    frozen = ultravox_model.apply_lora(torch.nn.Linear(3, 3), dataclasses.asdict(simp(r=0)))
    meta["r0_trainable"] = [n for n, p in frozen.named_parameters() if p.requires_grad]
    np.savez_compressed(os.path.join(HERE, "lora_reference.npz"), **arrays)
    with open(os.path.join(HERE, "lora_reference.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("lora_reference:", meta["encoder"]["adapted"], meta["llm"]["adapted"])


def lora_targets_cases():
    """apply_lora with target_modules BEYOND the default (ultravox_config.py:19-21 is only a default; the reference hands the list to peft,
    ultravox_model.py:695, 707): q / k / v / out_proj of an HF WhisperEncoder, q / k / v / o_proj of an HF LlamaForCausalLM (GQA), and a
    v + o-only list - which modules get adapted, under which names, forward and adapter gradients.  Same towers / seeds as lora_cases()."""
    from transformers import LlamaConfig, LlamaForCausalLM, WhisperConfig
    from transformers.models.whisper.modeling_whisper import WhisperEncoder
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.weights import random_state_dict
    tiny = dict(audio_config=dict(d_model=64, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=128, num_mel_bins=80,
                                  max_source_positions=1500),
                text_config=dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                                 num_key_value_heads=2, vocab_size=256, rope_theta=10000.0, max_position_embeddings=512,
                                 rms_norm_eps=1e-5),
                hidden_size=128, stack_factor=8, projector_ln_mid=True)
    cfg = UltravoxConfig(**tiny)
    a, t = cfg.audio_config, cfg.text_config
    sd = random_state_dict(cfg, seed=17)
    simp = ultravox_config.LoraConfigSimplified
    arrays, meta = {}, {"tiny": tiny, "seed": 17, "cases": {}}

    def randomise_b(model, seed):
        g = torch.Generator().manual_seed(seed)
        for n, p in model.named_parameters():
            if "lora_B" in n:
                p.data = 0.05 * torch.randn(p.shape, generator=g)

    for case, targets, r in (("all", ["q_proj", "k_proj", "v_proj", "out_proj", "o_proj"], 4), ("vo", ["v_proj", "out_proj", "o_proj"], 2),
                             ("mlp", ["q_proj", "fc1", "fc2", "gate_proj", "up_proj", "down_proj"], 4)):      # the MLP's linears next to one attention projection
        lcfg = dataclasses.asdict(simp(r=r, lora_alpha=6, target_modules=targets))
        cm = meta["cases"][case] = {"lora_config": dict(lcfg)}
        # ---- encoder ----
        enc = WhisperEncoder(WhisperConfig(d_model=a.d_model, encoder_layers=a.encoder_layers, encoder_attention_heads=a.encoder_attention_heads,
                                           encoder_ffn_dim=a.encoder_ffn_dim, num_mel_bins=a.num_mel_bins,
                                           max_source_positions=a.max_source_positions, attn_implementation="eager")).eval()
        enc.load_state_dict({k[len("audio_tower."):]: v for k, v in sd.items() if k.startswith("audio_tower.")}, strict=False)
        wrapped = ultravox_model.apply_lora(enc, dict(lcfg))
        randomise_b(wrapped, 1)
        names = [n for n, p in wrapped.named_parameters() if p.requires_grad]
        cm["encoder"] = {"trainable": names, "adapted": sorted({n.split(".lora_")[0] for n in names})}
        torch.manual_seed(0)
        x, audio_len = torch.randn(2, 80, 120), torch.tensor([120, 75])
        inner = wrapped.base_model.model
        h = torch.nn.functional.gelu(inner.conv1(x))
        h = torch.nn.functional.gelu(inner.conv2(h)).permute(0, 2, 1)
        h = h + inner.embed_positions.weight[: h.size(-2)]
        keep = torch.arange(h.shape[1])[None, :].lt(((audio_len - 1) // 2 + 1).view(-1, 1))
        mask = (1.0 - keep[:, None, None, :].float()) * torch.finfo(torch.float32).min
        for layer in inner.layers:
            out = layer(h, mask)
            h = out[0] if isinstance(out, tuple) else out
        y = inner.layer_norm(h)
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(2))
        (y * gy).sum().backward()
        arrays.update({f"{case}.enc.x": x.numpy(), f"{case}.enc.audio_len": audio_len.numpy(), f"{case}.enc.y": y.detach().numpy(),
                       f"{case}.enc.gy": gy.numpy()})
        for n, p in wrapped.named_parameters():
            if p.requires_grad:
                arrays[f"{case}.enc.w." + n], arrays[f"{case}.enc.g." + n] = p.detach().numpy(), p.grad.numpy()
        # ---- language model ----
        llm = LlamaForCausalLM(LlamaConfig(hidden_size=t.hidden_size, intermediate_size=t.intermediate_size, num_hidden_layers=t.num_hidden_layers,
                                           num_attention_heads=t.num_attention_heads, num_key_value_heads=t.num_key_value_heads,
                                           vocab_size=t.vocab_size, rms_norm_eps=t.rms_norm_eps, rope_theta=t.rope_theta,
                                           max_position_embeddings=t.max_position_embeddings, tie_word_embeddings=False,
                                           attn_implementation="eager")).eval()
        llm.load_state_dict({k[len("language_model."):]: v for k, v in sd.items() if k.startswith("language_model.")}, strict=False)
        wl = ultravox_model.apply_lora(llm, dict(lcfg))
        randomise_b(wl, 3)
        names = [n for n, p in wl.named_parameters() if p.requires_grad]
        cm["llm"] = {"trainable": names, "adapted": sorted({n.split(".lora_")[0] for n in names})}
        emb = 0.5 * torch.randn(2, 21, t.hidden_size, generator=torch.Generator().manual_seed(4))
        am = torch.ones(2, 21, dtype=torch.long)
        am[1, -5:] = 0
        logits = wl(inputs_embeds=emb, attention_mask=am).logits
        gl = torch.randn(logits.shape, generator=torch.Generator().manual_seed(5)) * am[..., None]
        (logits * gl).sum().backward()
        arrays.update({f"{case}.llm.emb": emb.numpy(), f"{case}.llm.mask": am.numpy(), f"{case}.llm.logits": logits.detach().numpy(),
                       f"{case}.llm.gl": gl.numpy()})
        for n, p in wl.named_parameters():
            if p.requires_grad:
                arrays[f"{case}.llm.w." + n], arrays[f"{case}.llm.g." + n] = p.detach().numpy(), p.grad.numpy()
    # a list none of whose names exist in the tower: peft raises (the text our config check repeats)
    try:
        ultravox_model.apply_lora(torch.nn.Sequential(torch.nn.Linear(3, 3)), dataclasses.asdict(simp(r=2, target_modules=["linear_k"])))
        meta["no_hit_error"] = None
    except ValueError as e:
        meta["no_hit_error"] = str(e)
    np.savez_compressed(os.path.join(HERE, "lora_targets_reference.npz"), **arrays)
    with open(os.path.join(HERE, "lora_targets_reference.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("lora_targets_reference:", {c: (m["encoder"]["adapted"], m["llm"]["adapted"]) for c, m in meta["cases"].items()})


def lora_w2v_cases():
    """apply_lora on the AutoModel branch's tower (ultravox_model.py:460-467: whatever tower was loaded is wrapped): the reference's apply_lora with
    its default target_modules and with q / k / v / out_proj on an installed-HF Wav2Vec2Model, both encoder families (post-LN; do_stable_layer_norm) -
    which modules are adapted, under which names, last_hidden_state and the adapter gradients."""
    from transformers import Wav2Vec2Config, Wav2Vec2Model
    from ultravox_amd.config import UltravoxConfig
    from ultravox_amd.weights import random_state_dict
    w2v_tiny = dict(model_type="wav2vec2", hidden_size=64, num_hidden_layers=2, num_attention_heads=2, intermediate_size=128, conv_dim=[64] * 7,
                    num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4)
    text_tiny = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, vocab_size=256)
    simp = ultravox_config.LoraConfigSimplified
    arrays, meta = {}, {"w2v_tiny": w2v_tiny, "text_tiny": text_tiny, "seed": 19, "cases": {}}
    for case, stable, norm, bias, targets in (("post_ln_default", False, "group", False, None),
                                              ("stable_all", True, "layer", True, ["q_proj", "k_proj", "v_proj", "out_proj"]),
                                              ("post_ln_all", False, "group", False, ["q_proj", "k_proj", "v_proj", "out_proj"])):
        fam = dict(feat_extract_norm=norm, conv_bias=bias, do_stable_layer_norm=stable)
        cfg = UltravoxConfig(audio_config={**w2v_tiny, **fam}, text_config=text_tiny, hidden_size=64)
        sd = random_state_dict(cfg, seed=19)
        hf = Wav2Vec2Model(Wav2Vec2Config(hidden_size=64, num_hidden_layers=2, num_attention_heads=2, intermediate_size=128, conv_dim=[64] * 7,
                                          num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4, attn_implementation="eager", **fam)).eval()
        missing, unexpected = hf.load_state_dict({k[len("audio_tower."):]: v for k, v in sd.items() if k.startswith("audio_tower.")}, strict=False)
        assert not unexpected and not missing      # (random_state_dict carries masked_spec_embed since round 6; the eval forward does not read it)
        lcfg = dataclasses.asdict(simp(r=4, lora_alpha=6) if targets is None else simp(r=4, lora_alpha=6, target_modules=targets))
        wrapped = ultravox_model.apply_lora(hf, dict(lcfg))
        g = torch.Generator().manual_seed(1)
        for n, p in wrapped.named_parameters():
            if "lora_B" in n:
                p.data = 0.05 * torch.randn(p.shape, generator=g)
        names = [n for n, p in wrapped.named_parameters() if p.requires_grad]
        torch.manual_seed(0)
        x = torch.randn(2, 6000) * 0.1 + 0.02
        x = (x - x.mean(-1, keepdim=True)) / torch.sqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-7)
        y = wrapped(x).last_hidden_state
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(2))
        (y * gy).sum().backward()
        meta["cases"][case] = {"family": fam, "lora_config": lcfg, "trainable": names, "adapted": sorted({n.split(".lora_")[0] for n in names})}
        arrays.update({f"{case}.x": x.numpy(), f"{case}.y": y.detach().numpy(), f"{case}.gy": gy.numpy()})
        for n, p in wrapped.named_parameters():
            if p.requires_grad:
                arrays[f"{case}.w." + n], arrays[f"{case}.g." + n] = p.detach().numpy(), p.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "lora_w2v_reference.npz"), **arrays)
    with open(os.path.join(HERE, "lora_w2v_reference.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("lora_w2v_reference:", {c: m["adapted"][:2] for c, m in meta["cases"].items()})


CONFIG_SCALARS = ["ignore_index", "audio_model_id", "text_model_id", "audio_token_index", "hidden_size", "stack_factor", "norm_init",
                  "projector_act", "projector_ln_mid", "llm_only_training", "audio_latency_block_size", "vocab_size", "initializer_range"]
CONFIG_TEXT = ["model_type", "hidden_size", "num_hidden_layers", "num_attention_heads", "num_key_value_heads", "intermediate_size",
               "vocab_size", "rms_norm_eps", "head_dim", "max_position_embeddings", "initializer_range", "tie_word_embeddings"]
CONFIG_AUDIO = ["model_type", "d_model", "encoder_layers", "encoder_attention_heads", "encoder_ffn_dim", "num_mel_bins",
                "max_source_positions", "hidden_size", "num_hidden_layers", "num_attention_heads", "intermediate_size",
                "conv_dim", "conv_stride", "conv_kernel", "num_conv_pos_embeddings", "num_conv_pos_embedding_groups"]


def config_summary(c):
    """What the hot path reads from an UltravoxConfig - the same function runs on the REFERENCE object here and on
    ultravox_amd.config.UltravoxConfig in tests/test_checkpoint_cpu.py (attributes a class does not have are skipped there)."""
    out = {k: getattr(c, k) for k in CONFIG_SCALARS}
    for name in ("text_model_lora_config", "audio_model_lora_config"):
        v = getattr(c, name)
        out[name] = dataclasses.asdict(v) if dataclasses.is_dataclass(v) else v
    out["loss"] = None
    out["text"] = {k: getattr(c.text_config, k) for k in CONFIG_TEXT if hasattr(c.text_config, k)}
    out["audio"] = {k: list(v) if isinstance(v, tuple) else v
                    for k in CONFIG_AUDIO if (v := getattr(c.audio_config, k, None)) is not None}
    out["diff_keys"] = sorted(k for k in c.to_diff_dict() if k not in ("transformers_version", "torch_dtype"))
    return out


def config_cases():
    """The REFERENCE UltravoxConfig (ultravox_config.py:56-203) built from keyword sets that need no network (sub-configs
    as dicts: AutoConfig.for_model fills the [3P] family defaults), including partial dicts - what a missing field means."""
    llama_small = {"model_type": "llama", "hidden_size": 256, "intermediate_size": 512, "num_hidden_layers": 2,
                   "num_attention_heads": 4, "num_key_value_heads": 2, "vocab_size": 512}
    whisper_small = {"model_type": "whisper", "d_model": 128, "encoder_layers": 2, "encoder_attention_heads": 4,
                     "encoder_ffn_dim": 256, "num_mel_bins": 80, "max_source_positions": 1500}
    cases = {
        "defaults": {},
        "small_dicts": dict(text_config=llama_small, audio_config=whisper_small, hidden_size=384, stack_factor=4,
                            projector_ln_mid=True, audio_latency_block_size=50),
        "partial_text_dict": dict(text_config={"model_type": "llama", "hidden_size": 256, "vocab_size": 512},
                                  audio_model_lora_config={"r": 8, "lora_alpha": 16}, text_model_lora_config={"r": 4}),
        "projector_variants": dict(text_config=llama_small, audio_config=whisper_small, projector_act="swiglu", norm_init=0.4,
                                   projector_ln_mid=False, stack_factor=8, audio_token_index=32000, ignore_index=-100),
        "gemma_wav2vec2": dict(text_config={"model_type": "gemma", "hidden_size": 192, "head_dim": 32, "vocab_size": 512,
                                            "num_hidden_layers": 2, "num_attention_heads": 4, "num_key_value_heads": 4,
                                            "intermediate_size": 384},
                               audio_config={"model_type": "wav2vec2", "hidden_size": 64, "num_hidden_layers": 2,
                                             "num_attention_heads": 2, "intermediate_size": 128}),
        "partial_gemma_dict": dict(text_config={"model_type": "gemma", "vocab_size": 512}),
        "llm_only": dict(text_config=llama_small, llm_only_training=True),
        # ultravox_config.py:68 "any of LlamaConfig or MistralConfig": the family defaults a partial Mistral dict resolves to
        "partial_mistral_dict": dict(text_config={"model_type": "mistral", "hidden_size": 256, "num_attention_heads": 4, "vocab_size": 512}),
    }
    out = {}
    for name, kw in cases.items():
        c = ultravox_config.UltravoxConfig(**json.loads(json.dumps(kw)))
        out[name] = {"kwargs": kw, "expect": config_summary(c)}
    lc = ultravox_config.LossConfig()
    out["_loss_config_defaults"] = {"loss_function": str(lc.loss_function.value), "kl_temperature": lc.kl_temperature,
                                    "requires_alt_fields": lc.requires_alt_fields}
    with open(os.path.join(HERE, "config.json"), "w") as f:
        json.dump(out, f, indent=1, default=str)
    print("config:", sorted(out))


def forward_cases():
    """The REFERENCE UltravoxModel.forward (ultravox_model.py:277-396) END TO END on CPU: its own _prepare_audio_embeds,
    _audio_iter, UltravoxProjector, the in-place merge loop, the installed-HF LlamaForCausalLM + ForCausalLMLoss, and
    loss.backward() into the projector - with random tiny towers built from dict configs (no network) and the audio tower's
    forward replaced by recorded hidden states (the installed transformers 5.x encoder layer no longer takes the 4.51.3
    mask the reference's ModifiedWhisperEncoder.forward builds; the tower itself is pinned separately, against HF blocks).
    Weights and inputs are the seeded tensors of tests/forward_fixture_util.py (functions of name and shape), so only the
    outputs are stored.  Compatibility shims, all outside the arithmetic: transformers.modeling_utils._init_weights (removed
    in 5.x: the reference reads it at :447), tie_weights(**kwargs), rotary inv_freq recomputed after to_empty()."""
    import forward_fixture_util as U
    transformers.modeling_utils._init_weights = True
    tw = ultravox_model.UltravoxModel.tie_weights
    ultravox_model.UltravoxModel.tie_weights = lambda self, *a, **k: tw(self)
    arrays, meta = {}, {"cases": {}}
    for name, ln_mid in (("ln_mid", True), ("ln_post", False), ("ln_mid_mixed", True)):
        kw = U.config_kwargs(ln_mid)
        kw["audio_config"].update({"_name_or_path": "random/whisper-nano", "decoder_layers": 1, "decoder_attention_heads": 2,
                                   "decoder_ffn_dim": 64, "vocab_size": 100, "pad_token_id": 0, "bos_token_id": 1,
                                   "eos_token_id": 2, "decoder_start_token_id": 1})
        m = ultravox_model.UltravoxModel(ultravox_config.UltravoxConfig(**json.loads(json.dumps(kw)))).to_empty(device="cpu")
        n_w = 0
        with torch.no_grad():
            for n, p in m.named_parameters():
                p.copy_(U.param(n, p.shape))
                n_w += n.startswith(("multi_modal_projector.", "language_model."))
        m.float()
        dim = U.TEXT["hidden_size"] // U.TEXT["num_attention_heads"]
        inv = 1.0 / (U.TEXT["rope_theta"] ** (torch.arange(0, dim, 2).float() / dim))
        m.language_model.model.rotary_emb.inv_freq = inv
        m.language_model.model.rotary_emb.original_inv_freq = inv.clone()
        enc = U.tower_output()
        m.audio_tower.forward = lambda audio_values, audio_len=None, **k: transformers.modeling_outputs.BaseModelOutput(
            last_hidden_state=enc[: audio_values.shape[0]])
        out = m(audio_values=torch.zeros(U.N_AUDIO, 80, 3000), **U.batch(mixed=name.endswith("_mixed")))
        out.loss.backward()
        for n, p in m.multi_modal_projector.named_parameters():
            arrays[f"{name}.g.multi_modal_projector.{n}"] = p.grad.numpy()
        arrays[f"{name}.logits"] = out.logits.detach().numpy()
        arrays[f"{name}.loss"] = np.array(out.loss.item(), np.float64)
        meta["cases"][name] = {"loss": out.loss.item(), "n_weights": int(n_w),
                               "weight_names": [n for n, _ in m.named_parameters() if n.startswith(("multi_modal_projector.", "language_model."))]}
    np.savez_compressed(os.path.join(HERE, "forward_reference.npz"), **arrays)
    with open(os.path.join(HERE, "forward_reference.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("forward_reference:", {k: v["loss"] for k, v in meta["cases"].items()})


def _seeded_reference_model(ln_mid=True, extra=None, base_kwargs=None):
    """The reference UltravoxModel on the seeded tiny towers of tests/forward_fixture_util.py, audio tower stubbed."""
    import forward_fixture_util as U
    transformers.modeling_utils._init_weights = True
    if not getattr(ultravox_model.UltravoxModel.tie_weights, "_shimmed", False):
        tw = ultravox_model.UltravoxModel.tie_weights
        f = lambda self, *a, **k: tw(self)
        f._shimmed = True
        ultravox_model.UltravoxModel.tie_weights = f
    kw = base_kwargs if base_kwargs is not None else U.config_kwargs(ln_mid)
    kw["audio_config"].update({"_name_or_path": "random/whisper-nano", "decoder_layers": 1, "decoder_attention_heads": 2,
                               "decoder_ffn_dim": 64, "vocab_size": 100, "pad_token_id": 0, "bos_token_id": 1,
                               "eos_token_id": 2, "decoder_start_token_id": 1})
    kw.update(extra or {})
    m = ultravox_model.UltravoxModel(ultravox_config.UltravoxConfig(**json.loads(json.dumps(kw)))).to_empty(device="cpu")
    with torch.no_grad():
        for n, p in m.named_parameters():
            p.copy_(U.param(n, p.shape))
    m.float()
    tc = kw["text_config"]
    if tc.get("tie_word_embeddings"):      # to_empty() + per-name seeding broke the tie: share the tensor again (HF tie_weights)
        m.language_model.lm_head.weight = m.language_model.model.embed_tokens.weight
    dim = tc.get("head_dim") or tc["hidden_size"] // tc["num_attention_heads"]
    inv = 1.0 / (tc["rope_theta"] ** (torch.arange(0, dim, 2).float() / dim))
    for mod in m.language_model.modules():
        if hasattr(mod, "inv_freq"):
            mod.inv_freq = inv
            mod.original_inv_freq = inv.clone()
    return m


def kl_forward_cases():
    """The REFERENCE UltravoxModel.forward in TRAINING mode with LossFunction.KL_Divergence (ultravox_model.py:335-351 ->
    _compute_kl_loss :200-256): text-only teacher pass of the same LM over alt_* without gradient, student = the audio path,
    KL over the prediction positions + the end-of-turn term; loss and projector gradients on the seeded tiny model."""
    import forward_fixture_util as U
    arrays, meta = {}, {}
    for tag, lc in (("t2_eot1", dict(kl_temperature=2.0, eot_loss_weight=1.0)), ("t1_eot0", dict(kl_temperature=1.0, eot_loss_weight=0.0))):
        m = _seeded_reference_model(True)
        m.set_loss_config(ultravox_config.LossConfig(loss_function=ultravox_config.LossFunction.KL_Divergence, **lc))
        m.train()
        enc = U.tower_output()
        m.audio_tower.forward = lambda audio_values, audio_len=None, **k: transformers.modeling_outputs.BaseModelOutput(
            last_hidden_state=enc[: audio_values.shape[0]])
        out = m(audio_values=torch.zeros(U.N_AUDIO, 80, 3000), **U.batch(), **U.alt_batch())
        out.loss.backward()
        for n, p in m.multi_modal_projector.named_parameters():
            arrays[f"{tag}.g.multi_modal_projector.{n}"] = p.grad.numpy()
        arrays[f"{tag}.loss"] = np.array(out.loss.item(), np.float64)
        meta[tag] = {"loss": out.loss.item(), **lc}
    np.savez_compressed(os.path.join(HERE, "kl_forward_reference.npz"), **arrays)
    with open(os.path.join(HERE, "kl_forward_reference.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("kl_forward_reference:", meta)


def lora_forward_cases():
    """The REFERENCE UltravoxModel with text_model_lora_config r = 4 (apply_lora through tests/peft_stub.py on the LLM's q_proj /
    k_proj, ultravox_model.py:499-526, 690-709), forward + backward on the seeded tiny model with NON-ZERO lora_B: loss, logits
    and the gradients of the projector and of every adapter matrix.  Parameter names are recorded: the base weights sit under
    peft's `base_model.model.` / `.base_layer` names there and are seeded by those names."""
    import forward_fixture_util as U
    m = _seeded_reference_model(True, extra={"text_model_lora_config": dataclasses.asdict(ultravox_config.LoraConfigSimplified(r=4))})
    enc = U.tower_output()
    m.audio_tower.forward = lambda audio_values, audio_len=None, **k: transformers.modeling_outputs.BaseModelOutput(
        last_hidden_state=enc[: audio_values.shape[0]])
    out = m(audio_values=torch.zeros(U.N_AUDIO, 80, 3000), **U.batch())
    out.loss.backward()
    arrays = {"logits": out.logits.detach().numpy(), "loss": np.array(out.loss.item(), np.float64)}
    trainable = [n for n, p in m.named_parameters() if p.requires_grad]
    for n, p in m.named_parameters():
        if p.requires_grad:
            arrays["g." + n] = p.grad.numpy()
    meta = {"loss": out.loss.item(), "trainable": trainable, "lora_config": dataclasses.asdict(ultravox_config.LoraConfigSimplified(r=4)),
            "weight_names": [n for n, _ in m.named_parameters() if n.startswith(("multi_modal_projector.", "language_model."))]}
    np.savez_compressed(os.path.join(HERE, "lora_forward_reference.npz"), **arrays)
    with open(os.path.join(HERE, "lora_forward_reference.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("lora_forward_reference:", meta["loss"], len(trainable))


def kl_lora_forward_cases():
    """The REFERENCE model with text_model_lora_config r = 4 in TRAINING mode under LossFunction.KL_Divergence: the teacher pass of
    _compute_kl_loss goes through the SAME self.language_model (ultravox_model.py:212-222), i.e. with the adapters ACTIVE, under
    no_grad; the student is the audio path through the same adapted model.  Loss, projector and adapter gradients, non-zero lora_B."""
    import forward_fixture_util as U
    lcfg = dataclasses.asdict(ultravox_config.LoraConfigSimplified(r=4))
    m = _seeded_reference_model(True, extra={"text_model_lora_config": lcfg})
    lc = dict(kl_temperature=2.0, eot_loss_weight=1.0)
    m.set_loss_config(ultravox_config.LossConfig(loss_function=ultravox_config.LossFunction.KL_Divergence, **lc))
    m.train()
    enc = U.tower_output()
    m.audio_tower.forward = lambda audio_values, audio_len=None, **k: transformers.modeling_outputs.BaseModelOutput(
        last_hidden_state=enc[: audio_values.shape[0]])
    out = m(audio_values=torch.zeros(U.N_AUDIO, 80, 3000), **U.batch(), **U.alt_batch())
    out.loss.backward()
    arrays = {"loss": np.array(out.loss.item(), np.float64)}
    trainable = [n for n, p in m.named_parameters() if p.requires_grad]
    for n, p in m.named_parameters():
        if p.requires_grad:
            arrays["g." + n] = p.grad.numpy()
    meta = {"loss": out.loss.item(), "trainable": trainable, "lora_config": lcfg, **lc,
            "weight_names": [n for n, _ in m.named_parameters() if n.startswith(("multi_modal_projector.", "language_model."))]}
    np.savez_compressed(os.path.join(HERE, "kl_lora_forward_reference.npz"), **arrays)
    with open(os.path.join(HERE, "kl_lora_forward_reference.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("kl_lora_forward_reference:", meta["loss"], len(trainable))


def generate_cases():
    """The REFERENCE UltravoxModel.generate (ultravox_model.py:398-426 -> [3P] GenerationMixin greedy search) on the seeded
    tiny model of forward_cases: audio merged once before the prefill, a left-padded prompt next to an unpadded one,
    (a) 10 new tokens with no EOS, (b) the same with an EOS id chosen so that row 0 stops after 3 tokens and is padded."""
    import forward_fixture_util as U
    transformers.modeling_utils._init_weights = True
    if not getattr(ultravox_model.UltravoxModel.tie_weights, "_shimmed", False):
        tw = ultravox_model.UltravoxModel.tie_weights
        f = lambda self, *a, **k: tw(self)
        f._shimmed = True
        ultravox_model.UltravoxModel.tie_weights = f
    kw = U.config_kwargs(True)
    kw["audio_config"].update({"_name_or_path": "random/whisper-nano", "decoder_layers": 1, "decoder_attention_heads": 2,
                               "decoder_ffn_dim": 64, "vocab_size": 100, "pad_token_id": 0, "bos_token_id": 1,
                               "eos_token_id": 2, "decoder_start_token_id": 1})
    m = ultravox_model.UltravoxModel(ultravox_config.UltravoxConfig(**json.loads(json.dumps(kw)))).to_empty(device="cpu")
    with torch.no_grad():
        for n, p in m.named_parameters():
            p.copy_(U.param(n, p.shape))
    m.float().eval()
    dim = U.TEXT["hidden_size"] // U.TEXT["num_attention_heads"]
    inv = 1.0 / (U.TEXT["rope_theta"] ** (torch.arange(0, dim, 2).float() / dim))
    m.language_model.model.rotary_emb.inv_freq = inv
    m.language_model.model.rotary_emb.original_inv_freq = inv.clone()
    enc = U.tower_output()[U.GEN_AUDIO_ROWS]
    m.audio_tower.forward = lambda audio_values, audio_len=None, **k: transformers.modeling_outputs.BaseModelOutput(
        last_hidden_state=enc[: audio_values.shape[0]])
    b = U.generate_batch()
    T = b["input_ids"].shape[1]
    free = m.generate(audio_values=torch.zeros(3, 80, 3000), max_new_tokens=10, do_sample=False, pad_token_id=0,
                      eos_token_id=None, **b)
    eos = int(free[0, T + 2])
    assert eos not in free[0, T:T + 2].tolist() and eos not in free[1, T:].tolist()
    stop = m.generate(audio_values=torch.zeros(3, 80, 3000), max_new_tokens=10, do_sample=False, pad_token_id=0,
                      eos_token_id=eos, **b)
    out = {"prompt_len": T, "free": free[:, T:].tolist(), "eos": eos, "with_eos": stop[:, T:].tolist(), "pad_token_id": 0}
    with open(os.path.join(HERE, "generate_reference.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("generate_reference:", out)


class _TupleLayer(torch.nn.Module):
    """Adapter between the reference's ModifiedWhisperEncoder.forward (written for transformers 4.51.3: it calls
    `encoder_layer(hidden, mask, layer_head_mask=..., output_attentions=...)` and indexes the result with [0],
    ultravox_model.py:966-975) and the installed 5.x WhisperEncoderLayer (no layer_head_mask, returns the tensor).  It touches
    no arithmetic: the mask the REFERENCE built (:915-936) is handed to the layer unchanged."""

    def __init__(self, layer):
        super().__init__()
        self.layer = layer

    def forward(self, hidden_states, attention_mask, layer_head_mask=None, output_attentions=False):
        assert layer_head_mask is None and not output_attentions
        out = self.layer(hidden_states, attention_mask)
        return out if isinstance(out, tuple) else (out,)


def _wrap_encoder_layers(enc):
    enc.layers = torch.nn.ModuleList([_TupleLayer(l) for l in enc.layers])
    return enc


def real_tower_cases():
    """The REFERENCE ModifiedWhisperEncoder.forward ITSELF (ultravox_model.py:865-994: conv stem, positional slice, the
    audio_len key-padding mask :915-926, the latency mask of init_latency_mask :834-863 merged at :928-936, the layer loop,
    final LayerNorm) on seeded tiny weights, and the REFERENCE UltravoxModel.forward + loss.backward() with that tower in place
    (nothing stubbed: mel -> tower -> projector -> merge -> Llama -> loss).  Only shim: _TupleLayer above."""
    import forward_fixture_util as U
    arrays, meta = {}, {"encoder_cases": {}, "model_cases": {}}
    wcfg = transformers.WhisperConfig(**{k: v for k, v in U.AUDIO_REAL.items() if k != "model_type"}, attn_implementation="eager")
    for name, (frames, audio_len, block) in U.ENCODER_CASES.items():
        enc = ultravox_model.ModifiedWhisperEncoder(wcfg).eval()
        names = sorted("audio_tower." + n for n, _ in enc.named_parameters())      # incl. embed_positions.weight
        with torch.no_grad():
            for n, p in enc.named_parameters():
                p.copy_(U.param("audio_tower." + n, p.shape))
        enc.init_latency_mask(block, torch.float32)
        _wrap_encoder_layers(enc)
        n_items = 3 if audio_len is None else len(audio_len)
        x = U.mel(n_items, frames)
        with torch.no_grad():
            out = enc(x, audio_len=None if audio_len is None else torch.tensor(audio_len)).last_hidden_state
        arrays[f"enc.{name}"] = out.numpy()
        meta["encoder_cases"][name] = {"frames": frames, "audio_len": audio_len, "audio_latency_block_size": block,
                                       "weight_names": names}
    for name, ln_mid, latency in (("real_tower", True, None), ("real_tower_latency", True, 100)):
        kw = U.real_config_kwargs(ln_mid, latency)
        kw["torch_dtype"] = "float32"           # init_latency_mask reads config.torch_dtype (ultravox_model.py:472-475)
        m = _seeded_reference_model(ln_mid, base_kwargs=kw)
        names = sorted(n for n, _ in m.named_parameters())
        assert isinstance(m.audio_tower, ultravox_model.ModifiedWhisperEncoder)
        assert (m.audio_tower.audio_streaming_mask is None) == (latency is None)
        if latency is not None:     # _seeded_reference_model's to_empty() wiped the (non-persistent) mask buffer: let the
            del m.audio_tower.audio_streaming_mask          # reference build it again
            m.audio_tower.init_latency_mask(latency, torch.float32)
        _wrap_encoder_layers(m.audio_tower)
        b = U.batch()
        melx = U.mel(U.N_AUDIO, 3000)
        out = m(audio_values=melx, **b)
        out.loss.backward()
        for n, p in m.multi_modal_projector.named_parameters():
            arrays[f"{name}.g.multi_modal_projector.{n}"] = p.grad.numpy()
        arrays[f"{name}.logits"] = out.logits.detach().numpy()
        arrays[f"{name}.loss"] = np.array(out.loss.item(), np.float64)
        with torch.no_grad():
            tower = m.audio_tower(melx, audio_len=b["audio_lens"]).last_hidden_state
        arrays[f"{name}.tower_rows"] = tower[:, :80].numpy()          # the rows the projector's first 10 outputs read
        meta["model_cases"][name] = {"loss": out.loss.item(), "audio_latency_block_size": latency,
                                     "weight_names": names}
    np.savez_compressed(os.path.join(HERE, "real_tower_reference.npz"), **arrays)
    with open(os.path.join(HERE, "real_tower_reference.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("real_tower_reference:", {k: v["loss"] for k, v in meta["model_cases"].items()}, list(meta["encoder_cases"]))


if __name__ == "__main__":
    if len(sys.argv) > 1:          # python make_golden.py kl_lora_forward_cases ...: only the named generators
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    real_tower_cases()
    kl_lora_forward_cases()
    lora_forward_cases()
    kl_forward_cases()
    generate_cases()
    forward_cases()
    config_cases()
    lora_cases()
    lora_targets_cases()
    lora_w2v_cases()
    processor_cases()
    projector_cases()
    projector_act_cases()
    latency_mask_cases()
    logmel_cases()
    logmel_speech_cases()
    kl_cases()
    diff_state_dict_cases()
    dataproc_cases()
