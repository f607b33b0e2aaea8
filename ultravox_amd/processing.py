"""Host-side mirror of ultravox/model/ultravox_processing.py: UltravoxProcessor.__call__ (:217-370),
_chunk_and_pad_audio (:153-215) and DataCollatorForSeq2SeqWithAudio.__call__ (:17-64).

Everything here is integer / index arithmetic and must be bit-exact with the reference: mel-frame
lengths, chunking at `audio_context_size` frames, audio_token_len = ceil(frames / (ds * stack)),
placeholder expansion and audio_token_start_idx, right/left padding with the left-pad displacement.
The mel spectrogram itself comes from the `audio_processor` object (ultravox_amd.frontend runs it on the
GPU; any object with the HF feature-extractor call contract works).
"""
from __future__ import annotations

import math
from typing import Any, Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

AUDIO_PLACEHOLDER = "<|audio|>"


class BatchFeature(dict):
    """Minimal stand-in for transformers.BatchFeature: a dict with attribute access and .to()."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def to(self, device):
        return BatchFeature({k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in self.items()})


def _as_tensor_dict(data: Dict[str, Any], return_tensors) -> BatchFeature:
    if return_tensors is None:
        return BatchFeature(data)
    kind = getattr(return_tensors, "value", return_tensors)
    out = {}
    for k, v in data.items():
        if kind == "pt":
            out[k] = v if isinstance(v, torch.Tensor) else torch.as_tensor(v)
        elif kind == "np":
            out[k] = v.cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
        else:
            raise ValueError(f"return_tensors={return_tensors!r} is not supported (pt / np)")
    return BatchFeature(out)


class UltravoxProcessor:
    """Wraps an audio feature extractor and a tokenizer (ultravox_processing.py:67-128)."""

    attributes = ["audio_processor", "tokenizer"]

    def __init__(self, audio_processor=None, tokenizer=None, audio_padding: str = "longest",
                 encoder_ds_factor: int = 2, stack_factor: int = 8, audio_placeholder: str = AUDIO_PLACEHOLDER,
                 audio_context_size: Optional[int] = 3000, audio_frames_fn=None):
        """audio_frames_fn (not in the reference): samples -> encoder frames for a RAW-WAVEFORM tower (wav2vec2, BASELINE
        config 5), e.g. `config.audio_config.feat_extract_output_length`; default = wav2vec2's conv stack."""
        self.audio_frames_fn = audio_frames_fn or _wav2vec2_frames
        self.audio_padding = audio_padding
        self.encoder_ds_factor = encoder_ds_factor
        self.stack_factor = stack_factor
        self.audio_placeholder = audio_placeholder
        self.audio_context_size = audio_context_size
        assert tokenizer is not None and tokenizer.eos_token is not None, \
            "The tokenizer has no EOS token. Cannot recover."
        self.vocab = tokenizer.get_vocab()
        self.audio_token_replacement = tokenizer.eos_token
        if tokenizer.pad_token_id is None:
            tokenizer.pad_token_id = tokenizer.eos_token_id
        if audio_processor is None:
            from .frontend import WhisperFeatureExtractor
            audio_processor = WhisperFeatureExtractor()
        self.audio_processor = audio_processor
        self.tokenizer = tokenizer

    # -- ultravox_processing.py:153-215
    def _chunk_and_pad_audio(self, audio_values: torch.Tensor, audio_lens: torch.Tensor,
                             include_audio_num_chunks: bool = False) -> Dict[str, Any]:
        ctx = self.audio_context_size or audio_values.shape[-1]
        pieces: List[torch.Tensor] = []
        piece_lens: List[int] = []
        continuation: List[bool] = []
        chunks_per_item: List[int] = []
        for i in range(audio_values.shape[0]):
            n = int(audio_lens[i])
            chunks_per_item.append(int(math.ceil(n / ctx)))
            for off in range(0, n, ctx):
                piece = audio_values[i, :, off:off + ctx]
                cont = off > 0
                # only continuation chunks are padded up to the context size (see reference comment :187-191)
                if cont and piece.shape[-1] < ctx:
                    piece = F.pad(piece, (0, ctx - piece.shape[-1]))
                pieces.append(piece)
                piece_lens.append(min(n - off, ctx))
                continuation.append(cont)
        # index tensors live on the host (in the reference everything here is a CPU tensor; with the device
        # log-mel frontend only `audio_values` stays on the GPU)
        dev = torch.device("cpu")
        data = {
            "audio_values": torch.stack(pieces, dim=0),
            "audio_lens": torch.tensor(piece_lens, dtype=torch.int64, device=dev),
            "audio_is_continuation": torch.tensor(continuation, dtype=torch.bool, device=dev),
            "audio_batch_size": torch.tensor([len(pieces)], device=dev),
        }
        if include_audio_num_chunks:
            data["audio_num_chunks"] = torch.tensor(chunks_per_item, dtype=torch.int64, device=dev)
        return data

    # -- ultravox_processing.py:217-370
    def __call__(self, text: Optional[str] = None, audio=None, audios=None, sampling_rate: Optional[int] = None,
                 return_tensors: Optional[str] = "pt", include_audio_num_chunks: bool = False, **kwargs) -> BatchFeature:
        if audio is not None and audios is not None:
            raise ValueError("Only one of `audio` or `audios` should be provided.")
        if audio is not None:
            audios = audio if isinstance(audio, list) or audio.ndim == 2 else [audio]
        elif audios is None:
            audios = []

        data: Dict[str, Any] = {}
        is_cont: Sequence[bool] = []
        if len(audios) > 0:
            audios = [x.numpy() if isinstance(x, torch.Tensor) else x for x in audios]
        if len(audios) > 0 and getattr(self.audio_processor.feature_extractor, "hop_length", None) is None:
            # Raw-waveform tower (the `input_values` fallback of :308).  The reference cannot run this branch - it reads
            # feature_extractor.hop_length unconditionally (:284) and chunks [B, mels, F] features - so the contract is the
            # third-party module's: one un-chunked item per audio, audio_lens = ENCODER frames, audio_token_len =
            # ceil(frames / stack_factor) = the rows the projector produces for it.
            feats = self.audio_processor(audios, sampling_rate=sampling_rate, padding="longest", truncation=False,
                                         return_attention_mask=True, **kwargs)
            n_samples = torch.as_tensor(feats["attention_mask"]).sum(-1).tolist()
            frames = torch.tensor([self.audio_frames_fn(int(n)) for n in n_samples], dtype=torch.int64)
            if int(frames.min()) <= 0:
                raise ValueError("an audio clip is shorter than the encoder's receptive field")
            data["audio_values"] = torch.as_tensor(feats["input_values"])
            data["audio_lens"] = frames
            data["audio_batch_size"] = torch.tensor([len(audios)])
            if include_audio_num_chunks:
                data["audio_num_chunks"] = torch.ones(len(audios), dtype=torch.int64)
            is_cont = [False] * len(audios)
            data["audio_token_len"] = torch.ceil(frames / self.stack_factor).to(dtype=torch.int)
        elif len(audios) > 0:
            hop = self.audio_processor.feature_extractor.hop_length
            # at least two hops of samples, the feature extractor's minimum (:283-292)
            audios = [np.pad(x, (0, 2 * hop - len(x)), mode="constant") if len(x) < 2 * hop else x for x in audios]
            feats = self.audio_processor(audios, sampling_rate=sampling_rate, padding="longest",
                                         pad_to_multiple_of=hop, truncation=False, return_attention_mask=True,
                                         **kwargs)
            values = feats["input_features"] if "input_features" in feats else feats["input_values"]
            values = torch.as_tensor(values)
            frame_lens = torch.as_tensor(feats["attention_mask"]).sum(-1)
            data.update(self._chunk_and_pad_audio(values, frame_lens.cpu(), include_audio_num_chunks))
            is_cont = data.pop("audio_is_continuation").tolist()
            data["audio_token_len"] = torch.ceil(
                data["audio_lens"] / (self.encoder_ds_factor * self.stack_factor)).to(dtype=torch.int)

        if text is not None:
            if not isinstance(text, str):
                raise ValueError("Text must be a string. Batch mode not supported yet.")
            parts = self.tokenizer(text.split(AUDIO_PLACEHOLDER), add_special_tokens=False, **kwargs)["input_ids"]
            fill_id = self.vocab[self.audio_token_replacement]
            ids: List[int] = []
            starts: List[int] = []
            part = -1
            n_audio = len(audios)
            for i, tok_len in enumerate(data.get("audio_token_len", [])):
                if not is_cont[i]:
                    part += 1
                    if part >= len(parts):
                        raise ValueError(f"Text contains too few audio placeholders. (Expected {n_audio} placeholders)")
                    ids.extend(parts[part])
                starts.append(len(ids))
                ids.extend([fill_id] * int(tok_len))
            part += 1
            if part != len(parts) - 1:
                raise ValueError(f"Text contains too many audio placeholders. (Expected {n_audio} placeholders)")
            ids.extend(parts[part])
            if "audio_token_len" in data:
                data["audio_token_start_idx"] = torch.as_tensor(starts)
            data["input_ids"] = [ids]
            data["attention_mask"] = [[1] * len(ids)]
        return _as_tensor_dict(data, return_tensors)

    def batch_decode(self, *args, **kwargs):
        return self.tokenizer.batch_decode(*args, **kwargs)

    def decode(self, *args, **kwargs):
        return self.tokenizer.decode(*args, **kwargs)

    @property
    def model_input_names(self):
        return list(set(self.tokenizer.model_input_names + self.audio_processor.model_input_names))


def _wav2vec2_frames(n_samples: int) -> int:
    """[3P] Wav2Vec2Model._get_feat_extract_output_lengths for the standard conv stack (kernels 10,3,3,3,3,2,2; strides 5,2,...)."""
    n = int(n_samples)
    for k, st in zip((10, 3, 3, 3, 3, 2, 2), (5, 2, 2, 2, 2, 2, 2)):
        n = (n - k) // st + 1
    return n


def _pad_1d(seqs: List[Any], value: int, side: str) -> torch.Tensor:
    seqs = [torch.as_tensor(s).reshape(-1) for s in seqs]      # UltravoxDataproc hands labels over as plain lists
    n = max(int(s.shape[0]) for s in seqs)
    out = []
    for s in seqs:
        gap = n - s.shape[0]
        out.append(F.pad(s, (gap, 0) if side == "left" else (0, gap), value=value))
    return torch.stack(out)


class DataCollatorForSeq2SeqWithAudio:
    """ultravox_processing.py:12-64 on top of the [3P] DataCollatorForSeq2Seq behaviour it inherits:
    input_ids padded with pad_token_id, attention_mask with 0, labels with -100, on tokenizer.padding_side."""

    def __init__(self, tokenizer, include_alt_fields: bool = False, label_pad_token_id: int = -100):
        self.tokenizer = tokenizer
        self.include_alt_fields = include_alt_fields
        self.label_pad_token_id = label_pad_token_id

    def _pad_text(self, feats: List[Dict[str, Any]]) -> Dict[str, torch.Tensor]:
        side = getattr(self.tokenizer, "padding_side", "right")
        out = {"input_ids": _pad_1d([f["input_ids"] for f in feats], self.tokenizer.pad_token_id, side),
               "attention_mask": _pad_1d([f["attention_mask"] for f in feats], 0, side)}
        if all("labels" in f and f["labels"] is not None for f in feats):
            out["labels"] = _pad_1d([f["labels"] for f in feats], self.label_pad_token_id, side)
        for k in feats[0]:
            if k not in ("input_ids", "attention_mask", "labels") and isinstance(feats[0][k], torch.Tensor):
                out[k] = torch.stack([f[k] for f in feats])
        return out

    def __call__(self, features: List[Dict[str, Any]], *args, **kwargs) -> Dict[str, torch.Tensor]:
        features = [dict(f) for f in features]
        take = lambda key: [x for f in features for x in f.pop(key, [])]
        audio_values, audio_lens = take("audio_values"), take("audio_lens")
        audio_token_len, audio_token_start_idx = take("audio_token_len"), take("audio_token_start_idx")
        alt = None
        if self.include_alt_fields:
            alt = [{"input_ids": f.pop("alt_input_ids"), "attention_mask": f.pop("alt_attention_mask"),
                    "labels": f.pop("alt_labels")} for f in features]
        batch = self._pad_text(features)
        if alt is not None:
            ab = self._pad_text(alt)
            batch["alt_input_ids"], batch["alt_attention_mask"], batch["alt_labels"] = \
                ab["input_ids"], ab["attention_mask"], ab["labels"]
        if audio_values and len(audio_values) > 0 and len(audio_values[0]) > 0:
            batch["audio_token_start_idx"] = torch.stack(audio_token_start_idx)
            batch["audio_lens"] = torch.stack(audio_lens)
            batch["audio_token_len"] = torch.stack(audio_token_len)
            width = max(x.shape[-1] for x in audio_values)
            batch["audio_values"] = torch.stack([F.pad(x, (0, width - x.shape[-1])) for x in audio_values])
            if getattr(self.tokenizer, "padding_side", "right") == "left":
                lens = torch.LongTensor([torch.as_tensor(f["input_ids"]).shape[-1] for f in features])
                shift = (batch["input_ids"].shape[-1] - lens).repeat_interleave(batch["audio_batch_size"].squeeze(-1))
                batch["audio_token_start_idx"] += shift.to(batch["audio_token_start_idx"].device)
        return batch
