"""Sample -> model-input preprocessing: the mirror of the reference's `UltravoxDataproc`
(ultravox/model/ultravox_data_proc.py:10-154), the stage between a dataset of voice samples and
`DataCollatorForSeq2SeqWithAudio`.  Host-side integer work only: chat template -> `UltravoxProcessor` -> labels with the
prompt masked out, plus the text-only "alt" fields the KL-distillation loss feeds to the teacher pass.

A sample is any object with `messages`, `audio` (1-D float array or None), `sample_rate` and `audio_transcript`
(`ultravox_amd.inference.VoiceSample` has the first three; the transcript defaults to "").  Pinned against vectors the
reference class itself produced (tests/golden/dataproc.json)."""
import copy
from typing import Any, Dict, Iterable, Optional

import numpy as np

from .config import LossMaskType
from .processing import AUDIO_PLACEHOLDER, UltravoxProcessor

IGNORE_INDEX = -100


class UltravoxDataproc:
    def __init__(self, dataset: Iterable, processor: UltravoxProcessor, loss_mask_type: LossMaskType, augmentation=None,
                 inference_mode: bool = False, include_alt_fields: bool = False, max_response_tokens: Optional[int] = None,
                 chat_template: Optional[str] = None) -> None:
        self._dataset = dataset
        self.processor = processor
        self.loss_mask_type = LossMaskType(loss_mask_type)
        self.augmentation = augmentation
        self.inference_mode = inference_mode                  # drop the assistant turn: the model is to generate it
        self.include_alt_fields = include_alt_fields          # alt_* = the same dialogue with <|audio|> -> transcript
        self.max_response_tokens = max_response_tokens
        self.chat_template = chat_template

    # the reference's Dataproc base (ultravox/data/datasets.py:592-615)
    def __iter__(self):
        for sample in self._dataset:
            yield self._process(sample)

    def __len__(self) -> int:
        return len(self._dataset)

    def __str__(self) -> str:
        return f"Dataproc({self._dataset})"

    @property
    def name(self):
        return self._dataset.name

    def _template(self, messages) -> str:
        return self.processor.tokenizer.apply_chat_template(messages, tokenize=False, chat_template=self.chat_template)

    def _prompt_len(self, sample, audio) -> int:
        """Number of leading positions whose labels are masked (:46-79).  The prompt is re-tokenised WITH the audio so the
        expanded placeholder run is counted; the processor raises if text and audio disagree about placeholders."""
        if self.loss_mask_type == LossMaskType.ALL:
            return 0
        if self.loss_mask_type == LossMaskType.AFTER_AUDIO:
            prompt = self._template(sample.messages).split(AUDIO_PLACEHOLDER)[0] + AUDIO_PLACEHOLDER
        else:                                                  # LAST_ASSISTANT: everything before the final message
            prompt = self._template(sample.messages[:-1])
        return self.processor(text=prompt, audios=audio, sampling_rate=sample.sample_rate)["input_ids"].shape[-1]

    def _process(self, sample) -> Dict[str, Any]:
        if self.augmentation:
            sample = self.augmentation.apply_sample(sample)
        if self.inference_mode:
            sample = copy.copy(sample)
            sample.messages = sample.messages[:-1]
        text = self._template(sample.messages)
        audio = np.expand_dims(sample.audio, axis=0) if sample.audio is not None else None      # [channels = 1, samples]
        inputs = self.processor(text=text, audios=audio, return_tensors="pt", sampling_rate=sample.sample_rate)
        input_ids = inputs["input_ids"] = inputs["input_ids"].squeeze(0)
        inputs["attention_mask"] = inputs["attention_mask"].squeeze(0)
        n_masked = self._prompt_len(sample, audio)
        labels = input_ids.clone()                             # unshifted: the loss shifts (ultravox_model.py / HF)
        labels[:n_masked] = IGNORE_INDEX
        keep = None
        if self.max_response_tokens and n_masked + self.max_response_tokens < len(input_ids):
            keep = n_masked + self.max_response_tokens
        if self.include_alt_fields:
            alt = self.processor(text=text.replace(AUDIO_PLACEHOLDER, getattr(sample, "audio_transcript", None) or ""),
                                 audio=None, return_tensors="pt")
            alt_ids = alt["input_ids"].squeeze(0)
            alt_masked = n_masked + len(alt_ids) - len(input_ids)      # the two prompts differ only in the audio span
            alt_labels = alt_ids.clone()
            alt_labels[:alt_masked] = IGNORE_INDEX
            alt_keep = None if keep is None else alt_masked + self.max_response_tokens
            inputs["alt_input_ids"] = alt_ids[:alt_keep]
            inputs["alt_attention_mask"] = alt["attention_mask"].squeeze(0)[:alt_keep]
            inputs["alt_labels"] = alt_labels[:alt_keep].tolist()
        if keep is not None:
            inputs["input_ids"] = input_ids[:keep]
            inputs["attention_mask"] = inputs["attention_mask"][:keep]
            labels = labels[:keep]
        return {**inputs, "labels": labels.tolist()}
