"""Host-side inference wrapper over `UltravoxModel.generate` — the mirror of the reference's `LocalInference`
(ultravox/inference/infer.py:20-342) and its message types (ultravox/inference/base.py:9-32): one sample in, text + token
counts out, optional token streaming and a conversation mode that carries the dialogue between calls.

What differs from the reference, on purpose:
  * `past_key_values` is the `KVState` that `UltravoxModel.generate(return_dict_in_generate=True)` returns (the cache
    `uvx_llm_prefill` filled, owned by the caller) instead of an HF `Cache`; the next turn runs only the new tokens through
    `uvx_llm_prefill_chunk`.  `generate` re-checks that the cached ids are still a prefix of the new prompt and falls back
    to a full prefill when a re-tokenised reply no longer matches (the reference trusts the caller);
    As in the reference, the cache is what keeps an earlier AUDIO turn audible: the past message holds only
    `eos * audio_token_len` placeholders, the cached keys / values of those positions were computed from the audio
    embeddings.  The wrapper therefore marks its states `partial_ok`: when the re-templated dialogue departs from the
    cached ids part-way (a reply that re-tokenises differently), the rows of the longest common prefix are still reused;
  * `infer_stream` needs one pass, not the reference's two (`:205-223` exist only to snapshot the cache before HF's
    in-place cache grows; a `KVState`'s rows below `cur_len` are never rewritten);
  * resampling uses scipy's polyphase filter (librosa is not a dependency here).
"""
import copy
import dataclasses
import math
import queue
import re
import threading
from typing import Any, Dict, Generator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .processing import DataCollatorForSeq2SeqWithAudio, UltravoxProcessor

SAMPLE_RATE = 16000
MAX_NEW_TOKENS = 1024
AUDIO_PLACEHOLDER = "<|audio|>"


@dataclasses.dataclass
class VoiceOutput:
    text: str
    input_tokens: int
    output_tokens: int
    thinking_content: Optional[str] = None


class InferenceMessage:
    pass


@dataclasses.dataclass
class InferenceChunk(InferenceMessage):
    text: str


@dataclasses.dataclass
class InferenceStats(InferenceMessage):
    input_tokens: int
    output_tokens: int


InferenceGenerator = Generator[InferenceMessage, None, None]


@dataclasses.dataclass
class VoiceSample:
    """The part of the reference's `VoiceSample` (ultravox/data/types.py) inference consumes: chat messages whose last
    user turn may hold one `<|audio|>` placeholder, plus the raw waveform."""
    messages: List[Dict[str, str]]
    audio: Optional[np.ndarray] = None
    sample_rate: int = SAMPLE_RATE

    @classmethod
    def from_prompt(cls, prompt: str) -> "VoiceSample":
        return cls([{"role": "user", "content": prompt}])

    @classmethod
    def from_prompt_and_raw(cls, prompt: str, audio: np.ndarray, sample_rate: int) -> "VoiceSample":
        return cls([{"role": "user", "content": prompt}], audio, sample_rate)

    def add_past_messages(self, past_messages: List[Dict[str, str]]) -> None:
        self.messages = list(past_messages) + list(self.messages)


class _TokenQueueStreamer:
    """HF streamer protocol (`put(ids)`, `end()`): skips the prompt, decodes the running completion and queues the new
    text each time it is printable (a trailing U+FFFD means a multi-byte character is still incomplete)."""

    def __init__(self, tokenizer):
        self.tokenizer, self.q = tokenizer, queue.Queue()
        self._ids: List[int] = []
        self._emitted = 0
        self._prompt_seen = False

    def put(self, value: torch.Tensor) -> None:
        if not self._prompt_seen:
            self._prompt_seen = True
            return
        self._ids.extend(int(t) for t in value.reshape(-1).tolist())
        text = self.tokenizer.decode(self._ids, skip_special_tokens=True)
        if text.endswith("�"):
            return
        self.q.put((text[self._emitted:], len(self._ids)))
        self._emitted = len(text)

    def end(self) -> None:
        self.q.put(None)


class LocalInference:
    def __init__(self, model, processor: UltravoxProcessor, tokenizer, dtype: Optional[torch.dtype] = None,
                 conversation_mode: bool = False, chat_template: Optional[str] = None, enable_thinking: bool = False,
                 thinking_regex: Optional[str] = None):
        self.model, self.processor, self.tokenizer = model, processor, tokenizer
        self.dtype = dtype if dtype is not None else getattr(model, "dtype", torch.bfloat16)
        self.conversation_mode = conversation_mode
        self.past_messages: List[Dict[str, str]] = []
        self.past_key_values: Any = None
        self.data_collator = DataCollatorForSeq2SeqWithAudio(tokenizer=tokenizer, include_alt_fields=False)
        self.chat_template, self.enable_thinking, self.thinking_regex = chat_template, enable_thinking, thinking_regex
        assert self.tokenizer.padding_side == "left", "batched generation needs a left-padding tokenizer (infer.py:47)"

    # ---- conversation state (infer.py:49-91) ----
    def update_conversation(self, past_messages: Optional[List[Dict[str, str]]] = None, past_key_values: Any = None) -> None:
        self.past_messages = list(past_messages or [])
        self.past_key_values = past_key_values

    def _get_sample_with_past(self, sample: Optional[VoiceSample]) -> VoiceSample:
        if sample is None:
            if not self.past_messages:
                raise ValueError("No past messages available to generate a response.")
            return VoiceSample(list(self.past_messages))
        sample = copy.copy(sample)
        sample.add_past_messages(self.past_messages)
        return sample

    def _build_past_messages(self, query_messages: List[Dict[str, str]], audio_token_len: int,
                             response_content: str) -> List[Dict[str, str]]:
        messages = [dict(m) for m in query_messages]
        if audio_token_len > 0:
            user_content = messages[-1]["content"]
            n = user_content.count(AUDIO_PLACEHOLDER)
            if n != 1:
                raise ValueError(f"Expected 1 audio placeholder, found {n}")
            messages[-1]["content"] = user_content.replace(AUDIO_PLACEHOLDER, self.tokenizer.eos_token * audio_token_len)
        messages.append({"role": "assistant", "content": response_content})
        return messages

    def _postprocess_response(self, text: str) -> Tuple[str, Optional[str]]:
        """Split `<think>`-style content off the reply (infer.py:93-122)."""
        if not self.enable_thinking:
            return text, None
        if not self.thinking_regex:
            raise ValueError("thinking_regex is not set while enable_thinking is True")
        m = re.search(self.thinking_regex, text, re.DOTALL)
        if not m:
            raise ValueError(f"{self.thinking_regex} not matched in the response while thinking is enabled: {text}")
        return re.sub(self.thinking_regex, "", text, flags=re.DOTALL).strip(), m.group(1).strip()

    # ---- one sample -> model inputs (infer.py:267-307) ----
    def _dataproc(self, sample: VoiceSample, add_generation_prompt: bool = True) -> Dict[str, torch.Tensor]:
        kw = {} if self.chat_template is None else {"chat_template": self.chat_template}
        if self.enable_thinking:
            kw["enable_thinking"] = True
        text_input = self.tokenizer.apply_chat_template(sample.messages, add_generation_prompt=add_generation_prompt,
                                                        tokenize=False, **kw)
        audio_input = None
        if sample.audio is not None:
            audio = sample.audio
            if audio.dtype == np.int16:
                audio = audio / np.float32(32768.0)
            if audio.dtype not in (np.float64, np.float32):
                raise ValueError("Audio must be float64 or float32 or int16")
            if sample.sample_rate != SAMPLE_RATE:
                from scipy.signal import resample_poly
                g = math.gcd(SAMPLE_RATE, int(sample.sample_rate))
                audio = resample_poly(audio, SAMPLE_RATE // g, int(sample.sample_rate) // g, axis=-1).astype(audio.dtype)
            audio_input = torch.from_numpy(np.ascontiguousarray(audio))
            if audio_input.ndim == 2:
                audio_input = audio_input.squeeze(0)
        inputs = self.processor(audio=audio_input, text=text_input, return_tensors="pt", sampling_rate=SAMPLE_RATE)
        inputs = {k: v.to(self.model.device) for k, v in inputs.items()}
        if "audio_values" in inputs:
            inputs["audio_values"] = inputs["audio_values"].to(dtype=self.dtype)
        return inputs

    def _terminators(self) -> List[int]:
        ids = [self.tokenizer.eos_token_id]
        if "<|eot_id|>" in getattr(self.tokenizer, "added_tokens_encoder", {}):
            eot = self.tokenizer.convert_tokens_to_ids("<|eot_id|>")
            if eot not in ids:
                ids.append(eot)
        return ids

    def _generate(self, inputs: Dict[str, torch.Tensor], max_new_tokens: Optional[int] = None,
                  temperature: Optional[float] = None, streamer=None) -> Tuple[torch.Tensor, Any]:
        """-> (sequences, past_key_values or None); the cache is requested and threaded through in conversation mode only."""
        args: Dict[str, Any] = {"max_new_tokens": max_new_tokens or MAX_NEW_TOKENS}
        if temperature is not None and temperature > 0:
            args.update(do_sample=True, temperature=temperature)
        else:
            args.update(do_sample=False, top_p=None, top_k=None)
        if self.conversation_mode:
            args.update(past_key_values=self.past_key_values, return_dict_in_generate=True)
        out = self.model.generate(**inputs, **args, pad_token_id=self.tokenizer.eos_token_id,
                                  eos_token_id=self._terminators(), streamer=streamer)
        past = getattr(out, "past_key_values", None)
        if past is not None and hasattr(past, "partial_ok"):
            past.partial_ok = True      # a reply that re-tokenises differently must not cost the earlier (audio) turns
        return getattr(out, "sequences", out), past

    def _remember(self, sample: VoiceSample, inputs: Dict[str, torch.Tensor], response_text: str, past_key_values) -> None:
        if self.conversation_mode:
            atl = inputs.get("audio_token_len")
            n_audio = int(atl.reshape(-1)[0]) if atl is not None and atl.numel() else 0
            self.update_conversation(self._build_past_messages(sample.messages, n_audio, response_text), past_key_values)

    # ---- public API (infer.py:125-265) ----
    def infer(self, sample: Optional[VoiceSample] = None, max_tokens: Optional[int] = None,
              temperature: Optional[float] = None) -> VoiceOutput:
        extended = self._get_sample_with_past(sample)
        inputs = self._dataproc(extended)
        input_len = inputs["input_ids"].shape[1]
        sequences, past = self._generate(inputs, max_tokens, temperature)
        output_tokens = self._strip(sequences[0][input_len:])
        text, thinking = self._postprocess_response(self.tokenizer.decode(output_tokens, skip_special_tokens=True))
        self._remember(extended, inputs, text, past)
        return VoiceOutput(text, input_len, len(output_tokens), thinking_content=thinking)

    def _strip(self, tokens: Sequence[int]) -> List[int]:
        """Completion up to and including its first terminator (finished rows are padded with eos by `generate`)."""
        toks = [int(t) for t in tokens]
        stop = set(self._terminators())
        for i, t in enumerate(toks):
            if t in stop:
                return toks[:i + 1]
        return toks

    def infer_batch(self, samples: List[VoiceSample], max_tokens: Optional[int] = None,
                    temperature: Optional[float] = None) -> List[VoiceOutput]:
        assert not self.conversation_mode, "infer_batch does not support conversation mode (infer.py:152-159)"
        inputs = [self._dataproc(s) for s in samples]
        for item in inputs:
            for key, val in item.items():
                if not key.startswith("audio"):
                    item[key] = val.squeeze(0)
        tensors = {k: (v.to(self.model.device) if v is not None else v) for k, v in self.data_collator(inputs).items()}
        input_len = tensors["input_ids"].shape[1]
        out = []
        for row in self._generate(tensors, max_tokens, temperature)[0]:
            toks = self._strip(row[input_len:])
            text, thinking = self._postprocess_response(self.tokenizer.decode(toks, skip_special_tokens=True))
            out.append(VoiceOutput(text, input_len, len(toks), thinking_content=thinking))
        return out

    def infer_stream(self, sample: Optional[VoiceSample] = None, max_tokens: Optional[int] = None,
                     temperature: Optional[float] = None) -> InferenceGenerator:
        extended = self._get_sample_with_past(sample)
        inputs = self._dataproc(extended)
        input_tokens = inputs["input_ids"].shape[1]
        streamer = _TokenQueueStreamer(self.tokenizer)
        failure: List[BaseException] = []
        result: List[Any] = [None]

        def thunk():
            try:
                result[0] = self._generate(inputs, max_tokens, temperature, streamer=streamer)[1]
            except BaseException as e:      # surface the failure on the consumer's side, never hang the queue
                failure.append(e)
                streamer.end()

        thread = threading.Thread(target=thunk)
        thread.start()
        output_text, output_token_len = "", 0
        while True:
            item = streamer.q.get()
            if item is None:
                break
            chunk, output_token_len = item
            if chunk:
                output_text += chunk
                yield InferenceChunk(chunk)
        thread.join()
        if failure:
            raise failure[0]
        response_text, _ = self._postprocess_response(output_text)
        self._remember(extended, inputs, response_text, result[0])
        yield InferenceStats(input_tokens, output_token_len)
