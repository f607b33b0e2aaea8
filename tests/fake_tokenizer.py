"""Deterministic stand-in for the Llama-3 tokenizer (the real one is an un-materialised LFS pointer in
the reference snapshot and cannot be fetched offline).  Splits like a byte-level BPE does around
spaces — "Test with " -> ["Test", " with", " "], " and " -> [" and", " "] — so the reference's index
vectors (ultravox_processing_test.py:46-137: starts [3], [3, 191], [3, 12, 200, 234] ...) replay verbatim;
token ids are stable hashes, only their structure is asserted."""
import re
import zlib
from typing import Dict, List


class FakeTokenizer:
    eos_token = "<|eot_id|>"
    eos_token_id = 128009
    pad_token_id = None
    padding_side = "right"
    model_input_names = ["input_ids", "attention_mask"]

    def __init__(self, padding_side: str = "right"):
        self.padding_side = padding_side
        self._vocab: Dict[str, int] = {self.eos_token: self.eos_token_id}

    def get_vocab(self) -> Dict[str, int]:
        return self._vocab

    def _tok(self, text: str) -> List[int]:
        return [10 + zlib.crc32(p.encode()) % 100000 for p in re.findall(r"\s?\S+|\s+", text)]

    def __call__(self, texts, add_special_tokens: bool = False, **kw):
        if isinstance(texts, str):
            texts = [texts]
        return {"input_ids": [self._tok(t) for t in texts]}


class FakeChatTokenizer(FakeTokenizer):
    """FakeTokenizer + what the inference wrapper touches: a Llama-3-shaped chat template, decode, terminators."""
    added_tokens_encoder = {"<|eot_id|>": 128009}

    def __init__(self, padding_side: str = "left"):
        super().__init__(padding_side)
        self._text: Dict[int, str] = {}

    def _tok(self, text: str) -> List[int]:
        ids = []
        for piece in re.findall(r"<\|[a-z_]+\|>|\s?[^\s<]+|\s+|<", text):
            tid = self._vocab.get(piece)
            if tid is None:
                tid = 10 + zlib.crc32(piece.encode()) % 100000
            self._text[tid] = piece
            ids.append(tid)
        return ids

    def apply_chat_template(self, messages, add_generation_prompt=False, tokenize=False, **kw):
        s = "<|begin_of_text|>" + "".join(f"<|start_header_id|>{m['role']}<|end_header_id|>\n\n{m['content']}<|eot_id|>"
                                           for m in messages)
        return s + ("<|start_header_id|>assistant<|end_header_id|>\n\n" if add_generation_prompt else "")

    def convert_tokens_to_ids(self, tok: str) -> int:
        return self.added_tokens_encoder[tok]

    def decode(self, ids, skip_special_tokens: bool = False) -> str:
        out = []
        for t in ids:
            piece = self._text.get(int(t), f"<{int(t)}>")
            if skip_special_tokens and (int(t) == self.eos_token_id or re.fullmatch(r"<\|[a-z_]+\|>", piece)):
                continue
            out.append(piece)
        return "".join(out)

    def pad(self, *a, **k):
        raise NotImplementedError
