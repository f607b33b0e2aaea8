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
