"""Tokenizer access for the entry points. The reference uses `LlamaTokenizer.from_pretrained("lmsys/vicuna-7b-v1.3")`
(demo.py:224, test.py:291) with pad = unk (id 0), left padding, and `<IMG>` added as id 32000. No tokenizer model is
reachable offline, so `load_tokenizer` falls back to a deterministic stand-in with the same interface and the same
special ids (BOS 1, EOS 2, pad/unk 0, <IMG> 32000); generated ids then decode to placeholder words."""
from __future__ import annotations

import os
import re
import zlib
from typing import List

import torch

IMG_TOKEN, IMG_ID = "<IMG>", 32000


class SyntheticTokenizer:
    pad_token_id, unk_token_id, bos_token_id, eos_token_id = 0, 0, 1, 2
    pad_token = unk_token = "<unk>"
    padding_side = "left"

    def __init__(self, vocab_size: int = 32000):
        self.vocab_size = vocab_size
        self._seen = {}                                 # id -> piece for everything this instance has encoded

    def __len__(self):
        return self.vocab_size + 1                      # + <IMG>

    def add_special_tokens(self, d):
        return 1

    def _encode(self, text: str) -> List[int]:
        ids = [self.bos_token_id]
        for piece in re.findall(r"<IMG>|\w+:?|[^\w\s]", text):
            i = IMG_ID if piece == IMG_TOKEN else 3 + zlib.crc32(piece.encode()) % (self.vocab_size - 3)
            self._seen.setdefault(i, piece)
            ids.append(i)
        return ids

    def __call__(self, text, return_tensors="pt", padding=False, **_):
        texts = [text] if isinstance(text, str) else list(text)
        enc = [self._encode(t) for t in texts]
        T = max(len(e) for e in enc)
        ids = torch.tensor([[self.pad_token_id] * (T - len(e)) + e for e in enc], dtype=torch.long)
        return {"input_ids": ids, "attention_mask": ids.ne(self.pad_token_id).long()}

    batch_encode_plus = __call__

    def batch_decode(self, sequences, skip_special_tokens=True):
        out = []
        for row in sequences.tolist():
            words = []
            for t in row:
                if skip_special_tokens and t in (0, 1, 2):
                    continue
                words.append(IMG_TOKEN if t == IMG_ID else self._seen.get(t, f"tok{t}"))
            out.append(" ".join(words))
        return out


def load_tokenizer(path: str = None):
    if path and os.path.isdir(path):
        from transformers import LlamaTokenizer
        tok = LlamaTokenizer.from_pretrained(path, use_fast=False, truncation_side="left", padding_side="left")
        tok.pad_token = tok.unk_token
        tok.add_special_tokens({"additional_special_tokens": [IMG_TOKEN]})
        return tok
    return SyntheticTokenizer()
