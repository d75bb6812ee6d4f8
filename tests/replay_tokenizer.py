"""The real Qwen2 tokenizer of the checkpoint, REPLAYED from tests/golden/tokenizer.json: every (text, ids) pair the
reference code tokenised while the fixture was made (tests/golden/make_golden.py::golden_tokenizer). Lookup only --
a text that is not in the record raises KeyError, which is exactly the failure wanted when the product builds a
prompt that differs from the reference's by one character. The vocabulary itself (vocab.json / merges.txt, 5 MB of
reference data) does not ship."""
from __future__ import annotations

import json
import os
from typing import List

import torch

FIXTURE = os.path.join(os.path.dirname(__file__), "golden", "tokenizer.json")


def unrle(rle) -> List[int]:
    out: List[int] = []
    for i, n in rle:
        out.extend([i] * n)
    return out


class _Enc(dict):
    def to(self, device):
        return _Enc({k: v.to(device) for k, v in self.items()})


class ReplayTokenizer:
    init_kwargs = {}

    def __init__(self, path: str = FIXTURE):
        self.data = json.load(open(path, encoding="utf-8"))
        self.by_text = {p["text"]: unrle(p["ids_rle"]) for p in self.data["pairs"]}
        for s, ids in self.data["strings"].items():
            self.by_text.setdefault(s, ids)
        self.by_ids = {tuple(v): k for k, v in self.by_text.items()}
        self.pad_token_id = self.data["endoftext"]

    def encode(self, text: str, add_special_tokens: bool = False) -> List[int]:
        return list(self.by_text[text])

    def __call__(self, text, return_tensors="pt", **kw):
        t = torch.tensor([self.encode(text)], dtype=torch.long)
        return _Enc(input_ids=t, attention_mask=torch.ones_like(t))

    def decode(self, ids, skip_special_tokens: bool = False) -> str:
        ids = ids.tolist() if hasattr(ids, "tolist") else list(ids)
        text = self.by_ids[tuple(ids)]
        if skip_special_tokens:
            for s in ("<|im_start|>", "<|im_end|>", "<|endoftext|>", "<image>"):
                text = text.replace(s, "")
        return text

    def batch_decode(self, batch, skip_special_tokens: bool = False):
        return [self.decode(b, skip_special_tokens) for b in batch]
