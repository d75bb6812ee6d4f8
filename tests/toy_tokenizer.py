"""Tiny deterministic tokenizer with the slice of the HF interface the hot path uses (the real Qwen2
vocab/merges are reference data that does not ship). Byte-level: ids 0..255 are UTF-8
bytes, the chat specials follow. Used by the tests on both sides (the reference takes `tokenizer=` as an
argument, so the same object drove the golden generation)."""
from __future__ import annotations

import re
from typing import List

import torch

SPECIALS = ["<|im_start|>", "<|im_end|>", "<image>", "<|endoftext|>"]
IM_START, IM_END, IMAGE, EOT = 256, 257, 258, 259
VOCAB = 512


class _Enc(dict):
    def to(self, device):
        return _Enc({k: v.to(device) for k, v in self.items()})


class ToyTokenizer:
    init_kwargs = {}
    pad_token_id = EOT

    def __init__(self):
        self._split = re.compile("(" + "|".join(re.escape(s) for s in SPECIALS) + ")")

    def encode(self, text: str, add_special_tokens: bool = False) -> List[int]:
        out: List[int] = []
        for part in self._split.split(text):
            if not part:
                continue
            if part in SPECIALS:
                out.append(256 + SPECIALS.index(part))
            else:
                out.extend(part.encode("utf-8"))
        return out

    def __call__(self, text, return_tensors="pt", padding=False, truncation=False, max_length=None, **kw):
        ids = self.encode(text)
        if truncation and max_length is not None:
            ids = ids[:max_length]
        t = torch.tensor([ids], dtype=torch.long)
        return _Enc(input_ids=t, attention_mask=torch.ones_like(t))

    def decode(self, ids, skip_special_tokens: bool = False) -> str:
        ids = ids.tolist() if hasattr(ids, "tolist") else list(ids)
        buf, out = bytearray(), []
        for i in ids:
            if i < 256:
                buf.append(i)
            else:
                out.append(buf.decode("utf-8", errors="replace"))
                buf = bytearray()
                if not skip_special_tokens and i - 256 < len(SPECIALS):
                    out.append(SPECIALS[i - 256])
        out.append(buf.decode("utf-8", errors="replace"))
        return "".join(out)

    def batch_decode(self, batch, skip_special_tokens: bool = False):
        return [self.decode(b, skip_special_tokens) for b in batch]
