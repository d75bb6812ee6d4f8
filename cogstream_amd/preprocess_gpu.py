"""GPU version of Videollama3ImageProcessor.preprocess for in-memory clips:
the token-budget size logic stays on the host (cogstream_amd.processing.simple_batched_resize), the per-pixel
work -- Pillow-exact bicubic resize, rescale/normalise, merge-window patchify -- runs in csrc/preprocess.hip and
writes pixel_values directly in the encoder's dtype, so frames never round-trip through host fp32."""
from __future__ import annotations

import ctypes as C
from functools import lru_cache
from typing import Dict, Sequence

import numpy as np
import torch

from . import _lib as L
from .processing import PATCH, media_target_sizes, pixel_value_table, resample_coeffs
from .runtime import get_handle


@lru_cache(maxsize=64)
def _tables(in_size: int, out_size: int):
    b, k, ks = resample_coeffs(in_size, out_size)
    return np.ascontiguousarray(b), np.ascontiguousarray(k), ks


def preprocess_media_gpu(items: Sequence[torch.Tensor], merge_sizes: Sequence[int], min_tokens: int = 16,
                         max_tokens: int = 16384, out_dtype=torch.bfloat16) -> Dict[str, torch.Tensor]:
    """items: uint8 device tensors [t, H, W, 3] (an image = a one-frame item). Same result as
    processing.preprocess_media (bit-identical in fp32; bf16 output = that fp32 rounded once)."""
    dev = items[0].device
    if dev.type != "cuda":
        raise L.CogsError("preprocess_media_gpu needs frames on the GPU")
    shapes = [(int(v.shape[0]), int(v.shape[1]), int(v.shape[2])) for v in items]
    targets = media_target_sizes(shapes, list(merge_sizes), min_tokens, max_tokens)
    handle = get_handle(dev)
    table = torch.from_numpy(pixel_value_table()).to(dev)
    outs, grids = [], []
    for v, merge_size, (th, tw) in zip(items, merge_sizes, targets):
        assert v.dtype == torch.uint8 and v.is_contiguous() and v.shape[-1] == 3
        t, H, W = int(v.shape[0]), int(v.shape[1]), int(v.shape[2])
        bx, kx, ksx = _tables(W, tw)
        by, ky, ksy = _tables(H, th)
        dbx, dkx = torch.from_numpy(bx).to(dev), torch.from_numpy(kx).to(dev)
        dby, dky = torch.from_numpy(by).to(dev), torch.from_numpy(ky).to(dev)
        gh, gw = th // PATCH, tw // PATCH
        out = torch.empty(t * gh * gw, 3 * PATCH * PATCH, device=dev, dtype=out_dtype)
        n = C.c_size_t()
        L.check(L.lib.cogs_preprocess_workspace_bytes(t, H, tw, C.byref(n)))
        ws = handle.workspace("preprocess", n.value)
        L.check(L.lib.cogs_preprocess_frames(L.current_stream(), v.data_ptr(), t, H, W, th, tw, int(merge_size),
                                             dbx.data_ptr(), dkx.data_ptr(), ksx, dby.data_ptr(), dky.data_ptr(), ksy,
                                             table.data_ptr(), out.data_ptr(), L.dtype_code(out_dtype), ws.data_ptr(), ws.numel()),
                "cogs_preprocess_frames")
        outs.append(out)
        grids.append((t, gh, gw))
    return {"pixel_values": torch.cat(outs, dim=0) if len(outs) > 1 else outs[0],
            "grid_sizes": torch.tensor(grids, dtype=torch.int64),
            "merge_sizes": torch.tensor([int(m) for m in merge_sizes], dtype=torch.int64)}


def preprocess_videos_gpu(videos: Sequence[torch.Tensor], merge_size: int = 2, min_tokens: int = 16,
                          max_tokens: int = 16384, out_dtype=torch.bfloat16) -> Dict[str, torch.Tensor]:
    """preprocess_media_gpu for clips that all use one merge size"""
    return preprocess_media_gpu(videos, [merge_size] * len(videos), min_tokens, max_tokens, out_dtype)
