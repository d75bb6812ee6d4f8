"""Oracle: question-aware visual token compression (model/cogreasoner_chat.py:336-476,513-584).
TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py."""
from __future__ import annotations

from typing import List, Optional

import torch


def compression_mask(pixel_values, grid_sizes, merge_sizes, modals: List[str], threshold: float = 0.1,
                     min_tokens: int = 1, minor_frame_indices: Optional[List[int]] = None) -> torch.Tensor:
    """_get_compression_mask (cogreasoner_chat.py:383-432); arithmetic runs in pixel_values' dtype"""
    minor = set(minor_frame_indices or [])
    num_patches = grid_sizes.prod(dim=1).div(merge_sizes ** 2).long()
    masks = []
    gcount = 0
    for images, n, gs, ms, modal in zip(pixel_values.split(grid_sizes.prod(dim=1).tolist(), dim=0), num_patches.tolist(),
                                        grid_sizes.tolist(), merge_sizes.tolist(), modals):
        t, h, w = gs
        if modal == "image" or (modal == "video" and t == 1):
            masks.append(torch.ones((n,), dtype=torch.bool))
        elif modal == "video":
            im = images.view(t, (h // ms) * (w // ms), -1)
            diff = torch.abs(im[1:] - im[:-1]).mean(dim=-1) * 255
            diff = torch.cat([torch.full_like(diff[0:1], threshold + 1), diff], dim=0)
            m = diff > threshold
            pad = torch.nonzero(m.sum(dim=1) < min_tokens)[:, 0]
            m[pad, :min_tokens] = 1
            for f in range(t):
                if gcount + f in minor:
                    m[f, 0] = True
                    m[f, 1:] = False
            masks.append(m.flatten())
        else:
            masks.append(torch.ones((0,), dtype=torch.bool))
        gcount += t
    return torch.cat(masks)


def compress_unimportant_events(mm_features, patch_num: int, minor_frame_indices: List[int]) -> torch.Tensor:
    """cogreasoner_chat.py:434-447"""
    total, dim = mm_features.shape
    if total % patch_num != 0:
        raise ValueError("patch count not divisible")
    f = mm_features.view(total // patch_num, patch_num, dim).clone()
    for i in minor_frame_indices:
        f[i, 0, :] = f[i, :, :].mean(dim=0)
    return f.view(-1, dim)


def compress_visual_tokens(mask, mm_features, input_ids, attention_mask, image_token_index: int):
    """_compress_visual_tokens (cogreasoner_chat.py:449-476), inference subset (no labels/position_ids)"""
    mm = mm_features[mask]
    sel = input_ids == image_token_index
    text = torch.logical_not(sel)
    text[sel] = mask
    return mm, input_ids[text], (attention_mask[text] if attention_mask is not None else None)


def scatter_embeds(embed_table, input_ids, mm_features, image_token_index: int) -> torch.Tensor:
    """cogreasoner_chat.py:567-572"""
    e = embed_table[input_ids].clone()
    sel = input_ids == image_token_index
    e[sel] = e[sel] * 0.0 + mm_features.to(e.dtype)
    return e
