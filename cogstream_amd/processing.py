"""Host-side processor: the tensor/text contract of Videollama3Qwen2Processor for in-memory clips.

Restates (not imports) the reference's host preprocessing:
  simple_batched_resize      model/image_processing_videollama3.py:93-131   token-budgeted target sizes
  _preprocess / patchify     model/image_processing_videollama3.py:235-347  PIL bicubic, x/255, (x-.5)/.5,
                                                                            merge-window-major [N,588] rows
  chat template / <image>    model/chat_template.json, model/processing_cogreasoner.py:707-730,752-801
  history extraction         model/processing_cogreasoner.py:936-956
Video DECODING (ffmpeg/decord, processing_cogreasoner.py:326-429) is out of scope: clips arrive as
uint8 arrays + timestamps. This module is numpy/PIL/str only; nothing here
touches the GPU, and nothing here imports the oracle."""
from __future__ import annotations

import math
import re
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

PATCH = 14
IMAGE_TOKEN = "<image>"
DEFAULT_SYSTEM = ("You are VideoLLaMA3 created by Alibaba DAMO Academy, a helpful assistant to help people "
                  "understand images and videos.")


def simple_batched_resize(sizes: Sequence[Tuple[int, int]], num_images: int, factor: int = 28,
                          min_tokens: int = 16, max_tokens: int = 16384) -> List[Tuple[int, int]]:
    """image_processing_videollama3.py:93-131; `sizes` = (height, width) of each image / first video frame,
    `num_images` = total frame count over the request (the budget is shared)."""
    min_pixels = min_tokens * factor * factor
    max_pixels = max_tokens * factor * factor
    out = []
    for height, width in sizes:
        h_bar = round(height / factor) * factor
        w_bar = round(width / factor) * factor
        if h_bar * w_bar > (max_pixels // num_images):
            beta = math.sqrt((height * width) / (max_pixels // num_images))
            h_bar = math.floor(height / beta / factor) * factor
            w_bar = math.floor(width / beta / factor) * factor
        if h_bar * w_bar < min_pixels:
            beta = math.sqrt(min_pixels / (height * width))
            h_bar = math.ceil(height * beta / factor) * factor
            w_bar = math.ceil(width * beta / factor) * factor
        out.append((h_bar, w_bar))
    return out


def patchify(frames: np.ndarray, merge_size: int) -> np.ndarray:
    """[t, c, H, W] -> [t*gh*gw, c*14*14], rows ordered frame -> merge-row -> merge-col -> window (dy,dx),
    elements (c, py, px)   (image_processing_videollama3.py:326-345)"""
    t, c, H, W = frames.shape
    gh, gw = H // PATCH, W // PATCH
    p = frames.reshape(t, c, gh // merge_size, merge_size, PATCH, gw // merge_size, merge_size, PATCH)
    p = p.transpose(0, 2, 5, 3, 6, 1, 4, 7)
    return p.reshape(t * gh * gw, c * PATCH * PATCH)


def _resize_bicubic(frame_hwc: np.ndarray, size: Tuple[int, int]) -> np.ndarray:
    from PIL import Image

    h, w = size
    return np.asarray(Image.fromarray(frame_hwc).resize((w, h), resample=Image.BICUBIC))


def preprocess_videos(videos: Sequence[np.ndarray], merge_size: int = 2, min_tokens: int = 16,
                      max_tokens: int = 16384) -> Dict[str, np.ndarray]:
    """Videollama3ImageProcessor.preprocess (:349-473) for a list of clips, each uint8 [t, H, W, 3].
    Returns pixel_values fp32 [N,588], grid_sizes int64 [V,3], merge_sizes int64 [V]."""
    num_images = sum(int(v.shape[0]) for v in videos)
    sizes = [(int(v.shape[1]), int(v.shape[2])) for v in videos]
    targets = simple_batched_resize(sizes, num_images, PATCH * merge_size, min_tokens, max_tokens)
    pix, grids = [], []
    for v, (th, tw) in zip(videos, targets):
        fr = np.stack([_resize_bicubic(f, (th, tw)) for f in v])                # [t, th, tw, 3] uint8
        x = fr.astype(np.float32) * np.float32(0.00392156862745098)            # rescale_factor
        x = (x - np.float32(0.5)) / np.float32(0.5)                              # image_mean/std = 0.5
        x = x.transpose(0, 3, 1, 2)
        pix.append(patchify(x, merge_size))
        grids.append((v.shape[0], th // PATCH, tw // PATCH))
    return {
        "pixel_values": np.concatenate(pix, axis=0).astype(np.float32),
        "grid_sizes": np.asarray(grids, dtype=np.int64),
        "merge_sizes": np.asarray([merge_size] * len(videos), dtype=np.int64),
    }


def synthetic_clip(num_frames: int, height: int = 480, width: int = 854, kind: str = "noise",
                   clip_idx: int = 0) -> Tuple[np.ndarray, List[float]]:
    """Synthetic 480p clips of SURVEY.md section 8d: 'noise' = iid U[0,255] (every token survives the
    pixel-diff pruning), 'drift' = static base + moving 25%-area square + unit noise. 1 fps timestamps."""
    rng = np.random.default_rng(20250824 + clip_idx)
    if kind == "noise":
        frames = rng.integers(0, 256, size=(num_frames, height, width, 3), dtype=np.uint8)
    elif kind == "drift":
        base = rng.integers(0, 256, size=(height, width, 3)).astype(np.int16)
        side = int(math.sqrt(0.25 * height * width))
        sq = rng.integers(0, 256, size=(side, side, 3)).astype(np.int16)
        frames = np.empty((num_frames, height, width, 3), dtype=np.uint8)
        for i in range(num_frames):
            f = base.copy()
            x0 = (8 * i) % max(width - side, 1)
            y0 = (height - side) // 2
            f[y0:y0 + side, x0:x0 + side] = sq
            f = f + np.rint(rng.normal(0, 1, size=f.shape)).astype(np.int16)
            frames[i] = np.clip(f, 0, 255).astype(np.uint8)
    else:
        raise ValueError(kind)
    return frames, [float(i) for i in range(num_frames)]


# ----------------------------------------------------------------------------- text side

def _round1(x: float) -> str:
    # jinja `| round(1) | string`
    return str(round(float(x), 1))


def render_conversation(conversation: List[Dict[str, Any]], add_system_prompt: bool = True,
                        add_generation_prompt: bool = True) -> str:
    """model/chat_template.json rendered for role/content dict lists: video items
    carry num_frames (+ timestamps), image items an optional timestamp."""
    out = []
    for i, msg in enumerate(conversation):
        if add_system_prompt and i == 0 and msg["role"] != "system":
            out.append("<|im_start|>system\n" + DEFAULT_SYSTEM + "<|im_end|>\n")
        ident = "stream" if msg["role"] == "stream" else "im"
        out.append(f"<|{ident}_start|>" + msg["role"] + "\n")
        content = msg["content"]
        if isinstance(content, str):
            out.append(content + f"<|{ident}_end|>\n")
            continue
        for c in content:
            if isinstance(c, str):
                out.append(c)
            elif c.get("type") == "text" or "text" in c:
                out.append(c["text"])
            elif c.get("type") == "image" or "image" in c:
                if "timestamp" in c:
                    out.append("Time " + _round1(c["timestamp"]) + "s: ")
                out.append(IMAGE_TOKEN + "\n")
            elif c.get("type") == "video" or "video" in c:
                n = c["num_frames"]
                for k in range(n):
                    if "timestamps" in c:
                        out.append("Time " + _round1(c["timestamps"][k]) + "s:")
                    out.append(IMAGE_TOKEN + ("," if k < n - 1 else "\n"))
        out.append(f"<|{ident}_end|>" + ("" if ident == "stream" else "\n"))
    if add_generation_prompt:
        out.append("<|im_start|>assistant\n")
    return "".join(out)


def expand_image_tokens(text: str, tokens_per_image: Sequence[int]) -> str:
    """process_text (processing_cogreasoner.py:707-730): the i-th <image> becomes tokens_per_image[i] copies"""
    parts = text.split(IMAGE_TOKEN)
    if len(parts) - 1 != len(tokens_per_image):
        raise AssertionError(f"{len(parts) - 1} <image> placeholders vs {len(tokens_per_image)} images")
    out = [parts[0]]
    for n, rest in zip(tokens_per_image, parts[1:]):
        out.append(IMAGE_TOKEN * int(n))
        out.append(rest)
    return "".join(out)


def process_history_qas(conversation: List[Dict[str, Any]]) -> Tuple[List[str], List[str], str]:
    """processing_cogreasoner.py:936-956: past (question, answer) strings and the current question"""
    qs: List[str] = []
    ans: List[Any] = []
    for msg in conversation:
        role, content = msg.get("role"), msg.get("content")
        if role == "user":
            if isinstance(content, str):
                qs.append(content)
            elif isinstance(content, list):
                qs.extend(c.get("text") for c in content if isinstance(c, dict) and c.get("type") == "text")
        elif role == "assistant":
            ans.append(content)
    return qs[:-1], ans, (qs[-1] if qs else "")


class CogStreamProcessor:
    """Videollama3Qwen2Processor.__call__ for in-memory clips (processing_cogreasoner.py:732-744,665):
    conversation items of type 'video' carry {'video': uint8 [t,H,W,3], 'timestamps': [...]}."""

    def __init__(self, tokenizer, video_merge_size: int = 2, max_tokens: int = 16384, min_tokens: int = 16):
        self.tokenizer = tokenizer
        self.video_merge_size = video_merge_size
        self.max_tokens, self.min_tokens = max_tokens, min_tokens

    def __call__(self, conversation: List[Dict[str, Any]], add_system_prompt: bool = True,
                 add_generation_prompt: bool = True, return_tensors: str = "pt") -> Dict[str, Any]:
        import torch

        videos, all_ts, conv = [], [], []
        for msg in conversation:
            if isinstance(msg["content"], str):
                conv.append(msg)
                continue
            items = []
            for c in msg["content"]:
                if isinstance(c, dict) and c.get("type") == "video":
                    v = np.asarray(c["video"])
                    ts = [float(t) for t in c.get("timestamps", range(len(v)))]
                    videos.append(v)
                    all_ts.extend(ts)
                    items.append({"type": "video", "num_frames": len(v), "timestamps": ts})
                else:
                    items.append(c)
            conv.append({"role": msg["role"], "content": items})
        feats = preprocess_videos(videos, self.video_merge_size, self.min_tokens, self.max_tokens) if videos else None
        text = render_conversation(conv, add_system_prompt, add_generation_prompt)
        per_image: List[int] = []
        if feats is not None:
            for (t, gh, gw), ms in zip(feats["grid_sizes"].tolist(), feats["merge_sizes"].tolist()):
                per_image.extend([(gh // ms) * (gw // ms)] * t)
        text = expand_image_tokens(text, per_image)
        enc = self.tokenizer(text, return_tensors="pt")
        hist_qs, hist_as, cur_q = process_history_qas(conversation)
        out = {
            "input_ids": enc["input_ids"], "attention_mask": enc["attention_mask"],
            "modals": ["video"] * len(videos), "tokenizer": self.tokenizer,
            "hist_qs": hist_qs, "hist_as": hist_as, "current_question": cur_q,
            "all_timestamps": all_ts, "total_image_num": len(all_ts), "original_text": text,
        }
        if feats is not None:
            out["pixel_values"] = torch.from_numpy(feats["pixel_values"])
            out["grid_sizes"] = torch.from_numpy(feats["grid_sizes"])
            out["merge_sizes"] = torch.from_numpy(feats["merge_sizes"])
        return out

    def batch_decode(self, ids, **kw):
        return self.tokenizer.batch_decode(ids, **kw)
