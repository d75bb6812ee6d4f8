"""Host-side processor: the tensor/text contract of Videollama3Qwen2Processor for in-memory clips.

Restates (not imports) the reference's host preprocessing:
  simple_batched_resize      model/image_processing_videollama3.py:93-131   token-budgeted target sizes
  _preprocess / patchify     model/image_processing_videollama3.py:235-347  PIL bicubic, x/255, (x-.5)/.5,
                                                                            merge-window-major [N,588] rows
  chat template / <image>    model/chat_template.json, model/processing_cogreasoner.py:707-730,752-801
  history extraction         model/processing_cogreasoner.py:936-956
Video DECODING (ffmpeg/decord, processing_cogreasoner.py:326-429) is out of scope: clips arrive as
uint8 arrays + timestamps. The functions are numpy/PIL/str only; the one GPU hook is
CogStreamProcessor(device=...), which hands the frames to preprocess_gpu.py (same values, bit for bit) instead of
PIL. Nothing here imports the oracle."""
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


# ---- Pillow's bicubic resample, restated exactly (third-party arithmetic the reference calls through
# transformers.image_transforms.resize -> PIL.Image.resize(resample=BICUBIC); Pillow 10.4 src/libImaging/Resample.c:
# precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc). Two passes
# (horizontal first), double-precision coefficients normalised then fixed to 22 bits, uint8 rounding after EACH
# pass. The GPU preprocessing kernels (csrc/preprocess.hip) consume the tables built here. ----
_PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_coeffs(in_size: int, out_size: int):
    """-> (bounds int32 [out,2] = (first input index, tap count), coeffs int32 [out, ksize], ksize)"""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << _PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << _PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def resize_bicubic_exact(frame_hwc: np.ndarray, size: Tuple[int, int]) -> np.ndarray:
    """numpy restatement of PIL's RGB bicubic resize (bit-exact; tests pin it against PIL itself)"""
    th, tw = size
    h, w, c = frame_hwc.shape
    img = frame_hwc.astype(np.int64)
    half = 1 << (_PRECISION_BITS - 1)
    if tw != w:
        bx, kx, _ = resample_coeffs(w, tw)
        out = np.empty((h, tw, c), dtype=np.int64)
        for xx in range(tw):
            x0, n = bx[xx]
            out[:, xx, :] = half + (img[:, x0:x0 + n, :] * kx[xx, :n, None].astype(np.int64)).sum(axis=1)
        img = np.clip(out >> _PRECISION_BITS, 0, 255)
    if th != h:
        by, ky, _ = resample_coeffs(h, th)
        out = np.empty((th, img.shape[1], c), dtype=np.int64)
        for yy in range(th):
            y0, n = by[yy]
            out[yy] = half + (img[y0:y0 + n] * ky[yy, :n, None, None].astype(np.int64)).sum(axis=0)
        img = np.clip(out >> _PRECISION_BITS, 0, 255)
    return img.astype(np.uint8)


def pixel_value_table(rescale_factor: float = 0.00392156862745098, image_mean=(0.5, 0.5, 0.5),
                      image_std=(0.5, 0.5, 0.5)) -> np.ndarray:
    """fp32 [3,256]: what rescale + normalize make of each byte value per channel, in the reference's arithmetic
    (transformers image_transforms.rescale: float64 product rounded to fp32; normalize: fp32 (x-mean)/std;
    settings from model/preprocessor_config.json). Both the host path and the HIP kernel look values up here."""
    u = np.arange(256, dtype=np.uint8)
    x = (u.astype(np.float64) * rescale_factor).astype(np.float32)
    mean = np.array(image_mean, dtype=np.float32)[:, None]
    std = np.array(image_std, dtype=np.float32)[:, None]
    return np.ascontiguousarray(((x[None, :] - mean) / std).astype(np.float32))


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
        x = pixel_value_table()[np.arange(3), fr]                               # [t, th, tw, 3] fp32
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

    def __init__(self, tokenizer, video_merge_size: int = 2, max_tokens: int = 16384, min_tokens: int = 16,
                 device=None, pixel_dtype=None):
        """device=None: host PIL path (fp32 pixel_values on the CPU, as the reference returns them).
        device='cuda:N': frames are uploaded as bytes and pre-processed by the HIP kernels (same values bit for
        bit); pixel_values stay on the GPU in pixel_dtype (default bf16, the encoder's input dtype)."""
        self.tokenizer = tokenizer
        self.video_merge_size = video_merge_size
        self.max_tokens, self.min_tokens = max_tokens, min_tokens
        self.device, self.pixel_dtype = device, pixel_dtype
        self.video_loader = None      # path -> video_io.DecodedVideo; default video_io.read_decoded_video (.npz)

    @classmethod
    def from_pretrained(cls, path: str, device=None, pixel_dtype=None, trust_remote_code: bool = True, tokenizer=None,
                        **unused):
        """AutoProcessor.from_pretrained(model_path, trust_remote_code=True) (evaluate/answer_generate.py:179): the
        checkpoint's tokenizer + preprocessor_config.json / processor_config.json settings. `tokenizer`: use this
        object instead of loading vocab.json / merges.txt from the directory (tests: the replayed real tokenizer)"""
        from . import checkpoint as ck
        p = ck.load_configs(path)["processor"]
        if (p["patch_size"], p["resample"], list(p["image_mean"]), list(p["image_std"])) != (14, 3, [0.5] * 3, [0.5] * 3):
            raise ValueError(f"{path}: pre-processing settings other than the reference's (patch 14, bicubic, "
                             "mean/std 0.5) are not implemented")
        return cls(tokenizer if tokenizer is not None else ck.load_tokenizer(path), video_merge_size=p["video_merge_size"], max_tokens=p["max_tokens"],
                   min_tokens=p["min_tokens"], device=device, pixel_dtype=pixel_dtype)

    def __call__(self, conversation: List[Dict[str, Any]], add_system_prompt: bool = True,
                 add_generation_prompt: bool = True, return_tensors: str = "pt") -> Dict[str, Any]:
        import torch

        # file-backed clips ({"video": {"video_path": ..., "fps": 1, "max_frames": 180}}, evaluate/answer_generate.py:126):
        # sampled at fps, cut to max_frames, timestamps stitched across segments -- _load_multimodal_data (:431-509)
        if any(isinstance(c, dict) and c.get("type") == "video" and isinstance(c.get("video"), dict)
               for m in conversation if isinstance(m["content"], (list, tuple)) for c in m["content"]):
            from .video_io import load_multimodal_data, read_decoded_video
            paths = {c["video"]["video_path"] for m in conversation if isinstance(m["content"], (list, tuple))
                     for c in m["content"] if isinstance(c, dict) and c.get("type") == "video" and isinstance(c.get("video"), dict)}
            loader = self.video_loader or read_decoded_video
            conversation, _ = load_multimodal_data(conversation, {p: loader(p) for p in sorted(paths)})
        videos, all_ts, conv = [], [], []
        for msg in conversation:
            if isinstance(msg["content"], str):
                conv.append(msg)
                continue
            items = []
            for c in msg["content"]:
                if isinstance(c, dict) and c.get("type") == "video":
                    v = c["video"] if isinstance(c["video"], torch.Tensor) else np.asarray(c["video"])
                    ts = [float(t) for t in c.get("timestamps", range(len(v)))]
                    videos.append(v)
                    all_ts.extend(ts)
                    items.append({"type": "video", "num_frames": len(v), "timestamps": ts})
                else:
                    items.append(c)
            conv.append({"role": msg["role"], "content": items})
        feats = None
        if videos and self.device is not None:
            from .preprocess_gpu import preprocess_videos_gpu
            dev_videos = [(v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v)))
                          .to(self.device, non_blocking=True).contiguous() for v in videos]
            feats = preprocess_videos_gpu(dev_videos, self.video_merge_size, self.min_tokens, self.max_tokens,
                                          out_dtype=self.pixel_dtype or torch.bfloat16)
        elif videos:
            host = preprocess_videos([v.cpu().numpy() if isinstance(v, torch.Tensor) else v for v in videos],
                                     self.video_merge_size, self.min_tokens, self.max_tokens)
            feats = {k: torch.from_numpy(a) for k, a in host.items()}
        text = render_conversation(conv, add_system_prompt, add_generation_prompt)
        per_image: List[int] = []
        if feats is not None:
            for (t, gh, gw), ms in zip(feats["grid_sizes"].tolist(), feats["merge_sizes"].tolist()):
                per_image.extend([(gh // ms) * (gw // ms)] * t)
        text = expand_image_tokens(text, per_image)
        enc = self.tokenizer(text, return_tensors="pt")
        hist_qs, hist_as, cur_q = process_history_qas(conversation)
        # content keys of the video segments (xxh64 of the raw frame bytes) for the model's visual-token cache
        import xxhash
        video_keys = [xxhash.xxh64(memoryview(np.ascontiguousarray(v.cpu().numpy() if isinstance(v, torch.Tensor) else v))
                                   .cast("B")).hexdigest() for v in videos]   # hashed in place: no copy of the frames
        out = {
            "video_keys": video_keys,
            "input_ids": enc["input_ids"], "attention_mask": enc["attention_mask"],
            "modals": ["video"] * len(videos), "tokenizer": self.tokenizer,
            "hist_qs": hist_qs, "hist_as": hist_as, "current_question": cur_q,
            "all_timestamps": all_ts, "total_image_num": len(all_ts), "original_text": text,
        }
        if feats is not None:
            out["pixel_values"] = feats["pixel_values"]
            out["grid_sizes"] = feats["grid_sizes"]
            out["merge_sizes"] = feats["merge_sizes"]
        return out

    def batch_decode(self, ids, **kw):
        return self.tokenizer.batch_decode(ids, **kw)
