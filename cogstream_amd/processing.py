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


def batched_resize(shapes: Sequence[Tuple[int, int, int]], factors: Sequence[int], min_tokens: int = 16,
                   max_tokens: int = 16384) -> List[Tuple[int, int]]:
    """image_processing_videollama3.py:134-173 -- the size rule for a batch whose items have DIFFERENT merge sizes (an
    image next to a clip): `shapes` = (frames, height, width) per item, `factors` = 14 * merge_size per item. One
    common scale sqrt(total_tokens / max_tokens) when the batch is over budget, else round to the factor."""
    total = sum(n * math.ceil(h / f) * math.ceil(w / f) for (n, h, w), f in zip(shapes, factors))
    out = []
    if total > max_tokens:
        beta = math.sqrt(total / max_tokens)
        for (_, h, w), f in zip(shapes, factors):
            out.append((math.floor(h / beta / f) * f, math.floor(w / beta / f) * f))
    else:
        for (_, h, w), f in zip(shapes, factors):
            out.append((round(h / f) * f, round(w / f) * f))
    return out


def media_target_sizes(shapes: Sequence[Tuple[int, int, int]], merge_sizes: Sequence[int], min_tokens: int = 16,
                       max_tokens: int = 16384) -> List[Tuple[int, int]]:
    """Videollama3ImageProcessor.preprocess's choice (:424-439): one merge size for every item -> simple_batched_resize
    with factor 14 * merge (the budget is shared by all frames); mixed merge sizes -> batched_resize"""
    if all(m == merge_sizes[0] for m in merge_sizes):
        return simple_batched_resize([(h, w) for _, h, w in shapes], sum(n for n, _, _ in shapes),
                                     PATCH * merge_sizes[0], min_tokens, max_tokens)
    return batched_resize(shapes, [PATCH * m for m in merge_sizes], min_tokens, max_tokens)


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


def preprocess_media(items: Sequence[np.ndarray], merge_sizes: Sequence[int], min_tokens: int = 16,
                     max_tokens: int = 16384) -> Dict[str, np.ndarray]:
    """Videollama3ImageProcessor.preprocess (:349-473) for a list of clips / images, each uint8 [t, H, W, 3] (an image
    is a one-frame item, merge size 1 in the shipped processor: processing_cogreasoner.py:235-236).
    Returns pixel_values fp32 [N,588], grid_sizes int64 [V,3], merge_sizes int64 [V]."""
    shapes = [(int(v.shape[0]), int(v.shape[1]), int(v.shape[2])) for v in items]
    targets = media_target_sizes(shapes, list(merge_sizes), min_tokens, max_tokens)
    pix, grids = [], []
    for v, ms, (th, tw) in zip(items, merge_sizes, targets):
        fr = np.stack([_resize_bicubic(f, (th, tw)) for f in v])                # [t, th, tw, 3] uint8
        x = pixel_value_table()[np.arange(3), fr]                               # [t, th, tw, 3] fp32
        x = x.transpose(0, 3, 1, 2)
        pix.append(patchify(x, ms))
        grids.append((v.shape[0], th // PATCH, tw // PATCH))
    return {
        "pixel_values": np.concatenate(pix, axis=0).astype(np.float32),
        "grid_sizes": np.asarray(grids, dtype=np.int64),
        "merge_sizes": np.asarray(list(merge_sizes), dtype=np.int64),
    }


def preprocess_videos(videos: Sequence[np.ndarray], merge_size: int = 2, min_tokens: int = 16,
                      max_tokens: int = 16384) -> Dict[str, np.ndarray]:
    """preprocess_media for clips that all use one merge size"""
    return preprocess_media(videos, [merge_size] * len(videos), min_tokens, max_tokens)


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


def content_key(v) -> str:
    """content key of one clip / image for the model's visual-token cache (CogReasoner.enable_visual_cache). Host arrays:
    xxh64 of the raw bytes, hashed in place. Tensors already on the GPU are fingerprinted THERE (two position-weighted
    64-bit sums over the bytes, ~0.1 ms for a 256-frame 480p clip) -- copying 315 MB back to the host just to hash them
    cost 58 ms per request, more than encoding the clip."""
    import torch
    if isinstance(v, torch.Tensor) and v.is_cuda:
        b = v.contiguous().view(torch.uint8).reshape(-1)
        n = b.numel()
        pad = (-n) % 8
        if pad:
            b = torch.cat([b, b.new_zeros(pad)])
        w = b.view(torch.int64)
        i = torch.arange(w.numel(), device=w.device, dtype=torch.int64)
        # odd multipliers that depend on the position (wrapping int64 arithmetic): a swap or a change of any word moves both sums
        s1 = (w * (2 * i + 0x9E3779B97F4A7C15 % (1 << 62) | 1)).sum()
        s2 = ((w ^ (w >> 29)) * ((i * 0x2545F4914F6CDD1D % (1 << 62)) | 1)).sum()
        a, c = (int(x) & 0xFFFFFFFFFFFFFFFF for x in torch.stack([s1, s2]).tolist())
        return f"gpu:{a:016x}{c:016x}:{tuple(v.shape)}"
    import xxhash
    arr = np.ascontiguousarray(v.cpu().numpy() if isinstance(v, torch.Tensor) else v)
    return xxhash.xxh64(memoryview(arr).cast("B")).hexdigest()


def _as_frames(x) -> Any:
    """an image / clip as handed over by a caller -> uint8 [t, H, W, 3] (numpy, or a torch tensor left where it is)"""
    import torch
    if isinstance(x, torch.Tensor):
        return x if x.ndim == 4 else x[None]
    if isinstance(x, (list, tuple)):
        x = np.stack([np.asarray(f) for f in x])
    x = np.asarray(x)
    return x if x.ndim == 4 else x[None]


class CogStreamProcessor:
    """Videollama3Qwen2Processor (processing_cogreasoner.py:222-744) for in-memory media:

        __call__(text=None, conversation=None, images=None, return_labels=False, **kwargs)             :732-744
            conversation -> _process_conversation (:633-665): input_ids, attention_mask, pixel_values, grid_sizes,
                            merge_sizes, modals, tokenizer, hist_qs, hist_as, current_question, all_timestamps,
                            total_image_num, original_text
            text         -> _process_plain (:667-694): input_ids, attention_mask (+ the image keys when images are given)
        kwargs: add_system_prompt / add_generation_prompt (default False, the reference's chat_template_kwargs
        defaults :213-217; the driver passes True, evaluate/answer_generate.py:62-67), return_tensors ("pt" only).

    Conversation items: {"type": "video", "video": uint8 [t,H,W,3] | {"video_path": ..}, "timestamps": [...]},
    {"type": "image", "image": uint8 [H,W,3], "timestamp": optional}, {"type": "text", "text": ...}. A video uses
    video_merge_size (2), an image image_merge_size (1), as the shipped processor does (:235-236,:697-701).
    `images=`: the media themselves in conversation order -- [("video", frames), ("image", img), ...] or bare arrays
    (4-D = video) -- for callers that keep them out of the conversation, whose video items then carry "num_frames"
    (and "timestamps"). In the reference this argument dies with a NameError (:639-641,655-665: all_timestamps / text
    are unbound on that path); here it works. return_labels=True builds training targets (:520-608): training is
    outside the inference path this package rebuilds, so it raises NotImplementedError."""

    def __init__(self, tokenizer, video_merge_size: int = 2, max_tokens: int = 16384, min_tokens: int = 16,
                 device=None, pixel_dtype=None, image_merge_size: int = 1):
        """device=None: host PIL path (fp32 pixel_values on the CPU, as the reference returns them).
        device='cuda:N': frames are uploaded as bytes and pre-processed by the HIP kernels (same values bit for
        bit); pixel_values stay on the GPU in pixel_dtype (default bf16, the encoder's input dtype)."""
        self.tokenizer = tokenizer
        self.video_merge_size, self.image_merge_size = video_merge_size, image_merge_size
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
                   min_tokens=p["min_tokens"], device=device, pixel_dtype=pixel_dtype,
                   image_merge_size=p.get("image_merge_size", 1))

    # ------------------------------------------------------------------ images
    def process_images(self, images, merge_size=None) -> Dict[str, Any]:
        """:689-705 -- images: [("video"|"image", data), ...] or bare arrays / tensors (4-D = video, 3-D = image)
        -> pixel_values, grid_sizes, merge_sizes, modals"""
        import torch
        named = []
        if isinstance(images, tuple) and len(images) == 2 and isinstance(images[0], str):
            images = [images]
        elif not isinstance(images, (list, tuple)):
            images = [images]
        for it in images:
            if isinstance(it, (list, tuple)) and len(it) == 2 and isinstance(it[0], str):
                if it[0] not in ("image", "video"):
                    raise ValueError(f"Could not make batched images from {it[0]!r}")
                named.append((it[0], _as_frames(it[1])))
            else:
                fr = it if isinstance(it, torch.Tensor) else (np.stack([np.asarray(f) for f in it]) if isinstance(it, (list, tuple)) else np.asarray(it))
                if fr.ndim not in (3, 4):
                    raise ValueError(f"Could not make batched images from an array of shape {tuple(fr.shape)}")
                named.append(("video" if fr.ndim == 4 else "image", _as_frames(fr)))
        modals = [m for m, _ in named]
        merges = ([self.image_merge_size if m == "image" else self.video_merge_size for m in modals] if merge_size is None
                  else ([int(merge_size)] * len(named) if isinstance(merge_size, int) else [int(x) for x in merge_size]))
        if len(merges) != len(named):
            raise AssertionError("Merge size must be the same length as images.")
        frames = [f for _, f in named]
        if self.device is not None:
            from .preprocess_gpu import preprocess_media_gpu
            dev_items = [(v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v)))
                         .to(self.device, non_blocking=True).contiguous() for v in frames]
            feats = preprocess_media_gpu(dev_items, merges, self.min_tokens, self.max_tokens,
                                         out_dtype=self.pixel_dtype or torch.bfloat16)
        else:
            host = preprocess_media([v.cpu().numpy() if isinstance(v, torch.Tensor) else v for v in frames], merges,
                                    self.min_tokens, self.max_tokens)
            feats = {k: torch.from_numpy(a) for k, a in host.items()}
        feats["modals"] = modals
        feats["_frames"] = frames
        return feats

    def process_text(self, text: str, image_inputs: Dict[str, Any]):
        """:707-730 -- the i-th <image> becomes as many placeholders as image i has merged tokens -> (encoding, text)"""
        per_image: List[int] = []
        if image_inputs:
            for (t, gh, gw), ms in zip(image_inputs["grid_sizes"].tolist(), image_inputs["merge_sizes"].tolist()):
                per_image.extend([(gh // ms) * (gw // ms)] * t)
        text = expand_image_tokens(text, per_image)
        return self.tokenizer(text, return_tensors="pt"), text

    # ------------------------------------------------------------------ entry points
    def __call__(self, text: Optional[str] = None, conversation: Optional[List[Dict[str, Any]]] = None, images=None,
                 return_labels: bool = False, **kwargs) -> Dict[str, Any]:
        """:732-744"""
        if conversation is not None:
            if text is not None:
                raise ValueError("You cannot provide 'message' with 'text'.")
            return self._process_conversation(conversation, images, return_labels, **kwargs)
        return self._process_plain(text, images, return_labels, **kwargs)

    def _process_plain(self, text=None, images=None, return_labels: bool = False, **kwargs) -> Dict[str, Any]:
        """:667-694 (the reference unpacks process_text's (encoding, text) pair as a mapping there and cannot return;
        this is what it evidently means to return)"""
        if text is None:
            raise ValueError("You must provide 'text' or 'message'.")
        if return_labels:
            raise ValueError("return_labels is not supported for plain text processing.")
        if kwargs.get("return_tensors", "pt") != "pt":
            raise ValueError("only return_tensors='pt' is supported")
        image_inputs = self.process_images(images, kwargs.get("merge_size")) if images is not None else {}
        enc, _ = self.process_text(text, image_inputs)
        image_inputs.pop("_frames", None)
        return {"input_ids": enc["input_ids"], "attention_mask": enc["attention_mask"], **image_inputs}

    def _process_conversation(self, conversation, images=None, return_labels: bool = False, add_system_prompt: bool = False,
                              add_generation_prompt: bool = False, return_tensors: str = "pt", **kwargs) -> Dict[str, Any]:
        """:633-665"""
        import torch
        assert isinstance(conversation, list), "Conversation must be a list of messages."
        if return_labels:
            raise NotImplementedError("return_labels=True builds training targets (processing_cogreasoner.py:520-608); "
                                      "training is outside the inference path of this package")
        if return_tensors != "pt":
            raise ValueError("only return_tensors='pt' is supported")
        # file-backed clips ({"video": {"video_path": ..., "fps": 1, "max_frames": 180}}, evaluate/answer_generate.py:126):
        # sampled at fps, cut to max_frames, timestamps stitched across segments -- _load_multimodal_data (:431-509)
        if any(isinstance(c, dict) and c.get("type") == "video" and isinstance(c.get("video"), dict)
               for m in conversation if isinstance(m["content"], (list, tuple)) for c in m["content"]):
            from .video_io import load_multimodal_data, read_decoded_video
            paths = {c["video"]["video_path"] for m in conversation if isinstance(m["content"], (list, tuple))
                     for c in m["content"] if isinstance(c, dict) and c.get("type") == "video" and isinstance(c.get("video"), dict)}
            loader = self.video_loader or read_decoded_video
            conversation, _ = load_multimodal_data(conversation, {p: loader(p) for p in sorted(paths)})
        media, all_ts, conv = [], [], []      # _gather_multimodal_data (:511-530) + the template's view of each item
        given = list(images) if images is not None else None
        for msg in conversation:
            if isinstance(msg["content"], str):
                conv.append(msg)
                continue
            items = []
            for c in msg["content"]:
                if isinstance(c, dict) and c.get("type") == "video":
                    if given is not None:
                        v = _as_frames(given[len(media)][1] if isinstance(given[len(media)], (list, tuple)) and
                                       isinstance(given[len(media)][0], str) else given[len(media)])
                    else:
                        v = _as_frames(c["video"])
                    ts = [float(t) for t in c.get("timestamps", range(len(v)))]
                    media.append(("video", v))
                    all_ts.extend(ts)
                    items.append({"type": "video", "num_frames": len(v), "timestamps": ts})
                elif isinstance(c, dict) and c.get("type") == "image":
                    if given is not None:
                        im = given[len(media)][1] if isinstance(given[len(media)], (list, tuple)) and isinstance(given[len(media)][0], str) else given[len(media)]
                    else:
                        im = c["image"]
                    media.append(("image", _as_frames(im)))
                    item = {"type": "image"}
                    if "timestamp" in c:
                        item["timestamp"] = c["timestamp"]
                        all_ts.append(float(c["timestamp"]))
                    items.append(item)
                else:
                    items.append(c)
            conv.append({"role": msg["role"], "content": items})
        if given is not None and len(given) != len(media):
            raise AssertionError("Number of images does not match the number of image tokens in the text.")
        feats = self.process_images(media) if media else {}
        frames = feats.pop("_frames", [])
        text = render_conversation(conv, add_system_prompt, add_generation_prompt)
        enc, text = self.process_text(text, feats)
        hist_qs, hist_as, cur_q = process_history_qas(conversation)
        video_keys = [content_key(v) for v in frames]      # for the model's visual-token cache
        total_image_num = sum(int(v.shape[0]) for v in frames)     # :656-658
        out = {
            "video_keys": video_keys,
            "input_ids": enc["input_ids"], "attention_mask": enc["attention_mask"],
            "modals": feats.get("modals", []), "tokenizer": self.tokenizer,
            "hist_qs": hist_qs, "hist_as": hist_as, "current_question": cur_q,
            "all_timestamps": all_ts, "total_image_num": total_image_num, "original_text": text,
        }
        for k in ("pixel_values", "grid_sizes", "merge_sizes"):
            if k in feats:
                out[k] = feats[k]
        return out

    def batch_decode(self, *args, **kwargs):
        return self.tokenizer.batch_decode(*args, **kwargs)

    def decode(self, *args, **kwargs):
        return self.tokenizer.decode(*args, **kwargs)
