"""Oracle: VideoLLaMA3 vision encoder + projector (model/modeling_videollama3_encoder.py,
model/cogreasoner_chat.py:179-211). TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py."""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

BLOCK_DIAG, REF_EAGER_GLOBAL = 0, 1


def rot_pos_ids(grid_sizes: torch.Tensor, merge_sizes: torch.Tensor) -> torch.Tensor:
    """(h, w) position of every patch row in merge-window order (modeling_videollama3_encoder.py:405-429)"""
    out = []
    for (t, h, w), ms in zip(grid_sizes.tolist(), merge_sizes.tolist()):
        hp = torch.arange(h).unsqueeze(1).expand(-1, w).reshape(h // ms, ms, w // ms, ms).permute(0, 2, 1, 3).flatten()
        wp = torch.arange(w).unsqueeze(0).expand(h, -1).reshape(h // ms, ms, w // ms, ms).permute(0, 2, 1, 3).flatten()
        out.append(torch.stack([hp, wp], dim=-1).repeat(t, 1))
    return torch.cat(out, dim=0)


def rotary_freqs(grid_sizes, merge_sizes, head_dim: int) -> torch.Tensor:
    """[N, head_dim/2] angles: VisionRotaryEmbedding(head_dim//2) (:173-183), rot_pos_emb (:430-432)"""
    dim = head_dim // 2
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, dim, 2, dtype=torch.float) / dim))
    seq = torch.arange(int(grid_sizes[:, 1:].max()), dtype=inv_freq.dtype)
    full = torch.outer(seq, inv_freq)
    return full[rot_pos_ids(grid_sizes, merge_sizes)].flatten(1)


def apply_rope(x: torch.Tensor, freqs: torch.Tensor) -> torch.Tensor:
    """apply_rotary_pos_emb_vision (:161-170), x [N, heads, hd]"""
    dt = x.dtype
    x = x.float()
    cos = freqs.cos().unsqueeze(1).repeat(1, 1, 2).float()
    sin = freqs.sin().unsqueeze(1).repeat(1, 1, 2).float()
    half = x.shape[-1] // 2
    rot = torch.cat((-x[..., half:], x[..., :half]), dim=-1)
    return (x * cos + rot * sin).to(dt)


def cu_seqlens(grid_sizes: torch.Tensor) -> torch.Tensor:
    """(:439-440)"""
    cu = torch.repeat_interleave(grid_sizes[:, 1] * grid_sizes[:, 2], grid_sizes[:, 0]).cumsum(dim=0, dtype=torch.int32)
    return F.pad(cu, (1, 0), value=0)


def attention(x, w: Dict[str, torch.Tensor], pre: str, heads: int, cu, freqs, mode: int) -> torch.Tensor:
    """VisionAttention.forward (:236-277, eager: ADDS the bool mask) / VisionFlashAttention2 (:286-315,
    block-diagonal per frame)"""
    n, hdim = x.shape
    hd = hdim // heads
    q = F.linear(x, w[pre + "q_proj.weight"], w[pre + "q_proj.bias"]).view(n, heads, hd)
    k = F.linear(x, w[pre + "k_proj.weight"], w[pre + "k_proj.bias"]).view(n, heads, hd)
    v = F.linear(x, w[pre + "v_proj.weight"], w[pre + "v_proj.bias"]).view(n, heads, hd)
    q, k = apply_rope(q, freqs), apply_rope(k, freqs)
    out = torch.empty_like(q)
    if mode == REF_EAGER_GLOBAL:
        mask = torch.zeros(n, n, dtype=torch.bool)
        for i in range(1, len(cu)):
            mask[cu[i - 1]:cu[i], cu[i - 1]:cu[i]] = True
        a = torch.matmul(q.transpose(0, 1), k.transpose(0, 1).transpose(1, 2)) / math.sqrt(hd)
        a = a + mask  # sic: bool added as 0/1, not used as a mask
        a = F.softmax(a, dim=-1, dtype=torch.float32).to(q.dtype)
        out = torch.matmul(a, v.transpose(0, 1)).transpose(0, 1)
    else:
        for i in range(1, len(cu)):
            s, e = int(cu[i - 1]), int(cu[i])
            a = torch.matmul(q[s:e].transpose(0, 1), k[s:e].transpose(0, 1).transpose(1, 2)) / math.sqrt(hd)
            a = F.softmax(a, dim=-1, dtype=torch.float32).to(q.dtype)
            out[s:e] = torch.matmul(a, v[s:e].transpose(0, 1)).transpose(0, 1)
    return F.linear(out.reshape(n, -1), w[pre + "out_proj.weight"], w[pre + "out_proj.bias"])


def encode(w: Dict[str, torch.Tensor], pixel_values, grid_sizes, merge_sizes, *, heads: int, layers: int,
           eps: float = 1e-6, mode: int = BLOCK_DIAG, return_hidden: bool = False) -> torch.Tensor:
    """Videollama3VisionEncoderModel.forward (:479-510); weights keyed by HF names without the
    'model.vision_encoder.' prefix; dtype = dtype of the weights"""
    dt = w["embeddings.patch_embedding.weight"].dtype
    hdim = w["embeddings.patch_embedding.weight"].shape[0]
    x = pixel_values.to(dt)
    # conv2d k=s=14 on [N,3,14,14] == GEMM on the flattened rows (:202-210)
    x = F.linear(x, w["embeddings.patch_embedding.weight"].reshape(hdim, -1), w["embeddings.patch_embedding.bias"])
    freqs = rotary_freqs(grid_sizes, merge_sizes, hdim // heads)
    cu = cu_seqlens(grid_sizes)
    for i in range(layers):
        p = f"encoder.layers.{i}."
        h = F.layer_norm(x, (hdim,), w[p + "layer_norm1.weight"], w[p + "layer_norm1.bias"], eps)
        x = x + attention(h, w, p + "self_attn.", heads, cu, freqs, mode)
        h = F.layer_norm(x, (hdim,), w[p + "layer_norm2.weight"], w[p + "layer_norm2.bias"], eps)
        h = F.linear(h, w[p + "mlp.fc1.weight"], w[p + "mlp.fc1.bias"])
        h = F.gelu(h, approximate="tanh")
        x = x + F.linear(h, w[p + "mlp.fc2.weight"], w[p + "mlp.fc2.bias"])
    if return_hidden:
        return x
    x = F.layer_norm(x, (hdim,), w["post_layernorm.weight"], w["post_layernorm.bias"], eps)
    outs = []
    for chunk, gs, ms in zip(x.split(grid_sizes.prod(dim=1).tolist(), dim=0), grid_sizes.tolist(), merge_sizes.tolist()):
        t, gh, gw = gs
        c = chunk.shape[-1]
        y = chunk.view(t, gh // ms, gw // ms, ms, ms, c).permute(0, 1, 3, 2, 4, 5).reshape(t, gh, gw, c).permute(0, 3, 1, 2)
        y = F.interpolate(y, size=(gh // ms, gw // ms), mode="bilinear")
        outs.append(y.permute(0, 2, 3, 1).reshape(-1, c))
    return torch.cat(outs, dim=0)


def project(w: Dict[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """MlpGeluProjector (cogreasoner_chat.py:179-211); keys readout.{0,2}.{weight,bias}"""
    h = F.gelu(F.linear(x, w["readout.0.weight"], w["readout.0.bias"]))
    return F.linear(h, w["readout.2.weight"], w["readout.2.bias"])


def patchify_order(t: int, gh: int, gw: int, ms: int) -> torch.Tensor:
    """row r of pixel_values -> (frame, patch_row, patch_col), restating the transpose of
    image_processing_videollama3.py:326-345 (frame -> merge-row -> merge-col -> 2x2 window)"""
    idx = torch.stack(torch.meshgrid(torch.arange(t), torch.arange(gh), torch.arange(gw), indexing="ij"), dim=-1)
    idx = idx.view(t, gh // ms, ms, gw // ms, ms, 3).permute(0, 1, 3, 2, 4, 5)
    return idx.reshape(-1, 3)
