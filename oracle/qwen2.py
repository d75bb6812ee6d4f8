"""Oracle: Qwen2 decoder forward, greedy decode and HF logits processors (third-party arithmetic the
reference calls: transformers==4.46.3 modeling_qwen2 / generation; call sites
model/cogreasoner_chat.py:312-316,322,802-807). TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py."""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F


def rms_norm(x, weight, eps):
    """Qwen2RMSNorm.forward"""
    dt = x.dtype
    xf = x.float()
    var = xf.pow(2).mean(-1, keepdim=True)
    return weight * (xf * torch.rsqrt(var + eps)).to(dt)


def rope_cos_sin(positions: torch.Tensor, head_dim: int, theta: float):
    """Qwen2RotaryEmbedding"""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))
    freqs = positions.float()[:, None] * inv_freq[None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def lora_linear(x, w, name: str, lora: Optional[Dict[str, torch.Tensor]], scaling: float):
    """nn.Linear `name` of the state dict w, plus -- if the adapter targets it -- the UNMERGED low-rank branch of
    peft==0.15.2 (environment.yml:335) `lora.Linear.forward` in eval mode: result + lora_B(lora_A(x)) * scaling,
    scaling = lora_alpha / r (second_stage_training.py:257-264: r = 8, alpha = 16). peft is not installed in the
    build container (and cannot be: no network), so this branch is restated from its published algorithm. It is pinned
    by tests/golden/lora.npz -- the reference's own model object with the linears of the reference's target-module
    list (second_stage_training.py:241-254) wrapped by independently written nn.Linear-based modules computing that
    forward (tests/golden/make_golden.py::golden_lora) -- i.e. pinned to an independent evaluation inside the reference
    model, not to peft's code."""
    y = F.linear(x, w[name + ".weight"], w.get(name + ".bias"))
    if lora is not None and (name + ".lora_A.weight") in lora:
        y = y + F.linear(F.linear(x, lora[name + ".lora_A.weight"]), lora[name + ".lora_B.weight"]) * scaling
    return y


def forward(w: Dict[str, torch.Tensor], embeds: torch.Tensor, *, heads: int, kv_heads: int, layers: int,
            eps: float = 1e-6, theta: float = 1e6, past: Optional[List[Tuple[torch.Tensor, torch.Tensor]]] = None,
            lora: Optional[Dict[str, torch.Tensor]] = None, lora_scaling: float = 2.0,
            ) -> Tuple[torch.Tensor, List[Tuple[torch.Tensor, torch.Tensor]]]:
    """Qwen2Model.forward on embeddings [S, H] -> (last_hidden_state [S, H] after final norm, kv)"""
    lin = lambda t, name: lora_linear(t, w, name, lora, lora_scaling)
    S, H = embeds.shape
    hd = H // heads
    dt = embeds.dtype
    pos0 = past[0][0].shape[1] if past else 0
    cos, sin = rope_cos_sin(torch.arange(pos0, pos0 + S), hd, theta)
    cos, sin = cos.to(dt), sin.to(dt)
    x = embeds
    new_kv = []
    for i in range(layers):
        p = f"layers.{i}."
        h = rms_norm(x, w[p + "input_layernorm.weight"], eps)
        q = lin(h, p + "self_attn.q_proj").view(S, heads, hd).transpose(0, 1)
        k = lin(h, p + "self_attn.k_proj").view(S, kv_heads, hd).transpose(0, 1)
        v = lin(h, p + "self_attn.v_proj").view(S, kv_heads, hd).transpose(0, 1)
        q = q * cos + rotate_half(q) * sin
        k = k * cos + rotate_half(k) * sin
        if past:
            k = torch.cat([past[i][0], k], dim=1)
            v = torch.cat([past[i][1], v], dim=1)
        new_kv.append((k, v))
        rep = heads // kv_heads
        kk = k.repeat_interleave(rep, dim=0)
        vv = v.repeat_interleave(rep, dim=0)
        a = torch.matmul(q, kk.transpose(1, 2)) / math.sqrt(hd)
        ctx = k.shape[1]
        causal = torch.arange(ctx)[None, :] <= (torch.arange(S)[:, None] + pos0)
        a = a.masked_fill(~causal, float("-inf"))
        a = F.softmax(a, dim=-1, dtype=torch.float32).to(dt)
        o = torch.matmul(a, vv).transpose(0, 1).reshape(S, H)
        x = x + lin(o, p + "self_attn.o_proj")
        h = rms_norm(x, w[p + "post_attention_layernorm.weight"], eps)
        g = lin(h, p + "mlp.gate_proj")
        u = lin(h, p + "mlp.up_proj")
        x = x + lin(F.silu(g) * u, p + "mlp.down_proj")
    return rms_norm(x, w["norm.weight"], eps), new_kv


def logits(w, hidden_last: torch.Tensor) -> torch.Tensor:
    return F.linear(hidden_last, w["lm_head.weight"]).float()


def repetition_penalty(scores: torch.Tensor, prev: torch.Tensor, penalty: float) -> torch.Tensor:
    """RepetitionPenaltyLogitsProcessor"""
    if prev.numel() == 0 or penalty == 1.0:
        return scores
    s = scores.clone()
    g = s[prev]
    s[prev] = torch.where(g < 0, g * penalty, g / penalty)
    return s


def allowed_mask(scores: torch.Tensor, allowed: List[int]) -> torch.Tensor:
    """StructuredLogitsProcessor (model/qaselect_module_predict.py:86-103)"""
    mask = torch.full_like(scores, float("-inf"))
    mask[allowed] = 0
    return scores + mask


def greedy_generate(w, embeds, *, heads, kv_heads, layers, max_new_tokens, eos: List[int], eps=1e-6, theta=1e6,
                    rep_penalty: float = 1.0, allowed: Optional[List[int]] = None,
                    prompt_ids: Optional[torch.Tensor] = None) -> Tuple[List[int], torch.Tensor]:
    """GenerationMixin greedy search with inputs_embeds: returns (new token ids, first-step logits).
    The repetition penalty sees prompt_ids (if the caller passed input_ids) + generated ids."""
    hid, kv = forward(w, embeds, heads=heads, kv_heads=kv_heads, layers=layers, eps=eps, theta=theta)
    out: List[int] = []
    first = None
    seen = prompt_ids.clone() if prompt_ids is not None else torch.empty(0, dtype=torch.long)
    for _ in range(max_new_tokens):
        lg = logits(w, hid[-1])
        if first is None:
            first = lg.clone()
        lg = repetition_penalty(lg, seen, rep_penalty)
        if allowed is not None:
            lg = allowed_mask(lg, allowed)
        tok = int(torch.argmax(lg))
        out.append(tok)
        seen = torch.cat([seen, torch.tensor([tok])])
        if tok in eos:
            break
        e = w["embed_tokens.weight"][tok][None, :]
        hid, kv = forward(w, e, heads=heads, kv_heads=kv_heads, layers=layers, eps=eps, theta=theta, past=kv)
    return out, first
