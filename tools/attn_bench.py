#!/usr/bin/env python3
"""Attention timing (HIP events, interleaved rounds): the ViT block-diagonal shape (hd 72, 16 heads, 924-token
frames), the same tokens in longer segments, and the Qwen2 causal prefill (hd 128, 28/4 heads)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogstream_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16
cases = []


PRE = os.environ.get("ATTN_PRE") == "1"      # Q pre-multiplied by scale*log2(e), deferred-max softmax kernels


def vit_case(nseg, seglen, heads=16, hd=72):
    L = nseg * seglen
    qkv = torch.randn(L, 3 * heads * hd, device=dev)
    if PRE:
        qkv[:, :heads * hd] *= 1.4426950408889634 / hd ** 0.5
    qkv = qkv.to(bf)
    cu = torch.arange(0, L + 1, seglen, device=dev, dtype=torch.int32)
    out = torch.empty(L, heads * hd, device=dev, dtype=bf)
    H = heads * hd
    fl = 4.0 * nseg * heads * seglen * seglen * hd

    def run():
        ops.attention(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], hq=heads, hkv=heads, head_dim=hd, cu_seqlens=cu,
                      max_seqlen=seglen, out=out, q_prescaled=PRE)
    return (f"vit hd{hd} {nseg}x{seglen}", run, fl)


def llm_case(S, hq=28, hkv=4, hd=128):
    q = (torch.randn(S, hq * hd, device=dev) * (1.4426950408889634 / hd ** 0.5 if PRE else 1.0)).to(bf)
    k = torch.randn(S, hkv * hd, device=dev).to(bf)
    v = torch.randn(S, hkv * hd, device=dev).to(bf)
    out = torch.empty(S, hq * hd, device=dev, dtype=bf)
    fl = 4.0 * hq * S * S * hd / 2

    def run():
        ops.attention(q, k, v, hq=hq, hkv=hkv, head_dim=hd, causal=True, out=out, q_prescaled=PRE)
    return (f"llm hd{hd} causal S={S}", run, fl)


cases = [vit_case(64, 924), vit_case(66, 896), vit_case(16, 3696), vit_case(64, 924, hd=128), vit_case(64, 924, heads=18, hd=64),
         llm_case(15396)]
times = {c[0]: [] for c in cases}
for r in range(6):
    for name, run, fl in cases:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run()
        e1.record()
        torch.cuda.synchronize()
        if r > 0:
            times[name].append(e0.elapsed_time(e1))
for name, run, fl in cases:
    t = sorted(times[name])
    med = t[len(t) // 2]
    print(f"{name:28s} median {med:7.3f} ms  min {t[0]:7.3f} ms  {fl / med / 1e9:7.1f} TFLOP/s (algorithmic)")
