#!/usr/bin/env python3
"""Qwen2 causal prefill attention (28/4 heads, hd 128, S = 15396, pre-scaled Q) a few times: target of tools/pmc.sh"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogstream_amd import ops
dev = torch.device("cuda:0")
S, hq, hkv, hd = 15396, 28, 4, 128
q = (torch.randn(S, hq * hd, device=dev) * (1.4426950408889634 / hd ** 0.5)).to(torch.bfloat16)
k = torch.randn(S, hkv * hd, device=dev).to(torch.bfloat16)
v = torch.randn(S, hkv * hd, device=dev).to(torch.bfloat16)
for _ in range(4):
    ops.attention(q, k, v, hq=hq, hkv=hkv, head_dim=hd, causal=True, q_prescaled=True)
torch.cuda.synchronize()
