#!/usr/bin/env python3
"""One attention shape, a few launches (for rocprofv3 --pmc): python tools/attn_one.py [nseg seglen heads hd]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogstream_amd import ops  # noqa: E402

a = [int(x) for x in sys.argv[1:]] + [64, 924, 16, 72][len(sys.argv) - 1:]
nseg, seglen, heads, hd = a
dev = torch.device("cuda:0")
L = nseg * seglen
qkv = torch.randn(L, 3 * heads * hd, device=dev)
qkv[:, :heads * hd] *= 1.4426950408889634 / hd ** 0.5          # pre-scaled Q, as the production path hands it over
qkv = qkv.to(torch.bfloat16)
cu = torch.arange(0, L + 1, seglen, device=dev, dtype=torch.int32)
H = heads * hd
for _ in range(5):
    ops.attention(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], hq=heads, hkv=heads, head_dim=hd, cu_seqlens=cu, max_seqlen=seglen, q_prescaled=True)
torch.cuda.synchronize()
