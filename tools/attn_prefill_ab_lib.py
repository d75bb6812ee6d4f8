#!/usr/bin/env python3
"""Qwen2 prompt attention (hd 128, 28 / 4 heads, causal, pre-scaled Q, 15 395 tokens): the default library against
cogstream_amd/libcogs_hip_<name>.so (ALT_NAME=<name> tools/build_alt.sh attn -D...), shuffled rounds in one process.
    python tools/attn_prefill_ab_lib.py name [name ...]"""
import ctypes as C
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogstream_amd import ops  # noqa: E402

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = sys.argv[1:]
alts = [(n, C.CDLL(os.path.join(root, "cogstream_amd", f"libcogs_hip_{n}.so"))) for n in names]
dev = torch.device("cuda:0")
S, hq, hkv, hd = 15395, 28, 4, 128
torch.manual_seed(0)
q = (torch.randn(S, hq * hd, device=dev) * 0.5).bfloat16()
k = (torch.randn(S, hkv * hd, device=dev) * 0.5).bfloat16()
v = torch.randn(S, hkv * hd, device=dev).bfloat16()
ts = {n: [] for n in ["default"] + names}
outs = {}
random.seed(1)
for r in range(int(os.environ.get("AB_ROUNDS", "15"))):
    order = [("default", None)] + alts
    random.shuffle(order)
    for tag, lib in order:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        o = ops.attention(q, k, v, hq=hq, hkv=hkv, head_dim=hd, causal=True, q_prescaled=True, lib=lib)
        e1.record()
        torch.cuda.synchronize()
        if r:
            ts[tag].append(e0.elapsed_time(e1))
        outs[tag] = o
fl = 4.0 * S * S * hd * hq / 2
for n in ts:
    t = sorted(ts[n])[len(ts[n]) // 2]
    print(f"{n:12s} {t:.4f} ms (min {min(ts[n]):.4f})  {fl / t / 1e9:6.0f} TFLOP/s  same bits as default: {bool(torch.equal(outs[n], outs['default']))}")
