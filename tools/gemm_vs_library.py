#!/usr/bin/env python3
"""Yardstick only (nothing in the product calls it): the plain bf16 GEMMs of the ViT layer and of the Qwen2 prefill
through torch.nn.functional.linear (hipBLASLt / rocBLAS behind PyTorch-ROCm) next to cogs_gemm on the same random
operands, interleaved rounds in one process, HIP events. Usage: python tools/gemm_vs_library.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogstream_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
bf = torch.bfloat16
shapes = [("vit qkv", 59136, 3456, 1152), ("vit o", 59136, 1152, 1152), ("vit fc1", 59136, 4352, 1152),
          ("vit fc2", 59136, 1152, 4352), ("llm qkv", 15396, 4608, 3584), ("llm o", 15396, 3584, 3584),
          ("llm gate/up", 15396, 37888, 3584), ("llm down", 15396, 3584, 18944)]
prep = []
for name, M, N, K in shapes:
    a = (torch.rand(M, K, device=dev) * 2 - 1).to(bf)
    w = ((torch.rand(N, K, device=dev) * 2 - 1) * 0.05).to(bf)
    prep.append((name, a, w, torch.empty(M, N, device=dev, dtype=bf), 2.0 * M * N * K))
t_lib = {p[0]: [] for p in prep}
t_own = {p[0]: [] for p in prep}
for r in range(6):
    for name, a, w, out, fl in prep:
        for which, store in (("lib", t_lib), ("own", t_own)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if which == "lib":
                F.linear(a, w)
            else:
                ops.gemm(a, w, out=out)
            e1.record()
            torch.cuda.synchronize()
            if r > 0:
                store[name].append(e0.elapsed_time(e1))
for name, a, w, out, fl in prep:
    lib = sorted(t_lib[name])[len(t_lib[name]) // 2]
    own = sorted(t_own[name])[len(t_own[name]) // 2]
    ops.gemm(a, w, out=out)
    digest = int(out.view(torch.int16).to(torch.int64).mul_(torch.arange(1, out.numel() + 1, device=dev).view_as(out) % 8191).sum())
    print(f"{name:12s} digest {digest:x}")     # equal digests across runs (COGS_GEMM_* A/B builds) = bit-identical outputs
    print(f"{name:12s} M{a.shape[0]} N{w.shape[0]} K{a.shape[1]}: library {lib:7.3f} ms {fl / lib / 1e9:7.1f} TFLOP/s | cogs_gemm {own:7.3f} ms "
          f"{fl / own / 1e9:7.1f} TFLOP/s | cogs/library time {own / lib:5.2f}")
