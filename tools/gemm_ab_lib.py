#!/usr/bin/env python3
"""Default library vs cogstream_amd/libcogs_hip_<name>.so (tools/build_alt.sh, ALT_NAME=<name>; default name: alt) on the bf16
GEMM shapes of the path, interleaved in one process; results must be bit-identical.   python tools/gemm_ab_lib.py [name ...]"""
import ctypes as C
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogstream_amd import _lib as L  # noqa: E402
from cogstream_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
names = sys.argv[1:] or ["alt"]
alts = [(n, C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cogstream_amd", f"libcogs_hip_{n}.so")))
        for n in names]
shapes = [("ViT qkv", 59136, 3456, 1152), ("ViT o", 59136, 1152, 1152), ("ViT fc1", 59136, 4352, 1152), ("ViT fc2", 59136, 1152, 4352),
          ("Qwen2 qkv", 15396, 4608, 3584), ("Qwen2 o", 15396, 3584, 3584), ("Qwen2 gate/up", 15396, 37888, 3584),
          ("Qwen2 down", 15396, 3584, 18944)]
for item in filter(None, os.environ.get("AB_EXTRA", "").split(",")):      # AB_EXTRA="ring fc2:3696:1152:4352,..." adds shapes
    nm, m_, n_, k_ = item.split(":")
    shapes.append((nm, int(m_), int(n_), int(k_)))
ROUNDS = int(os.environ.get("AB_ROUNDS", "7"))     # AB_ROUNDS=25 resolves ~0.5 %
only = os.environ.get("AB_ONLY")                    # comma-separated substrings of shape names
for name, M, N, K in shapes:
    if only and not any(o in name for o in only.split(",")):
        continue
    a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
    w = ((torch.rand(N, K, device=dev) * 2 - 1) * 0.05).bfloat16()
    bias = torch.rand(N, device=dev).bfloat16()
    res = torch.rand(M, N, device=dev).bfloat16() if (os.environ.get("AB_RESIDUAL") == "1" and N <= 4608) else None
    swiglu = os.environ.get("AB_SWIGLU") == "1" and "gate" in name        # the fused SwiGLU epilogue of the Qwen2 MLP (no bias)
    rope = os.environ.get("AB_ROPE") == "1" and name == "ViT qkv"         # bias + rotary (interleaved (cos, sin) table) on q | k
    table = torch.rand(M, 36, 2, device=dev) if rope else None
    hm = 1152 if (rope and os.environ.get("AB_HM") == "1") else 0          # head-major q | k | v output as the encoder runs it
    outs, ts = {}, {n: [] for n in ["default"] + names}
    random.seed(M + N)
    for r in range(ROUNDS):
        order = [("default", None)] + alts
        random.shuffle(order)                 # no variant always runs first in a round
        for tag, lib in order:
            out = torch.empty(M, N // 2 if swiglu else N, device=dev, dtype=torch.bfloat16)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if swiglu:
                ops.gemm(a, w, act=L.ACT_SWIGLU, out=out, lib=lib)
            elif rope:
                out = ops.gemm(a, w, bias=bias, rope_cos=table, rope_cols=2304, head_dim=72, hm_cols=hm, out=(None if hm else out), lib=lib)
            else:
                ops.gemm(a, w, bias=bias, residual=res, out=out, lib=lib)
            e1.record()
            torch.cuda.synchronize()
            if r:
                ts[tag].append(e0.elapsed_time(e1))
            outs[tag] = out
    md = sorted(ts["default"])[len(ts["default"]) // 2]
    fl = 2.0 * M * N * K
    line = f"{name:14s} {M}x{N}x{K}: default {md:.3f} ms {fl / md / 1e9:6.0f} TF"
    for n in names:
        ma = sorted(ts[n])[len(ts[n]) // 2]
        line += f" | {n} {ma:.3f} ms {fl / ma / 1e9:6.0f} TF, x{ma / md:.3f}, same bits {bool(torch.equal(outs['default'], outs[n]))}"
    print(line, flush=True)
