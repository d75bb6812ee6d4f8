#!/usr/bin/env python3
"""Per-shape GEMM timing (HIP events, interleaved rounds in one process): the ViT layer's four GEMMs and the
Qwen2 prefill GEMMs, with and without their epilogues. Usage: python tools/gemm_bench.py [vit|llm]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogstream_amd import _lib as L  # noqa: E402
from cogstream_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "vit"
M = 59136 if which == "vit" else 15396
bf = torch.bfloat16


def rnd(*s):
    return (torch.rand(*s, device=dev) * 2 - 1).to(bf)


cases = []
if which == "vit":
    H, I = 1152, 4352
    cos = torch.rand(M, 36, 2, device=dev)   # interleaved (cos, sin)
    cases = [
        ("qkv  N3456 K1152 plain", dict(N=3456, K=H)),
        ("qkv  +bias+rope", dict(N=3456, K=H, bias=True, rope=(cos, 2304, 72))),
        ("o    N1152 K1152 plain", dict(N=H, K=H)),
        ("o    +bias+res", dict(N=H, K=H, bias=True, res=True)),
        ("fc1  N4352 K1152 plain", dict(N=I, K=H)),
        ("fc1  +bias+gelu_tanh", dict(N=I, K=H, bias=True, act=L.ACT_GELU_TANH)),
        ("fc2  N1152 K4352 plain", dict(N=H, K=I)),
        ("fc2  +bias+res", dict(N=H, K=I, bias=True, res=True)),
    ]
else:
    H, I = 3584, 18944
    cases = [
        ("qkv  N4608 K3584 plain", dict(N=4608, K=H)),
        ("o    N3584 K3584 +res", dict(N=H, K=H, res=True)),
        ("gu   N37888 K3584 swiglu", dict(N=2 * I, K=H, act=L.ACT_SWIGLU)),
        ("down N3584 K18944 +res", dict(N=H, K=I, res=True)),
    ]

prepared = []
for name, c in cases:
    a = rnd(M, c["K"])
    w = rnd(c["N"], c["K"]) * 0.05
    kw = {}
    if c.get("bias"):
        kw["bias"] = rnd(c["N"])
    if c.get("res"):
        kw["residual"] = rnd(M, c["N"])
    if c.get("act"):
        kw["act"] = c["act"]
    if c.get("rope"):
        t, cols, hd = c["rope"]
        kw.update(rope_cos=t, rope_sin=None, rope_cols=cols, head_dim=hd)
    out = torch.empty(M, c["N"] // 2 if c.get("act") == L.ACT_SWIGLU else c["N"], device=dev, dtype=bf)
    prepared.append((name, a, w, kw, out, 2.0 * M * c["N"] * c["K"]))

rounds = 5
times = {n: [] for n, *_ in prepared}
for r in range(rounds + 1):
    for name, a, w, kw, out, fl in prepared:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.gemm(a, w, out=out, **kw)
        e1.record()
        torch.cuda.synchronize()
        if r > 0:
            times[name].append(e0.elapsed_time(e1))
for name, a, w, kw, out, fl in prepared:
    t = sorted(times[name])
    med = t[len(t) // 2]
    print(f"{name:28s} median {med:7.3f} ms  min {t[0]:7.3f} ms  {fl / med / 1e9:7.1f} TFLOP/s (median)")
