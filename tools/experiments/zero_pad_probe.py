#!/usr/bin/env python3
"""Does it matter WHAT the padded columns of a half-filled 256-wide tile multiply? N = 1 152 is 4.5 tiles of 256: the last tile's
upper 128 weight rows are clamped copies of row N - 1. Same GEMM with that row random and with that row zero (the 128 garbage
columns then multiply zeros), interleaved in one process."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cogstream_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M, N = 59136, 1152
for K in (1152, 4352):
    torch.manual_seed(K)
    a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
    w = ((torch.rand(N, K, device=dev) * 2 - 1) * 0.05).bfloat16()
    wz = w.clone()
    wz[N - 1] = 0
    bias = torch.rand(N, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ts = {"last row random": [], "last row zero": []}
    for r in range(11):
        for name, ww in (("last row random", w), ("last row zero", wz)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                ops.gemm(a, ww, bias=bias, out=out)
            e1.record()
            torch.cuda.synchronize()
            if r:
                ts[name].append(e0.elapsed_time(e1) / 4)
    for name in ts:
        t = sorted(ts[name])[len(ts[name]) // 2]
        print(f"K={K} {name:18s} {t:.4f} ms  {2.0 * M * N * K / t / 1e9:6.0f} TFLOP/s")
