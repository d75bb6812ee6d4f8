#!/usr/bin/env python3
"""What each epilogue feature costs on the ViT out-projection / fc2 shapes (59 136 rows, N = 1 152, K = 1 152 / 4 352):
bias | bias + residual | bias + residual + row statistics, interleaved in one process (HIP events, medians)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogstream_amd import _lib as L  # noqa: E402
from cogstream_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M, N = 59136, 1152
for K in (1152, 4352):
    torch.manual_seed(K)
    a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
    w = ((torch.rand(N, K, device=dev) * 2 - 1) * 0.05).bfloat16()
    bias = torch.rand(N, device=dev).bfloat16()
    res = torch.rand(M, N, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    stats = torch.empty(M, N // 64, 2, device=dev, dtype=torch.float32)
    variants = {"bias": {}, "bias+residual": dict(residual=res), "bias+residual+row statistics": dict(residual=res, row_stats=stats)}
    ts = {k: [] for k in variants}
    for r in range(9):
        for name, kw in variants.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.gemm(a, w, bias=bias, out=out, **kw)
            e1.record()
            torch.cuda.synchronize()
            if r:
                ts[name].append(e0.elapsed_time(e1))
            body = L.debug_get("gemm_last_body")
    fl = 2.0 * M * N * K
    for name in variants:
        t = sorted(ts[name])[len(ts[name]) // 2]
        print(f"K={K} {name:32s} {t:.4f} ms  {fl / t / 1e9:6.0f} TFLOP/s (body {body})")
