#!/usr/bin/env python3
"""Would splitting the N = 1152 GEMMs by COLUMNS pay? 1152 = 4 x 256 + 128: the ping-pong kernel pads the fifth
column tile to 256 (11 % of its MFMAs). Times, with the out-proj / fc2 epilogue (bias + residual + row statistics):
the whole GEMM, its first 1024 columns (ping-pong kernel) and its last 128 columns (256x128 ring kernel)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogstream_amd import ops
dev = torch.device("cuda:0")
bf = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 59136
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for K in (1152, 4352):
    a = (torch.rand(M, K, device=dev) * 2 - 1).to(bf)
    w = ((torch.rand(1152, K, device=dev) * 2 - 1) * 0.05).to(bf)
    b = torch.rand(1152, device=dev).to(bf)
    r = torch.rand(M, 1152, device=dev).to(bf)
    out = torch.empty(M, 1152, device=dev, dtype=bf)
    st = torch.zeros(M, 18, 2, device=dev)
    whole = t(lambda: ops.gemm(a, w, b, residual=r, out=out, row_stats=st))
    left = t(lambda: ops.gemm(a, w[:1024], b[:1024], residual=r[:, :1024], out=out[:, :1024], row_stats=st[:, :16]))
    right = t(lambda: ops.gemm(a, w[1024:], b[1024:], residual=r[:, 1024:], out=out[:, 1024:], row_stats=st[:, 16:]))
    print(f"M={M} K={K}: whole {whole:.4f} ms | 1024 cols {left:.4f} + 128 cols {right:.4f} = {left + right:.4f} ms ({(left + right) / whole:.3f})")
