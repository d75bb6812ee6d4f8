#!/usr/bin/env python3
"""one GEMM shape, a few launches (for rocprofv3 --pmc): python tools/gemm_one.py M N K [gelu|res|stat|plain] [--debug name=value,...]
prints the median launch time (HIP events) and the body the dispatch took"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogstream_amd import _lib as L, ops
dbg = L.debug_from_argv(sys.argv)
M, N, K = (int(v) for v in sys.argv[1:4])
mode = sys.argv[4] if len(sys.argv) > 4 else "plain"
dev = torch.device("cuda:0")
a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
w = ((torch.rand(N, K, device=dev) * 2 - 1) * 0.05).bfloat16()
kw = {}
if mode == "gelu":
    kw = dict(bias=torch.rand(N, device=dev).bfloat16(), act=L.ACT_GELU_TANH)
elif mode == "res":
    kw = dict(bias=torch.rand(N, device=dev).bfloat16(), residual=torch.rand(M, N, device=dev).bfloat16())
elif mode == "stat":
    kw = dict(bias=torch.rand(N, device=dev).bfloat16(), residual=torch.rand(M, N, device=dev).bfloat16(),
              row_stats=torch.empty(M, N // 64, 2, device=dev, dtype=torch.float32))
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
ts = []
for _ in range(8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.gemm(a, w, out=out, **kw)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
t = sorted(ts[2:])[len(ts[2:]) // 2]
print(f"M={M} N={N} K={K} {mode} {dbg}: {t:.4f} ms, {2.0 * M * N * K / t / 1e9:.0f} TFLOP/s, body {L.debug_get('gemm_last_body')}")
