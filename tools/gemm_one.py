#!/usr/bin/env python3
"""one GEMM shape, a few launches (for rocprofv3 --pmc): python tools/gemm_one.py M N K [gelu|res|plain] [small]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogstream_amd import _lib as L, ops
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
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(6):
    ops.gemm(a, w, out=out, **kw)
torch.cuda.synchronize()
