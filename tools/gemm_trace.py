#!/usr/bin/env python3
"""Per-tile cycle stamps of the ping-pong GEMM (debug switch gemm_trace): K-loop / epilogue cycles of workgroup 0."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogstream_amd import _lib as L  # noqa: E402
from cogstream_amd import ops  # noqa: E402

L.debug_from_argv(sys.argv)
L.debug_set("gemm_trace", 1)
L.debug_set("gemm_split", 0)

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "vit"
M = 59136 if which == "vit" else 15396
bf = torch.bfloat16
shapes = ([("qkv plain", 3456, 1152, {}), ("qkv bias+rope", 3456, 1152, dict(bias=True, rope=True)), ("fc1 plain", 4352, 1152, {}),
           ("fc1 gelu", 4352, 1152, dict(bias=True, act=L.ACT_GELU_TANH)), ("fc2 res", 1152, 4352, dict(res=True))]
          if which == "vit" else
          [("gate/up swiglu", 37888, 3584, dict(act=L.ACT_SWIGLU)), ("gate/up plain", 37888, 3584, {}),
           ("down res", 3584, 18944, dict(res=True))])
for name, N, K, kw in shapes:
    a = (torch.rand(M, K, device=dev) - 0.5).to(bf)
    w = ((torch.rand(N, K, device=dev) - 0.5) * 0.05).to(bf)
    args = {}
    if kw.get("bias"):
        args["bias"] = torch.zeros(N, device=dev, dtype=bf)
    if kw.get("res"):
        args["residual"] = torch.zeros(M, N, device=dev, dtype=bf)
    if kw.get("act"):
        args["act"] = kw["act"]
    if kw.get("rope"):
        args.update(rope_cos=torch.rand(M, 36, 2, device=dev), rope_sin=None, rope_cols=2304, head_dim=72)
    print("==", name, file=sys.stderr)
    for _ in range(2):
        ops.gemm(a, w, **args)
    torch.cuda.synchronize()
