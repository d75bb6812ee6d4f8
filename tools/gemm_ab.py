#!/usr/bin/env python3
"""In-process A/B of library debug switches on single GEMM shapes (interleaved rounds of every variant in one process
on one device, HIP events around each launch, medians):

    python tools/gemm_ab.py [--shapes vit|share|llm|M,N,K,mode ...] [--rounds 9] SPEC [SPEC ...]

SPEC = name=value[,name=value...] (csrc/debug.h); the shipped defaults are always variant 0.
mode: plain | gelu (bias + tanh GELU) | res (bias + residual) | stat (bias + residual + row statistics)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogstream_amd import _lib as L  # noqa: E402
from cogstream_amd import ops  # noqa: E402

SETS = {
    "vit": ["59136,3456,1152,plain", "59136,1152,1152,stat", "59136,4352,1152,gelu", "59136,1152,4352,stat"],
    "half": ["29568,3456,1152,plain", "29568,1152,1152,stat", "29568,4352,1152,gelu", "29568,1152,4352,stat"],
    "share": ["7392,3456,1152,plain", "7392,1152,1152,stat", "7392,4352,1152,gelu", "7392,1152,4352,stat"],
    "llm": ["15395,4608,3584,plain", "15395,3584,3584,res", "15395,3584,18944,res"],
}
ap = argparse.ArgumentParser()
ap.add_argument("--shapes", nargs="*", default=["vit"])
ap.add_argument("--rounds", type=int, default=9)
ap.add_argument("specs", nargs="*")
args = ap.parse_args()
shapes = []
for s in args.shapes:
    shapes += SETS.get(s, [s])
variants = [("default", {})] + [(s, dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in s.split(","))) for s in args.specs]
dev = torch.device("cuda:0")
for sh in shapes:
    M, N, K, mode = sh.split(",")
    M, N, K = int(M), int(N), int(K)
    torch.manual_seed(K + N)
    a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
    w = ((torch.rand(N, K, device=dev) * 2 - 1) * 0.05).bfloat16()
    kw = {}
    if mode != "plain":
        kw["bias"] = torch.rand(N, device=dev).bfloat16()
    if mode == "gelu":
        kw["act"] = L.ACT_GELU_TANH
    if mode in ("res", "stat"):
        kw["residual"] = torch.rand(M, N, device=dev).bfloat16()
    if mode == "stat":
        kw["row_stats"] = torch.empty(M, N // 64, 2, device=dev, dtype=torch.float32)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ts = {name: [] for name, _ in variants}
    body, digest = {}, {}
    for r in range(args.rounds + 1):
        for name, sw in variants:
            old = {k: L.debug_get(k) for k in sw}
            for k, v in sw.items():
                L.debug_set(k, v)
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.gemm(a, w, out=out, **kw)
                e1.record()
                torch.cuda.synchronize()
                if r and rep:
                    ts[name].append(e0.elapsed_time(e1))
            body[name] = L.debug_get("gemm_last_body")
            if r == 0:
                digest[name] = float(out.float().sum())
            for k, v in old.items():
                L.debug_set(k, v)
    base = sorted(ts["default"])[len(ts["default"]) // 2]
    for name, _ in variants:
        t = sorted(ts[name])[len(ts[name]) // 2]
        print(f"M={M} N={N} K={K} {mode:5s} {name:40s} {t:.4f} ms (min {min(ts[name]):.4f}) {100 * (t / base - 1):+5.1f} %  "
              f"{2.0 * M * N * K / t / 1e9:6.0f} TFLOP/s  body {body[name]}  sum {digest[name]:.6e}", flush=True)
