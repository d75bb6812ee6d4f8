#!/usr/bin/env python3
"""Would a fixed split-K pay for a frame-sharded share's fc2 (N = 1152, K = 4352, 3 696 rows per stream = 70 ping-pong tiles on 256
CUs)? Emulation with what exists: the K = 4352 GEMM as shipped (ring kernel, residual + statistics) against `split` GEMMs of K / split
stacked as rows of ONE launch (M x split rows, fp32 output = the partial tiles) plus a torch sum over the partials as a stand-in for
the reduce + epilogue kernel. Alone on the GPU, HIP events, medians."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cogstream_amd import _lib as L  # noqa: E402
from cogstream_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
N, K = 1152, 4352


def med(fn, n=30):
    ts = []
    for i in range(n + 3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        if i >= 3:
            ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


for M in (3696, 7392):
    a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
    w = ((torch.rand(N, K, device=dev) * 2 - 1) * 0.05).bfloat16()
    bias = torch.rand(N, device=dev).bfloat16()
    res = torch.rand(M, N, device=dev).bfloat16()
    stats = torch.empty(M, N // 64, 2, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for S in (1, 2):
        with L.debug_switch("gemm_co_streams", S):
            t0 = med(lambda: ops.gemm(a, w, bias=bias, residual=res, row_stats=stats, out=out))
            body = L.debug_get("gemm_last_body")
        print(f"M={M} as shipped (co-streams hint {S}): {t0 * 1e3:7.1f} us, body {body}")
    for split in (2, 4):
        kc = K // split
        # rows [s*M, (s+1)*M) of the stacked operand = columns [s*kc, (s+1)*kc) of A; the weight slice of split s only matches its own
        # rows, so this is NOT the product -- it has the shape, the tile count and the traffic of the split launch, which is what is timed
        a2 = torch.cat([a[:, s * kc:(s + 1) * kc] for s in range(split)], 0).contiguous()
        w2 = w[:, :kc].contiguous()
        part = torch.empty(split * M, N, device=dev, dtype=torch.float32)
        with L.debug_switch("gemm_ring_cost_permille", 4000), L.debug_switch("gemm_split", 0):
            t1 = med(lambda: ops.gemm(a2, w2, out=part, out_f32=True))
            body = L.debug_get("gemm_last_body")
        t2 = med(lambda: part.view(split, M, N).sum(0).add_(res.float()).bfloat16())
        print(f"M={M} split-K {split}: partial launch {t1 * 1e3:7.1f} us (body {body}, {split * ((M + 255) // 256) * 5} padded tiles) + torch reduce stand-in "
              f"{t2 * 1e3:6.1f} us")
