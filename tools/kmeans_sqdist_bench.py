#!/usr/bin/env python3
"""cogs_kmeans_sqdist alone at the cfg3 size ([256, 179200] bf16): K = 18 (a Lloyd pass) and K = 1 (a k-means++ pass),
device time per call from HIP events over back-to-back calls. COGS_KM_RG=<n> overrides the row groups."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogstream_amd import ops
dev = torch.device("cuda:0")
T, PD = 256, 50 * 3584
x = (torch.randn(T, PD, device=dev) * 0.5).to(torch.bfloat16)
for K in (18, 1, 32, 40):
    ws = ops.kmeans_workspace(T, PD, K, dev)
    c = torch.randn(K, PD, device=dev)
    for _ in range(3):
        ops.kmeans_sqdist(x, c, None, K, ws)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    n = 20
    for _ in range(n):
        ops.kmeans_sqdist(x, c, None, K, ws)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"K={K:3d}: {us:7.1f} us per sqdist+reduce call, features read at {T * PD * 2 / us / 1e6:.2f} TB/s")
