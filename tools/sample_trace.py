#!/usr/bin/env python3
"""per-kernel time of the device sampler on a Qwen2-sized logits row: rocprofv3 --kernel-trace --stats -- python3 tools/sample_trace.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogstream_amd import ops
dev = torch.device("cuda:0")
V = 152064
lg0 = torch.randn(V, device=dev) * 3
prev = torch.randint(0, V, (100,), device=dev)
out = torch.empty(1, dtype=torch.int64, device=dev)
for i in range(50):
    lg = lg0.clone()
    ops.logits_process(lg, prev, 1.05, None, 1.0)
    ops.sample(lg, 20, 0.8, seed=1, offset=i, out=out, temperature=0.7)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(200):
    ops.logits_process(lg, prev, 1.05, None, 1.0)
    ops.sample(lg, 20, 0.8, seed=1, offset=i, out=out, temperature=0.7)
torch.cuda.synchronize()
print("logits_process + sample: %.1f us per token" % ((time.perf_counter() - t0) / 200 * 1e6))
