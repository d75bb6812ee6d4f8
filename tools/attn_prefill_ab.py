#!/usr/bin/env python3
"""Qwen2 prompt attention alone (hd 128, 28 query / 4 kv heads, causal, pre-scaled Q) at the bench's prompt length:
python tools/attn_prefill_ab.py [S] [--debug attn_prefill_dma=0]. Every variant named on the command line is timed
in THIS process, interleaved with the default (the LDS-DMA kernel attn_prefill_dma_kernel): attn_prefill_dma=0 is the
register-staged general kernel (attn_fwd_bf16_kernel<128, 2, true>), attn_prefill_deep=0 the round-4 form of the kernel."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogstream_amd import _lib as L
from cogstream_amd import ops

variants = [("default", {})]
while "--debug" in sys.argv:
    i = sys.argv.index("--debug")
    spec = sys.argv[i + 1]
    variants.append((spec, dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in spec.split(","))))
    del sys.argv[i:i + 2]

dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 15395
hq, hkv, hd = 28, 4, 128
torch.manual_seed(0)
q = (torch.randn(S, hq * hd, device=dev) * 0.5).bfloat16()
k = (torch.randn(S, hkv * hd, device=dev) * 0.5).bfloat16()
v = torch.randn(S, hkv * hd, device=dev).bfloat16()


def run():
    return ops.attention(q, k, v, hq=hq, hkv=hkv, head_dim=hd, causal=True, q_prescaled=True)


fl = 4.0 * S * S * hd * hq / 2
times = {name: [] for name, _ in variants}
sums = {}
for rnd in range(7):                      # interleaved rounds: every variant sees the same device state
    for name, sw in variants:
        olds = {name_: L.debug_get(name_) for name_ in sw}
        for name_, val_ in sw.items():
            L.debug_set(name_, val_)
        o = run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        times[name].append((time.perf_counter() - t0) / 5)
        sums[name] = float(o.float().abs().sum())
        for name_, val_ in olds.items():
            L.debug_set(name_, val_)
for name, _ in variants:
    t = sorted(times[name])[len(times[name]) // 2]
    print(f"S={S} variant={name}: median {t * 1e3:.3f} ms (min {min(times[name]) * 1e3:.3f}), {fl / t / 1e12:.0f} TFLOP/s, "
          f"checksum {sums[name]:.6e}")
