#!/usr/bin/env python3
"""Qwen2 prompt attention alone (hd 128, 28 query / 4 kv heads, causal, pre-scaled Q) at the bench's prompt length:
python tools/attn_prefill_ab.py [S]. COGS_ATTN_PREFILL_DMA=0 in the environment selects the register-staged general
kernel (attn_fwd_bf16_kernel<128, 2, true>), the default is the LDS-DMA kernel (attn_prefill_dma_kernel)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogstream_amd import ops

dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 15395
hq, hkv, hd = 28, 4, 128
torch.manual_seed(0)
q = (torch.randn(S, hq * hd, device=dev) * 0.5).bfloat16()
k = (torch.randn(S, hkv * hd, device=dev) * 0.5).bfloat16()
v = torch.randn(S, hkv * hd, device=dev).bfloat16()


def run():
    return ops.attention(q, k, v, hq=hq, hkv=hkv, head_dim=hd, causal=True, q_prescaled=True)


o = run()
torch.cuda.synchronize()
ts = []
for _ in range(7):
    t0 = time.perf_counter()
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 5)
t = sorted(ts)[len(ts) // 2]
fl = 4.0 * S * S * hd * hq / 2
print(f"S={S} variant={'general' if os.environ.get('COGS_ATTN_PREFILL_DMA') == '0' else 'dma'}: {t * 1e3:.3f} ms, {fl / t / 1e12:.0f} TFLOP/s, "
      f"checksum {float(o.float().abs().sum()):.6e}")
