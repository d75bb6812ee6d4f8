#!/usr/bin/env python3
"""Qwen2-7B (random weights) prefill of 2048 tokens + 24 greedy decode steps: the workload behind
`rocprofv3 --kernel-trace -- python3 tools/decode_trace.py` + tools/trace_summary.py (per-token kernel anatomy)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogstream_amd.llm import Qwen2Engine  # noqa: E402
from cogstream_amd.weights import LlmConfig, random_llm_state  # noqa: E402

dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ndec = int(sys.argv[2]) if len(sys.argv) > 2 else 24
cfg = LlmConfig()
eng = Qwen2Engine(random_llm_state(cfg, seed=2, device=dev, dtype=torch.bfloat16), cfg, dtype=torch.bfloat16, device=dev)
emb = (torch.randn(S, cfg.hidden_size, device=dev) * 0.02).to(torch.bfloat16)
eng.generate(emb[:64], max_new_tokens=2, ignore_eos=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
toks = eng.generate(emb, max_new_tokens=ndec, repetition_penalty=1.05, ignore_eos=True)
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
t0 = time.perf_counter()
eng.generate(emb, max_new_tokens=1, repetition_penalty=1.05, ignore_eos=True)
torch.cuda.synchronize()
t_pre = time.perf_counter() - t0
print("prefill+decode %.1f ms for %d tokens; decode alone %.3f ms/token at context %d" % (
    t_all * 1e3, len(toks), (t_all - t_pre) * 1e3 / max(1, ndec - 1), S))
