#!/usr/bin/env python3
"""In-process A/B of library debug switches on the Qwen2-7B token loop (random weights): prefill S tokens once, then time
`ndec` greedy decode steps per variant, interleaved rounds (csrc/debug.h; cdna guide rule 24).
    python tools/decode_ab.py [S=15395] [ndec=64] SPEC [SPEC ...]        SPEC = name=value[,name=value...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogstream_amd import _lib as L  # noqa: E402
from cogstream_amd.llm import Qwen2Engine  # noqa: E402
from cogstream_amd.weights import LlmConfig, random_llm_state  # noqa: E402

nums = [a for a in sys.argv[1:] if a.isdigit()]
specs = [a for a in sys.argv[1:] if "=" in a]
S = int(nums[0]) if nums else 15395
ndec = int(nums[1]) if len(nums) > 1 else 64
dev = torch.device("cuda:0")
cfg = LlmConfig()
eng = Qwen2Engine(random_llm_state(cfg, seed=2, device=dev, dtype=torch.bfloat16), cfg, dtype=torch.bfloat16, device=dev)
torch.manual_seed(0)
emb = (torch.randn(S, cfg.hidden_size, device=dev) * 0.02).to(torch.bfloat16)
variants = [("default", {})] + [(s, dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in s.split(","))) for s in specs]
cache = eng.new_cache(S + ndec + 8)
times = {n: [] for n, _ in variants}
toks = {}
for rnd in range(4):
    for name, sw in variants:
        old = {k: L.debug_get(k) for k in sw}
        for k, v in sw.items():
            L.debug_set(k, v)
        cache.reset(0)
        res = eng.forward(emb, cache)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = eng.generate(emb, max_new_tokens=ndec, repetition_penalty=1.05, ignore_eos=True, cache=cache, prefilled=res)
        torch.cuda.synchronize()
        times[name].append((time.perf_counter() - t0) / (len(out) - 1) * 1e3)
        toks[name] = [int(t) for t in out[:8]]
        for k, v in old.items():
            L.debug_set(k, v)
for name, _ in variants:
    t = sorted(times[name])[len(times[name]) // 2]
    print(f"context {S}, {ndec} tokens, {name:32s}: median {t:.4f} ms/token (min {min(times[name]):.4f}) = {1e3 / t:.1f} tok/s; first tokens {toks[name]}")
