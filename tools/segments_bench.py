#!/usr/bin/env python3
"""cogs_llm_forward_segments vs one causal prefill of the same token count (Qwen2-7B dims), with the library's
per-class event timing (gemm / attention / norm / other)."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogstream_amd import _lib as L  # noqa: E402
from cogstream_amd.llm import Qwen2Engine  # noqa: E402
from cogstream_amd.weights import LlmConfig, random_llm_state  # noqa: E402

dev = torch.device("cuda:0")
cfg = LlmConfig()
eng = Qwen2Engine(random_llm_state(cfg, 2, dev, torch.bfloat16), cfg, dtype=torch.bfloat16, device=dev)
lens = [int(x) for x in sys.argv[1:]] or [960, 1100, 700, 1300, 640, 900, 1000, 820, 760, 1180, 940, 880, 700, 1020, 860, 900, 640, 1080, 12]
S = sum(lens)
emb = (torch.randn(S, cfg.hidden_size, device=dev) * 0.02).to(torch.bfloat16)


def prof(fn, name):
    fn()
    torch.cuda.synchronize()
    ms = (C.c_float * 4)()
    cnt = (C.c_int * 4)()
    L.check(L.lib.cogs_profile_begin(eng.handle.h))
    t0 = time.perf_counter()
    fn()
    L.check(L.lib.cogs_profile_end(eng.handle.h, L.current_stream(), ms, cnt))
    dt = (time.perf_counter() - t0) * 1e3
    print(f"{name:34s} wall {dt:7.1f} ms | gemm {ms[0]:7.1f} ({cnt[0]}) attn {ms[1]:7.1f} ({cnt[1]}) norm {ms[2]:6.1f} other {ms[3]:6.1f} ({cnt[3]})")


print(f"S = {S} tokens in {len(lens)} sequences (max {max(lens)})")
prof(lambda: eng.forward_segments(emb, lens), "forward_segments")
prof(lambda: eng.forward(emb, None, want_logits=False, want_pooled=True), "one causal prefill, same S")
prof(lambda: [eng.forward(emb[sum(lens[:i]):sum(lens[:i + 1])], None, want_logits=False, want_pooled=True) for i in range(len(lens))],
     "one forward per sequence")
