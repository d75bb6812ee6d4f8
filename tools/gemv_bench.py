#!/usr/bin/env python3
"""Decode GEMV timing per Qwen2-7B shape (HIP events, interleaved rounds, weights rotated through > L2+MALL).
NOTE: every call goes through Python + ctypes (~12 us), so shapes that take less than that (qkv, o) read as ~12 us whatever
the kernel does; use tools/micro/gemv_micro.cpp (launches from C++) or a rocprofv3 trace for those."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogstream_amd import _lib as L  # noqa: E402
from cogstream_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16
ALT = None
if os.environ.get("COGS_ALT_LIB") == "1":      # tools/build_alt.sh gemv <flags>: time that build instead
    import ctypes
    ALT = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cogstream_amd", "libcogs_hip_alt.so"))
H, I, V = 3584, 18944, 152064
shapes = [("qkv  N4608  K3584", 4608, H, {}), ("o    N3584  K3584 +res", H, H, {"res": True}),
          ("gu   N37888 K3584 swiglu", 2 * I, H, {"act": L.ACT_SWIGLU}), ("down N3584  K18944 +res", H, I, {"res": True}),
          ("head N152064 K3584 f32", V, H, {"f32": True})]
prep = []
for name, N, K, kw in shapes:
    copies = max(2, int(1.2e9 // (N * K * 2)) + 1)       # rotate through > 1 GB of weights so nothing stays cached
    ws = [(torch.randn(N, K, device=dev) * 0.02).to(bf) for _ in range(copies)]
    x = torch.randn(1, K, device=dev).to(bf)
    args = {}
    if kw.get("res"):
        args["residual"] = torch.randn(1, N, device=dev).to(bf)
    if kw.get("act"):
        args["act"] = kw["act"]
    if kw.get("f32"):
        args["out_f32"] = True
    prep.append((name, ws, x, args, N * K * 2))
times = {p[0]: [] for p in prep}
REP = 12
for r in range(5):
    for name, ws, x, args, nbytes in prep:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(REP):                      # back to back, as inside a decode step
            ops.gemm(x, ws[(r * REP + i) % len(ws)], lib=ALT, **args)
        e1.record()
        torch.cuda.synchronize()
        if r > 0:
            times[name].append(e0.elapsed_time(e1) / REP)
tot = 0.0
for name, ws, x, args, nbytes in prep:
    t = sorted(times[name])
    med = t[len(t) // 2]
    per_tok = med * (1 if name.startswith("head") else 28)
    tot += per_tok
    print(f"{name:28s} median {med * 1e3:8.1f} us  {nbytes / med / 1e9:7.2f} TB/s   x{1 if name.startswith('head') else 28} = {per_tok:6.3f} ms/token")
print(f"GEMV total per token: {tot:.3f} ms")
