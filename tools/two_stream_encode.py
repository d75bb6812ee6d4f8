#!/usr/bin/env python3
"""Experiment: encode the cfg2 clip as ONE batch vs as two half-clips on two HIP streams (tails of one half's
GEMMs filled by the other half's kernels?)."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogstream_amd import _lib as L  # noqa: E402
from cogstream_amd.vision import VisionEncoder  # noqa: E402
from cogstream_amd.weights import VisionConfig, random_vit_state  # noqa: E402

dev = torch.device("cuda:0")
cfg = VisionConfig()
enc = VisionEncoder(random_vit_state(cfg, 0, dev, torch.bfloat16), cfg, dtype=torch.bfloat16, device=dev)
T, gh, gw = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 22, 42)
per = gh * gw
pix = (torch.randn(T * per, 588, device=dev) * 0.5).to(torch.bfloat16)


def raw_encode(px, t, stream, ws, out):
    gs = (C.c_int64 * 3)(t, gh, gw)
    ms = (C.c_int64 * 1)(2)
    L.check(L.lib.cogs_vit_encode(enc.handle.h, C.c_void_p(stream.cuda_stream), px.data_ptr(), L.dtype_code(px.dtype), gs, ms, 1,
                                  0, out.data_ptr(), ws.data_ptr(), ws.numel()), "cogs_vit_encode")


def ws_for(n):
    nb = C.c_size_t()
    L.check(L.lib.cogs_vit_workspace_bytes(enc.handle.h, n, C.byref(nb)))
    return torch.empty(nb.value, dtype=torch.uint8, device=dev)


def bench(parts, label):
    streams = [torch.cuda.Stream() for _ in parts]
    wss = [ws_for((e - b) * per) for b, e in parts]
    outs = [torch.empty((e - b) * per // 4, 1152, device=dev, dtype=torch.bfloat16) for b, e in parts]
    def run():
        for (b, e), s, w, o in zip(parts, streams, wss, outs):
            raw_encode(pix[b * per:e * per], e - b, s, w, o)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    print(f"{label:34s} {(time.perf_counter() - t0) / 10 * 1e3:7.2f} ms per clip")
    return torch.cat(outs)


a = bench([(0, T)], f"one batch of {T} frames")
b = bench([(0, T // 2), (T // 2, T)], f"two streams x {T // 2} frames")
c = bench([(i * T // 4, (i + 1) * T // 4) for i in range(4)], f"four streams x {T // 4} frames")
print("bit-identical:", bool(torch.equal(a, b)), bool(torch.equal(a, c)))
