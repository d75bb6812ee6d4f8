#!/usr/bin/env python3
"""In-process A/B of library debug switches on the encoder step (csrc/debug.h; cdna guide rule 24: interleaved rounds
of every variant in ONE process on ONE device).

    python tools/encoder_ab.py [--frames 64] [--grid 22x42] [--rounds 5] [--steps 5] [--streams 2] SPEC [SPEC ...]

SPEC = name=value[,name=value...]; the shipped defaults are always variant 0. Per variant: median / min step time of
encode + project, and the per-class kernel time of one profiled step (GEMM / attention / norm; HIP events around every
launch, one stream). Shapes: cfg2 = --frames 64 --grid 22x42 (default), a rank's 1/8 share = --frames 8, cfg3 = --frames
256 --grid 10x20, its share = --frames 32 --grid 10x20."""
import argparse
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogstream_amd import _lib as L  # noqa: E402
from cogstream_amd.vision import Projector, VisionEncoder  # noqa: E402
from cogstream_amd.weights import VisionConfig, random_proj_state, random_vit_state  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=64)
ap.add_argument("--grid", default="22x42")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--separate", action="store_true", help="projector as its own call behind the encoder (the form before round 6)")
ap.add_argument("--streams", type=int, default=2)
ap.add_argument("specs", nargs="*")
args = ap.parse_args()
gh, gw = (int(v) for v in args.grid.split("x"))
dev = torch.device("cuda:0")
vcfg = VisionConfig()
enc = VisionEncoder(random_vit_state(vcfg, 0, dev, torch.bfloat16), vcfg, device=dev)
proj = Projector(random_proj_state(1152, 3584, 1, dev, torch.bfloat16), device=dev)
T = args.frames
torch.manual_seed(0)
pix = (torch.rand(T * gh * gw, 588, device=dev) * 2 - 1).to(torch.bfloat16)
grid, merge = torch.tensor([[T, gh, gw]]), torch.tensor([2])
variants = [("default", {})] + [(s, dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in s.split(","))) for s in args.specs]


SEPARATE = [bool(args.separate)]      # the pseudo-switch separate=1 of a SPEC: projector as its own call behind the encoder


def step():
    return enc.encode_project(pix, grid, merge, proj)[1] if not SEPARATE[0] else proj(enc(pix, grid, merge))


class applied:
    def __init__(self, sw):
        self.sw = sw

    def __enter__(self):
        self.sep = SEPARATE[0]
        self.old = {k: L.debug_get(k) for k in self.sw if k != "separate"}
        for k, v in self.sw.items():
            if k == "separate":
                SEPARATE[0] = bool(v)
            else:
                L.debug_set(k, v)

    def __exit__(self, *a):
        SEPARATE[0] = self.sep
        for k, v in self.old.items():
            L.debug_set(k, v)


L.check(L.lib.cogs_vit_set_streams(enc.handle.h, args.streams))
times = {n: [] for n, _ in variants}
prof = {n: [] for n, _ in variants}
sums = {}
for n, sw in variants:
    with applied(sw):
        o = step()
        torch.cuda.synchronize()
        sums[n] = float(o.float().abs().sum())
for rnd in range(args.rounds):
    for n, sw in variants:
        with applied(sw):
            step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            times[n].append((time.perf_counter() - t0) / args.steps * 1e3)
            ms, cnt = (C.c_float * 4)(), (C.c_int * 4)()
            L.check(L.lib.cogs_profile_begin(enc.handle.h))
            step()
            L.check(L.lib.cogs_profile_end(enc.handle.h, L.current_stream(), ms, cnt))
            prof[n].append((float(ms[0]), float(ms[1]), float(ms[2])))
med = lambda xs: sorted(xs)[len(xs) // 2]
base = med(times["default"])
print(f"frames={T} grid={gh}x{gw} patches={T * gh * gw} streams={args.streams}")
for n, _ in variants:
    t = med(times[n])
    print(f"{n:40s} step median {t:7.3f} ms (min {min(times[n]):7.3f}) {100 * (t / base - 1):+5.1f} % | profiled step, one stream: "
          f"gemm {med([p[0] for p in prof[n]]):6.2f} attn {med([p[1] for p in prof[n]]):6.2f} norm {med([p[2] for p in prof[n]]):5.2f} ms"
          f" | checksum {sums[n]:.6e}")
