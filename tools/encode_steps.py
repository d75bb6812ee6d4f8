#!/usr/bin/env python3
"""N encoder steps (encode + project) of one clip with S frame-range streams, a pause between steps -- the workload behind
tools/step_timeline.py:  python tools/encode_steps.py [--frames 64] [--grid 22x42] [--streams 2] [--steps 3] [--debug name=value,...]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogstream_amd import _lib as L  # noqa: E402
from cogstream_amd.vision import Projector, VisionEncoder  # noqa: E402
from cogstream_amd.weights import VisionConfig, random_proj_state, random_vit_state  # noqa: E402

dbg = L.debug_from_argv(sys.argv)
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=64)
ap.add_argument("--grid", default="22x42")
ap.add_argument("--streams", type=int, default=2)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--separate", action="store_true", help="projector as its own call behind the encoder (the form before round 6)")
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
L.check(L.lib.cogs_vit_set_streams(enc.handle.h, args.streams))
for i in range(args.steps + 2):
    torch.cuda.synchronize()
    time.sleep(0.02)
    t0 = time.perf_counter()
    o = enc.encode_project(pix, grid, merge, proj)[1] if not args.separate else proj(enc(pix, grid, merge))
    torch.cuda.synchronize()
    print(f"step {i}: {(time.perf_counter() - t0) * 1e3:.3f} ms {dbg}", flush=True)
