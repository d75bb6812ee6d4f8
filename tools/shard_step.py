#!/usr/bin/env python3
"""One rank's share of a frame-sharded encode, alone on one GPU (for rocprofv3 --kernel-trace --stats):
    shard_step.py <frames> <gh> <gw> [steps] [--debug name=value,...]     e.g. 32 10 20 (cfg3 / 8 ranks), 8 22 42 (cfg2 / 8 ranks)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogstream_amd import _lib as L
from cogstream_amd.vision import Projector, VisionEncoder
from cogstream_amd.weights import VisionConfig, random_proj_state, random_vit_state

L.debug_from_argv(sys.argv)       # e.g. --debug gemm_pingpong=0
T, gh, gw = (int(v) for v in sys.argv[1:4])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda:0")
vcfg = VisionConfig()
enc = VisionEncoder(random_vit_state(vcfg, 0, dev, torch.bfloat16), vcfg, device=dev)
proj = Projector(random_proj_state(1152, 3584, 1, dev, torch.bfloat16), device=dev)
pix = (torch.rand(T * gh * gw, 588, device=dev) * 2 - 1).to(torch.bfloat16)
grid, merge = torch.tensor([[T, gh, gw]]), torch.tensor([2])
for _ in range(3):
    proj(enc(pix, grid, merge))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    proj(enc(pix, grid, merge))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"frames={T} grid={gh}x{gw} patches={T * gh * gw}: {dt * 1e3:.3f} ms per step, {T / dt:.0f} frames/s")
