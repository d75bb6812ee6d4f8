#!/usr/bin/env python3
"""cfg3-scale pieces of the path: 256-frame clip -> encode -> k-means(K=18) on [256, 50*3584] features ->
near-centroid picks -> pixel-diff mask -> event pooling, wall timings. (The oracle cross-check of the integer
products at this scale lives in tests/test_gpu_golden.py::test_full_size_kmeans_and_mask_match_oracle.)"""
import os, random, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogstream_amd import ops, processing
from cogstream_amd.kmeans import kmeans_with_time_min_max, select_additional_frames
from cogstream_amd.vision import Projector, VisionEncoder
from cogstream_amd.weights import LlmConfig, VisionConfig, random_proj_state, random_vit_state

dev = torch.device("cuda:0")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 256

def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return r, sorted(ts)[len(ts) // 2] * 1e3

frames, ts = processing.synthetic_clip(T, kind="drift")
feats = processing.preprocess_videos([frames])
grid, merge = torch.from_numpy(feats["grid_sizes"]), torch.from_numpy(feats["merge_sizes"])
t, gh, gw = grid[0].tolist()
P = gh * gw // 4
pix = torch.from_numpy(feats["pixel_values"]).to(dev, torch.bfloat16)
print(f"T={T} grid={gh}x{gw} patches={pix.shape[0]} tokens/frame={P}")
vcfg, lcfg = VisionConfig(), LlmConfig()
enc = VisionEncoder(random_vit_state(vcfg, 0, dev, torch.bfloat16), vcfg, device=dev)
proj = Projector(random_proj_state(1152, 3584, 1, dev, torch.bfloat16), device=dev)
mm, ms = timed(lambda: proj(enc(pix, grid, merge)))
print(f"encode+project: {ms:.2f} ms  ({T / ms * 1e3:.0f} frames/s)")
tsd = torch.tensor(ts)
K = -(-T // 15)
feat3 = mm.view(T, P, 3584)
def km():
    random.seed(0); torch.manual_seed(0)
    return kmeans_with_time_min_max(feat3, tsd, K)
(cf, ct, assign), ms = timed(km, 2)
print(f"kmeans K={K} on [{T},{P * 3584}] bf16: {ms:.2f} ms; cluster sizes {torch.bincount(assign.cpu(), minlength=K).tolist()}")
sel, ms = timed(lambda: select_additional_frames(feat3, cf, assign, 2), 2)
print(f"select_additional_frames: {ms:.2f} ms")
minor = torch.zeros(T, dtype=torch.uint8, device=dev); minor[::7] = 1
mask, ms = timed(lambda: ops.pixdiff_mask(pix, T, P, 0.1, 1, minor))
print(f"pixdiff mask: {ms:.3f} ms ({pix.numel() * 2 / ms / 1e6:.0f} GB/s), kept {int(mask.sum())}/{mask.numel()}")
fr = torch.arange(0, T, 7, dtype=torch.int32, device=dev)
mm2 = mm.clone()
_, ms = timed(lambda: ops.frame_mean_to_slot0(mm2, P, fr))
print(f"event pooling ({fr.numel()} frames): {ms:.3f} ms")
