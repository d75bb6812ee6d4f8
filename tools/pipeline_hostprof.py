#!/usr/bin/env python3
"""Where does the HOST spend its time in one answer through the product API at cfg3 (256 frames, k-means, 19 event passes)?
cProfile of the third bench.pipeline_once call; GPU work is asynchronous, so kernels show up only as the waits
(synchronize / item / tolist). Usage: python tools/pipeline_hostprof.py [frames]"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cogstream_amd import processing  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
model, processor, _ = bench.build_model(dev)
fr, _ = processing.synthetic_clip(T, kind="drift", clip_idx=0)
dfr = torch.from_numpy(fr).to(dev)
for _ in range(2):
    r = bench.pipeline_once(model, processor, dfr, 64)
print(r)
pr = cProfile.Profile()
pr.enable()
r = bench.pipeline_once(model, processor, dfr, 64)
pr.disable()
print(r)
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
