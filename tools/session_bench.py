#!/usr/bin/env python3
"""BASELINE.json configs[3]: a multi-turn streaming session (8 turns, one new 8-frame 480p segment and one
question per turn, growing history + historic-dialogue retrieval) through the whole product path at real
dimensions (random weights, toy byte tokenizer), with and without the visual-token cache.
    python tools/session_bench.py [turns=8] [frames_per_segment=8] [new_tokens=32]"""
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cogstream_amd import processing as pr  # noqa: E402
from cogstream_amd.answer_generate import infer  # noqa: E402
from cogstream_amd.chat import CogReasoner  # noqa: E402
from cogstream_amd.llm import Qwen2Engine  # noqa: E402
from cogstream_amd.vision import Projector, VisionEncoder  # noqa: E402
from cogstream_amd.weights import LlmConfig, VisionConfig, random_llm_state, random_proj_state, random_vit_state  # noqa: E402
from toy_tokenizer import IM_END, IMAGE, ToyTokenizer  # noqa: E402

a = [int(x) for x in sys.argv[1:]]
turns, fps_seg, ndec = (a + [8, 8, 32][len(a):])[:3]
dev = torch.device("cuda:0")
bf = torch.bfloat16
vcfg = VisionConfig()
lcfg = LlmConfig(image_token_index=IMAGE, eos_token_id=IM_END)
enc = VisionEncoder(random_vit_state(vcfg, 0, dev, bf), vcfg, dtype=bf, device=dev)
proj = Projector(random_proj_state(1152, 3584, 1, dev, bf), dtype=bf, device=dev)
eng = Qwen2Engine(random_llm_state(lcfg, 2, dev, bf), lcfg, dtype=bf, device=dev)
tok = ToyTokenizer()
procr = pr.CogStreamProcessor(tok, device=dev)
import json  # noqa: E402
hist = json.load(open(os.path.join(ROOT, "tests", "golden", "cfg4_history.json")))["turns"]
segs = []
for i in range(turns):
    fr, ts = pr.synthetic_clip(fps_seg, kind="drift", clip_idx=i)
    segs.append((fr, [t + fps_seg * i for t in ts], hist[i % len(hist)]["question"]))

for cached in (False, True):
    model = CogReasoner(enc, proj, eng, lcfg, generation_config=dict(do_sample=False, eos_token_id=[-1], repetition_penalty=1.05))
    if cached:
        model.enable_visual_cache()
    random.seed(0)
    torch.manual_seed(0)
    conv = [{"role": "system", "content": "You are a helpful assistant."}]
    lat = []
    for fr, ts, q in segs:
        conv.append({"role": "user", "content": [{"type": "video", "video": fr, "timestamps": ts}, {"type": "text", "text": q}]})
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out, sel = infer(conv, model, procr, max_new_tokens=ndec)
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t0)
        conv.append({"role": "assistant", "content": out})
    print(f"visual cache {'on ' if cached else 'off'}: per-turn answer latency (s) " + " ".join(f"{x:.3f}" for x in lat) +
          f" | session {sum(lat):.2f} s" + (f" | cache {model.visual_cache_stats}" if cached else ""))
