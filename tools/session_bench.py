#!/usr/bin/env python3
"""BASELINE.json configs[3]: a multi-turn streaming session (8 turns, one new 8-frame 480p segment and one
question per turn, growing history + historic-dialogue retrieval) through the whole product path at real
dimensions (random weights, toy byte tokenizer), with and without the visual-token cache and prefix-KV reuse.
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

def answer(conv, model, hist_n, keep_all):
    """infer() (evaluate/answer_generate.py:60-76); keep_all: the selection stage is replaced by its "gt" mode with
    every earlier turn selected -- what a trained selector that keeps the visual history does to the prompt (with
    random weights the FCC selection is noise and usually strips the history)"""
    if not keep_all:
        return infer(conv, model, procr, max_new_tokens=ndec)
    inputs = procr(conversation=conv, add_system_prompt=True, add_generation_prompt=True, return_tensors="pt")
    inputs["pixel_values"] = inputs["pixel_values"].to(dev, bf)
    inputs = model.qa_selection(**inputs, mode="gt", select_gt=list(range(hist_n)), if_visual=True)
    ids, sel = model.generate(**inputs, max_new_tokens=ndec)
    return procr.batch_decode(ids, skip_special_tokens=True)[0].strip(), sel


for cached, prefix, keep_all in ((False, False, False), (True, False, False), (True, True, False),
                                 (True, False, True), (True, True, True)):
    model = CogReasoner(enc, proj, eng, lcfg, generation_config=dict(do_sample=False, eos_token_id=[-1], repetition_penalty=1.05))
    if cached:
        model.enable_visual_cache()
    if prefix:
        model.enable_prefix_cache()
    random.seed(0)
    torch.manual_seed(0)
    conv = [{"role": "system", "content": "You are a helpful assistant."}]
    lat = []
    for i, (fr, ts, q) in enumerate(segs):
        conv.append({"role": "user", "content": [{"type": "video", "video": fr, "timestamps": ts}, {"type": "text", "text": q}]})
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out, sel = answer(conv, model, i, keep_all)
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t0)
        conv.append({"role": "assistant", "content": out})
    print(f"selection {'keep-all' if keep_all else 'FCC     '} visual cache {'on ' if cached else 'off'} prefix KV {'on ' if prefix else 'off'}: "
          "per-turn answer latency (s) " + " ".join(f"{x:.3f}" for x in lat) + f" | session {sum(lat):.2f} s" +
          (f" | cache {model.visual_cache_stats}" if cached else "") +
          (f" | prefix rows reused/seen {model.prefix_cache_stats()}" if prefix else ""))
