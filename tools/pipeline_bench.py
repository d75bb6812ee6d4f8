#!/usr/bin/env python3
"""The WHOLE product path on the cfg2 clip, real dimensions, random weights, toy byte tokenizer:
processor (GPU pre-processing) -> qa_selection -> generate (encode, k-means, event-summary passes, compression,
prefill, decode), timed per stage with torch.cuda.synchronize() around each.
    python tools/pipeline_bench.py [frames=64] [new_tokens=64] [--history]"""
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from cogstream_amd import processing as pr  # noqa: E402
from cogstream_amd.chat import CogReasoner  # noqa: E402
from cogstream_amd.llm import Qwen2Engine  # noqa: E402
from cogstream_amd.vision import Projector, VisionEncoder  # noqa: E402
from cogstream_amd.weights import LlmConfig, VisionConfig, random_llm_state, random_proj_state, random_vit_state  # noqa: E402
from toy_tokenizer import IM_END, IMAGE, ToyTokenizer  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
T = int(args[0]) if args else 64
ndec = int(args[1]) if len(args) > 1 else 64
dev = torch.device("cuda:0")
bf = torch.bfloat16
vcfg = VisionConfig()
lcfg = LlmConfig(image_token_index=IMAGE, eos_token_id=IM_END)   # real dimensions, the toy tokenizer's special ids
enc = VisionEncoder(random_vit_state(vcfg, 0, dev, bf), vcfg, dtype=bf, device=dev)
proj = Projector(random_proj_state(1152, 3584, 1, dev, bf), dtype=bf, device=dev)
eng = Qwen2Engine(random_llm_state(lcfg, 2, dev, bf), lcfg, dtype=bf, device=dev)
model = CogReasoner(enc, proj, eng, lcfg, generation_config=dict(do_sample=False, eos_token_id=[-1], repetition_penalty=1.05))
tok = ToyTokenizer()
procr = pr.CogStreamProcessor(tok, device=dev)
frames, ts = pr.synthetic_clip(T, kind="drift")
conv = [{"role": "system", "content": "You are a helpful assistant."}]
if "--history" in sys.argv:
    conv += [{"role": "user", "content": [{"type": "video", "video": frames[:T // 2], "timestamps": ts[:T // 2]},
                                          {"type": "text", "text": "What is in the first half?"}]},
             {"role": "assistant", "content": "A drifting pattern."},
             {"role": "user", "content": [{"type": "video", "video": frames[T // 2:], "timestamps": ts[T // 2:]},
                                          {"type": "text", "text": "And what happens afterwards?"}]}]
else:
    conv += [{"role": "user", "content": [{"type": "video", "video": frames, "timestamps": ts},
                                          {"type": "text", "text": "What moves across the clip?"}]}]


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


for rep in range(2):
    random.seed(0)
    torch.manual_seed(0)
    t0 = sync()
    inputs = procr(conversation=conv, add_system_prompt=True, add_generation_prompt=True, return_tensors="pt")
    t1 = sync()
    sel = model.qa_selection(**inputs, mode="FCC")
    t2 = sync()
    marks = {}
    orig = {}
    for name in ("encode_images", "select_events_based_on_summary", "compress_unimportant_events", "_get_compression_mask"):
        fn = getattr(model, name)
        orig[name] = fn

        def wrap(*a, _fn=fn, _n=name, **k):
            s = sync()
            r = _fn(*a, **k)
            marks[_n] = marks.get(_n, 0.0) + (sync() - s)
            return r
        setattr(model, name, wrap)
    import cogstream_amd.chat as chat_mod
    sub = {}

    def timed_fn(fn, name):
        def w(*a, **k):
            s = sync()
            r = fn(*a, **k)
            sub[name] = sub.get(name, 0.0) + (sync() - s)
            return r
        return w
    o_km, o_sel, o_fs, o_tok = chat_mod.kmeans_with_time_min_max, chat_mod.select_additional_frames, eng.forward_segments, tok.__call__
    chat_mod.kmeans_with_time_min_max = timed_fn(o_km, "kmeans")
    chat_mod.select_additional_frames = timed_fn(o_sel, "select_additional_frames")
    def fs_logged(e, lens):
        sub["_lens"] = [int(x) for x in lens]
        return o_fs(e, lens)
    eng.forward_segments = timed_fn(fs_logged, "forward_segments (K+1 sequences)")
    model.tokenizer = type("T", (), {"__call__": staticmethod(timed_fn(tok.__call__, "tokenizer calls")),
                                     "__getattr__": lambda self, n: getattr(tok, n)})()
    ids, _ = model.generate(**sel, max_new_tokens=ndec)
    chat_mod.kmeans_with_time_min_max, chat_mod.select_additional_frames, eng.forward_segments = o_km, o_sel, o_fs
    model.tokenizer = tok
    lens_seen = sub.pop("_lens", None)
    marks.update({"  . " + k: v for k, v in sub.items()})
    if lens_seen:
        print(f"   event-summary sequences: {len(lens_seen)} with {sum(lens_seen)} tokens, lengths {lens_seen}")
    t3 = sync()
    for name, fn in orig.items():
        setattr(model, name, fn)
    n_in = int(sel["new_input_ids"].numel())
    print(f"[run {rep}] frames {T}, prompt ids {n_in}, new tokens {ids.shape[1]}")
    print(f"   processor (H2D + GPU pre-processing + tokenise) {1e3 * (t1 - t0):8.1f} ms")
    print(f"   qa_selection                                    {1e3 * (t2 - t1):8.1f} ms")
    print(f"   generate                                        {1e3 * (t3 - t2):8.1f} ms, of which")
    for k, v in marks.items():
        print(f"        {k:40s} {1e3 * v:8.1f} ms")
    print(f"   whole answer                                    {1e3 * (t3 - t0):8.1f} ms")
